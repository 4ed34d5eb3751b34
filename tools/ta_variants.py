#!/usr/bin/env python
"""Tuning helper: time tri_attn_core (gemm mode 1) at the bench shape; run once per PRD_TA_VARIANT value
(unset / 0: second-generation core prd_tri2.hip; 10: first generation, 8 waves x 2 tiles; 1-3: its other variants)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from protein_redesign_amd import _lib, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 320
_lib.lib().prd_set_gemm_mode(1)
g = torch.Generator().manual_seed(0)
pair = torch.randn(1, N, N, 64, generator=g).cuda()
mask = torch.ones(1, N).cuda()
w = [torch.randn(64, 64, generator=g).cuda() / 8 for _ in range(4)] + [torch.randn(64, generator=g).cuda()]
og = torch.empty(1, N, N, 64, device="cuda")
for i in range(4):
    ops.tri_attn_core(pair, mask, w, 4, 16, ending=bool(i & 1), og=og)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    ops.tri_attn_core(pair, mask, w, 4, 16, ending=bool(i & 1), og=og)
e1.record()
torch.cuda.synchronize()
print(f"PRD_TA_VARIANT={os.environ.get('PRD_TA_VARIANT', '0')}: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per launch", flush=True)
