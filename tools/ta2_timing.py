#!/usr/bin/env python
"""Phase timing of tri_attn_core_v2_kernel (csrc/prd_tri2.hip) from in-kernel cycle stamps.  Needs the diagnostic library:
    python -m protein_redesign_amd.build --timing ; PRD_LIB=protein_redesign_amd/libprd_hip_timing.so python tools/ta2_timing.py [N]
Stamps per (workgroup, wave, row iteration): 0 loop top (after the barrier), 1 end of phase 1, 2 after the barrier,
3 end of the key loops, 4 after the barrier, 5 end of merge + store."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import _lib, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 320
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
pair = torch.randn(1, N, N, 64, generator=g).to(dev)
mask = torch.ones(1, N, device=dev)
wts = [torch.randn(64, 64, generator=g).to(dev) * 0.1 for _ in range(4)] + [torch.zeros(64, device=dev)]
og = torch.empty(1, N, N, 64, device=dev)
L = _lib.lib()
L.prd_debug_read2.argtypes = [ctypes.c_void_p]
for ending in (False, True):
    for _ in range(3):
        ops.tri_attn_core_v2(pair, mask, wts, 4, 16, ending=ending, og=og)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 12 * 8 * 16, dtype=np.uint64)
    assert L.prd_debug_read2(buf.ctypes.data) == 0
    full = buf.reshape(256, 12, 8, 16).astype(np.int64)
    nit = int((full[0, 0, :, 0] > 0).sum())
    full = full[:, :, :nit]
    t = full[..., :6]
    t0 = t[..., 0].min()
    print(f"ending={ending} N={N} rows/WG={nit}  kernel span (cycles) = {t[..., 5].max() - t0}")
    names = ["phase 1 (project)", "barrier", "phase 2 (key loops)", "barrier", "merge + store"]
    for k, nm in enumerate(names):
        d = t[..., k + 1] - t[..., k]
        print(f"  {nm:22s} per wave mean {d.mean():8.0f}   max over waves (mean over rows) {d.max(axis=1).mean():8.0f}   min over waves {d.min(axis=1).mean():8.0f}")
    top = t[:, :, 1:, 0] - t[:, :, :-1, 5] if nit > 1 else np.zeros(1)
    print(f"  top-of-loop wait        per wave mean {top.mean():8.0f}")
    span = t[..., 5].max(axis=1) - t[..., 0].min(axis=1)
    print(f"  row span per WG: mean {span.mean():.0f} min {span.min()} max {span.max()};  by wave, phase 2: " +
          " ".join(f"{(t[:, w, :, 3] - t[:, w, :, 2]).mean():.0f}" for w in range(12)))
    # finer stamp: 6 after LayerNorm + split of the wave's block
    print(f"  phase 1: LN + split {(full[..., 6] - full[..., 0]).mean():.0f}  by wave, phase 1: " +
          " ".join(f"{(t[:, w, :, 1] - t[:, w, :, 0]).mean():.0f}" for w in range(12)))
