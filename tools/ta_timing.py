#!/usr/bin/env python
"""Phase timing of tri_attn_core_kernel from in-kernel cycle stamps (library built with -DPRD_TIMING, PRD_LIB=...).
Stamps per (workgroup, wave, row iteration): 0 loop top, 1 end of projections, 2 after barrier, 3 end of key loops."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import _lib, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 320
if len(sys.argv) > 2:
    _lib.lib().prd_set_gemm_mode(int(sys.argv[2]))       # 1: the split-operand kernel
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
pair = torch.randn(1, N, N, 64, generator=g).to(dev)
mask = torch.ones(1, N, device=dev)
wts = [torch.randn(64, 64, generator=g).to(dev) * 0.1 for _ in range(4)] + [torch.zeros(64, device=dev)]
og = torch.empty(1, N, N, 64, device=dev)
for ending in (False, True):
    for _ in range(3):
        ops.tri_attn_core(pair, mask, wts, 4, 16, ending=ending, og=og)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 16 * 8 * 4, dtype=np.uint64)      # sizeof(prd_dbg)
    L = _lib.lib()
    L.prd_debug_read.argtypes = [ctypes.c_void_p]
    assert L.prd_debug_read(buf.ctypes.data) == 0
    t = buf[: 256 * 12 * 8 * 4].reshape(256, 12, 8, 4).astype(np.int64)
    nit = int((t[0, 0, :, 0] > 0).sum())
    nw = int((t[0, :, 0, 0] > 0).sum())                   # waves that stamped (8 for the split kernel, 12 for the fp32 one)
    t = t[:, :nw, :nit]
    t0 = t[..., 0].min()
    print(f"ending={ending} N={N} iterations/WG={nit}  kernel span (cycles) = {t[..., 3].max() - t0}")
    print("  first stamp offsets per WG: min %d max %d" % (t[:, 0, 0, 0].min() - t0, t[:, 0, 0, 0].max() - t0))
    p1 = t[..., 1] - t[..., 0]
    bw = t[..., 2] - t[..., 1]
    p2 = t[..., 3] - t[..., 2]
    it = t[:, :, 1:, 0] - t[:, :, :-1, 3] if nit > 1 else np.zeros(1)
    print("  waves/WG %d" % nw)
    print("  phase1 per wave:   mean %7.0f  max-over-waves mean %7.0f" % (p1.mean(), p1.max(axis=1).mean()))
    print("  barrier wait:      mean %7.0f" % bw.mean())
    print("  phase2 per wave:   mean %7.0f  max-over-waves mean %7.0f" % (p2.mean(), p2.max(axis=1).mean()))
    print("  top-of-loop wait:  mean %7.0f" % it.mean())
    wg = t[:, :, :, 3].max(axis=1) - t[:, :, :, 0].min(axis=1)
    print("  per-iteration WG span: mean %7.0f  min %d max %d ; phase1 span %7.0f phase2 span %7.0f" % (
        wg.mean(), wg.min(), wg.max(), (t[..., 2].max(axis=1) - t[..., 0].min(axis=1)).mean(),
        (t[..., 3].max(axis=1) - t[..., 2].min(axis=1)).mean()))
