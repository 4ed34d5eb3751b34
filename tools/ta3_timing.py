#!/usr/bin/env python
"""Phase timing of tri_attn_core_v3_kernel (csrc/prd_tri2.hip) from in-kernel cycle stamps.  Needs the diagnostic library:
    python -m protein_redesign_amd.build --timing ; PRD_LIB=protein_redesign_amd/libprd_hip_timing.so python tools/ta3_timing.py [N]
Stamps per (workgroup, wave, row iteration): 5 arrival at the barrier, 0 released, 1 merge of the previous row done, 2 key loops
done, 3 next row loaded + LayerNorm-ed + split, 4 projection of the next row done."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 320
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
pair = torch.randn(1, N, N, 64, generator=g).to(dev)
mask = torch.ones(1, N, device=dev)
model, _, _ = bench.build_model(dev, graph=False)
wts = model.Denoiser.folding_blocks[0].pair_attn_starting.attn.weights()[:5]
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
    t = full[:, :, :nit]
    t0 = t[..., 5].min()
    print(f"ending={ending} N={N} rows/WG={nit}  kernel span (cycles) = {t[..., 4].max() - t0}")
    segs = [("barrier wait (5 -> 0)", 5, 0), ("merge (0 -> 1)", 0, 1), ("key loops (1 -> 2)", 1, 2), ("row load + LN + split (2 -> 3)", 2, 3),
            ("projection GEMMs + stores (3 -> 4)", 3, 4)]
    for nm, a, b in segs:
        d = (t[..., b] - t[..., a])
        if nm.startswith("row load") or nm.startswith("projection"):
            d = d[:, :, :-1]                    # the last iteration projects nothing
        print(f"  {nm:36s} per wave mean {d.mean():8.0f}   by wave: " + " ".join(f"{d[:, w].mean():.0f}" for w in range(12)))
    it_span = t[:, :, 1:, 5] - t[:, :, :-1, 5]
    print(f"  iteration (barrier arrival to barrier arrival): mean {it_span.mean():.0f}  by wave: " + " ".join(f"{it_span[:, w].mean():.0f}" for w in range(12)))
    rel = t[:, :, :, 0].max(axis=1) - t[:, :, :, 5].min(axis=1)
    print(f"  first arrival -> release per iteration: mean {rel.mean():.0f}")
