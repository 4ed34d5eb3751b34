#!/usr/bin/env python
"""One LayerNorm-fused linear y = LN(x) W^T + b at a given shape through ops.gemm: the K-slab kernel + reduce launch (slab=True) or the tile kernels (gemm_h2 / gemm_ring)
(commit 6485dc3 also had 160 x 64 tiles walking all K slabs with the epilogue in place, gemm_slabfull_kernel: profiles/r05_gemm_head_variants.txt): HIP events back to back and the rel-L2 distance from float64.
usage: gemm_shape_bench.py M N K [M N K ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import ops  # noqa: E402

a = [int(v) for v in sys.argv[1:]] or [320, 8448, 512, 2560, 8448, 512, 2560, 2048, 512, 320, 2048, 512]
for M, N, K in zip(a[0::3], a[1::3], a[2::3]):
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.4).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    wsum = w.double().sum(1).float().contiguous()
    want = torch.nn.functional.layer_norm(x.double(), (K,)) @ w.double().t() + bias.double()
    for name, kw in (("K-slab kernels allowed", dict(slab=True, wsum=wsum)), ("tile kernels", dict())):
        y = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm(x, w, y, M, N, K, K, K, N, bias=bias, a_ln=True, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(x, w, y, M, N, K, K, K, N, bias=bias, a_ln=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        err = float((y.double() - want).norm() / want.norm())
        print(f"M={M:5d} N={N:5d} K={K:4d}  {name:28s} {e0.elapsed_time(e1) * 50:8.1f} us   rel-L2 vs float64 {err:.2e}", flush=True)
