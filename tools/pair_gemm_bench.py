#!/usr/bin/env python
"""Times the pair-position GEMMs of the training backward (204800 rows) per tile hint: tools/pair_gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from protein_redesign_amd import ops  # noqa: E402

M = 2 * 320 * 320
g = torch.Generator().manual_seed(0)


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, N, ln in ((64, 64, False), (64, 256, False), (64, 256, True), (256, 64, False)):
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    out = torch.empty(M, N, device="cuda")
    mb = (M * K + M * N) * 4 / 1e6
    line = f"K={K:3d} N={N:3d} ln={int(ln)} ({mb:.0f} MB, {mb / 8e6 * 1e6:.0f} us at 8 TB/s):"
    for th in (0, 32, 64, 128):
        try:
            t = bench(lambda: ops.gemm(x, w, out, M, N, K, K, K, N, a_ln=ln, tile_hint=th))
            line += f"  hint {th}: {t:6.1f} us"
        except Exception as e:  # noqa: BLE001
            line += f"  hint {th}: {type(e).__name__}"
    print(line)

print("row kernel (prd_pair_linear):")
for K, N, ln in ((64, 64, False), (64, 256, False), (64, 256, True), (256, 64, False)):
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    t = bench(lambda: ops.pair_linear(x, w, ln_in=ln))
    print(f"K={K:3d} N={N:3d} ln={int(ln)}: {t:6.1f} us")
x = torch.randn(M, 256, generator=g).cuda()
w = (torch.randn(64, 256, generator=g) / 16).cuda()
dy = torch.randn(M, 64, generator=g).cuda()
h = torch.randn(M, 256, generator=g).cuda()
w2 = (torch.randn(256, 64, generator=g) / 8).cuda()
print(f"K= 64 N=256 + ReLU mask: {bench(lambda: ops.pair_linear(dy, w2, relu_mask=h)):6.1f} us")
