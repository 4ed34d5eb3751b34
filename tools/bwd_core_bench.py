#!/usr/bin/env python
"""Times the triangle-attention backward cores (fp32 MFMA vs split-16) on one shape: tools/bwd_core_bench.py [b N]."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from protein_redesign_amd import ops
from protein_redesign_amd._lib import check, dptr, lib, stream

b, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 320)
P, H, c = 64, 4, 16
g = torch.Generator().manual_seed(0)
pair = torch.randn(b, N, N, P, generator=g).cuda()
mask = torch.ones(b, N).cuda()
wq, wk, wv, wg = [(torch.randn(64, P, generator=g) / math.sqrt(P)).cuda() for _ in range(4)]
bg = torch.zeros(64).cuda()
dog = (torch.randn(b, N, N, 64, generator=g) * 1e-3).cuda()
lse = torch.empty(b * N, H, N, 2, device="cuda")
og = ops.tri_attn_core_v2_lse(pair, mask, (wq, wk, wv, wg, bg), H, c, ending=False, lse=lse)
out = torch.empty(b, N, N, 4, 64, device="cuda")


def v2(stats=None, ending=0):
    check(lib().prd_tri_attn_bwd_core_v2(dptr(out), dptr(dog), dptr(og), dptr(pair), dptr(mask), dptr(wq), dptr(wk), dptr(wv), dptr(wg), dptr(bg),
                                         dptr(stats) if stats is not None else None, None, ending, b, N, P, H, c, stream()), "v2")


def v2s():
    v2(lse)




def v1():
    check(lib().prd_tri_attn_bwd_core(dptr(out), dptr(dog), dptr(pair), dptr(mask), dptr(wq), dptr(wk), dptr(wv), dptr(wg), dptr(bg),
                                      0, b, N, P, H, c, stream()), "v1")


res = {}
outs = {}
for rnd in range(3):                # arms alternate: the clock of a box drifts
    for name, fn in (("fp32", v1), ("split16", v2), ("split16 + kept statistics", v2s)):
        for _ in range(3):
            fn()
        if rnd == 0:
            outs[name] = out.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 10 * 1e3)
for name, v in res.items():
    print(f"{name}: {sorted(v)[1]:.1f} us  (b={b}, N={N}; median of 3 alternating rounds)")
a_ = outs["split16 + kept statistics"]
print(f"split16 + kept statistics vs fp32 core: rel-L2 {float((a_ - outs['fp32']).norm() / outs['fp32'].norm()):.2e}")
