#!/usr/bin/env python
"""The triangle-attention pair of a folding block (modules.py:338-339: starting attention in place, ending core) as three launches against the
ONE persistent launch prd_tri_attn_pair, HIP events, arms alternating, median of the rounds; results compared bit for bit.
    python tools/persistent_attn_bench.py [N] [b]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import _lib, ops  # noqa: E402
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 320
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda")
model, _, _ = bench.build_model(dev, graph=False)
blk = model.Denoiser.folding_blocks[0]
ts, te = blk.pair_attn_starting.attn, blk.pair_attn_ending.attn
g = torch.Generator().manual_seed(0)
pair0 = torch.randn(b, N, N, 64, generator=g).to(dev)
mask = torch.ones(b, N, device=dev)
og = torch.empty(b, N, N, 64, device=dev)
ws = torch.empty(ops.workspace_bytes("tri_attn", b, N, 0, 64) // 4, device=dev)
lib = _lib.lib()
tune0 = lib.prd_get_tune()


def three(pair):
    blk.pair_attn_starting.run(pair, mask, residual=True, out=pair, ws=ws)
    return ops.tri_attn_core(pair, mask, te.weights()[:5], 4, 16, ending=True, og=og)


def one(pair):
    return ops.tri_attn_pair_(pair, mask, ts.weights(), te.weights()[:5], 4, 16, og=og)


arms = [("three launches", three, tune0), ("one persistent launch, XCD-hierarchical barrier", one, tune0),
        ("one persistent launch, plain counter barrier", one, tune0 | (1 << 19))]
res, outs = {}, {}
for rnd in range(5):
    for name, fn, tune in arms:
        lib.prd_set_tune(tune)
        pair = pair0.clone()
        for _ in range(2):
            fn(pair)
        if rnd == 0:
            p = pair0.clone()
            o = fn(p).clone()
            outs[name] = (p, o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(pair)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
lib.prd_set_tune(tune0)
assert not ops.tri_attn_pair_timed_out(dev)
for name, v in res.items():
    same = all(torch.equal(x, y) for x, y in zip(outs[name], outs["three launches"]))
    print(f"N={N} b={b}  {name:<50s} {sorted(v)[len(v) // 2]:7.1f} us  (min {min(v):.1f})   {'bit-identical' if same else 'DIFFERS'}")

