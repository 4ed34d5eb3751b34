#!/usr/bin/env python
"""Same-box A/B of tri_attn_core_v3_kernel at N = 320: arms alternate inside one process (the clock of a box drifts), median of the rounds;
the results of all arms must agree bit for bit where the arm only re-deals work.
usage: ta_v3_ab.py [--N 320] [--b 1] [--reps 40] [--rounds 7] [--arms name=tune_xor_bits,...]   (default arms: round-6 switches)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=320)
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--arms", default="default=0,fixed helper pieces (PRD_TA2_TAIL=0)=524288")
    a = ap.parse_args()
    from protein_redesign_amd import _lib, ops
    import bench
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    P, H, c = 64, 4, 16
    pair = torch.randn(a.b, a.N, a.N, P, generator=g).to(dev)
    mask = torch.ones(a.b, a.N, device=dev)
    model, _, _ = bench.build_model(torch.device(dev), graph=False)
    wts = model.Denoiser.folding_blocks[0].pair_attn_starting.attn.weights()[:5]
    og = torch.empty(a.b, a.N, a.N, 64, device=dev)
    lib = _lib.lib()
    base = lib.prd_get_tune()
    arms = [(s.rsplit("=", 1)[0], int(s.rsplit("=", 1)[1])) for s in a.arms.split(",")]
    times = {n: [] for n, _ in arms}
    outs = {}
    for rnd in range(a.rounds):
        for name, bits in arms:
            lib.prd_set_tune(base ^ bits)
            for i in range(4):
                ops.tri_attn_core_v2(pair, mask, wts, H, c, ending=bool(i & 1), og=og)
            if rnd == 0:
                outs[name] = [ops.tri_attn_core_v2(pair, mask, wts, H, c, ending=e).clone() for e in (False, True)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.reps):
                ops.tri_attn_core_v2(pair, mask, wts, H, c, ending=bool(i & 1), og=og)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    lib.prd_set_tune(base)
    first = arms[0][0]
    for name, _ in arms:
        ts = sorted(times[name][1:])
        same = all(torch.equal(x, y) for x, y in zip(outs[name], outs[first]))
        rel = max(float((x - y).norm() / y.norm()) for x, y in zip(outs[name], outs[first]))
        print(f"N={a.N} b={a.b}  {name:<44s} median {ts[len(ts) // 2]:7.2f} us  min {ts[0]:7.2f}  max {ts[-1]:7.2f}   "
              f"{'bit-identical to' if same else f'rel-L2 {rel:.1e} vs'} '{first}'", flush=True)


if __name__ == "__main__":
    main()
