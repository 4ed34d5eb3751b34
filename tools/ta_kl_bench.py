#!/usr/bin/env python
"""Same-box A/B of the key-loop forms of tri_attn_core_v3_kernel (PRD_TA2_FLAGS bits 1-2: bit 1 = the next tile's Q K^T issued before
the split of this one, bit 2 = row sum on the matrix pipe) and of tri_attn_core_v2 for reference: average launch time with HIP events
(back-to-back launches, starting / ending alternating) and the rel-L2 distance of the result from form 0.
usage: PRD_LIB=protein_redesign_amd/libprd_hip_ab.so ta_kl_bench.py [--N 320] [--b 1] [--reps 40]
(since round 6 the forms other than the default exist in the -DPRD_AB library only: python -m protein_redesign_amd.build --ab; the shipped
library ignores the switches and runs the default form in every arm)"""
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
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    from protein_redesign_amd import _lib, ops
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    P, H, c = 64, 4, 16
    pair = torch.randn(a.b, a.N, a.N, P, generator=g).to(dev)
    mask = torch.ones(a.b, a.N, device=dev)
    # the weights of the bench model (bench.py: deterministic_state_dict seed 1): logits of realistic size -- with O(1) random
    # projections many rows leave the frozen-reference range and take the online redo, which is not what a step runs
    sys.path.insert(0, ROOT)
    import bench
    model, _, _ = bench.build_model(torch.device(dev), graph=False)
    wts = model.Denoiser.folding_blocks[0].pair_attn_starting.attn.weights()[:5]
    og = torch.empty(a.b, a.N, a.N, 64, device=dev)
    base_tune = _lib.lib().prd_get_tune()
    ref = None
    forms = [("v3 default: [G|V] as one GEMM, transposed V store", "gv"), ("v3 form 0 (round-3 phase 1 and order)", 0), ("v3 form 1 (next Q K^T before the split)", 2), ("v3 form 2 (row sum by mfma_4x4x4)", 4),
             ("v3 + key-loop priorities by remaining work", 1), ("v3 + static priority, youngest first", 8),
             ("v3 + shared blocks projected by the oldest waves", 16), ("v3 + oldest-wave projection + static priority", 24),
             ("v3 + oldest-wave projection + mfma row sum", 20), ("v2 (barrier per phase, priorities)", None)]
    times = {name: [] for name, _ in forms}
    errs = {}
    for rnd in range(a.rounds):             # interleaved rounds: the clock of a box drifts; the median over rounds is reported
        for name, f in forms:
            tune = base_tune & ~((1 << 6) | (31 << 7) | (1 << 4) | (1 << 21))
            if f is None:
                tune |= 1 << 4              # PRD_TUNE_TA2_NO_V3
            elif f == "gv":
                pass                        # default dispatch
            else:
                tune |= 1 << 21             # PRD_TUNE_TA2_NO_GV: the round-3 phase 1 (the A/B forms exist with it only)
                tune |= (1 << 6) | ((f & 31) << 7)
            _lib.lib().prd_set_tune(tune)
            for i in range(4):
                ops.tri_attn_core_v2(pair, mask, wts, H, c, ending=bool(i & 1), og=og)
            outs = [ops.tri_attn_core_v2(pair, mask, wts, H, c, ending=e).clone() for e in (False, True)]
            if f == 0 and ref is None:
                ref = outs
            errs[name] = max(float((o - r_).norm() / r_.norm()) for o, r_ in zip(outs, ref)) if ref is not None else float('nan')
            assert all(bool(torch.isfinite(o).all()) for o in outs), name
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.reps):
                ops.tri_attn_core_v2(pair, mask, wts, H, c, ending=bool(i & 1), og=og)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 1e3 / a.reps)
    for name, _ in forms:
        ts = sorted(times[name][1:]) if len(times[name]) > 1 else times[name]        # the first round warms the clock up
        print(f"N={a.N} b={a.b}  {name:<52s} median {ts[len(ts) // 2]:7.2f} us  min {ts[0]:7.2f}  max {ts[-1]:7.2f}   rel-L2 vs form 0 {errs[name]:.2e}", flush=True)
    _lib.lib().prd_set_tune(base_tune)


if __name__ == "__main__":
    main()
