#!/usr/bin/env python
"""SPAttention's core at the head of the trunk: the one-launch form (prd_spa_attn_core) against the GEMM-path form (logits GEMM +
softmax riding in the P V GEMM) on the same inputs, HIP-event timing.  usage: spa_bench.py [--N 320 --b 1 --c 512]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=320)
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--c", type=int, default=512)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--qscale", type=float, default=1.0, help="logit spread: q is scaled by this (x 1 / sqrt(c))")
    a = ap.parse_args()
    from protein_redesign_amd import ops
    dev, H = "cuda", 4
    HC = H * a.c
    g = torch.Generator().manual_seed(0)
    x = torch.randn(a.b, a.N, a.c, generator=g).to(dev)
    qkvg = torch.randn(a.b, a.N, 4 * HC, generator=g).to(dev)
    qkvg[..., :HC] *= a.qscale / a.c ** 0.5
    qkvg[..., 3 * HC:] = torch.sigmoid(qkvg[..., 3 * HC:])
    bias = torch.randn(a.b, H, a.N, a.N, generator=g).to(dev)
    mask = torch.ones(a.b, a.N, device=dev)
    wo, bo = (torch.randn(a.c, HC, generator=g) / HC ** 0.5).to(dev), torch.zeros(a.c, device=dev)
    packed = (None, None, None)
    res = {}
    for name, flag in (("one launch (prd_spa_attn_core)", True), ("logits GEMM + softmax | P V GEMM", False)):
        ops.SPA_CORE = flag
        run = lambda: ops.gated_attention_single(x, mask, bias, packed, wo, bo, H, a.c, key_mask=False, resid=None, qkvg=qkvg, logits_fp32=False)  # noqa: E731
        for _ in range(3):
            out = run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = out
        print(f"N={a.N} b={a.b} c={a.c}  {name:<36s} {e0.elapsed_time(e1) * 1e3 / a.reps:8.2f} us per call (core + output projection)")
    vals = list(res.values())
    print("rel-L2 between the two forms:", float((vals[0] - vals[1]).norm() / vals[1].norm()))
    # both against float64 (CPU)
    q, k, v, gt = [t.double().cpu().view(a.b, a.N, H, a.c).transpose(1, 2) for t in qkvg.split(HC, dim=-1)]
    att = torch.softmax(q @ k.transpose(-1, -2) + bias.double().cpu(), dim=-1) @ v
    want = (gt * att).transpose(1, 2).reshape(a.b, a.N, HC) @ wo.double().cpu().t() + bo.double().cpu()
    for name, out in res.items():
        err = (out.double().cpu() - want)
        print(f"   {name:<36s} vs float64: rel-L2 {float(err.norm() / want.norm()):.2e}, worst row {float((err.norm(dim=-1) / want.norm(dim=-1)).max()):.2e}")


if __name__ == "__main__":
    main()
