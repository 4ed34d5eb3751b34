#!/usr/bin/env python
"""Run-to-run bit-reproducibility of the attention cores: every kernel here is deterministic by construction (static task order, no
atomics in the data path), so two launches on the same inputs must agree bit for bit.  A mismatch is a race in the kernel or a faulty
box; used to tell the two apart when a parity test fails on one box only.
    python tools/determinism_stress.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import _lib, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
H, c = 4, 16
lib = _lib.lib()
prev = lib.prd_get_gemm_mode()
cases = [("fp32", 449, 32, False), ("fp32", 449, 64, False), ("fp32", 385, 64, True), ("split16", 1961, 64, True), ("split16", 449, 64, False),
         ("split16", 320, 64, False), ("fp32", 320, 64, False), ("split16", 769, 64, True)]
for mode, N, P, ending in cases:
    lib.prd_set_gemm_mode(_lib.GEMM_MODES[mode])
    g = torch.Generator().manual_seed(N + P)
    pair = torch.randn(1, N, N, P, generator=g).cuda()
    mask = torch.ones(1, N).cuda()
    mask[0, N - 18:] = 0
    wts = [(torch.randn(64, P, generator=g) / 8).cuda() for _ in range(4)] + [torch.zeros(64).cuda()]
    n = reps if N < 1000 else max(reps // 10, 3)
    ref = ops.tri_attn_core(pair, mask, wts, H, c, ending=ending).clone()
    bad, worst = 0, 0.0
    for i in range(n):
        out = ops.tri_attn_core(pair, mask, wts, H, c, ending=ending)
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, float((out - ref).norm() / ref.norm()))
    print(f"{mode:8s} N={N:5d} P={P} ending={int(ending)} variant={ops.tri_attn_variant(N, P)}: {bad} of {n} launches differ from the first"
          + (f" (worst rel-L2 {worst:.2e})" if bad else ""), flush=True)
lib.prd_set_gemm_mode(prev)

# Gaps: a position of og / the pair output that a kernel never writes keeps whatever the allocator hands out (another test's data in
# a long pytest session, zeros in a fresh process).  Fill every output and scratch buffer with NaN before the launch.
print("--- NaN-filled outputs ---")
for mode, N, P, ending in cases:
    lib.prd_set_gemm_mode(_lib.GEMM_MODES[mode])
    g = torch.Generator().manual_seed(N + P)
    pair = torch.randn(1, N, N, P, generator=g).cuda()
    mask = torch.ones(1, N).cuda()
    mask[0, N - 18:] = 0
    wts = [(torch.randn(64, P, generator=g) / 8).cuda() for _ in range(4)] + [torch.zeros(64).cuda()]
    wo, bo = (torch.randn(P, 64, generator=g) / 8).cuda(), torch.zeros(P).cuda()
    og = torch.full((1, N, N, 64), float("nan"), device="cuda")
    nst = ops.tri_attn_stats_floats(1, N, P, H)
    stats = torch.full((nst,), float("nan"), device="cuda") if nst else None
    ops.tri_attn_core(pair, mask, wts, H, c, ending=ending, og=og, stats=stats)
    bad_core = int((~torch.isfinite(og)).sum())
    out = torch.full_like(pair, float("nan"))
    ws = torch.full((ops.workspace_bytes("tri_attn", 1, N, 0, P) // 4,), float("nan"), device="cuda")
    ops.tri_attn(pair, mask, (*wts, wo, bo), H, c, ending=ending, residual=False, out=out, ws=ws)
    bad_full = int((~torch.isfinite(out)).sum())
    print(f"{mode:8s} N={N:5d} P={P} ending={int(ending)}: non-finite after core {bad_core}, after core + out-projection {bad_full}", flush=True)
lib.prd_set_gemm_mode(prev)
