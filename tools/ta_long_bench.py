#!/usr/bin/env python
"""Triangle attention core alone at long-row sizes (HIP events, back to back):  python tools/ta_long_bench.py [N ...]
Environment (read once per process by the library): PRD_TA2_LONG=0 keeps the first-generation long-row kernel,
PRD_TA2_FLAGS=9 adds the next-row prefetch to the round-3 kernel.  Every size is measured with the default dispatch and with the
tail-row split switched off (PRD_TUNE_TA2_NO_TAIL_SPLIT) and with the round-3 phase 1 (PRD_TUNE_TA2_NO_GV: a [K|Q] GEMM + a swapped V GEMM
instead of one [K|V] GEMM + transposed store) and with a ragged last key tile of <= 4 keys swept as a regular tile instead of
rank-1 updates (round 6), alternating, three rounds each (median)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import _lib, ops  # noqa: E402

P, H, c = 64, 4, 16
for N in [int(v) for v in sys.argv[1:]] or [449, 640, 769, 832]:
    g = torch.Generator().manual_seed(N)
    pair = torch.randn(1, N, N, P, generator=g).cuda()
    mask = torch.ones(1, N).cuda()
    wts = [(torch.randn(64, P, generator=g) / 8).cuda() for _ in range(4)] + [torch.zeros(64).cuda()]
    og = torch.empty(1, N, N, 64, device="cuda")
    lib = _lib.lib()
    tune0 = lib.prd_get_tune()
    for ending in (False, True):
        res = {}
        for rnd in range(3):
            for name, tune in (("default", tune0), ("no tail split", tune0 | (1 << 19)), ("round-3 phase 1 (A/B library only)", tune0 | (1 << 21)),
                               ("tail as a tile", tune0 | (1 << 6) | (3 << 7)),
                               ("r5 shared round", tune0 | (1 << 6) | (5 << 7))):       # PRD_TA2_FLAGS=5: shared last round projected / merged inside phase 2        # PRD_TA2_FLAGS=3: ragged last key tile swept as a 32-key tile (round 5)
                lib.prd_set_tune(tune)
                for _ in range(2):
                    ops.tri_attn_core(pair, mask, wts, H, c, ending=ending, og=og)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.tri_attn_core(pair, mask, wts, H, c, ending=ending, og=og)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) * 100)
        lib.prd_set_tune(tune0)
        gf = (8 * N * N * P * 64 + 4 * 64 * N ** 3) / 1e9
        for name, v in res.items():
            us = sorted(v)[1]
            print(f"N={N:4d} ending={int(ending)}  {name:16s} {us:8.1f} us  {gf / us * 1e3:6.1f} TF/s algorithmic  v2={ops.tri_attn_v2_supported(N, P)}", flush=True)
