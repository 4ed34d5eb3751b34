"""Run by tests/test_ab_build.py in a child process with PRD_LIB = libprd_hip_ab.so: the superseded forms of the second-generation attention
cores that only the A/B library carries -- key-loop forms 1-3 and the round-3 phase 1 of the short-row core, the round-3 phase 1 and the
next-row prefetch of the long-row core -- against the default form (same arithmetic, other summation orders: < 2e-6)."""
import sys

import torch

from protein_redesign_amd import _lib, ops

lib = _lib.lib()
assert lib.prd_set_gemm_mode(1) == 0
tune0 = lib.prd_get_tune()
H, c, P = 4, 16, 64
worst = 0.0
for N, arms in ((320, {"NO_GV": 1 << 21, "KL1": (1 << 21) | (1 << 6) | (2 << 7), "KL2": (1 << 21) | (1 << 6) | (4 << 7), "KL3": (1 << 21) | (1 << 6) | (6 << 7),
                       "prio+remap": (1 << 6) | (17 << 7)}),
                (449, {"NO_GV": 1 << 21, "prefetch": (1 << 6) | (9 << 7)})):
    g = torch.Generator().manual_seed(N)
    pair = torch.randn(1, N, N, P, generator=g).cuda()
    mask = torch.ones(1, N).cuda()
    mask[0, N - 11:] = 0
    wts = [(torch.randn(64, P, generator=g) / 8).cuda() for _ in range(4)] + [torch.zeros(64).cuda()]
    for ending in (False, True):
        lib.prd_set_tune(tune0)
        ref = ops.tri_attn_core_v2(pair, mask, wts, H, c, ending=ending).clone()
        for name, bits in arms.items():
            lib.prd_set_tune(tune0 | bits)
            got = ops.tri_attn_core_v2(pair, mask, wts, H, c, ending=ending)
            err = float((got - ref).norm() / ref.norm())
            worst = max(worst, err)
            assert torch.isfinite(got).all() and err < 2e-6, (N, ending, name, err)
lib.prd_set_tune(tune0)
print(f"ab forms ok, worst rel-L2 vs the default form {worst:.2e}")
sys.exit(0)
