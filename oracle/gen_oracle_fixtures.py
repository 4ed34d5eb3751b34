"""Outputs of the CPU oracle (oracle/prd_oracle.py) for the LARGE single-step parity cases, stored as fixtures so that the GPU
suite does not spend minutes of host time re-evaluating them on every run (test infrastructure; build container or any CPU box):

    python oracle/gen_oracle_fixtures.py [case ...]          # -> tests/golden/oracle_steps.npz

Each case is a pure function of seeds (weights, complex, redesign mask, z, seq_t -- `full_size_inputs`, shared with
tests/test_hip_parity.py), the fixture holds the oracle's (noise_pred [1,N,3], seq_pred [1,N,21]) per case.  The oracle itself is
pinned to the imported reference by tests/test_oracle_golden.py; tests/test_oracle_golden.py::test_stored_oracle_steps_match_a_live_run
re-evaluates the smallest stored case on every CPU run.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import prd_oracle as O  # noqa: E402

from protein_redesign_amd.constants import make_args  # noqa: E402
from protein_redesign_amd.synthetic import NoiseSource, deterministic_state_dict, synthetic_batch  # noqa: E402
from protein_redesign_amd.weights import spec_tensors  # noqa: E402

NOISE_SEED = 7
PATH = os.path.join(ROOT, "tests", "golden", "oracle_steps.npz")
# key -> (num_atoms, num_residues, num_blocks, weight seed)
CASES = {
    "n320_b1_s4": (64, 256, 1, 4),          # BASELINE configs[1] shape, one block
    "n320_b4_s4": (64, 256, 4, 4),          # ... and the network the bench replays
    "n769_b1_s7": (1, 768, 1, 7),           # BASELINE configs[4]: long rows
    "n1024_b1_s7": (24, 1000, 1, 7),        # rows beyond the LDS (key-chunked attention in fp32 arithmetic)
    "n769_b4_s11": (1, 768, 4, 11),         # configs[4] at its own depth
}


def full_size_inputs(na, nr, num_blocks, seed):
    """(args, params, prepared batch, z, seq_t, t) of a single-step case at single_dim 512 / pair_dim 64 -- CPU tensors only."""
    args = make_args(single_dim=512, pair_dim=64, num_blocks=num_blocks, num_steps=1000, mask_prob=0.3)
    params = deterministic_state_dict(spec_tensors(args), seed=seed)
    N = na + nr
    batch = synthetic_batch([(na, nr)], seed=0)
    pb = O.prepare_batch(batch, 0.3, [NoiseSource(NOISE_SEED, 0).randperm(nr)])
    g = torch.Generator().manual_seed(8)
    z, seq_t, t = torch.randn(1, N, 3, generator=g), torch.randn(1, N, 21, generator=g), torch.tensor([500])
    return args, params, pb, z, seq_t, t


def oracle_step(key):
    na, nr, nb, seed = CASES[key]
    args, params, pb, z, seq_t, t = full_size_inputs(na, nr, nb, seed)
    with torch.inference_mode():
        eps, logits = O.network_step(params, args, pb, z, seq_t, pb["residue_and_atom_mask"], t)
    return eps.numpy(), logits.numpy()


def load():
    return dict(np.load(PATH)) if os.path.exists(PATH) else {}


def main():
    import time
    names = sys.argv[1:] or list(CASES)
    out = load()
    for key in names:
        t0 = time.time()
        eps, logits = oracle_step(key)
        out[key + "_eps"], out[key + "_logits"] = eps, logits
        np.savez_compressed(PATH, **out)
        print(f"{key}: {time.time() - t0:.0f} s -> {PATH}", flush=True)


if __name__ == "__main__":
    main()
