"""Synthetic complexes and deterministic weights (no rdkit / ESM / checkpoints).

The batch dict reproduces the layout of the reference's ``collate_fn``
(reference ProteinReDiff/data.py:80-142): per sample the atom-keyed tensors
occupy ``[0:na]``, the residue-keyed tensors ``[na:na+nr]`` and the bond-keyed
tensors the ``[0:na, 0:na]`` corner; everything after ``na+nr`` is padding.

Weights are a pure function of ``(state_dict key order, shape, seed)`` so that
golden fixtures only have to store seeds and expected outputs, never weights.
"""
from __future__ import annotations

from typing import Dict, Mapping, Sequence, Tuple

import torch

from .constants import (ATOM_FEATURE_CARDS, BOND_FEATURE_CARDS,
                        NUM_RESIDUE_ATOMS, NUM_RESIDUE_CLASSES)


def synthetic_batch(sizes: Sequence[Tuple[int, int]], esm_dim: int = 1280,
                    seed: int = 0, n_total: int | None = None) -> Dict[str, torch.Tensor]:
    """Build a collated batch for ``len(sizes)`` complexes.

    sizes: per sample ``(num_atoms, num_residues)``.  ``n_total`` forces extra
    trailing padding (N defaults to ``max(na + nr)`` like data.py:81).
    """
    g = torch.Generator().manual_seed(seed)
    b = len(sizes)
    N = max(na + nr for na, nr in sizes)
    if n_total is not None:
        assert n_total >= N
        N = n_total

    def zeros(*shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype)

    batch = {
        "atom_feats": zeros(b, N, len(ATOM_FEATURE_CARDS), dtype=torch.long),
        "atom_mask": zeros(b, N),
        "atom_pos": zeros(b, N, 3),
        "bond_feats": zeros(b, N, N, len(BOND_FEATURE_CARDS), dtype=torch.long),
        "bond_mask": zeros(b, N, N),
        "bond_distance": zeros(b, N, N, dtype=torch.long),
        "residue_type": zeros(b, N, dtype=torch.long),
        "residue_mask": zeros(b, N),
        "residue_chain_index": zeros(b, N, dtype=torch.long),
        "residue_index": zeros(b, N, dtype=torch.long),
        "residue_atom_pos": zeros(b, N, NUM_RESIDUE_ATOMS, 3),
        "residue_atom_mask": zeros(b, N, NUM_RESIDUE_ATOMS),
        "residue_esm": zeros(b, N, esm_dim),
        "num_atoms": torch.tensor([na for na, _ in sizes], dtype=torch.long),
        "num_residues": torch.tensor([nr for _, nr in sizes], dtype=torch.long),
    }
    for k, (na, nr) in enumerate(sizes):
        a, r = slice(0, na), slice(na, na + nr)
        for f, card in enumerate(ATOM_FEATURE_CARDS):
            batch["atom_feats"][k, a, f] = torch.randint(0, card, (na,), generator=g)
        batch["atom_mask"][k, a] = 1.0
        batch["atom_pos"][k, a] = 5.0 * torch.randn(na, 3, generator=g)
        upper = (torch.rand(na, na, generator=g) < 0.05).float().triu(1)
        bmask = upper + upper.T
        batch["bond_mask"][k, a, a] = bmask
        for f, card in enumerate(BOND_FEATURE_CARDS):
            bf = torch.randint(0, card, (na, na), generator=g).triu(1)
            batch["bond_feats"][k, a, a, f] = ((bf + bf.T) * bmask.long())
        bd = torch.randint(0, 12, (na, na), generator=g).triu(1)
        batch["bond_distance"][k, a, a] = bd + bd.T
        batch["residue_type"][k, r] = torch.randint(1, NUM_RESIDUE_CLASSES, (nr,), generator=g)
        batch["residue_mask"][k, r] = 1.0
        batch["residue_index"][k, r] = torch.arange(nr)
        # two chains when the protein is long enough, to exercise the chain mask
        if nr >= 8:
            batch["residue_chain_index"][k, na + nr // 2: na + nr] = 1
        batch["residue_atom_pos"][k, r] = 10.0 * torch.randn(nr, NUM_RESIDUE_ATOMS, 3, generator=g)
        batch["residue_atom_mask"][k, r] = 1.0
        batch["residue_esm"][k, r] = torch.randn(nr, esm_dim, generator=g)
    return batch


def synthetic_sample(na: int, nr: int, esm_dim: int = 1280, seed: int = 0) -> Dict[str, torch.Tensor]:
    """One UN-collated complex: the union of the reference's ``ligand_to_data`` and ``protein_to_data`` dicts
    (data.py:28-77) without the rdkit molecule objects; ``residue_type`` is the raw aatype (collate adds 1)."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.stack([torch.randint(0, c, (na,), generator=g) for c in ATOM_FEATURE_CARDS], dim=-1)
    upper = (torch.rand(na, na, generator=g) < 0.05).float().triu(1)
    bmask = upper + upper.T
    bfeats = torch.stack([torch.randint(0, c, (na, na), generator=g).triu(1) for c in BOND_FEATURE_CARDS], dim=-1)
    bfeats = (bfeats + bfeats.transpose(0, 1)) * bmask.long().unsqueeze(-1)
    bd = torch.randint(0, 12, (na, na), generator=g).triu(1)
    return {
        "num_atoms": na, "atom_feats": feats, "atom_mask": torch.ones(na), "atom_pos": 5.0 * torch.randn(na, 3, generator=g),
        "bond_feats": bfeats, "bond_mask": bmask, "bond_distance": bd + bd.T,
        "num_residues": nr, "residue_type": torch.randint(0, NUM_RESIDUE_CLASSES - 1, (nr,), generator=g),
        "residue_mask": torch.ones(nr), "residue_chain_index": (torch.arange(nr) >= (nr + 1) // 2).long(),
        "residue_index": torch.arange(nr), "residue_atom_pos": 10.0 * torch.randn(nr, NUM_RESIDUE_ATOMS, 3, generator=g),
        "residue_atom_mask": torch.ones(nr, NUM_RESIDUE_ATOMS), "residue_esm": torch.randn(nr, esm_dim, generator=g),
    }


def clone_batch(batch: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}


def batch_to(batch: Mapping[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


# ---------------------------------------------------------------------------
# deterministic weights
# ---------------------------------------------------------------------------

_FROZEN = ("embed_beta.0.weight", "embed_dist.0.center")


# layers the reference initialises to ZERO ("final": modules.py:160-163, AF2_modules.py:151-153) or as pass-through gates
# ("gating": weight 0, bias 1): every residual update starts as the identity and the heads start at zero
_FINAL_INIT = ("out_proj.weight", "out_proj.bias", "single_fc.3.", "pair_fc.3.", "outer_linear.linear.", "linear_o.", "linear_out.",
               "weight_radial.3.", "seq_mlp.3.")
_GATING_INIT = ("gate_proj.", "ab_gate.", "out_gate.", "linear_g.")


def deterministic_state_dict(spec: Mapping[str, torch.Tensor], seed: int = 1, style: str = "random",
                             scales: Mapping[str, float] | None = None) -> Dict[str, torch.Tensor]:
    """Seeded weights for every key of ``spec`` (a state_dict used for names/shapes).

    ``style="random"``: matrices ~ N(0, 1/fan_in), biases ~ N(0, 0.1^2), LayerNorm scales 1 + N(0, 0.1^2), embedding
    tables ~ N(0, 1).  Nothing is left at the reference's zero ("final") initialisation, so every branch of the network is
    live at full strength -- the most demanding inputs for operator / step parity.  As a DENOISER such a network is a random
    map: iterated over hundreds of reverse-diffusion steps it amplifies round-off chaotically.

    ``style="near_init"``: the same, except that the layers the reference initialises to zero ("final") or to pass-through
    gates ("gating") sit an N(0, 0.02^2) perturbation away from that initialisation -- a network early in training, the
    recipe of SURVEY.md §8d.  Every branch is still live and residual updates are small.  Whether the reverse-diffusion LOOP
    built on such a network stays in the network's working range is a separate question, decided by the coordinate head: it is
    well conditioned at N = 140 (|z| contracts) and diverges at N = 320 (the head's update grows with the number of pairs) --
    see ``scales``.

    ``scales``: ``{key suffix: factor}`` applied to the generated tensors whose name ends with the suffix (after everything
    else; the random stream is unchanged).  The long-trajectory fixtures of the headline shape use it to put the coordinate
    head (``weight_radial.3.weight``) where the loop keeps the pair distances inside the support of the distance embedding
    (tools/conditioning_scan.py chooses the factor with the HIP path; oracle/gen_yardstick.py generates the fixtures with the
    imported reference).

    The two frozen buffers-as-parameters keep the values the reference constructs (modules.py:77-79, 91-93)."""
    if style not in ("random", "near_init"):
        raise ValueError(f"unknown weight style: {style}")
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name in sorted(spec):              # sorted: independent of module registration order
        ref = spec[name]
        shape = tuple(ref.shape)
        if name in _FROZEN:
            out[name] = ref.detach().clone().float()
            continue
        if name.endswith("bias"):
            w = 0.1 * torch.randn(shape, generator=g)
        elif ".embeddings." in name or name in ("embed_bond_distance.weight", "embed_relpos.weight"):
            w = torch.randn(shape, generator=g)
        elif len(shape) == 1:              # LayerNorm weight
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = shape[-1]
            w = torch.randn(shape, generator=g) / (fan_in ** 0.5)
        if style == "near_init":
            if any(t in name for t in _FINAL_INIT):
                w = 0.02 * torch.randn(shape, generator=g)
            elif any(t in name for t in _GATING_INIT):
                w = 0.02 * torch.randn(shape, generator=g) + (1.0 if name.endswith("bias") else 0.0)
        for suffix, factor in (scales or {}).items():
            if name.endswith(suffix):
                w = w * float(factor)
        out[name] = w.float()
    return out


_torch_randperm, _torch_randn = torch.randperm, torch.randn   # bound early: harnesses may patch torch.*


class NoiseSource:
    """CPU fp32 noise keyed by ``(seed, global sample index)``.

    RNG is device specific (torch.randn_like on CPU vs GPU), so parity runs and
    shard-invariant multi-GPU sampling draw every random tensor from here, in the
    order the reference's ``sample`` consumes them (model.py:399-418,
    mask_utils.py:87): permutation for the redesign mask, z_T, seq_T, then one
    ``[N,3]`` tensor per step with t > 0.
    """

    def __init__(self, seed: int, sample_index: int):
        self.g = torch.Generator().manual_seed((seed * 1_000_003 + sample_index) & 0x7FFFFFFF)

    def randperm(self, n: int) -> torch.Tensor:
        return _torch_randperm(n, generator=self.g)

    def randn(self, *shape) -> torch.Tensor:
        return _torch_randn(*shape, generator=self.g)
