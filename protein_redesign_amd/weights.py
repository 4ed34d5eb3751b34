"""Parameter inventory of the hot path (names, shapes, frozen values).

Key names and shapes are the reference's ``state_dict`` (reference
ProteinReDiff/model.py:82-122, modules.py:300-326, 366-386, models/AF2_modules.py:403-419, 498-501)
so a reference checkpoint's ``state_dict`` loads unchanged.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping

import torch

from .constants import ATOM_FEATURE_CARDS, BOND_FEATURE_CARDS, NUM_RESIDUE_CLASSES


def _get(args, k):
    return args[k] if isinstance(args, Mapping) else getattr(args, k)


def state_dict_spec(args) -> "OrderedDict[str, tuple]":
    """name -> shape for every tensor of ``ProteinReDiffModel(args).state_dict()``."""
    S, P = _get(args, "single_dim"), _get(args, "pair_dim")
    H, c = _get(args, "num_heads"), _get(args, "head_dim")
    tf, nb = _get(args, "transition_factor"), _get(args, "num_blocks")
    esm, dist, time = _get(args, "esm_dim"), _get(args, "dist_dim"), _get(args, "time_dim")
    spec: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, n_out, n_in, bias=True):
        spec[name + ".weight"] = (n_out, n_in)
        if bias:
            spec[name + ".bias"] = (n_out,)

    spa = "Denoiser.SPAAttnBlock"
    spec[spa + ".layer_norm_m.weight"] = (S,)
    spec[spa + ".layer_norm_m.bias"] = (S,)
    spec[spa + ".linear_z.0.weight"] = (P,)
    spec[spa + ".linear_z.0.bias"] = (P,)
    lin(spa + ".linear_z.1", H, P, bias=False)
    for n in "qkv":
        lin(spa + f".mha.linear_{n}", H * S, S, bias=False)
    lin(spa + ".mha.linear_o", S, H * S)
    lin(spa + ".mha.linear_g", H * S, S)
    opm = "Denoiser.opm"
    spec[opm + ".layer_norm.weight"] = (S,)
    spec[opm + ".layer_norm.bias"] = (S,)
    lin(opm + ".linear_1", S // 4, S)
    lin(opm + ".linear_2", S // 4, S)
    lin(opm + ".linear_out", P, S // 4)
    for i in range(nb):
        fb = f"Denoiser.folding_blocks.{i}"
        lin(fb + ".attn_bias.1", H, P)
        for n in "qkv":
            lin(fb + f".single_attn.{n}_proj", H * c, S, bias=False)
        lin(fb + ".single_attn.gate_proj", H * c, S)
        lin(fb + ".single_attn.out_proj", S, H * c)
        lin(fb + ".single_fc.1", S * tf, S)
        lin(fb + ".single_fc.3", S, S * tf)
        lin(fb + ".outer_linear.linear", P, 2 * S)
        for mode in ("outgoing", "incoming"):
            tm = fb + f".pair_mul_{mode}"
            lin(tm + ".ab_proj", 2 * P, P)
            lin(tm + ".ab_gate", 2 * P, P)
            lin(tm + ".out_proj", P, P)
            lin(tm + ".out_gate", P, P)
        for mode in ("starting", "ending"):
            ta = fb + f".pair_attn_{mode}.attn"
            for n in "qkv":
                lin(ta + f".{n}_proj", H * c, P, bias=False)
            lin(ta + ".gate_proj", H * c, P)
            lin(ta + ".out_proj", P, H * c)
        lin(fb + ".pair_fc.1", P * tf, P)
        lin(fb + ".pair_fc.3", P, P * tf)
    for f, card in enumerate(ATOM_FEATURE_CARDS):
        spec[f"embed_atom_feats.embeddings.{f}.weight"] = (card, S)
    spec["embed_beta.0.weight"] = (time // 2,)
    lin("embed_beta.1", P, time, bias=False)
    lin("embed_residue_type.1", S, NUM_RESIDUE_CLASSES, bias=False)
    for f, card in enumerate(BOND_FEATURE_CARDS):
        spec[f"embed_bond_feats.embeddings.{f}.weight"] = (card, P)
    spec["embed_bond_distance.weight"] = (_get(args, "max_bond_distance") + 1, P)
    lin("embed_residue_esm.1", S, esm, bias=False)
    spec["embed_relpos.weight"] = (2 * _get(args, "max_relpos") + 1, P)
    spec["embed_dist.0.center"] = (dist,)
    lin("embed_dist.1", P, dist, bias=False)
    lin("weight_radial.1", P, P)
    lin("weight_radial.3", 1, P, bias=False)
    lin("seq_mlp.1", S, S)
    lin("seq_mlp.3", NUM_RESIDUE_CLASSES, S, bias=False)
    return spec


def frozen_values(args) -> Dict[str, torch.Tensor]:
    """The two requires_grad=False parameters (reference modules.py:77-79, 91-93)."""
    return {
        "embed_beta.0.weight": torch.logspace(-4.0, 0.0, _get(args, "time_dim") // 2),
        "embed_dist.0.center": torch.linspace(0.0, 2.0, _get(args, "dist_dim")),
    }


def spec_tensors(args) -> "OrderedDict[str, torch.Tensor]":
    """Zero tensors of every shape (frozen ones at their values): a template for
    ``synthetic.deterministic_state_dict``."""
    frozen = frozen_values(args)
    return OrderedDict((k, frozen[k] if k in frozen else torch.zeros(shape))
                       for k, shape in state_dict_spec(args).items())
