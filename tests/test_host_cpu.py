"""CPU-only checks of the host side: C-ABI export list, boundary error behaviour, schedule,
batch preparation and parameter inventory.  No compute call is made (there is no GPU here)."""
import os
import re

import numpy as np
import pytest
import torch

import prd_oracle as O
from conftest import ROOT
from protein_redesign_amd import _lib, ops
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.diffusion_model import ProteinReDiffModel
from protein_redesign_amd.schedule import get_betas, schedule_tables
from protein_redesign_amd.synthetic import NoiseSource, clone_batch, synthetic_batch
from protein_redesign_amd.weights import state_dict_spec

TINY = dict(single_dim=32, pair_dim=32, head_dim=16, num_heads=4, num_blocks=1, esm_dim=16, num_steps=6, mask_prob=0.3)


def header_functions():
    text = open(os.path.join(ROOT, "include", "prd_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|size_t)\s+(prd_\w+)\s*\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 19
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/prd_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding table out of sync with the header"
    assert L.prd_version() == 100


def test_workspace_query_is_host_only():
    assert ops.workspace_bytes("tri_mul", 1, 320, 512, 64) == 3 * 64 * 320 * 320 * 4
    assert ops.workspace_bytes("tri_attn", 2, 100, 512, 32) == 2 * 100 * 100 * 64 * 4


def test_ops_refuse_cpu_tensors():
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.layer_norm(torch.zeros(4, 8))
    m = ProteinReDiffModel(make_args(**TINY))
    batch = synthetic_batch([(2, 5)], esm_dim=16, seed=3)
    with pytest.raises(RuntimeError, match="GPU only"):
        m.sample(batch)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libprd_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_state_dict_matches_reference_inventory():
    args = make_args(num_blocks=3)
    m = ProteinReDiffModel(args)
    spec = state_dict_spec(args)
    sd = m.state_dict()
    assert list(sd) == list(spec)
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    assert sum(v.numel() for v in ProteinReDiffModel(make_args(num_blocks=4)).state_dict().values()) == 16_275_280
    assert not m.embed_beta[0].weight.requires_grad and not m.embed_dist[0].center.requires_grad


def test_schedule_tables_match_oracle():
    for T, sched in ((10, "linear"), (1000, "linear"), (64, "cosine")):
        mine, ref = schedule_tables(T, sched), O.schedule_tables(T, sched)
        for k in ref:
            assert torch.equal(mine[k], ref[k]), (T, sched, k)
    with pytest.raises(ValueError):
        get_betas(10, "quadratic")


def test_prepare_batch_matches_oracle():
    args = make_args(**TINY)
    m = ProteinReDiffModel(args)
    batch = synthetic_batch([(3, 9), (2, 6)], esm_dim=16, seed=5, n_total=14)
    src = [NoiseSource(7, k) for k in range(2)]
    perms = [NoiseSource(7, k).randperm(n) for k, n in enumerate((9, 6))]
    mine = m.prepare_batch(clone_batch(batch), sources=src)
    ref = O.prepare_batch(clone_batch(batch), args["mask_prob"], perms)
    for k in ("residue_one_hot", "residue_esm", "residue_type_masked", "residue_extra_mask",
              "residue_inv_extra_mask", "x", "residue_and_atom_mask"):
        assert torch.equal(mine[k], ref[k]), k
    assert int((1 - mine["residue_extra_mask"])[0, 3:12].sum()) == int(9 * 0.3)


def test_argparse_surface():
    from argparse import ArgumentParser
    p = ProteinReDiffModel.add_argparse_args(ArgumentParser())
    ns = p.parse_args(["--num_blocks", "4", "--num_steps", "1000"])
    assert ns.single_dim == 512 and ns.pair_dim == 64 and ns.num_blocks == 4 and ns.n_recycles == 4
    ProteinReDiffModel(ns)
