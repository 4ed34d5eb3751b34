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
    assert L.prd_version() == 101      # 101: prd_step_boundary's sync buffer is two int32 (include/prd_hip.h)


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


# ---------------------------------------------------------------------------------------------------
# checkpoint / EMA surface (generate.py:103-105, model.py:124,197-201,215-217; torch_ema==0.3 key set)
# ---------------------------------------------------------------------------------------------------

def _pl17_checkpoint(model, shadow, nested: bool, as_namespace: bool):
    """A checkpoint with the key set PyTorch-Lightning 1.7 writes for ``save_hyperparameters(args)``: state_dict +
    hyper_parameters (either the flat argument dict or nested under the constructor's parameter name ``args``) + the
    ``ema_state_dict`` the reference's on_save_checkpoint adds (torch-ema 0.3 keys)."""
    from argparse import Namespace
    hp = dict(make_args(**TINY))
    hp = Namespace(**hp) if as_namespace else hp
    return {
        "epoch": 3, "global_step": 1234, "pytorch-lightning_version": "1.7.7",
        "state_dict": {k: v.clone() for k, v in model.state_dict().items()},
        "hyper_parameters": {"args": hp} if nested else hp,
        "ema_state_dict": {"decay": 0.999, "num_updates": 1234, "shadow_params": shadow, "collected_params": None},
        "optimizer_states": [], "lr_schedulers": [], "loops": {}, "callbacks": {},
    }


@pytest.mark.parametrize("nested,as_namespace,only_trainable", [(False, False, False), (True, True, False), (True, False, True)])
def test_load_from_lightning_style_checkpoint(tmp_path, nested, as_namespace, only_trainable):
    torch.manual_seed(0)
    src = ProteinReDiffModel(make_args(**TINY))
    params = list(src.parameters())
    assert len(params) - len([p for p in params if p.requires_grad]) == 2       # the two frozen projection tables
    g = torch.Generator().manual_seed(1)
    kept = [p for p in params if p.requires_grad or not only_trainable]          # 242-style (all) or 240-style (trainable) list
    shadow = [p.detach() + 0.25 * torch.randn(p.shape, generator=g) for p in kept]
    path = tmp_path / "last.ckpt"
    torch.save(_pl17_checkpoint(src, shadow, nested, as_namespace), path)
    m = ProteinReDiffModel.load_from_checkpoint(str(path), num_steps=17)         # generate.py:103-105 overrides num_steps
    assert m.num_steps == 17 and m.single_dim == TINY["single_dim"]
    for (k, a), b in zip(m.state_dict().items(), src.state_dict().values()):
        assert torch.equal(a, b), k
    assert m.ema.num_updates == 1234 and len(m.ema.shadow) == len(kept)
    mine = list(m.parameters())
    before = [p.detach().clone() for p in mine]
    with m.ema.average_parameters(m.parameters()):                               # predict_step / validation_step context
        tgt = [p for p in mine if p.requires_grad or not only_trainable]
        assert all(torch.equal(p, s) for p, s in zip(tgt, shadow))
    assert all(torch.equal(p, q) for p, q in zip(mine, before))                  # restored afterwards


def test_ema_state_round_trip_and_warmup():
    """on_save_checkpoint -> on_load_checkpoint round trip with torch-ema 0.3's key set, and its warm-up decay."""
    torch.manual_seed(0)
    m = ProteinReDiffModel(make_args(**TINY))
    ck = {}
    m.on_save_checkpoint(ck)
    assert set(ck["ema_state_dict"]) == {"decay", "num_updates", "shadow_params", "collected_params"}
    assert len(ck["ema_state_dict"]["shadow_params"]) == len(list(m.parameters()))
    p0 = next(p for p in m.parameters() if p.requires_grad)
    s0 = m.ema.shadow[0].clone()
    with torch.no_grad():
        p0.add_(1.0)
    m.optimizer_step()                                   # model.py:215-217: EMA update after the optimizer step
    d = min(0.999, 2.0 / 11.0)                           # first update: min(decay, (1 + 1) / (10 + 1))
    assert torch.allclose(m.ema.shadow[0], s0 - (1 - d) * (s0 - p0.detach()), atol=1e-6)
    assert m.ema.num_updates == 1
    ck = {}
    m.on_save_checkpoint(ck)
    m2 = ProteinReDiffModel(make_args(**TINY))
    m2.on_load_checkpoint(ck)
    assert m2.ema.num_updates == 1 and all(torch.equal(a, b) for a, b in zip(m2.ema.shadow, m.ema.shadow))


def test_default_noise_follows_the_global_seed():
    """Without explicit sources the noise key comes from torch's seeded global RNG (pl.seed_everything / generate.py --seed):
    different seeds -> different samples, same seed -> same samples; successive batches never repeat an index."""
    m = ProteinReDiffModel(make_args(**TINY))

    def draw(seed, batch_idx=None):
        torch.manual_seed(seed)
        return [s.randn(4) for s in m._sources(2, batch_idx)]

    a, b, c = draw(11, 0), draw(12, 0), draw(11, 0)
    assert all(torch.equal(x, y) for x, y in zip(a, c))
    assert not any(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[0], a[1]) and not torch.equal(draw(11, 1)[0], a[0])
    torch.manual_seed(5)
    first, second = m._sources(2), m._sources(2)         # direct sample() calls: running sample counter
    assert not torch.equal(first[0].randn(3), second[0].randn(3))
    m.sample_seed = 3                                    # pinned: independent of the global seed
    torch.manual_seed(1)
    x = m._sources(1, 0)[0].randn(3)
    torch.manual_seed(2)
    assert torch.equal(x, m._sources(1, 0)[0].randn(3))


def test_c_abi_argument_checks_under_asan():
    """the host side of the C ABI under AddressSanitizer (python -m protein_redesign_amd.build --asan): every
    entry point is called with an invalid argument set and must return its PRD_ERR_* code without touching the GPU, and ASan
    must stay silent.  Needs hipcc only (no GPU)."""
    import shutil
    import subprocess
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    from protein_redesign_amd.build import build_asan
    exe = build_asan(verbose=False)
    out = subprocess.run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0 and "host ABI check: OK" in text and "AddressSanitizer" not in text, text[-2000:]
