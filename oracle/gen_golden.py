"""Generate tests/golden/*.npz by running the IMPORTED reference (build container only).

    python oracle/gen_golden.py            # writes tests/golden/{tiny,small32,small64,cfg1}.npz

Each fixture stores the case description (JSON), the few random inputs that are not a pure function
of a seed, and the reference's outputs.  Weights and batches are regenerated from seeds by
``protein_redesign_amd.synthetic`` on both sides, so the fixtures stay small.  The reference source
itself never enters the repository: it is imported from /root/reference (oracle/ref_import.py).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ref_import import import_reference  # noqa: E402

from protein_redesign_amd.constants import make_args  # noqa: E402
from protein_redesign_amd.synthetic import (NoiseSource, clone_batch, deterministic_state_dict,  # noqa: E402
                                            synthetic_batch)

CASES = {
    # leaf modules + block + step + 5-step trajectory, ragged batch with padding
    "tiny": dict(
        args=dict(single_dim=32, pair_dim=8, head_dim=4, num_heads=2, num_blocks=2, esm_dim=16,
                  dist_dim=16, time_dim=16, num_steps=5, mask_prob=0.3),
        sizes=[(3, 6), (2, 5)], n_total=11, batch_seed=11, weight_seed=1, leaves=True, traj_sample=(3, 6)),
    # HIP-compatible small shapes (pair_dim 32 / 64, 4 heads x 16)
    "small32": dict(
        args=dict(single_dim=64, pair_dim=32, head_dim=16, num_heads=4, num_blocks=2, esm_dim=32,
                  num_steps=8, mask_prob=0.3),
        sizes=[(5, 20), (4, 17)], n_total=27, batch_seed=12, weight_seed=2, leaves=False, traj_sample=(5, 20)),
    "small64": dict(
        args=dict(single_dim=64, pair_dim=64, head_dim=16, num_heads=4, num_blocks=2, esm_dim=32,
                  num_steps=8, mask_prob=0.3),
        sizes=[(6, 30), (3, 22)], n_total=37, batch_seed=13, weight_seed=3, leaves=False, traj_sample=(6, 30)),
    # BASELINE.json configs[0] without the chemistry: 110 residues + 30 ligand atoms, 256/32, 4 blocks, 10 steps
    "cfg1": dict(
        args=dict(single_dim=256, pair_dim=32, head_dim=16, num_heads=4, num_blocks=4, esm_dim=1280,
                  num_steps=10, mask_prob=0.3),
        sizes=[(30, 110)], n_total=None, batch_seed=0, weight_seed=1, leaves=False, traj_sample=(30, 110)),
    # the same complex through LONG reverse-diffusion loops (trajectory only): T = 200 and the T = 1000 of BASELINE configs[1], with
    # weights an N(0, 0.02^2) perturbation away from the reference's zero / gating initialisation (a network early in training:
    # synthetic.deterministic_state_dict(style="near_init")) so that the loop is well conditioned
    "cfg1_t200": dict(
        args=dict(single_dim=256, pair_dim=32, head_dim=16, num_heads=4, num_blocks=4, esm_dim=1280,
                  num_steps=200, mask_prob=0.3),
        sizes=[(30, 110)], n_total=None, batch_seed=0, weight_seed=1, leaves=False, traj_sample=(30, 110), traj_only=True,
        weight_style="near_init", traj_every=10),
    "cfg1_t1000": dict(
        args=dict(single_dim=256, pair_dim=32, head_dim=16, num_heads=4, num_blocks=4, esm_dim=1280,
                  num_steps=1000, mask_prob=0.3),
        sizes=[(30, 110)], n_total=None, batch_seed=0, weight_seed=1, leaves=False, traj_sample=(30, 110), traj_only=True,
        weight_style="near_init", traj_every=25),
    # ... and with full-strength random weights, where the loop amplifies round-off chaotically: the fixture holds the state every
    # 10 steps so that implementations can be compared SEGMENT by segment, each restarted from the reference's own state
    "cfg1_t200_random": dict(
        args=dict(single_dim=256, pair_dim=32, head_dim=16, num_heads=4, num_blocks=4, esm_dim=1280,
                  num_steps=200, mask_prob=0.3),
        sizes=[(30, 110)], n_total=None, batch_seed=0, weight_seed=1, leaves=False, traj_sample=(30, 110), traj_only=True,
        weight_style="random", traj_every=10),
}

NOISE_SEED = 7


class _Injected:
    """Route the reference's global RNG calls inside ``sample`` through NoiseSource objects."""

    def __init__(self, sources):
        self.sources = sources

    def __enter__(self):
        self._randn_like, self._randperm = torch.randn_like, torch.randperm
        src = self.sources

        def randn_like(x, **kw):
            return torch.stack([src[k].randn(*x.shape[1:]) for k in range(x.shape[0])]).to(x.dtype)

        def randperm(n, **kw):
            assert len(src) == 1, "reference draws one permutation for the whole batch"
            return src[0].randperm(n)

        torch.randn_like, torch.randperm = randn_like, randperm
        return self

    def __exit__(self, *exc):
        torch.randn_like, torch.randperm = self._randn_like, self._randperm


def build_reference(ref_model, case):
    args = make_args(**case["args"])
    model = ref_model.ProteinReDiffModel(args).eval()
    sd = deterministic_state_dict(model.state_dict(), seed=case["weight_seed"], style=case.get("weight_style", "random"), scales=case.get("weight_scales"))
    model.load_state_dict(sd)
    model.run_setup_schedule()
    model.setup_schedule = True
    return model, args


@torch.inference_mode()
def run_case(name, case, ref_model):
    model, args = build_reference(ref_model, case)
    out = {"case": np.array(json.dumps(dict(case, name=name)))}
    out["state_dict_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in model.state_dict().items()}))
    esm_dim = args["esm_dim"]
    if case.get("traj_only"):
        one = synthetic_batch([case["traj_sample"]], esm_dim=esm_dim, seed=case["batch_seed"] + 500)
        states = []                      # (z, seq_t) entering every traj_every-th step, captured at the sample_step call
        every = case.get("traj_every", 0)
        inner = model.sample_step

        def spy(batch, z, seq_t, mask, t):
            step = args["num_steps"] - 1 - int(t[0])
            if every and step % every == 0:
                states.append((step, z.clone(), seq_t.clone()))
            return inner(batch, z, seq_t, mask, t)

        model.sample_step = spy
        with _Injected([NoiseSource(NOISE_SEED, 0)]):
            pos, logits = model.sample(clone_batch(one))
        model.sample_step = inner
        out.update(traj_pos=pos.numpy(), traj_logits=logits.numpy())
        if states:
            out.update(seg_step=np.array([s_[0] for s_ in states]), seg_z=torch.cat([s_[1] for s_ in states]).numpy(),
                       seg_seq_t=torch.cat([s_[2] for s_ in states]).numpy())
        return out
    batch = synthetic_batch(case["sizes"], esm_dim=esm_dim, seed=case["batch_seed"], n_total=case["n_total"])
    b, N = batch["atom_mask"].shape
    S, P = args["single_dim"], args["pair_dim"]
    g = torch.Generator().manual_seed(1000 + case["batch_seed"])

    # ---- one network step on a hand-prepared batch (no masking randomness) ----
    n_res = batch["residue_mask"].sum(-1).long().tolist()
    perm_src = [NoiseSource(NOISE_SEED, 100 + k) for k in range(b)]
    perms = [perm_src[k].randperm(n_res[k]) for k in range(b)]
    # replicate model.prepare_batch (eval) with per-sample permutations via the oracle's own helper is
    # NOT used here: we drive the reference's prepare_batch itself with an injected permutation.
    prepared = []
    for k in range(b):
        one = {kk: (vv[k:k + 1].clone() if torch.is_tensor(vv) else vv) for kk, vv in batch.items()}
        model.mask_prob = args["mask_prob"]
        with _Injected([_FixedPerm(perms[k])]):
            prepared.append(model.prepare_batch(one))
    pb = {kk: torch.cat([p_[kk] for p_ in prepared]) for kk in prepared[0] if torch.is_tensor(prepared[0][kk])}
    mask = pb["residue_and_atom_mask"]
    z = torch.randn(b, N, 3, generator=g)
    seq_t = torch.randn(b, N, 21, generator=g)
    t = torch.tensor([(3 + 2 * k) % args["num_steps"] for k in range(b)], dtype=torch.long)
    eps, logits = model.sample_step(pb, z, seq_t, mask, t)
    out.update(step_z=z.numpy(), step_seq_t=seq_t.numpy(), step_t=t.numpy(),
               step_noise_pred=eps.numpy(), step_seq_pred=logits.numpy(),
               prep_extra_mask=pb["residue_extra_mask"].numpy(), prep_x=pb["x"].numpy())

    # ---- diffusion loss (validation / training forward value), injected noise ----
    nz = O_remove_mean(torch.randn(b, N, 3, generator=g), mask)
    ns = O_remove_mean(torch.randn(b, N, 21, generator=g), pb["residue_mask"])
    with _Sequence([nz, ns]):
        out["loss_value"] = model.diffusion_loss(pb, pb["x"], mask, t).numpy()
    out.update(loss_noise_z=nz.numpy(), loss_noise_seq=ns.numpy())

    # ---- schedule tables ----
    for T, sched in ((10, "linear"), (64, "linear"), (1000, "linear"), (64, "cosine")):
        model.num_steps, model.diffusion_schedule = T, sched
        model.run_setup_schedule()
        for key in ("betas", "sqrt_alphas", "sqrt_one_minus_alphas_cumprod", "sqrt_betas", "alphas"):
            out[f"sched_{sched}_{T}_{key}"] = getattr(model, key).numpy()
    model.num_steps, model.diffusion_schedule = args["num_steps"], args["diffusion_schedule"]
    model.run_setup_schedule()

    # ---- leaf modules on random single / pair tensors ----
    if case["leaves"]:
        single = torch.randn(b, N, S, generator=g)
        pair = torch.randn(b, N, N, P, generator=g)
        mask2d = mask.unsqueeze(-1) * mask.unsqueeze(-2)
        out.update(leaf_single=single.numpy(), leaf_pair=pair.numpy())
        blk = model.Denoiser.folding_blocks[0]
        bias = blk.attn_bias(pair)
        out["leaf_pair_bias"] = bias.numpy()
        out["leaf_single_attn"] = blk.single_attn(single, mask, attn_bias=bias).numpy()
        out["leaf_single_fc"] = blk.single_fc(single).numpy()
        out["leaf_outer_linear"] = blk.outer_linear(single).numpy()
        out["leaf_tri_mul_out"] = blk.pair_mul_outgoing(pair, mask2d).numpy()
        out["leaf_tri_mul_in"] = blk.pair_mul_incoming(pair, mask2d).numpy()
        out["leaf_tri_attn_start"] = blk.pair_attn_starting(pair, mask2d).numpy()
        out["leaf_tri_attn_end"] = blk.pair_attn_ending(pair, mask2d).numpy()
        out["leaf_pair_fc"] = blk.pair_fc(pair).numpy()
        s2, p2 = blk(single, pair, mask)
        out.update(leaf_block_single=s2.numpy(), leaf_block_pair=p2.numpy())
        out["leaf_opm"] = model.Denoiser.opm(single, mask).numpy()
        out["leaf_spa"] = model.Denoiser.SPAAttnBlock(single, pair, mask).numpy()
        s3, p3, _ = model.Denoiser(pb, None, None, single.clone(), pair.clone(), None)
        out.update(leaf_denoiser_single=s3.numpy(), leaf_denoiser_pair=p3.numpy())

    # ---- full reverse-diffusion trajectory, batch size 1 (generate.py default) ----
    one = synthetic_batch([case["traj_sample"]], esm_dim=esm_dim, seed=case["batch_seed"] + 500)
    with _Injected([NoiseSource(NOISE_SEED, 0)]):
        pos, logits = model.sample(clone_batch(one))
    out.update(traj_pos=pos.numpy(), traj_logits=logits.numpy())
    return out


GRAD_PROJECTIONS = 4


def grad_fingerprint(named_grads):
    """Per trainable tensor: L2 norm of the gradient and its projections on GRAD_PROJECTIONS seeded Gaussian directions
    (direction k of tensor i: generator seed 4242 + 16 i + k, drawn in float64) -- 5 numbers pin a gradient without storing it."""
    norms, projs = [], []
    for i, (name, g) in enumerate(named_grads):
        g = g.detach().double().reshape(-1)
        norms.append(float(g.norm()))
        row = []
        for k in range(GRAD_PROJECTIONS):
            gen = torch.Generator().manual_seed(4242 + 16 * i + k)
            row.append(float(torch.dot(g, torch.randn(g.numel(), generator=gen, dtype=torch.float64))))
        projs.append(row)
    return np.array(norms), np.array(projs)


def run_grad_case(name, case, ref_model):
    """training_step of the imported reference (model.py:528-549) with injected t / noise: loss and gradient fingerprints of
    all trainable tensors.  Runs with autograd enabled (not under inference_mode), per-block checkpointing included."""
    model, args = build_reference(ref_model, case)
    model.train()
    esm_dim = args["esm_dim"]
    batch = synthetic_batch(case["sizes"], esm_dim=esm_dim, seed=case["batch_seed"], n_total=case["n_total"])
    b, N = batch["atom_mask"].shape
    n_res = batch["residue_mask"].sum(-1).long().tolist()
    perms = [NoiseSource(NOISE_SEED, 100 + k).randperm(n_res[k]) for k in range(b)]
    prepared = []
    for k in range(b):
        one = {kk: (vv[k:k + 1].clone() if torch.is_tensor(vv) else vv) for kk, vv in batch.items()}
        model.mask_prob = args["mask_prob"]
        with _Injected([_FixedPerm(perms[k])]):
            prepared.append(model.prepare_batch(one))
    pb = {kk: torch.cat([p_[kk] for p_ in prepared]) for kk in prepared[0] if torch.is_tensor(prepared[0][kk])}
    mask = pb["residue_and_atom_mask"]
    g = torch.Generator().manual_seed(2000 + case["batch_seed"])
    t = torch.tensor([(3 + 2 * k) % args["num_steps"] for k in range(b)], dtype=torch.long)
    nz = O_remove_mean(torch.randn(b, N, 3, generator=g), mask)
    ns = O_remove_mean(torch.randn(b, N, 21, generator=g), pb["residue_mask"])
    with _Sequence([nz, ns]):
        diff = model.diffusion_loss(pb, pb["x"], mask, t)
    loss = torch.mean(diff / (mask > 0.5).sum(-1))                 # model.py:538-540
    loss.backward()
    named = [(n_, p_.grad) for n_, p_ in model.named_parameters() if p_.requires_grad]
    assert all(g_ is not None for _, g_ in named)
    norms, projs = grad_fingerprint(named)
    return {"train_t": t.numpy(), "train_noise_z": nz.numpy(), "train_noise_seq": ns.numpy(),
            "train_loss": np.array(float(loss)), "train_grad_names": np.array(json.dumps([n_ for n_, _ in named])),
            "train_grad_norm": norms, "train_grad_proj": projs}


def O_remove_mean(x, mask):
    m = mask.unsqueeze(-1).expand_as(x)
    return x - m * (m * x).sum(1, keepdim=True) / m.sum(1, keepdim=True)


class _Sequence:
    """torch.randn_like returns the given tensors in order (their mean is already removed, and remove_mean is
    idempotent up to round-off, so the reference's remove_mean(randn_like(.)) reproduces them)."""

    def __init__(self, tensors):
        self.tensors = list(tensors)

    def __enter__(self):
        self._randn_like = torch.randn_like
        it = iter(self.tensors)
        torch.randn_like = lambda x, **kw: next(it).to(x.dtype)
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._randn_like


class _FixedPerm:
    def __init__(self, perm):
        self.perm = perm

    def randperm(self, n):
        assert n == self.perm.numel()
        return self.perm

    def randn(self, *shape):
        raise RuntimeError("prepare_batch must not draw normals")


def run_host_case():
    """Callers / data formats around the hot path: reference collate_fn, PDB text, sequence decoding."""
    import dataclasses

    import generate as ref_generate                     # /root/reference/generate.py (stubs make it importable)
    import ProteinReDiff.data as ref_data
    import ProteinReDiff.protein as ref_protein
    from protein_redesign_amd.synthetic import synthetic_sample
    out = {}
    samples = [synthetic_sample(4, 9, esm_dim=8, seed=31), synthetic_sample(6, 5, esm_dim=8, seed=32)]
    batch = ref_data.collate_fn(samples)
    for k, v in batch.items():
        if torch.is_tensor(v):
            out["collate_" + k] = v.numpy()
    g = torch.Generator().manual_seed(33)
    nres = 7
    prot = ref_protein.Protein(
        chain_index=np.array([0, 0, 0, 1, 1, 1, 1]), residue_index=np.array([3, 4, 5, 1, 2, 3, 12]),
        aatype=torch.randint(0, 20, (nres,), generator=g).numpy(),
        atom_pos=(30 * torch.randn(nres, 37, 3, generator=g)).numpy().astype(np.float32),
        atom_mask=(torch.rand(nres, 37, generator=g) < 0.3).float().numpy())
    out.update(pdb_chain_index=prot.chain_index, pdb_residue_index=prot.residue_index, pdb_aatype=prot.aatype,
               pdb_atom_pos=prot.atom_pos, pdb_atom_mask=prot.atom_mask,
               pdb_text=np.array(ref_protein.protein_to_pdb_string(prot)))
    logits = torch.randn(12, 21, generator=g)
    logits[:2, 0] += 10.0                                  # leading X's are stripped
    logits[-1, 0] += 10.0
    logits[2:-1, 0] -= 10.0
    out["seq_logits"] = logits.numpy()
    out["seq_pred"] = np.array("".join(ref_generate.predict_seq(logits.numpy())))
    seq_prot = ref_generate.update_seq(ref_protein.protein_from_sequence("A" * 9), logits.numpy())
    out["seq_aatype"] = seq_prot.aatype
    fs = ref_protein.protein_from_sequence("ACDXW")
    out.update(fromseq_aatype=fs.aatype, fromseq_mask=fs.atom_mask)
    return out


def main():
    ref_model, _ = import_reference()
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    names = sys.argv[1:] or list(CASES) + ["host"]
    if "host" in names:
        names = [n for n in names if n != "host"]
        path = os.path.join(ROOT, "tests", "golden", "host.npz")
        np.savez_compressed(path, **run_host_case())
        print("host ->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")
    for name in names:
        torch.manual_seed(0)
        res = run_case(name, CASES[name], ref_model)
        if not CASES[name].get("traj_only"):
            torch.manual_seed(0)
            res.update(run_grad_case(name, CASES[name], ref_model))
        path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
        np.savez_compressed(path, **res)
        print(name, "->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
