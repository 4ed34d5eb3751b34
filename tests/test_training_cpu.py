"""CPU checks of the training step's backward machinery (SURVEY.md §8f "next" #1), without a GPU:

* torch_ref.py (the differentiable restatement a HIP operator's backward recomputes) + the HipOp / per-block checkpoint wiring
  of training.py against the ORACLE's autograd: loss value and the gradient of every trainable tensor.  The HIP forwards
  cannot run here, so the test swaps ``training.HipOp`` for a double that evaluates the restatement in ``forward`` as well --
  the product code itself has no such switch;
* the oracle's loss + gradients against the fingerprints captured from the imported reference (tests/golden/*.npz);
* the data-parallel gradient average over a world-size-2 gloo group.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import prd_oracle as O
from conftest import rel_l2
from protein_redesign_amd import training
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.diffusion_model import ProteinReDiffModel
from protein_redesign_amd.synthetic import NoiseSource, deterministic_state_dict, synthetic_batch
from protein_redesign_amd.weights import spec_tensors

NOISE_SEED = 7
GRAD_PROJECTIONS = 4
FINGERPRINT_TOL = 1e-3


class _RefOp(torch.autograd.Function):
    """Test double of training.HipOp: same contract, but ``forward`` evaluates the torch restatement (no GPU here)."""

    @staticmethod
    def forward(ctx, fwd, ref, *tensors):
        ctx.ref = ref
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            return ref(*[t.detach() for t in tensors])

    backward = training.HipOp.backward


def oracle_training_loss(params, args, pb, t, nz, ns):
    """model.py:528-549 on the oracle: mean over the batch of diffusion_loss / node count."""
    diff = O.diffusion_loss(params, args, pb, t, nz, ns)
    return torch.mean(diff / (pb["residue_and_atom_mask"] > 0.5).sum(-1))


def case_inputs(golden, name):
    case, z = golden(name)
    args = make_args(**case["args"])
    params = deterministic_state_dict(spec_tensors(args), seed=case["weight_seed"], style=case.get("weight_style", "random"), scales=case.get("weight_scales"))
    sizes = [tuple(s) for s in case["sizes"]]
    batch = synthetic_batch(sizes, esm_dim=args["esm_dim"], seed=case["batch_seed"], n_total=case["n_total"])
    perms = [NoiseSource(NOISE_SEED, 100 + k).randperm(n) for k, (_, n) in enumerate(sizes)]
    pb = O.prepare_batch(batch, args["mask_prob"], perms)
    return case, z, args, params, pb


def oracle_grads(args, params, pb, t, nz, ns):
    leaf = {k: v.clone().requires_grad_(k not in ("embed_beta.0.weight", "embed_dist.0.center")) for k, v in params.items()}
    loss = oracle_training_loss(leaf, args, pb, t, nz, ns)
    loss.backward()
    return float(loss), {k: v.grad for k, v in leaf.items() if v.requires_grad}


@pytest.mark.parametrize("name", ["tiny", "small32"])
def test_backward_wiring_matches_oracle_autograd(golden, name, monkeypatch):
    case, z, args, params, pb = case_inputs(golden, name)
    t = torch.from_numpy(z["train_t"])
    nz, ns = torch.from_numpy(z["train_noise_z"]), torch.from_numpy(z["train_noise_seq"])
    want_loss, want = oracle_grads(args, params, pb, t, nz, ns)
    monkeypatch.setattr(training, "HipOp", _RefOp)
    # TriangleMultiplication has a hand-written HIP backward (training.TriMulFn): on the CPU its restatement stands in
    from protein_redesign_amd import torch_ref as R
    monkeypatch.setattr(training, "tri_mul_update",
                        lambda tm, pair, mask: R.triangle_multiplication(pair, mask, *tm.weights(), incoming=tm.mode == "incoming"))
    monkeypatch.setattr(training, "tri_attn_update",
                        lambda ta, pair, mask: R.triangle_attention(pair, mask, *ta.attn.weights(), ta.attn.num_heads, ta.attn.head_dim,
                                                                    ending=ta.mode == "ending"))
    model = ProteinReDiffModel(args)
    model.load_state_dict(params)
    model.run_setup_schedule()
    model.setup_schedule = True
    diff = model.diffusion_loss(dict(pb), pb["x"], pb["residue_and_atom_mask"], t, nz, ns)
    loss = torch.mean(diff / (pb["residue_and_atom_mask"] > 0.5).sum(-1))
    loss.backward()
    assert abs(float(loss) - want_loss) < 1e-5 * abs(want_loss)
    got = {k: p.grad for k, p in model.named_parameters() if p.requires_grad}
    assert sorted(got) == sorted(want) and len(got) == len(list(model.parameters())) - 2
    scale = float(torch.cat([g.reshape(-1) for g in want.values()]).double().norm())
    for k in want:
        assert got[k] is not None, k
        # fp32 autograd of two op orders; tensors whose exact gradient is zero (e.g. a bias in front of a LayerNorm) hold round-off only
        err = float((got[k].double() - want[k].double()).norm())
        assert err < 2e-4 * float(want[k].double().norm()) + 1e-7 * scale, (k, err, float(want[k].norm()))


@pytest.mark.parametrize("name", ["tiny", "small32", "small64", "cfg1"])
def test_oracle_gradients_match_reference_fingerprints(golden, name):
    """The oracle's autograd against the imported reference's training_step: loss and, for each of the trainable tensors, the
    gradient norm and four seeded random projections (oracle/gen_golden.py: grad_fingerprint)."""
    case, z, args, params, pb = case_inputs(golden, name)
    t = torch.from_numpy(z["train_t"])
    loss, grads = oracle_grads(args, params, pb, t, torch.from_numpy(z["train_noise_z"]), torch.from_numpy(z["train_noise_seq"]))
    assert abs(loss - float(z["train_loss"])) < 1e-5 * abs(float(z["train_loss"]))
    names = json.loads(str(z["train_grad_names"]))
    assert sorted(names) == sorted(grads)
    scale = float(np.linalg.norm(z["train_grad_norm"]))
    for i, k in enumerate(names):
        g = grads[k].double().reshape(-1)
        n_ref = float(z["train_grad_norm"][i])
        # FINGERPRINT_TOL: the reference's own fp32 CPU autograd moves by up to ~2e-4 of a tensor's gradient norm with the number
        # of BLAS threads (N^2-term reductions with cancellation, e.g. Denoiser.opm.layer_norm.weight), so the fixture pins
        # identity of the gradients at 1e-3; the tensor-by-tensor 1e-4 comparison is HIP vs the oracle's autograd (GPU suite)
        assert abs(float(g.norm()) - n_ref) < FINGERPRINT_TOL * n_ref + 1e-7 * scale, (k, float(g.norm()), n_ref)
        for j in range(GRAD_PROJECTIONS):
            gen = torch.Generator().manual_seed(4242 + 16 * i + j)
            proj = float(torch.dot(g, torch.randn(g.numel(), generator=gen, dtype=torch.float64)))
            assert abs(proj - float(z["train_grad_proj"][i, j])) < FINGERPRINT_TOL * n_ref + 1e-7 * scale, (k, j)


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)
    extra = torch.nn.Parameter(torch.zeros(4))                   # gradient 1 on rank 0, 0 on rank 1 -> 0.5 after averaging
    unused = torch.nn.Parameter(torch.zeros(3))                  # never touched: like DDP(find_unused_parameters=False) an error
    x = torch.arange(10.0).view(2, 5) + rank
    loss = lin(x).square().sum() + extra.sum() * (1.0 if rank == 0 else 0.0)
    loss.backward()
    training.all_reduce_gradients([lin.weight, lin.bias, frozen, extra])
    training.all_reduce_gradients([lin.weight, lin.bias, frozen, extra])      # persistent flat buffer: a second call averages equal values
    raised = False
    try:
        training.all_reduce_gradients([lin.weight, unused])
    except RuntimeError:
        raised = True
    torch.save({"w": lin.weight.grad, "b": lin.bias.grad, "u": extra.grad, "raised": raised}, os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_average_over_two_ranks(tmp_path):
    port = 29700 + (os.getpid() % 500)
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = (torch.load(os.path.join(str(tmp_path), f"g{r}.pt")) for r in range(2))
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    want_w = torch.zeros_like(lin.weight)
    for rank in range(2):
        lin.zero_grad()
        lin((torch.arange(10.0).view(2, 5) + rank)).square().sum().backward()
        want_w += lin.weight.grad / 2
    assert torch.allclose(g0["w"], want_w, rtol=1e-6) and torch.equal(g0["w"], g1["w"]) and torch.equal(g0["b"], g1["b"])
    assert torch.allclose(g0["u"], torch.full((4,), 0.5)) and torch.equal(g0["u"], g1["u"])
    assert g0["raised"] and g1["raised"]


# ---------------------------------------------------------------------------------------------------
# accumulate_grad_batches (train.py:57; README.md:136-169 use 8 / 10): gradients accumulate locally, ONE collective and ONE
# optimiser step per k micro-batches
# ---------------------------------------------------------------------------------------------------

class _StubEma:
    def __init__(self):
        self.updates = 0

    def update(self, params):
        self.updates += 1


class _StubModel(torch.nn.Module):
    """What training.Fitter touches of a ProteinReDiffModel: training_step(batch, idx, check_finite=...), parameters(), ema."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.lin = torch.nn.Linear(6, 4)
        self.ema = _StubEma()
        self.nonfinite_policy = "raise"

    def training_step(self, batch, batch_idx, check_finite=True):
        return torch.tanh(self.lin(batch["x"])).square().sum(-1).mean()          # mean over the batch, like model.py:546


def _accum_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}
    real = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    dist.all_reduce = counting
    data = torch.randn(world, 8, 6, generator=torch.Generator().manual_seed(11))[rank]     # 8 samples per rank: two groups of k = 4
    out = {}
    for slices in (1, 3):
        model = _StubModel()
        opt = torch.optim.SGD(model.parameters(), lr=1.0)                    # lr 1: the parameter change IS the averaged gradient
        fitter = training.Fitter(model, opt, accumulate_grad_batches=4, reduce_slices=slices)
        p0 = [p.detach().clone() for p in model.parameters()]
        calls["n"] = 0
        for i in range(4):
            fitter.step({"x": data[i:i + 1]}, i)
        out[f"delta{slices}"] = [a - b.detach() for a, b in zip(p0, model.parameters())]
        out[f"calls{slices}"] = calls["n"]
        out[f"steps{slices}"] = (fitter.optimizer_steps, model.ema.updates)
        for i in range(4, 7):                                                # an incomplete group: flushed at the end of the epoch
            fitter.step({"x": data[i:i + 1]}, i)
        assert fitter.optimizer_steps == 1 and fitter.micro == 3
        fitter.finish_accumulation()
        out[f"flush{slices}"] = (fitter.optimizer_steps, fitter.micro)
    # the stateless form: fit_step with accumulate_grad_batches
    model = _StubModel()
    opt = torch.optim.SGD(model.parameters(), lr=1.0)
    p0 = [p.detach().clone() for p in model.parameters()]
    for i in range(4):
        training.fit_step(model, {"x": data[i:i + 1]}, i, opt, accumulate_grad_batches=4)
    out["delta_fit_step"] = [a - b.detach() for a, b in zip(p0, model.parameters())]
    # group boundaries follow the count of micro-batches, not batch_idx % k: a loader that starts at index 5 still steps after 4 calls;
    # a trailing incomplete group is stepped by fit_flush (Lightning flushes it at the end of the epoch)
    model = _StubModel()
    opt = torch.optim.SGD(model.parameters(), lr=1.0)
    p0 = [p.detach().clone() for p in model.parameters()]
    for i in range(4):
        training.fit_step(model, {"x": data[i:i + 1]}, 5 + i, opt, accumulate_grad_batches=4)
    out["delta_fit_step_offset"] = [a - b.detach() for a, b in zip(p0, model.parameters())]
    for i in range(4, 6):
        training.fit_step(model, {"x": data[i:i + 1]}, 5 + i, opt, accumulate_grad_batches=4)
    out["fit_flush"] = (model.ema.updates, training.fit_flush(model, opt), model.ema.updates, training.fit_flush(model, opt))
    torch.save(out, os.path.join(out_dir, f"a{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_accumulation_over_two_ranks(tmp_path):
    """k = 4 micro-batches of one sample on each of two gloo ranks == the gradient of the single batch of 4 per rank, averaged over
    the ranks; ONE all-reduce (or ``reduce_slices`` pieces of it) and ONE optimiser / EMA step per 4 micro-batches."""
    port = 29200 + (os.getpid() % 500)
    mp.spawn(_accum_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"a{r}.pt")) for r in range(2))
    data = torch.randn(2, 8, 6, generator=torch.Generator().manual_seed(11))
    model = _StubModel()
    want = [torch.zeros_like(p) for p in model.parameters()]
    for rank in range(2):
        model.zero_grad()
        model.training_step({"x": data[rank, :4]}, 0).backward()              # the batch of 4 in one piece
        for w, p in zip(want, model.parameters()):
            w += p.grad / 2
    assert r0["fit_flush"] == (1, 2, 2, 2)
    for key in ("delta1", "delta3", "delta_fit_step", "delta_fit_step_offset"):
        for got0, got1, w in zip(r0[key], r1[key], want):
            assert torch.allclose(got0, w, rtol=1e-5, atol=1e-7), key
            assert torch.equal(got0, got1), key                               # both ranks hold the same averaged gradient
    assert r0["calls1"] == 1 and 1 < r0["calls3"] <= 3                        # per optimiser step, not per micro-batch (slices are parameter-aligned)
    assert r0["steps1"] == (1, 1) and r0["flush1"] == (2, 0) and r0["flush3"] == (2, 0)


def test_accumulated_micro_batches_on_the_network(golden, monkeypatch):
    """The same on the real network (CPU stand-ins for the HIP forwards, as in test_backward_wiring_matches_oracle_autograd): with
    accumulate_grad_batches = 2 the optimiser sees the MEAN of the two micro-batch gradients -- what Lightning's loss / k leaves in
    p.grad (train.py:57).  (A batch of two is NOT the reference for this network: model.py:515-525 adds the KL and cross-entropy
    terms summed over the WHOLE batch to every complex's loss, so the loss of a batch is not the mean of its complexes' losses.)"""
    case, z, args, params, pb = case_inputs(golden, "tiny")
    from protein_redesign_amd import torch_ref as R
    monkeypatch.setattr(training, "HipOp", _RefOp)
    monkeypatch.setattr(training, "tri_mul_update",
                        lambda tm, pair, mask, residual=False: R.triangle_multiplication(pair, mask, *tm.weights(), incoming=tm.mode == "incoming"))
    monkeypatch.setattr(training, "tri_attn_update",
                        lambda ta, pair, mask, residual=False: R.triangle_attention(pair, mask, *ta.attn.weights(), ta.attn.num_heads,
                                                                                    ta.attn.head_dim, ending=ta.mode == "ending"))
    sizes = [tuple(s) for s in case["sizes"]]
    assert len(sizes) >= 2
    t = torch.from_numpy(z["train_t"])[:2]
    nz, ns = torch.from_numpy(z["train_noise_z"])[:2], torch.from_numpy(z["train_noise_seq"])[:2]

    def fresh():
        m = ProteinReDiffModel(args)
        m.load_state_dict(params)
        m.nonfinite_policy = "off"
        return m

    full = synthetic_batch(sizes, esm_dim=args["esm_dim"], seed=case["batch_seed"], n_total=case["n_total"])

    def micro(k):
        return ({kk: (v[k:k + 1].clone() if torch.is_tensor(v) else v) for kk, v in full.items()},
                dict(t=t[k:k + 1], noise_z=nz[k:k + 1], noise_seq=ns[k:k + 1], sources=[NoiseSource(NOISE_SEED, 100 + k)]))

    singles = []
    for k in range(2):
        m = fresh()
        b, kw = micro(k)
        m.training_step(b, k, **kw).backward()
        singles.append([p.grad for p in m.parameters()])
    acc = fresh()
    fitter = training.Fitter(acc, torch.optim.SGD(acc.parameters(), lr=0.0), accumulate_grad_batches=2)
    for k in range(2):
        b, kw = micro(k)
        fitter.step(b, k, **kw)
    assert fitter.optimizer_steps == 1 and fitter.micro == 0
    want = [None if g0 is None else (g0 + g1) / 2 for g0, g1 in zip(*singles)]
    scale = float(torch.cat([w.reshape(-1) for w in want if w is not None]).double().norm())
    for (name, p), w in zip(acc.named_parameters(), want):
        if w is None:
            assert p.grad is None, name
            continue
        assert float((p.grad.double() - w.double()).norm()) < 1e-5 * float(w.double().norm()) + 1e-7 * scale, name


def _verdict_worker(rank, world, port, out_dir):
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stub = type("M", (), {"device": torch.device("cpu"), "nonfinite_group": None})()
    any_rank = ProteinReDiffModel._any_rank
    out = {"one_bad": any_rank(stub, rank == 1), "none_bad": any_rank(stub, False), "all_bad": any_rank(stub, True)}
    torch.save(out, os.path.join(out_dir, f"v{rank}.pt"))
    dist.destroy_process_group()


def test_nonfinite_verdict_is_agreed_over_the_ranks(tmp_path):
    """the fp32 fallback / NonFiniteError of training_step was decided from the LOCAL loss: one rank switched arithmetic (or
    raised before the gradient all-reduce) while the others went on.  The verdict is OR-ed over the group first (gloo, world size 2)."""
    port = 29400 + (os.getpid() % 200)
    mp.spawn(_verdict_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        v = torch.load(os.path.join(str(tmp_path), f"v{r}.pt"))
        assert v == {"one_bad": True, "none_bad": False, "all_bad": True}, (r, v)
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel
    stub = type("M", (), {"device": torch.device("cpu"), "nonfinite_group": None})()
    assert ProteinReDiffModel._any_rank(stub, True) is True and ProteinReDiffModel._any_rank(stub, False) is False   # no group: local
