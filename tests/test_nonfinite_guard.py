"""The split-16 default must not hand back NaNs with return code 0 (missing #2): the reference is fp32 end to end
(model.py:377-422) and cannot overflow at 65504; PRD_ARITH_SPLIT16 can (include/prd_hip.h, OPERAND RANGE).  ``ab_proj`` x 1e5 in one
triangle multiplication (contraction operands of magnitude 1e5) is driven through ``model.sample()``, ``model.training_step()`` and the
``Fitter`` loop: with the default policy the call is repeated under PRD_ARITH_FP32 and meets the oracle at the suite's tolerances;
with policy "raise" it fails naming the arithmetic; nothing is ever returned non-finite."""
import warnings

import pytest
import torch

import prd_oracle as O
from conftest import rel_l2
from protein_redesign_amd import _lib, training
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.diffusion_model import ProteinReDiffModel
from protein_redesign_amd.synthetic import NoiseSource, batch_to, clone_batch, deterministic_state_dict, synthetic_batch
from protein_redesign_amd.weights import spec_tensors
from test_training_cpu import oracle_training_loss

pytestmark = pytest.mark.gpu
DEV = "cuda"
SCALE = 1e5
PFX = "Denoiser.folding_blocks.0.pair_mul_outgoing.ab_proj"


@pytest.fixture
def split16():
    prev = _lib.lib().prd_get_gemm_mode()
    assert _lib.lib().prd_set_gemm_mode(1) == 0
    yield
    assert _lib.lib().prd_set_gemm_mode(prev) == 0


def out_of_range_model(num_steps=6, train=False):
    args = make_args(single_dim=64, pair_dim=64, num_blocks=2, esm_dim=16, num_steps=num_steps, mask_prob=0.4)
    params = dict(deterministic_state_dict(spec_tensors(args), seed=31))        # full-strength random weights: ab_proj x 1e5 -> operands ~1e5
    params[PFX + ".weight"] = params[PFX + ".weight"] * SCALE
    params[PFX + ".bias"] = params[PFX + ".bias"] * SCALE
    model = ProteinReDiffModel(args)
    model.load_state_dict(params)
    model = model.to(DEV)
    return (model.train() if train else model.eval()), args, params


def test_sample_repeats_in_fp32_and_meets_the_oracle(split16):
    model, args, params = out_of_range_model()
    one = synthetic_batch([(5, 27)], esm_dim=16, seed=3)
    want_pos, want_logits = O.sample(params, args, clone_batch(one), [NoiseSource(9, 0)])
    assert torch.isfinite(want_pos).all()
    with pytest.warns(RuntimeWarning, match="PRD_ARITH_FP32"):
        pos, logits = model.sample(batch_to(clone_batch(one), DEV), sources=[NoiseSource(9, 0)])
    assert torch.isfinite(pos).all() and torch.isfinite(logits).all()
    assert rel_l2(pos.cpu(), want_pos) < 1e-4 and rel_l2(logits.cpu(), want_logits) < 1e-4
    assert model.arith_fallbacks == 1 and model.arithmetic == "fp32"
    assert _lib.lib().prd_get_gemm_mode() == 1, "the repeat must not leak the arithmetic into the process default"
    # later calls of this model go straight to fp32: no second warning, same result
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pos2, _ = model.sample(batch_to(clone_batch(one), DEV), sources=[NoiseSource(9, 0)])
    assert torch.equal(pos, pos2) and model.arith_fallbacks == 1


def test_sample_policy_raise_names_the_arithmetic(split16):
    model, args, params = out_of_range_model()
    model.nonfinite_policy = "raise"
    one = batch_to(synthetic_batch([(5, 27)], esm_dim=16, seed=3), DEV)
    with pytest.raises(_lib.NonFiniteError, match="split16"):
        model.sample(one, sources=[NoiseSource(9, 0)])
    # the sticky flag of the step-boundary kernel alone is enough (the loop's state is not inspected when it is set)
    from protein_redesign_amd.diffusion_model import ReverseDiffusion
    loop = ReverseDiffusion(model, batch_to(synthetic_batch([(5, 27)], esm_dim=16, seed=3), DEV), [NoiseSource(9, 0)])
    loop.run()
    assert int(loop.sync[1]) == 1 and not loop.finite()
    # an in-range model leaves the flag alone
    model.nonfinite_policy = "off"
    assert not torch.isfinite(model.sample(batch_to(synthetic_batch([(5, 27)], esm_dim=16, seed=3), DEV), sources=[NoiseSource(9, 0)])[0]).all()


def test_in_range_model_never_trips(split16):
    args = make_args(single_dim=64, pair_dim=64, num_blocks=2, esm_dim=16, num_steps=6, mask_prob=0.4)
    model = ProteinReDiffModel(args)
    model.load_state_dict(deterministic_state_dict(spec_tensors(args), seed=31))
    model = model.to(DEV).eval()
    model.nonfinite_policy = "raise"
    from protein_redesign_amd.diffusion_model import ReverseDiffusion
    loop = ReverseDiffusion(model, batch_to(synthetic_batch([(5, 27), (3, 20)], esm_dim=16, seed=3), DEV), [NoiseSource(9, k) for k in range(2)])
    loop.run()
    assert int(loop.sync[1]) == 0 and loop.finite()


def _oracle_step(args, params):
    """One training step of the out-of-range network on the oracle: (batch, t, noises, leaf tensors with .grad, loss)."""
    batch = synthetic_batch([(4, 18), (3, 14)], esm_dim=16, seed=6, n_total=24)
    perms = [NoiseSource(1, k).randperm(n) for k, n in enumerate((18, 14))]
    pb = O.prepare_batch(clone_batch(batch), args["mask_prob"], perms)
    g = torch.Generator().manual_seed(3)
    t = torch.tensor([11, 30])
    nz = O.remove_mean(torch.randn(2, 24, 3, generator=g), pb["residue_and_atom_mask"])
    ns = O.remove_mean(torch.randn(2, 24, 21, generator=g), pb["residue_mask"])
    leaf = {k: v.clone().requires_grad_(k not in ("embed_beta.0.weight", "embed_dist.0.center")) for k, v in params.items()}
    want = oracle_training_loss(leaf, args, pb, t, nz, ns)
    want.backward()
    assert torch.isfinite(want)
    return batch, t, nz, ns, leaf, want


def _assert_gradients_meet_the_oracle(model, leaf, what):
    worst = 0.0
    scale = float(torch.cat([v.grad.reshape(-1) for v in leaf.values() if v.grad is not None]).double().norm())
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), (what, k)
        w = leaf[k].grad.double()
        err = float((p.grad.detach().cpu().double() - w).norm())
        assert err < 1e-4 * float(w.norm()) + 1e-6 * scale, (what, k, err, float(w.norm()))
        worst = max(worst, err / max(float(w.norm()), 1e-3 * scale))
    print(f"\n{what}: worst gradient rel-L2 vs oracle autograd {worst:.2e}")


@pytest.mark.parametrize("checkpoint", [False, True])
def test_model_pinned_to_fp32_runs_its_backward_in_fp32(split16, checkpoint, monkeypatch):
    """``model.arithmetic = "fp32"`` used to cover the forward only -- autograd runs the backward after training_step's
    ``with`` has exited, and the recompute / the hand-written backward kernels then read the split-16 process default: finite loss, NaN
    gradients for exactly the models the pin is for.  Every node now carries the arithmetic of its forward.  Tolerance as everywhere:
    each gradient <= 1e-4 of its norm against the oracle's autograd."""
    monkeypatch.setattr(training, "USE_CHECKPOINT", checkpoint)
    model, args, params = out_of_range_model(num_steps=50, train=True)
    model.arithmetic = "fp32"
    model.nonfinite_policy = "raise"
    batch, t, nz, ns, leaf, want = _oracle_step(args, params)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        loss = model.training_step(batch_to(clone_batch(batch), DEV), 0, t=t.to(DEV), noise_z=nz.to(DEV), noise_seq=ns.to(DEV),
                                   sources=[NoiseSource(1, k) for k in range(2)])
    assert abs(float(loss.detach()) - float(want.detach())) < 1e-4 * abs(float(want.detach()))
    assert _lib.lib().prd_get_gemm_mode() == 1          # the process default is split-16 while the backward runs
    loss.backward()
    assert _lib.lib().prd_get_gemm_mode() == 1
    _assert_gradients_meet_the_oracle(model, leaf, f"model pinned to fp32, checkpoint={checkpoint}")


def test_sample_fallback_then_training_stays_in_fp32(split16):
    """sample()'s fallback pins the model; a training step of the same object afterwards must be fp32 end to end (no second warning,
    finite gradients that meet the oracle)."""
    model, args, params = out_of_range_model(num_steps=50)
    one = synthetic_batch([(5, 27)], esm_dim=16, seed=3)
    with pytest.warns(RuntimeWarning, match="PRD_ARITH_FP32"):
        model.sample(batch_to(clone_batch(one), DEV), sources=[NoiseSource(9, 0)])
    assert model.arithmetic == "fp32"
    model.train()
    batch, t, nz, ns, leaf, want = _oracle_step(args, params)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        loss = model.training_step(batch_to(clone_batch(batch), DEV), 0, t=t.to(DEV), noise_z=nz.to(DEV), noise_seq=ns.to(DEV),
                                   sources=[NoiseSource(1, k) for k in range(2)])
        loss.backward()
    assert model.arith_fallbacks == 1
    _assert_gradients_meet_the_oracle(model, leaf, "sample() fallback, then training_step")


def test_training_step_repeats_in_fp32_and_meets_the_oracle(split16):
    model, args, params = out_of_range_model(num_steps=50, train=True)
    batch, t, nz, ns, leaf, want = _oracle_step(args, params)
    try:
        with pytest.warns(RuntimeWarning, match="PRD_ARITH_FP32"):
            loss = model.training_step(batch_to(clone_batch(batch), DEV), 0, t=t.to(DEV), noise_z=nz.to(DEV), noise_seq=ns.to(DEV),
                                       sources=[NoiseSource(1, k) for k in range(2)])
        assert torch.isfinite(loss) and abs(float(loss.detach()) - float(want.detach())) < 1e-4 * abs(float(want.detach()))
        assert model.arithmetic == "fp32" and model.arith_fallbacks == 1
        assert _lib.lib().prd_get_gemm_mode() == 1, "the repeat pins the MODEL; the process default stays (the nodes carry their arithmetic)"
        loss.backward()
        _assert_gradients_meet_the_oracle(model, leaf, "repeated training step")
    finally:
        _lib.lib().prd_set_gemm_mode(1)


def test_training_step_policy_raise(split16):
    model, args, params = out_of_range_model(num_steps=50, train=True)
    model.nonfinite_policy = "raise"
    batch = batch_to(synthetic_batch([(4, 18), (3, 14)], esm_dim=16, seed=6, n_total=24), DEV)
    with pytest.raises(_lib.NonFiniteError, match="split16"):
        model.training_step(batch, 0)


def test_fitter_skips_the_step_on_the_device_and_moves_to_fp32(split16):
    """No host round trip per step: the non-finite step is skipped on the device (fused Adam's found_inf), the flag is read one
    step later, the parameters were never touched, and the loop goes on in fp32."""
    model, args, params = out_of_range_model(num_steps=50, train=True)
    cfg = model.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    assert opt.defaults.get("fused")
    batch = batch_to(synthetic_batch([(4, 18), (3, 14)], esm_dim=16, seed=6, n_total=24), DEV)
    fitter = training.Fitter(model, opt, sched)
    before = [p.detach().clone() for p in model.parameters()]
    try:
        l0 = fitter.step(clone_batch(batch), 0)
        assert not torch.isfinite(l0)
        assert all(torch.equal(a, b) for a, b in zip(before, model.parameters())), "a non-finite step must not move the parameters"
        with pytest.warns(RuntimeWarning, match="PRD_ARITH_FP32"):
            l1 = fitter.step(clone_batch(batch), 1)
        assert fitter.skipped_steps == 1 and model.arithmetic == "fp32" and _lib.lib().prd_get_gemm_mode() == 1
        l2 = fitter.step(clone_batch(batch), 2)
        assert torch.isfinite(l1) and torch.isfinite(l2)
        assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
        assert all(torch.isfinite(p).all() for p in model.parameters())
    finally:
        _lib.lib().prd_set_gemm_mode(1)
