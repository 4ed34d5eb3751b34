"""GPU: the optimisation step (SURVEY.md §8f "next" #1, BASELINE configs[3]).  ``training_step`` runs the HIP forward with one
autograd node per operator and per-block recompute (training.py); its loss and the gradient of EVERY trainable tensor are held
against (a) the fingerprints captured from the imported reference's ``training_step`` (tests/golden, oracle/gen_golden.py) and
(b) the oracle's autograd, tensor by tensor."""
import json
import math

import numpy as np
import pytest
import torch

import prd_oracle as O
from conftest import rel_l2
from protein_redesign_amd import training
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.diffusion_model import ProteinReDiffModel
from protein_redesign_amd.synthetic import NoiseSource, batch_to, deterministic_state_dict, synthetic_batch
from protein_redesign_amd.weights import spec_tensors
from test_training_cpu import FINGERPRINT_TOL, GRAD_PROJECTIONS, NOISE_SEED, case_inputs, oracle_grads

pytestmark = pytest.mark.gpu
DEV = "cuda"
GRAD_TOL = 1e-4


@pytest.fixture(params=["fp32", "split16"])
def gemm_mode(request):
    from protein_redesign_amd import _lib
    prev = _lib.lib().prd_get_gemm_mode()
    assert _lib.lib().prd_set_gemm_mode(_lib.GEMM_MODES[request.param]) == 0
    yield request.param
    assert _lib.lib().prd_set_gemm_mode(prev) == 0


def hip_model(args, params):
    model = ProteinReDiffModel(args)
    model.load_state_dict(params)
    model = model.to(DEV).train()
    model.run_setup_schedule()
    model.setup_schedule = True
    return model


@pytest.mark.parametrize("name", ["small32", "small64", "cfg1"])
def test_training_step_gradients_vs_reference_and_oracle(golden, name, gemm_mode, monkeypatch):
    case, z, args, params, pb = case_inputs(golden, name)
    if name != "small32":       # the pair-position weight gradients through prd_linear_wgrad also at these small row counts
        from protein_redesign_amd import ops
        monkeypatch.setattr(ops, "WGRAD_MIN_ROWS", 1)
    t = torch.from_numpy(z["train_t"])
    nz, ns = torch.from_numpy(z["train_noise_z"]), torch.from_numpy(z["train_noise_seq"])
    want_loss, want = oracle_grads(args, params, pb, t, nz, ns)
    model = hip_model(args, params)
    dpb = batch_to(pb, DEV)
    mask = dpb["residue_and_atom_mask"]
    diff = model.diffusion_loss(dpb, dpb["x"], mask, t.to(DEV), nz.to(DEV), ns.to(DEV))
    loss = torch.mean(diff / (mask > 0.5).sum(-1))
    loss.backward()
    assert abs(float(loss) - float(z["train_loss"])) < GRAD_TOL * abs(float(z["train_loss"]))
    got = {k: p.grad for k, p in model.named_parameters() if p.requires_grad}
    names = json.loads(str(z["train_grad_names"]))
    assert sorted(got) == sorted(names) and all(g is not None for g in got.values())
    scale = float(np.linalg.norm(z["train_grad_norm"]))
    worst = 0.0
    for i, k in enumerate(names):
        g = got[k].detach().cpu().double().reshape(-1)
        # (b) the whole tensor against the oracle's autograd; exactly-zero gradients hold round-off only
        err = float((g - want[k].double().reshape(-1)).norm())
        ref = float(want[k].double().norm())
        assert err < GRAD_TOL * ref + 1e-6 * scale, (k, err, ref)
        worst = max(worst, err / max(ref, 1e-3 * scale))
        # (a) norm and seeded projections against the imported reference's training_step
        n_ref = float(z["train_grad_norm"][i])
        assert abs(float(g.norm()) - n_ref) < FINGERPRINT_TOL * n_ref + 1e-6 * scale, (k, float(g.norm()), n_ref)
        for j in range(GRAD_PROJECTIONS):
            gen = torch.Generator().manual_seed(4242 + 16 * i + j)
            proj = float(torch.dot(g, torch.randn(g.numel(), generator=gen, dtype=torch.float64)))
            assert abs(proj - float(z["train_grad_proj"][i, j])) < FINGERPRINT_TOL * n_ref + 1e-6 * scale, (k, j)
    print(f"\n{name} [{gemm_mode}]: {len(names)} gradients, worst rel-L2 vs oracle autograd {worst:.2e}")


def test_training_step_api_and_optimizer_step():
    """training_step(batch, batch_idx) -> scalar loss (model.py:528-549), Adam + LinearLR + EMA as configure_optimizers /
    optimizer_step wire them (model.py:203-217): a few steps on one synthetic batch lower the loss and move the EMA."""
    args = make_args(single_dim=64, pair_dim=32, num_blocks=2, esm_dim=16, num_steps=50, mask_prob=0.3, learning_rate=1e-3,
                     warmup_steps=2)
    params = deterministic_state_dict(spec_tensors(args), seed=5, style="near_init")
    model = hip_model(args, params)
    cfg = model.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    batch = batch_to(synthetic_batch([(4, 18), (3, 14)], esm_dim=16, seed=6, n_total=24), DEV)
    g = torch.Generator().manual_seed(3)
    t = torch.tensor([11, 30], device=DEV)
    nz = O.remove_mean(torch.randn(2, 24, 3, generator=g), (batch["atom_mask"] + batch["residue_mask"]).cpu()).to(DEV)
    ns = O.remove_mean(torch.randn(2, 24, 21, generator=g), batch["residue_mask"].cpu()).to(DEV)
    shadow0 = [s.clone() for s in model.ema.shadow]
    losses = []
    for step in range(4):
        src = [NoiseSource(1, k) for k in range(2)]
        losses.append(float(training.fit_step(model, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, step,
                                              opt, sched, t=t, noise_z=nz, noise_seq=ns, sources=src)))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert model.ema.num_updates == 4
    assert any(not torch.equal(a, b) for a, b in zip(shadow0, model.ema.shadow))
    # free-running call (noise, t and mask drawn internally) returns a scalar with a graph
    loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0)
    assert loss.dim() == 0 and loss.requires_grad


@pytest.mark.parametrize("mode", ["outgoing", "incoming"])
@pytest.mark.parametrize("P", [32, 64])
def test_triangle_multiplication_backward_kernels(mode, P, gemm_mode):
    """The hand-written backward of TriangleMultiplication (prd_tri_mul_out_bwd / prd_tri_mul_contract / prd_tri_mul_proj_bwd)
    against the oracle's autograd: gradient with respect to the pair input and all eight weight tensors, ragged masked batch."""
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(70 + P)
    b, N = 2, 45
    pair = torch.randn(b, N, N, P, generator=g)
    mask = torch.ones(b, N)
    mask[1, 38:] = 0
    names = ["ab_proj.weight", "ab_proj.bias", "ab_gate.weight", "ab_gate.bias", "out_proj.weight", "out_proj.bias", "out_gate.weight", "out_gate.bias"]
    shapes = [(2 * P, P), (2 * P,), (2 * P, P), (2 * P,), (P, P), (P,), (P, P), (P,)]
    wts = [torch.randn(s, generator=g) / (math.sqrt(P) if len(s) == 2 else 4.0) for s in shapes]
    dy = torch.randn(b, N, N, P, generator=g)
    pl = pair.clone().requires_grad_(True)
    leaf = {"tm." + n: w.clone().requires_grad_(True) for n, w in zip(names, wts)}
    m2 = mask.unsqueeze(-1) * mask.unsqueeze(-2)
    out = O.triangle_multiplication(leaf, "tm", pl, m2, mode == "incoming")
    out.backward(dy)
    dpair, grads = ops.tri_mul_backward(dy.to(DEV), pair.to(DEV), mask.to(DEV), [w.to(DEV) for w in wts], incoming=mode == "incoming")
    assert rel_l2(dpair.cpu(), pl.grad) < 1e-5
    for n, got in zip(names, grads):
        assert rel_l2(got.cpu(), leaf["tm." + n].grad) < 1e-5, n


@pytest.mark.parametrize("mode", ["starting", "ending"])
@pytest.mark.parametrize("P,b,N", [(32, 2, 45), (64, 2, 45), (64, 1, 100), (64, 1, 352), (64, 1, 384)])
def test_triangle_attention_backward_kernels(mode, P, b, N, gemm_mode):
    """The hand-written backward of TriangleAttention (prd_tri_attn_bwd_core + row GEMMs + prd_ln_rows_bwd) against the oracle's
    autograd: gradient with respect to the pair input and all seven weight tensors; ragged masked batch, one fully masked row;
    N = 384 is the largest complex BASELINE configs[3] draws (round 4: the kernel holds rows up to training.TRI_ATTN_BWD_MAX_N =
    416 positions; the gate of a row is parked in its own output slot instead of LDS)."""
    from protein_redesign_amd import ops, training
    assert N <= training.TRI_ATTN_BWD_MAX_N
    g = torch.Generator().manual_seed(80 + P)
    H, c = 4, 16
    pair = torch.randn(b, N, N, P, generator=g)
    mask = torch.ones(b, N)
    mask[b - 1, N - 7:] = 0
    names = ["attn.q_proj.weight", "attn.k_proj.weight", "attn.v_proj.weight", "attn.gate_proj.weight", "attn.gate_proj.bias",
             "attn.out_proj.weight", "attn.out_proj.bias"]
    shapes = [(64, P), (64, P), (64, P), (64, P), (64,), (P, 64), (P,)]
    wts = [torch.randn(s, generator=g) / (math.sqrt(s[-1]) if len(s) == 2 else 4.0) for s in shapes]
    dy = torch.randn(b, N, N, P, generator=g)
    pl = pair.clone().requires_grad_(True)
    leaf = {"ta." + n: w.clone().requires_grad_(True) for n, w in zip(names, wts)}
    m2 = mask.unsqueeze(-1) * mask.unsqueeze(-2)
    out = O.triangle_attention(leaf, "ta", pl, m2, H, c, mode == "ending")
    out.backward(dy)
    dpair, grads = ops.tri_attn_backward(dy.to(DEV), pair.to(DEV), mask.to(DEV), [w.to(DEV) for w in wts], H, c, ending=mode == "ending")
    assert rel_l2(dpair.cpu(), pl.grad) < 1e-5
    for n, got in zip(names, grads):
        assert rel_l2(got.cpu(), leaf["ta." + n].grad) < 1e-5, n


@pytest.mark.gpu
def test_pair_track_backward_is_bit_reproducible(gemm_mode):
    """The hand-written backward of the pair-track operators at the training shape (2 complexes of N = 320: every persistent
    workgroup runs several tasks) gives bit-identical gradients when repeated -- no atomics, fixed reduction orders, and no
    timing-dependent result anywhere (a fused LayerNorm-backward tail that failed exactly this check was not shipped)."""
    from protein_redesign_amd import ops
    b, N, P, H, c = 2, 320, 64, 4, 16
    g = torch.Generator().manual_seed(5)
    pair = torch.randn(b, N, N, P, generator=g).to(DEV)
    dy = (torch.randn(b, N, N, P, generator=g) * 1e-3).to(DEV)
    mask = torch.ones(b, N)
    mask[1, N - 9:] = 0
    mask = mask.to(DEV)

    def w(*shape):
        return (torch.randn(*shape, generator=g) / math.sqrt(shape[-1] if len(shape) == 2 else 16.0)).to(DEV)

    ta = [w(64, P), w(64, P), w(64, P), w(64, P), w(64), w(P, 64), w(P)]
    tm = [w(2 * P, P), w(2 * P), w(2 * P, P), w(2 * P), w(P, P), w(P), w(P, P), w(P)]

    def run():
        d1, g1 = ops.tri_attn_backward(dy, pair, mask, ta, H, c, ending=False, residual=True)
        d2, g2 = ops.tri_mul_backward(dy, pair, mask, tm, incoming=True)
        return [d1, *g1, d2, *g2]

    first = [t.clone() for t in run()]
    for _ in range(3):
        for k, (a, r) in enumerate(zip(run(), first)):
            assert torch.equal(a, r), f"output {k} differs between runs"


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C", [(40 * 40 * 2, 64), (204800, 64), (9001, 32)])
def test_multi_table_embedding_gradient(rows, C):
    """prd_embed_wgrad_multi (several small tables looked up at the same rows, one pass over dy, per-set row scales) against the
    single-table kernel and a float64 scatter-sum; out-of-range indices are ignored; bit-identical when repeated."""
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    cards = [5, 6, 2, 8, 65]
    dy = torch.randn(rows, C, generator=g).to(DEV)
    idxs = [torch.randint(0, c, (rows,), generator=g).to(DEV) for c in cards]
    idxs[1][:7] = -1
    scales = [torch.rand(rows, generator=g).to(DEV), None, torch.rand(rows, generator=g).to(DEV), None, (torch.rand(rows, generator=g) > 0.3).float().to(DEV)]
    got = ops.embed_wgrad_multi(idxs, dy, cards, scales)
    for k, c in enumerate(cards):
        v = dy.double() * (scales[k].double().unsqueeze(1) if scales[k] is not None else 1.0)
        ok = idxs[k] >= 0
        want = torch.zeros(c, C, dtype=torch.float64, device=DEV).index_add_(0, idxs[k][ok], v[ok])
        assert got[k].shape == (c, C) and float((got[k].double() - want).norm() / want.norm()) < 2e-6, k
        one = ops.embed_wgrad(idxs[k], dy, c, scale=scales[k])
        assert float((got[k] - one).norm() / one.norm()) < 2e-6
    again = ops.embed_wgrad_multi(idxs, dy, cards, scales)
    assert all(torch.equal(a, b_) for a, b_ in zip(got, again))


@pytest.mark.gpu
@pytest.mark.parametrize("b,N", [(2, 45), (1, 320)])
def test_radial_basis_rows_and_pair_symmetrisation_kernels(b, N):
    """prd_rbf_rows against the reference's formula (modules.py:73-82: exp(-(R-1)/2 (|z_i - z_j| - c_r)^2), here times mask_i mask_j)
    and prd_sym_rows against scale * (x + x^T), bit-exact."""
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(N)
    z = (torch.randn(b, N, 3, generator=g) * 0.7).to(DEV)
    mask = (torch.rand(b, N, generator=g) > 0.2).float().to(DEV)
    centers = torch.linspace(0.0, 2.0, 256).to(DEV)
    got = ops.rbf_rows(z, centers, mask)
    dist = torch.linalg.norm(z.double().unsqueeze(-2) - z.double().unsqueeze(-3), dim=-1)
    want = torch.exp(-(255 / 2.0) * torch.square(dist.unsqueeze(-1) - centers.double())) * (mask.unsqueeze(-1) * mask.unsqueeze(-2)).unsqueeze(-1)
    assert got.shape == (b, N, N, 256) and float((got.double() - want).abs().max()) < 2e-5
    x = torch.randn(b, N, N, 64, generator=g).to(DEV)
    assert torch.equal(ops.sym_rows(x, 0.5), 0.5 * (x + x.transpose(1, 2)))


@pytest.mark.gpu
@pytest.mark.parametrize("R,P,S", [(90, 32, 64), (640, 64, 512), (301, 64, 130)])
def test_outer_linear_backward_reductions(R, P, S):
    """prd_outer_linear_bwd_reduce against float64: dx = sum_p T w1 and dw1 = sum_r T x over T [R, P, S]; repeatable bit for bit."""
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(R + S)
    T = torch.randn(R, P, S, generator=g).to(DEV)
    w1 = torch.randn(P, 2 * S, generator=g).to(DEV)[:, :S]              # a column block, as the backward passes it
    x = torch.randn(R, S, generator=g).to(DEV)
    dx, dw1 = ops.outer_linear_bwd_reduce(T, w1, x)
    wdx = (T.double() * w1.double()).sum(1)
    wdw = (T.double() * x.double().unsqueeze(1)).sum(0)
    assert float((dx.double() - wdx).norm() / wdx.norm()) < 1e-6 and float((dw1.double() - wdw).norm() / wdw.norm()) < 1e-6
    dx2, dw2 = ops.outer_linear_bwd_reduce(T, w1, x)
    assert torch.equal(dx, dx2) and torch.equal(dw1, dw2)


@pytest.mark.gpu
@pytest.mark.parametrize("b,N,P", [(2, 45, 64), (1, 130, 32), (2, 320, 64)])
def test_symmetrised_transpose_kernel(b, N, P):
    """prd_sym_transpose: out[b,i,p,j] = dy[b,i,j,p] + dy[b,j,i,p] (the operand of the outer-linear backward's GEMM), bit-exact
    against the torch expression; N not a multiple of 4 / of the 64-position tile."""
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(N + P)
    dy = torch.randn(b, N, N, P, generator=g).to(DEV)
    want = (dy + dy.transpose(1, 2)).permute(0, 1, 3, 2).contiguous()
    got = ops.sym_transpose(dy)
    assert got.shape == (b, N, P, N) and torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [8192 + 7, 204800])
@pytest.mark.parametrize("case", ["64to64", "64to256_ln_relu", "64to256_mask", "256to64"])
def test_pair_position_linear_row_kernel(case, rows):
    """prd_pair_linear (weights resident in LDS, rows streamed, split-16 arithmetic) against float64: the activation-gradient GEMMs of
    the attention / transition backward with their fused neighbours -- LayerNorm of the input rows (and the normalised rows as a
    side output), bias + ReLU, the ReLU mask from recomputed activations; a row count that is not a multiple of the 32-row task,
    and one where every wave runs several tasks; bit-identical between runs."""
    from protein_redesign_amd import _lib, ops
    g = torch.Generator().manual_seed(len(case) + rows)
    K, OUT = (256, 64) if case.startswith("256") else ((64, 256) if "256" in case else (64, 64))
    x = torch.randn(rows, K, generator=g) * 1.7 + 0.3
    w = torch.randn(OUT, K, generator=g) / math.sqrt(K)
    bias = torch.randn(OUT, generator=g) * 0.1 if "relu" in case else None
    hmask = torch.randn(rows, OUT, generator=g).clamp_min(0) if "mask" in case else None
    xd = x.double()
    if "ln" in case:
        xd = torch.nn.functional.layer_norm(xd, (K,))
    want = xd @ w.double().t() + (bias.double() if bias is not None else 0)
    if "relu" in case:
        want = want.clamp_min(0)
    if hmask is not None:
        want = want * (hmask.double() > 0)
    prev = _lib.lib().prd_get_gemm_mode()
    assert _lib.lib().prd_set_gemm_mode(_lib.GEMM_MODES["split16"]) == 0
    try:
        xn = torch.empty(rows, K, device=DEV) if "ln_relu" in case else None
        dev_args = (x.to(DEV), w.to(DEV), bias.to(DEV) if bias is not None else None)
        kw = dict(ln_in="ln_relu" in case, xn_out=xn, act=1 if "relu" in case else 0, relu_mask=hmask.to(DEV) if hmask is not None else None)
        got = ops.pair_linear(*dev_args, **kw)
        assert got is not None and got.shape == (rows, OUT)
        for _ in range(3):
            assert torch.equal(got, ops.pair_linear(*dev_args, **kw))
        # the weight as a transposed view of a [K, OUT] matrix (what the backward of a linear passes): staged as it lies in memory
        w_view = dev_args[1].t().contiguous().t()
        assert not w_view.is_contiguous() and torch.equal(got, ops.pair_linear(dev_args[0], w_view, dev_args[2], **kw))
        assert _lib.lib().prd_set_gemm_mode(_lib.GEMM_MODES["fp32"]) == 0
        assert ops.pair_linear(x.to(DEV), w.to(DEV)) is None           # fp32 arithmetic: the caller's GEMM path
    finally:
        assert _lib.lib().prd_set_gemm_mode(prev) == 0
    assert float((got.double().cpu() - want).norm() / want.norm()) < 2e-6
    if xn is not None:
        assert float((xn.double().cpu() - xd).norm() / xd.norm()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1e-6, 1e-9, 1e3, 1e-30])
@pytest.mark.parametrize("case", ["64to64", "64to256_mask", "256to64", "256to64_spread"])
def test_pair_position_linear_operand_range(case, scale):
    """the backward calls prd_pair_linear on RAW gradient rows (dy W2, g W1, d og, dqkvg W), whose entries are
    1e-4 ... 1e-9 near a minimum; fp16 hi | lo operands carry an absolute 3e-8 error below 6e-5 and flush below 6e-8.  Every 32-row
    task row is therefore normalised by an exact power of two before the split (per 64-channel piece with a running scale for
    K = 256) -- held here against float64 at the tolerance of the O(1) test, for rows scaled by 1e-6, 1e-9, 1e3 and 1e-30, with
    three decades between neighbouring rows, an all-zero row, and (``spread``) eight decades between the four pieces of a row."""
    from protein_redesign_amd import _lib, ops
    rows = 8192 + 7
    g = torch.Generator().manual_seed(len(case) + 17)
    K, OUT = (256, 64) if case.startswith("256") else ((64, 256) if "256" in case else (64, 64))
    x = torch.randn(rows, K, generator=g) * scale
    x[1::3] *= 1e-3                                              # decades between the rows of one 32-row task
    x[5] = 0.0
    if "spread" in case:
        x[:, :64] *= 1e-8
        x[:, 64:128] *= 1e-4
        x[7::2, 192:] *= 1e-6                                    # the running scale both rises and stays along a row
    w = torch.randn(OUT, K, generator=g) / math.sqrt(K)
    hmask = torch.randn(rows, OUT, generator=g).clamp_min(0) if "mask" in case else None
    want = x.double() @ w.double().t()
    if hmask is not None:
        want = want * (hmask.double() > 0)
    prev = _lib.lib().prd_get_gemm_mode()
    assert _lib.lib().prd_set_gemm_mode(_lib.GEMM_MODES["split16"]) == 0
    try:
        got = ops.pair_linear(x.to(DEV), w.to(DEV), relu_mask=hmask.to(DEV) if hmask is not None else None)
        assert got is not None and torch.equal(got, ops.pair_linear(x.to(DEV), w.to(DEV), relu_mask=hmask.to(DEV) if hmask is not None else None))
    finally:
        assert _lib.lib().prd_set_gemm_mode(prev) == 0
    got = got.double().cpu()
    assert torch.isfinite(got).all() and float(got[5].abs().max()) == 0.0
    # row by row: a small row must be as accurate as a large one (a whole-tensor norm would only see the large rows)
    err = (got - want).norm(dim=1) / want.norm(dim=1).clamp_min(1e-300)
    keep = want.norm(dim=1) > 0
    assert float(err[keep].max()) < 4e-6, (case, scale, float(err[keep].max()))


@pytest.mark.gpu
@pytest.mark.parametrize("O,I", [(256, 64), (64, 256), (64, 64), (128, 64)])
@pytest.mark.parametrize("profile", ["tiny", "huge", "rising", "falling", "spike", "zeros_then_data", "all_zero"])
def test_linear_weight_gradient_range_of_the_gradient(O, I, profile):
    """The split-16 weight-gradient kernel multiplies dy by a running power of two per wave (fp16 has five exponent bits, gradients no
    fixed magnitude).  Gradients of 1e-30 and 1e+20, magnitudes that rise or fall by sixty octaves along the rows of a slab (the
    scale is lowered on the fly / stays where the large rows put it), a single row 2^40 above the rest, leading all-zero rows, an
    all-zero gradient: the result against a float64 reduction, relative to the largest entry of dW."""
    from protein_redesign_amd import _lib, ops
    rows = 24000
    g = torch.Generator().manual_seed(O + I + len(profile))
    dy = torch.randn(rows, O, generator=g, dtype=torch.float64)
    x = torch.randn(rows, I, generator=g)
    t = torch.linspace(0, 1, rows, dtype=torch.float64).view(-1, 1)
    if profile == "tiny":
        dy = dy * 1e-30
    elif profile == "huge":
        dy = dy * 1e20
    elif profile == "rising":
        dy = dy * torch.exp2(-30 + 60 * ((t * 32) % 1.0))        # sixty octaves up inside every ~750-row stretch (slabs are ~94 rows)
    elif profile == "falling":
        dy = dy * torch.exp2(30 - 60 * ((t * 32) % 1.0))
    elif profile == "spike":
        dy[rows // 2 + 5] *= 2.0 ** 40
    elif profile == "zeros_then_data":
        dy[:4000] = 0
    elif profile == "all_zero":
        dy = dy * 0
    dy = dy.float().cuda()
    x = x.cuda()
    prev = _lib.lib().prd_get_gemm_mode()
    assert _lib.lib().prd_set_gemm_mode(_lib.GEMM_MODES["split16"]) == 0
    try:
        got, db = ops.linear_wgrad(dy, x, bias=True)
    finally:
        assert _lib.lib().prd_set_gemm_mode(prev) == 0
    want = dy.double().t() @ x.double()
    wantb = dy.double().sum(0)
    assert torch.isfinite(got).all() and torch.isfinite(db).all()
    if profile == "all_zero":
        assert float(got.abs().max()) == 0.0 and float(db.abs().max()) == 0.0
        return
    assert float((got.double() - want).norm() / want.norm()) < 2e-6
    assert float((got.double() - want).abs().max() / want.abs().max()) < 2e-6
    assert float((db.double() - wantb).norm() / wantb.norm()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("ending", [False, True])
@pytest.mark.parametrize("P,b,N,gscale", [(64, 2, 45, 1.0), (32, 1, 100, 1e-6), (64, 1, 320, 1e-4), (64, 1, 384, 1e3), (64, 2, 33, 0.0)])
def test_triangle_attention_backward_core_split16_vs_fp32(P, b, N, gscale, ending):
    """prd_tri_attn_bwd_core_v2 (16-bit matrix pipe, split operands, power-of-two scaling of the incoming gradient) against
    prd_tri_attn_bwd_core (fp32 MFMA) on the same inputs: d(W_q x) | d(W_k x) | d(W_v x) | d(gate), for gradients of very
    different magnitude (the scaling is per position), positions that receive no gradient at all, masked keys, a fully masked row."""
    from protein_redesign_amd import ops
    from protein_redesign_amd._lib import check, dptr, lib, stream
    g = torch.Generator().manual_seed(7 * N + P)
    H, c = 4, 16
    pair = torch.randn(b, N, N, P, generator=g).to(DEV)
    mask = torch.ones(b, N)
    mask[b - 1, N - 7:] = 0
    mask[0, 3] = 0
    mask = mask.to(DEV)
    wq, wk, wv, wg = [(torch.randn(64, P, generator=g) / math.sqrt(P)).to(DEV) for _ in range(4)]
    bg = (torch.randn(64, generator=g) / 4).to(DEV)
    dog = torch.randn(b, N, N, 64, generator=g) * gscale
    dog *= torch.logspace(-4, 0, N).view(1, 1, N, 1)            # four decades between the positions of a row
    dog[:, :, 5] = 0                                            # a position without gradient
    dog = dog.to(DEV)
    lse = torch.empty(b * N, H, N, 2, device=DEV)
    og = ops.tri_attn_core_v2_lse(pair, mask, (wq, wk, wv, wg, bg), H, c, ending=ending, lse=lse)
    r = torch.full((b, N, N, 4, 64), float("nan"), device=DEV)
    assert lib().prd_tri_attn_bwd_core_v2_supported(N, P) == 1
    check(lib().prd_tri_attn_bwd_core(dptr(r), dptr(dog), dptr(pair), dptr(mask), dptr(wq), dptr(wk), dptr(wv), dptr(wg), dptr(bg),
                                      int(ending), b, N, P, H, c, stream()), "fp32")
    for stats in (None, lse):                                   # statistics recomputed in the kernel / kept by the forward
        a = torch.full((b, N, N, 4, 64), float("nan"), device=DEV)
        check(lib().prd_tri_attn_bwd_core_v2(dptr(a), dptr(dog), dptr(og), dptr(pair), dptr(mask), dptr(wq), dptr(wk), dptr(wv), dptr(wg), dptr(bg),
                                             dptr(stats) if stats is not None else None, None, int(ending), b, N, P, H, c, stream()), "v2")
        assert torch.isfinite(a).all()
        for k, name in enumerate(["dq", "dk", "dv", "dgate"]):
            x, y = a[..., k, :].double(), r[..., k, :].double()
            if gscale == 0.0:
                assert float(x.abs().max()) == 0.0 and float(y.abs().max()) == 0.0, name
            else:
                assert float((x - y).norm() / y.norm()) < 3e-6, (name, stats is not None, float((x - y).norm() / y.norm()))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,O,I", [(40 * 40 * 2, 64, 64), (102400, 256, 64), (20000, 64, 256), (9001, 128, 128), (8192, 256, 256),
                                      (102400, 4, 64), (20001, 1, 64), (9000, 12, 128)])
def test_linear_weight_gradient_kernel(rows, O, I, gemm_mode):
    """prd_linear_wgrad (slab partials on fp32 MFMA, or in split-16 mode on the 16-bit matrix pipe, + ordered reduction) against a
    float64 reduction; also through strided views (a column slice of a wider tensor, as the attention backward passes them)."""
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(rows + O)
    wide = torch.randn(rows, O + 64, generator=g).cuda()
    dy = wide[:, 64:]                                   # row stride O + 64, offset 64
    x = torch.randn(rows, I, generator=g).cuda()
    got = ops.linear_wgrad(dy, x)
    want = (dy.double().t() @ x.double())
    err = (got.double() - want).norm() / want.norm()
    assert got.shape == (O, I) and err < 2e-6, err
    got2, db = ops.linear_wgrad(dy, x, bias=True)      # the bias gradient from the same pass
    wantb = dy.double().sum(0)
    assert db.shape == (O,) and (db.double() - wantb).norm() / wantb.norm() < 2e-6
    if rows >= ops.WGRAD_MIN_ROWS:                      # bit-reproducible (no atomics)
        assert torch.equal(got, ops.linear_wgrad(dy, x)) and torch.equal(got, got2)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,card,C", [(102400, 65, 64), (20011, 5, 64), (9000, 128, 32)])
def test_embedding_table_gradient_kernel(rows, card, C):
    """prd_embed_wgrad against index_add in float64; bit-reproducible."""
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(rows + card)
    idx = torch.randint(0, card, (rows,), generator=g)
    dy = torch.randn(rows, C, generator=g)
    want = torch.zeros(card, C, dtype=torch.float64).index_add_(0, idx, dy.double())
    got = ops.embed_wgrad(idx.cuda(), dy.cuda(), card)
    assert got.shape == (card, C) and (got.double().cpu() - want).norm() / want.norm() < 2e-6
    assert torch.equal(got, ops.embed_wgrad(idx.cuda(), dy.cuda(), card))


@pytest.mark.gpu
def test_full_size_backward_directional_derivative():
    """BASELINE configs[3]'s per-GPU share (2 complexes of N = 320, single_dim 512, pair_dim 64, 4 blocks): at this size every
    hand-written backward path is the one that runs (attention backward at N = 320, weight / bias / embedding-table gradients
    at 2 x 10^5 rows) and the oracle's autograd is out of reach on the host, so the check is size-independent: the gradient
    contracted with a random parameter direction equals the central finite difference of the HIP forward loss along it."""
    from protein_redesign_amd.constants import make_args
    args = make_args(single_dim=512, pair_dim=64, num_blocks=4, num_steps=1000, mask_prob=0.3)
    params = deterministic_state_dict(spec_tensors(args), seed=9, style="near_init")
    model = hip_model(args, params)
    batch = batch_to(synthetic_batch([(64, 256)] * 2, seed=4), DEV)
    N = 320
    g = torch.Generator().manual_seed(17)
    t = torch.tensor([400, 77], device=DEV)
    nz = O.remove_mean(torch.randn(2, N, 3, generator=g), (batch["atom_mask"] + batch["residue_mask"]).cpu()).to(DEV)
    ns = O.remove_mean(torch.randn(2, N, 21, generator=g), batch["residue_mask"].cpu()).to(DEV)

    def loss_of():
        src = [NoiseSource(2, k) for k in range(2)]
        return model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, t=t, noise_z=nz,
                                   noise_seq=ns, sources=src)

    loss = loss_of()
    loss.backward()
    named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for _, p in named)
    gd = torch.Generator().manual_seed(23)
    dirs = [torch.randn(p.shape, generator=gd).to(DEV) * (p.detach().norm() / math.sqrt(p.numel()) + 1e-3) for _, p in named]
    analytic = sum(float((p.grad.double() * d.double()).sum()) for (_, p), d in zip(named, dirs))
    eps = 1e-3              # measured: the difference quotient is stable to 1e-3 relative for eps in [2.5e-4, 4e-3]
    vals = []
    with torch.no_grad():
        for sgn in (1.0, -1.0):
            for (_, p), d in zip(named, dirs):
                p.add_(d, alpha=sgn * eps)
            vals.append(float(loss_of().detach().double()))
            for (_, p), d in zip(named, dirs):
                p.add_(d, alpha=-sgn * eps)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    print(f"\nfull-size directional derivative: analytic {analytic:.6e}  finite difference {numeric:.6e}  loss {float(loss):.4f}")
    assert abs(analytic) > 1.0 and abs(analytic - numeric) <= 5e-3 * abs(analytic)


def test_gradient_with_respect_to_the_coordinates(golden, gemm_mode):
    """the hand-written backwards of the input stage and of the heads return no gradient for z; a caller who needs
    d loss / d z (guidance, a gradient check on the positions) must get it from the differentiable restatement, not a silent None:
    d(<eps, a> + <logits, b>) / dz against the oracle's autograd."""
    case, zf, args, params, pb = case_inputs(golden, "small64")
    model = hip_model(args, params)
    dpb = batch_to(pb, DEV)
    mask = pb["residue_and_atom_mask"]
    g = torch.Generator().manual_seed(12)
    b, N = mask.shape
    z0 = O.remove_mean(torch.randn(b, N, 3, generator=g), mask)
    seq_t = torch.randn(b, N, 21, generator=g)
    t = torch.tensor([5] * b)
    a_, b_ = torch.randn(b, N, 3, generator=g), torch.randn(b, N, 21, generator=g)
    zo = z0.clone().requires_grad_(True)
    eps_o, log_o = O.network_step(params, args, pb, zo, seq_t, mask, t)
    ((eps_o * a_).sum() + (log_o * b_).sum()).backward()
    zh = z0.to(DEV).requires_grad_(True)
    eps_h, log_h = model(dpb, zh, seq_t.to(DEV), mask.to(DEV), t.to(DEV))
    ((eps_h * a_.to(DEV)).sum() + (log_h * b_.to(DEV)).sum()).backward()
    assert zh.grad is not None
    assert rel_l2(zh.grad.cpu(), zo.grad) < GRAD_TOL
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)


@pytest.mark.gpu
@pytest.mark.parametrize("H,affine", [(4, False), (8, True)])
def test_attention_bias_backward_in_one_pass(H, affine):
    """prd_pair_bias_bwd (dLN = dbias . W', LayerNorm backward, LN(x) rows and dbias by position in one pass over the pair rows)
    against float64 autograd of bias = (W diag(gamma)) LN(pair) (+ c), the weight gradients through PairBiasFn, and against the
    round-4 composition (permute copy + GEMM + LayerNorm-backward pass + LayerNorm pass)."""
    from protein_redesign_amd import ops
    g = torch.Generator().manual_seed(100 + H)
    b, N, P = 2, 96, 64
    pair = (torch.randn(b, N, N, P, generator=g) * 1.7 + 0.3).cuda()
    w = (torch.randn(H, P, generator=g) / 8).cuda()
    c = torch.randn(H, generator=g).cuda() if not affine else None
    gamma = (1 + 0.2 * torch.randn(P, generator=g)).cuda() if affine else None
    beta = (0.1 * torch.randn(P, generator=g)).cuda() if affine else None
    dbias = torch.randn(b, H, N, N, generator=g).cuda() * 1e-3
    leaves = [t.double().detach().requires_grad_(True) for t in (pair, w) + ((gamma, beta) if affine else (c,))]
    x64 = torch.nn.functional.layer_norm(leaves[0], (P,), leaves[2] if affine else None, leaves[3] if affine else None)
    bias64 = torch.einsum("bijc,hc->bhij", x64, leaves[1]) + (0 if affine else leaves[2].view(1, H, 1, 1))
    want = torch.autograd.grad(bias64, leaves, dbias.double())
    outs = {}
    for fused in (True, False):
        training.PAIR_BIAS_BWD = fused
        try:
            p_, w_ = pair.clone().requires_grad_(True), w.clone().requires_grad_(True)
            extra = [t.clone().requires_grad_(True) for t in ((gamma, beta) if affine else (c,))]
            out = training.PairBiasFn.apply(p_, w_, None, *extra) if affine else training.PairBiasFn.apply(p_, w_, extra[0])
            outs[fused] = torch.autograd.grad(out, [p_, w_] + extra, dbias)
        finally:
            training.PAIR_BIAS_BWD = True
    for got, ref in zip(outs[True], want):
        assert rel_l2(got.double().cpu(), ref.cpu()) < 2e-6
    for a_, b_ in zip(outs[True], outs[False]):
        assert rel_l2(a_.cpu(), b_.cpu()) < 2e-6
    fused = ops.pair_bias_bwd(dbias, w * gamma if affine else w, pair)
    assert fused is not None and torch.equal(fused[2], dbias.permute(0, 2, 3, 1).reshape(-1, H))
