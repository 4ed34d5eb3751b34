"""Differentiable network for the optimisation step (reference model.py:490-549, modules.py:391-404) -- SURVEY.md §8f "next" #1.

Every operator of the trunk becomes one ``torch.autograd.Function``: its FORWARD is the HIP operator of the inference path
(ops.py -> libprd_hip.so, out of place); its BACKWARD is a hand-written HIP backward (TriMulFn, TriAttnFn; csrc/prd_bwd.hip) or
-- ``HipOp``, the survey's first cut -- recomputes the operator with the differentiable torch-op restatement in torch_ref.py on
the GPU and lets autograd produce the gradients of its inputs and weights (the weight gradients of the pair-position linears
again on a HIP kernel).  The reference's per-block ``torch.utils.checkpoint`` (modules.py:399-401) is available
(``USE_CHECKPOINT``) but off by default.

fp32 throughout (the reference trains under fp16 autocast, train.py:37; parity is defined against its fp32 arithmetic).
"""
from __future__ import annotations

import math
import os

from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch.utils.checkpoint import checkpoint

from . import _lib
from . import ops
from . import torch_ref as R


# Per-block activation checkpointing (reference modules.py:399-401).  With it, between blocks only (single, pair) are kept, a
# block's forward is re-run on the HIP kernels inside the backward pass, and every operator recomputes what its backward needs.
# Without it the operators' inputs stay alive (six pair tensors per block, 26 MB each per complex at N = 320) and the triangle
# operators also keep their intermediates (operands a | b and the contraction output; the gated head outputs) instead of
# recomputing them: peak memory of a 2-complex step 1.9 -> 3.9 GB -- nothing against 288 GB -- and 56.8 -> 47.9 ms per step.
# Off by default; PRD_TRAIN_CHECKPOINT=1 restores the reference's memory behaviour (same gradients either way).
USE_CHECKPOINT = os.environ.get("PRD_TRAIN_CHECKPOINT", "0") == "1"
# 0: the backward of the pair transition, the attention bias and the outer-linear through the torch restatement (HipOp), as in
# round 2 -- A/B measurements only
LIBRARY_BWD = os.environ.get("PRD_LIBRARY_BWD", "1") != "0"
HEADS_BWD = os.environ.get("PRD_HEADS_BWD", "1") != "0"                 # 0: the heads' backward through the torch restatement (A/B)
INPUT_STAGE_BWD = os.environ.get("PRD_INPUT_STAGE_BWD", "1") != "0"     # 0: the input stage's backward through its torch restatement (A/B)
# 1: the residual adds of a folding block's pair updates ride in the kernels' own residual paths (pair + update written by the
# operator, dy added to its input gradient) instead of eight torch adds over the pair tensor per block; 0: A/B measurements
FUSED_RESIDUAL = os.environ.get("PRD_TRAIN_FUSED_RESIDUAL", "1") != "0"
# 0: the attention-bias backward as permute copy + K = H GEMM + LayerNorm-backward pass + LayerNorm pass (A/B; round 4)
PAIR_BIAS_BWD = os.environ.get("PRD_PAIR_BIAS_BWD", "1") != "0"


def _forward_arith() -> Optional[int]:
    try:
        return _lib.arith()
    except RuntimeError:            # library not built (host-only tests of the restatement): nothing to pin
        return None


def _pin_arithmetic(cls):
    """Class decorator for the operator nodes below: the arithmetic (prd_hip.h PRD_ARITH_*) in force when the node's FORWARD ran is
    recorded in ``ctx`` and put back in force around its BACKWARD.  The binding injects the *current* default into every library
    call, and autograd runs a backward long after the ``with _lib.arithmetic(...)`` of ``training_step`` has exited: without the pin a
    model held to "fp32" (``model.arithmetic``, or sample()'s fallback) ran its recompute and its hand-written backward kernels in
    split-16 -- the operand overflow the pin is there to avoid came back in the gradients."""
    import functools
    f, b = cls.__dict__["forward"].__func__, cls.__dict__["backward"].__func__

    @functools.wraps(f)
    def forward(ctx, *args, **kwargs):
        ctx.prd_arith = _forward_arith()
        return f(ctx, *args, **kwargs)

    @functools.wraps(b)
    def backward(ctx, *gouts):
        with _lib.arithmetic(getattr(ctx, "prd_arith", None)):
            return b(ctx, *gouts)

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


@_pin_arithmetic
class HipOp(torch.autograd.Function):
    """``HipOp.apply(fwd, ref, *tensors)``: ``fwd(*tensors)`` runs HIP kernels (no autograd), ``ref(*tensors)`` is the same
    operator in differentiable torch ops and is only evaluated inside ``backward``.  Both return one tensor or a tuple."""

    @staticmethod
    def forward(ctx, fwd: Callable, ref: Callable, *tensors):
        ctx.ref = ref
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            out = fwd(*[t.detach() for t in tensors])
        return out

    @staticmethod
    def backward(ctx, *gouts):
        need = ctx.needs_input_grad[2:]
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(True) if (n and t.is_floating_point()) else t.detach()
                   for t, n in zip(ctx.saved_tensors, need)]
            outs = ctx.ref(*ins)
            outs = outs if isinstance(outs, tuple) else (outs,)
            pairs = [(o, g) for o, g in zip(outs, gouts) if g is not None]
            wrt = [i for i in ins if i.requires_grad]
            grads = torch.autograd.grad([o for o, _ in pairs], wrt, [g.contiguous() for _, g in pairs], allow_unused=True)
        it = iter(grads)
        return (None, None) + tuple(next(it) if i.requires_grad else None for i in ins)


@_pin_arithmetic
class InputStageFn(torch.autograd.Function):
    """The input stage (model.py:332-361) with a hand-written backward of its PAIR half.  Forward: the HIP input kernels.  Backward:
    the single half (a few [b, N, S] operators) through its torch restatement; the pair half without an autograd graph over
    [b, N, N, *] tensors -- the three kinds of small tables (bond features, bond distance, relative position) by the slab scatter-sum
    prd_embed_wgrad with the forward's mask factors as a row scale, the distance embedding's weight as dy^T rbf with the radial-basis
    rows materialised once by prd_rbf_rows (they do not depend on a parameter), the time-embedding term from the masked sum of dy.
    (The torch restatement recomputed and differentiated a dozen [b, N, N, 64 / 256] tensors: 2.05 ms of a 25 ms step.)"""

    @staticmethod
    def forward(ctx, fwd: Callable, meta: dict, z, seq_t, *params):
        ctx.meta = meta
        ctx.save_for_backward(z, seq_t, *params)
        with torch.no_grad():
            return fwd(z.detach(), seq_t.detach(), *[p.detach() for p in params])

    @staticmethod
    def backward(ctx, dsingle, dpair):
        m = ctx.meta
        batch, mask, t, na, nb = m["batch"], m["mask"], m["t"], m["na"], m["nb"]
        z, seq_t, *params = ctx.saved_tensors
        need = ctx.needs_input_grad[2:]
        atom_tabs, bond_tabs = params[:na], params[na:na + nb]
        bd_tab, rp_tab, w_rt, w_esm, centers, w_dist, freqs, w_beta = params[na + nb:]
        grads = [None] * (2 + len(params))
        # ---- single half: torch restatement of [b, N, S]-sized operators ----
        if dsingle is not None:
            with torch.enable_grad():
                s_in = seq_t.detach().requires_grad_(True) if need[1] else seq_t.detach()
                a_in = [p.detach().requires_grad_(True) for p in atom_tabs]
                wr, we = w_rt.detach().requires_grad_(True), w_esm.detach().requires_grad_(True)
                single = R.input_stage_single(batch, s_in, a_in, wr, we)
                wrt = ([s_in] if need[1] else []) + a_in + [wr, we]
                g = list(torch.autograd.grad(single, wrt, dsingle.contiguous(), allow_unused=True))
            if need[1]:
                grads[1] = g.pop(0)
            for k in range(na):
                grads[2 + k] = g[k]
            grads[2 + na + nb + 2], grads[2 + na + nb + 3] = g[na], g[na + 1]
        # ---- pair half ----
        if dpair is not None:
            with torch.no_grad():
                b, N = mask.shape
                P = dpair.shape[-1]
                dyp = dpair.contiguous()
                dy2 = dyp.view(-1, P)
                am, rm = batch["atom_mask"], batch["residue_mask"]
                am2 = am.unsqueeze(-1) * am.unsqueeze(-2)
                ri, ci = batch["residue_index"], batch["residue_chain_index"]
                rel = (ri.unsqueeze(-1) - ri.unsqueeze(-2)).clamp(min=-m["max_relpos"], max=m["max_relpos"]) + m["max_relpos"]
                s_rel = (rm.unsqueeze(-1) * rm.unsqueeze(-2) * (ci.unsqueeze(-1) == ci.unsqueeze(-2)).float()).contiguous().view(-1)
                s_bond = (am2 * batch["bond_mask"] * (1.0 / math.sqrt(nb))).contiguous().view(-1)
                idxs = [batch["bond_feats"][..., f].contiguous().view(-1) for f in range(nb)]
                idxs += [batch["bond_distance"].clamp(max=m["max_bond_distance"]).contiguous().view(-1), rel.contiguous().view(-1)]
                cards = [tab.shape[0] for tab in (*bond_tabs, bd_tab, rp_tab)]
                scales = [s_bond] * nb + [am2.contiguous().view(-1), s_rel]
                if nb + 2 <= 8 and sum(cards) <= 128:           # all table gradients in one pass over dy
                    tg = ops.embed_wgrad_multi(idxs, dy2, cards, scales)
                else:
                    tg = [ops.embed_wgrad(i, dy2, c_, scale=s_) for i, c_, s_ in zip(idxs, cards, scales)]
                for f in range(nb + 2):
                    grads[2 + na + f] = tg[f]
                rbf = ops.rbf_rows(z.detach(), centers.detach(), mask)                    # [b, N, N, R], mask_i mask_j inside
                grads[2 + na + nb + 5] = ops.linear_wgrad(dy2, rbf.view(-1, rbf.shape[-1]))
                m2 = (mask.unsqueeze(-1) * mask.unsqueeze(-2)).unsqueeze(-1)
                S = (dyp * m2).sum(dim=(1, 2))                                            # [b, P]: gradient of the per-complex time term
            with torch.enable_grad():
                fr, wb = freqs.detach().requires_grad_(freqs.requires_grad), w_beta.detach().requires_grad_(True)
                wx = fr * (t / m["num_steps"])[:, None, None].unsqueeze(-1)
                sinus = torch.cat([torch.sin(wx), torch.cos(wx)], dim=-1)
                term = torch.nn.functional.linear(sinus, wb)                             # [b, 1, 1, P]
                gb = torch.autograd.grad(term, [fr, wb] if fr.requires_grad else [wb], S.view_as(term), allow_unused=True)
            if fr.requires_grad:
                grads[2 + na + nb + 6], grads[2 + na + nb + 7] = gb[0], gb[1]
            else:
                grads[2 + na + nb + 7] = gb[0]
        return (None, None, *grads)


@_pin_arithmetic
class HeadsFn(torch.autograd.Function):
    """The two heads (modules.py:403 + model.py:364-374) with a hand-written backward of the coordinate head.  Forward: the HIP
    kernels.  Backward: the sequence head (single-sized) through its torch restatement; the coordinate head without an autograd graph
    over pair-sized tensors -- psym = (pair + pair^T) / 2 (prd_sym_rows), h = relu(W1 LN(psym) + b1) recomputed by the row kernel,
    dw[b,i,j] = mask_i mask_j r_ij . d eps_i (r = the unit difference vectors), g = dw w2 [h > 0], dLN = g W1, LayerNorm backward,
    symmetrisation of the result; weight gradients from the slab reductions."""

    @staticmethod
    def forward(ctx, fwd: Callable, z, mask, single, pair, *w):
        ctx.save_for_backward(z, mask, single, pair, *w)
        with torch.no_grad():
            return fwd(single.detach(), pair.detach(), *[x.detach() for x in w])

    @staticmethod
    def backward(ctx, deps, dlogits):
        z, mask, single, pair, wr1, br1, wr2, ws1, bs1, ws2 = ctx.saved_tensors
        gs = gp = None
        gw = [None] * 6
        if dlogits is not None:
            with torch.enable_grad():
                s_in = single.detach().requires_grad_(True)
                ws = [x.detach().requires_grad_(True) for x in (ws1, bs1, ws2)]
                g = torch.autograd.grad(R.seq_head(s_in, *ws), [s_in, *ws], dlogits.contiguous())
            gs, gw[3], gw[4], gw[5] = g
        if deps is not None:
            with torch.no_grad():
                b, N, _, P = pair.shape
                m = mask.unsqueeze(-1)
                de = deps - m * (m * deps).sum(dim=1, keepdim=True) / m.sum(dim=1, keepdim=True)      # remove_mean is self-adjoint
                zij = z.unsqueeze(-2) - z.unsqueeze(-3)
                r = zij * torch.rsqrt(torch.sum(torch.square(zij), -1, keepdim=True) + 1e-4)
                dw = (mask.unsqueeze(-1) * mask.unsqueeze(-2)) * (r * de.unsqueeze(2)).sum(-1)         # [b, N, N]
                psym = ops.sym_rows(pair.detach().contiguous(), 0.5).view(-1, P)
                xn = torch.empty_like(psym)
                h = ops.pair_linear(psym, wr1, br1, ln_in=True, xn_out=xn, act=1)
                if h is None:
                    h = ops.linear(psym, wr1, br1, act=1, ln_a=True, ln_a_out=xn)
                dw2 = dw.reshape(-1, 1)
                gw[2] = (dw2 * h).sum(0, keepdim=True)                                                  # d wr2 [1, HID]
                g = torch.where(h > 0, dw2 * wr2, torch.zeros((), device=h.device))                     # [rows, HID]
                dxn = ops.pair_linear(g, wr1.t())
                if dxn is None:
                    dxn = ops.linear(g, wr1.t().contiguous())
                dps = ops.ln_rows_bwd(dxn, psym)
                gp = ops.sym_rows(dps.view(b, N, N, P), 0.5)
                gw[0], gw[1] = ops.linear_wgrad(g, xn, bias=True)
        return (None, None, None, gs, gp, *gw)


@_pin_arithmetic
class PairTransitionFn(torch.autograd.Function):
    """pair_fc (modules.py:321-326): y = W2 relu(W1 LN(x) + b1) + b2 at every pair position.  Forward: the row kernel of the
    inference path.  Backward entirely on the library's kernels (no torch restatement, no autograd graph over [b N N, 256]
    activations): h = relu(W1 LN(x) + b1) recomputed by the LayerNorm-fused GEMM, g = (dy W2) * [h > 0], dLN = g W1,
    dx = LN'(dLN; x) (prd_ln_rows_bwd), dW2 | db2 = dy^T h, dW1 | db1 = g^T LN(x) (prd_linear_wgrad slab reductions)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual=False):
        ctx.save_for_backward(x, w1, b1, w2, b2)
        ctx.residual = residual
        with torch.no_grad():
            return ops.pair_transition(x.detach().contiguous(), w1, b1, w2, b2, residual=residual)

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, b2 = ctx.saved_tensors
        P, HID = x.shape[-1], w1.shape[0]
        with torch.no_grad():
            x2 = x.detach().contiguous().view(-1, P)
            dy2 = dy.contiguous().view(-1, P)
            xn = torch.empty_like(x2)
            res = dy2 if ctx.residual else None                                # (+ dy: the residual path)
            h = ops.pair_linear(x2, w1, b1, ln_in=True, xn_out=xn, act=1)      # [rows, HID]; LN(x) rows on the side
            if h is None:
                h = ops.linear(x2, w1, b1, act=1, ln_a=True, ln_a_out=xn)
            g = ops.pair_linear(dy2, w2.t(), relu_mask=h)                      # (dy W2) * [h > 0]
            if g is None:
                g = ops.linear(dy2, w2.t().contiguous(), relu_mask=h)
            dxn = ops.pair_linear(g, w1.t())                                   # [rows, P]
            if dxn is None:
                dxn = ops.linear(g, w1.t().contiguous())
            dx = ops.ln_rows_bwd(dxn, x2, res=res)
            dw2, db2 = ops.linear_wgrad(dy2, h, bias=True)
            dw1, db1 = ops.linear_wgrad(g, xn, bias=True)
            dx = dx.view_as(x)
        return dx, dw1, db1, dw2, db2, None


@_pin_arithmetic
class PairBiasFn(torch.autograd.Function):
    """attn_bias of a folding block (modules.py:300-304): bias[b,h,i,j] = (W LN(pair[b,i,j]) + c)[h], and SPAttention's pair bias
    (AF2_modules.py:454-459): the same with an affine LayerNorm (gamma, beta) and no c.  Backward on the library's kernels:
    dLN = dbias (W gamma) (a K = H GEMM), dpair = LN'(dLN; pair), dW' | dc' = dbias^T LN(pair) from the narrow slab reduction; with
    the affine LayerNorm folded as W' = W diag(gamma), c' = c + W beta:  dW = dW' gamma + dc' (x) beta,  dgamma = sum_h W dW',
    dbeta = W^T dc'."""

    @staticmethod
    def forward(ctx, pair, w, c, gamma=None, beta=None):
        ctx.affine = gamma is not None
        ctx.has_c = c is not None
        ctx.save_for_backward(pair, w, *([gamma, beta] if ctx.affine else []))
        with torch.no_grad():
            return ops.pair_bias(pair.detach().contiguous(), w, c, gamma, beta) if ctx.affine else ops.pair_bias(pair.detach().contiguous(), w, c)

    @staticmethod
    def backward(ctx, dbias):
        pair, w, *aff = ctx.saved_tensors
        P, H = pair.shape[-1], w.shape[0]
        with torch.no_grad():
            x2 = pair.detach().contiguous().view(-1, P)
            wf = w * aff[0] if ctx.affine else w                              # W diag(gamma)
            fused = ops.pair_bias_bwd(dbias, wf, pair.detach()) if PAIR_BIAS_BWD else None
            if fused is not None:                                             # one pass over the pair rows (prd_pair_bias_bwd)
                dx, xn, d2 = fused
            else:
                d2 = dbias.permute(0, 2, 3, 1).contiguous().view(-1, H)       # [rows, H]
                dxn = d2 @ wf                                                 # [rows, P]: K = H = 4, not a matrix-pipe shape
                dx = ops.ln_rows_bwd(dxn, x2)
                xn = ops.layer_norm(x2)
            dwf, dcf = ops.linear_wgrad(d2, xn, bias=True)
            if not ctx.affine:
                return dx.view_as(pair), dwf, (dcf if ctx.has_c else None), None, None
            gamma, beta = aff
            dw = dwf * gamma + dcf.unsqueeze(1) * beta
            dgamma = (w * dwf).sum(0)
            dbeta = dcf @ w
        return dx.view_as(pair), dw, (dcf if ctx.has_c else None), dgamma, dbeta


@_pin_arithmetic
class OuterLinearFn(torch.autograd.Function):
    """OuterLinear (modules.py:283-287) in the split form out[i,j] = W1 (x_i * x_j) + u_i - u_j + c, x = LN(single), u = W2 x.
    Forward: the HIP kernels of the inference path.  Backward without re-running the forward and without a [b,N,N,S] tensor:
    T[b,i,p,s] = sum_j (dy[b,i,j,p] + dy[b,j,i,p]) x[b,j,s] (one batched GEMM), dx = sum_p W1[p,s] T + du W2,
    dW1 = 1/2 sum_{b,i} x T, du = sum_j dy[i,j] - sum_j dy[j,i], dW2 = du^T x, dc = sum dy, dsingle = LN'(dx; single)."""

    @staticmethod
    def forward(ctx, single, w, c, fwd, pair=None):
        ctx.save_for_backward(single, w)
        ctx.residual = pair is not None            # with ``pair``: returns pair + update (the kernel's residual path)
        with torch.no_grad():
            return fwd(single.detach(), w, c) if pair is None else fwd(single.detach(), w, c, pair.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        single, w = ctx.saved_tensors
        b, N, S = single.shape
        P = w.shape[0]
        with torch.no_grad():
            s2 = single.detach().contiguous()
            x = ops.layer_norm(s2)
            w1, w2 = w[:, :S], w[:, S:]
            if dy.is_cuda and P in (32, 64) and LIBRARY_BWD:
                dsym = ops.sym_transpose(dy.contiguous()).view(b, N * P, N)     # (dy + dy^T)[b, i, p, j] in one pass
            else:
                dsym = (dy + dy.transpose(1, 2)).permute(0, 1, 3, 2).reshape(b, N * P, N)
            if dy.is_cuda and ops.split16_gemm_ok(N * P, S, N):                 # the split-16 tile GEMM, one launch per complex
                xt = x.transpose(1, 2).contiguous()                             # [b, S, N]: B as [N_out][K]
                T = torch.empty(b, N * P, S, device=dy.device, dtype=torch.float32)
                for bb in range(b):
                    ops.gemm(dsym[bb], xt[bb], T[bb], N * P, S, N, N, N, S)
                T = T.view(b, N, P, S)
            else:
                T = torch.bmm(dsym, x).view(b, N, P, S)
            du = dy.sum(2) - dy.sum(1)                                          # [b, N, P]
            if dy.is_cuda and LIBRARY_BWD:                                      # both reductions over T without T-sized temporaries
                dxt, dw1 = ops.outer_linear_bwd_reduce(T.reshape(b * N, P, S), w1, x.reshape(b * N, S))
                dx = dxt.view(b, N, S) + du @ w2
                dw1 = 0.5 * dw1
            else:
                dx = (T * w1).sum(dim=2) + du @ w2
                dw1 = 0.5 * (T * x.unsqueeze(2)).sum(dim=(0, 1))
            dw2 = du.reshape(-1, P).t() @ x.reshape(-1, S)
            dc = dy.sum(dim=(0, 1, 2))
            dsingle = ops.ln_rows_bwd(dx.reshape(-1, S).contiguous(), s2.view(-1, S)).view_as(single)
        return dsingle, torch.cat([dw1, dw2], dim=1), dc, None, (dy if ctx.residual else None)


@_pin_arithmetic
class TriMulFn(torch.autograd.Function):
    """TriangleMultiplication update with a hand-written backward (csrc/prd_bwd.hip through ops.tri_mul_backward): output-stage and
    projection-stage row kernels, the two gradient contractions on the forward contraction kernel; only the weight-gradient
    reductions (tall-skinny GEMMs over all N^2 rows) go through the BLAS library.  Nothing is recomputed in torch ops."""

    @staticmethod
    def forward(ctx, pair, mask, incoming: bool, residual: bool, *wts):
        ctx.incoming = incoming
        ctx.residual = residual                    # True: returns pair + update from the kernel's own residual path (no torch add)
        ctx.keep_ws = not USE_CHECKPOINT           # the operands a | b and the contraction output (3 pair-sized tensors) stay for the backward
        with torch.no_grad():
            p = pair.detach().contiguous()
            b, N, _, P = p.shape
            ws = torch.empty(ops.workspace_bytes("tri_mul", b, N, 0, P) // 4, device=p.device, dtype=torch.float32) if ctx.keep_ws else None
            out = ops.tri_mul(p, mask, [w.detach() for w in wts], incoming=incoming, residual=residual, ws=ws)
        ctx.save_for_backward(pair, mask, *wts, *([ws] if ctx.keep_ws else []))
        return out

    @staticmethod
    def backward(ctx, dy):
        saved = list(ctx.saved_tensors)
        ws = saved.pop() if ctx.keep_ws else None
        pair, mask, *wts = saved
        with torch.no_grad():
            dpair, grads = ops.tri_mul_backward(dy, pair.detach().contiguous(), mask, [w.detach() for w in wts], incoming=ctx.incoming, ws=ws)
            if ctx.residual:
                dpair = dpair.add_(dy)
        return (dpair, None, None, None, *grads)


@_pin_arithmetic
class TriAttnFn(torch.autograd.Function):
    """TriangleAttention update with a hand-written backward (ops.tri_attn_backward: out-projection backward, flash-style attention
    core backward per (row, head), projection + LayerNorm backward on HIP kernels; weight-gradient reductions through BLAS).
    Rows longer than the backward core's LDS layout (N > ~400) fall back to the recompute-in-torch node (HipOp)."""

    @staticmethod
    def forward(ctx, pair, mask, ending: bool, H: int, c: int, residual: bool, *wts):
        ctx.cfg = (ending, H, c)
        ctx.residual = residual
        with torch.no_grad():
            p, w = pair.detach().contiguous(), [x.detach() for x in wts]
            # without per-block checkpointing the gated head outputs (64 floats per pair position) and the softmax statistics of
            # the queries (2 floats per query and head) are kept for the backward instead of being recomputed by a second core launch
            ctx.keep_og = not USE_CHECKPOINT
            b, N, _, P = p.shape
            lse = (torch.empty(b * N, H, N, 2, device=p.device, dtype=torch.float32)
                   if ctx.keep_og and p.is_cuda and ops.tri_attn_lse_supported(N, P) else None)
            og = ops.tri_attn_core(p, mask, w[:5], H, c, ending=ending, lse=lse)
            out = ops.tri_attn_out(p, og, w[5], w[6], residual=residual)
        ctx.has_lse = lse is not None
        ctx.save_for_backward(pair, mask, *wts, *([og] if ctx.keep_og else []), *([lse] if ctx.has_lse else []))
        return out

    @staticmethod
    def backward(ctx, dy):
        saved = list(ctx.saved_tensors)
        lse = saved.pop() if ctx.has_lse else None
        og = saved.pop() if ctx.keep_og else None
        pair, mask, *wts = saved
        ending, H, c = ctx.cfg
        with torch.no_grad():
            dpair, grads = ops.tri_attn_backward(dy, pair.detach().contiguous(), mask, [w.detach() for w in wts], H, c, ending=ending, og=og, lse=lse,
                                                 residual=ctx.residual)
        return (dpair, None, None, None, None, None, *grads)


TRI_ATTN_BWD_MAX_N = 416        # prd_tri_attn_bwd_core keeps q, k, v, do of a row (padded to 32) and the head's weights in LDS (the gate is
                                # parked in its own output slot): 156 KB at N = 416 (pitch 20 floats); BASELINE configs[3] draws N <= 384.
                                # Longer rows recompute through torch_ref


def tri_attn_update(ta, pair: torch.Tensor, mask: torch.Tensor, residual: bool = False) -> torch.Tensor:
    """The update, or with ``residual`` pair + update."""
    a = ta.attn
    H, c, end = a.num_heads, a.head_dim, ta.mode == "ending"
    if pair.shape[1] <= TRI_ATTN_BWD_MAX_N:
        return TriAttnFn.apply(pair, mask, end, H, c, residual, *a.weights())

    def ref(p, *w):
        return R.triangle_attention(p, mask, *w, H, c, ending=end)

    def hip(p, *w):
        return ops.tri_attn(p.contiguous(), mask, w, H, c, ending=end, residual=False)

    upd = HipOp.apply(hip, ref, pair, *a.weights())
    return pair + upd if residual else upd


def tri_mul_update(tm, pair: torch.Tensor, mask: torch.Tensor, residual: bool = False) -> torch.Tensor:
    return TriMulFn.apply(pair, mask, tm.mode == "incoming", residual, *tm.weights())


def _lin(m) -> Tuple[torch.Tensor, ...]:
    return (m.weight, m.bias) if m.bias is not None else (m.weight,)


# ---------------------------------------------------------------------------------------------------
# one FoldingBlock (modules.py:328-343), every update out of place
# ---------------------------------------------------------------------------------------------------

def folding_block(blk, single: torch.Tensor, pair: torch.Tensor, mask: torch.Tensor):
    sa, H, c = blk.single_attn, blk.single_attn.num_heads, blk.single_attn.head_dim
    wb, bb = blk.attn_bias[1].weight, blk.attn_bias[1].bias

    big = pair.is_cuda and pair.numel() // pair.shape[-1] >= ops.WGRAD_MIN_ROWS and LIBRARY_BWD
    if big:
        bias = PairBiasFn.apply(pair, wb, bb)
    else:
        bias = HipOp.apply(lambda p, w, b: ops.pair_bias(p.contiguous(), w, b), R.pair_bias, pair, wb, bb)

    def sa_ref(x, bias_, *w):
        return R.gated_attention(x, mask, *w, H, c, bias=bias_)

    def sa_hip(x, bias_, *w):
        packed = ops.pack_attention(*w[:5], sa.scale)
        return ops.gated_attention_single(x.contiguous(), mask, bias_.contiguous(), packed, w[5], w[6], H, c, key_mask=True, resid=None, ln_a=True)

    single = single + HipOp.apply(sa_hip, sa_ref, single, bias, *sa.weights())

    fc = blk.single_fc
    fcw = (fc[1].weight, fc[1].bias, fc[3].weight, fc[3].bias)
    single = single + HipOp.apply(lambda x, *w: ops.transition_single(x.contiguous(), *w, residual=False),
                                  R.transition, single, *fcw)

    ol = blk.outer_linear

    def ol_hip(x, w, b, pair_in=None):
        bsz, n, _ = x.shape
        out = torch.empty(bsz, n, n, ol.pair_dim, device=x.device, dtype=torch.float32)
        xn = ops.layer_norm(x.contiguous())
        u = torch.empty(bsz, n, ol.pair_dim, device=x.device, dtype=torch.float32)
        S = x.shape[-1]
        ops.gemm(xn, w, u, bsz * n, ol.pair_dim, S, S, 2 * S, ol.pair_dim, b_off=S)
        if pair_in is not None:
            return ops.outer_linear_pair(pair_in, xn, u, w, b, residual=True, out=out)
        return ops.outer_linear_pair(out, xn, u, w, b, residual=False, out=out)

    fuse = big and FUSED_RESIDUAL
    if fuse:
        pair = OuterLinearFn.apply(single, ol.linear.weight, ol.linear.bias, ol_hip, pair)
    elif big:
        pair = pair + OuterLinearFn.apply(single, ol.linear.weight, ol.linear.bias, ol_hip)
    else:
        pair = pair + HipOp.apply(ol_hip, R.outer_linear, single, ol.linear.weight, ol.linear.bias)

    for tm in (blk.pair_mul_outgoing, blk.pair_mul_incoming):
        pair = tri_mul_update(tm, pair, mask, residual=True) if fuse else pair + tri_mul_update(tm, pair, mask)

    for ta in (blk.pair_attn_starting, blk.pair_attn_ending):
        pair = tri_attn_update(ta, pair, mask, residual=True) if fuse else pair + tri_attn_update(ta, pair, mask)

    pf = blk.pair_fc
    pfw = (pf[1].weight, pf[1].bias, pf[3].weight, pf[3].bias)
    if fuse:
        pair = PairTransitionFn.apply(pair, *pfw, True)
    elif big:
        pair = pair + PairTransitionFn.apply(pair, *pfw)
    else:
        pair = pair + HipOp.apply(lambda x, *w: ops.pair_transition(x.contiguous(), *w, residual=False),
                                  R.transition, pair, *pfw)
    return single, pair


# ---------------------------------------------------------------------------------------------------
# the whole network (model.py:254-316), differentiable with respect to every trainable parameter
# ---------------------------------------------------------------------------------------------------

def network(model, batch: Dict[str, torch.Tensor], z: torch.Tensor, seq_t: torch.Tensor, mask: torch.Tensor, t: torch.Tensor,
            use_checkpoint: Optional[bool] = None):
    """(noise_pred [b,N,3], seq_pred [b,N,21]) with a backward; ``model`` is a ProteinReDiffModel."""
    if use_checkpoint is None:
        use_checkpoint = USE_CHECKPOINT
    mask = mask.contiguous()
    den = model.Denoiser
    H = den.num_heads

    # ---- input stage (model.py:332-361) ----
    atom_tabs = [e.weight for e in model.embed_atom_feats.embeddings]
    bond_tabs = [e.weight for e in model.embed_bond_feats.embeddings]
    na, nb = len(atom_tabs), len(bond_tabs)
    in_params = (*atom_tabs, *bond_tabs, model.embed_bond_distance.weight, model.embed_relpos.weight,
                 model.embed_residue_type[1].weight, model.embed_residue_esm[1].weight, model.embed_dist[0].center,
                 model.embed_dist[1].weight, model.embed_beta[0].weight, model.embed_beta[1].weight)

    def in_ref(z_, s_, *p):
        return R.input_stage(batch, z_, s_, mask, t, model.num_steps, model.max_bond_distance, model.max_relpos,
                             p[:na], p[na:na + nb], *p[na + nb:])

    def in_hip(z_, s_, *p):
        static = model._static_inputs(batch)
        rm = batch["residue_mask"].contiguous()
        single = ops.single_init(static["single"], s_.contiguous(), rm, model.embed_residue_type[1].weight)
        eb = ops.time_embed(t.contiguous(), model.embed_beta[0].weight, model.embed_beta[1].weight, model.num_steps)
        pair = ops.pair_init(static["pair"], z_.contiguous(), mask, model.embed_dist[0].center, model.embed_dist[1].weight, eb)
        return single, pair

    P_ = model.embed_dist[1].weight.shape[0]
    # (the hand-written backwards below return no gradient for the coordinates: a caller who differentiates with respect to z --
    # guidance, a gradient check on the positions -- gets the differentiable torch restatement instead of a silent None)
    hand = (z.is_cuda and not z.requires_grad and LIBRARY_BWD and INPUT_STAGE_BWD and P_ <= 64 and model.embed_dist[0].center.numel() % 4 == 0 and not model.embed_dist[0].center.requires_grad
            and all(tab.shape[0] <= 128 for tab in (*bond_tabs, model.embed_bond_distance.weight, model.embed_relpos.weight)))
    if hand:                                            # hand-written backward of the pair half (InputStageFn)
        meta = dict(batch=batch, mask=mask, t=t, na=na, nb=nb, num_steps=model.num_steps, max_bond_distance=model.max_bond_distance,
                    max_relpos=model.max_relpos)
        single, pair = InputStageFn.apply(in_hip, meta, z, seq_t, *in_params)
    else:
        single, pair = HipOp.apply(in_hip, in_ref, z, seq_t, *in_params)

    # ---- OuterProductUpdate (masked add) and SPAttention (modules.py:394-398) ----
    opm = den.opm
    opm_params = (opm.layer_norm.weight, opm.layer_norm.bias, *_lin(opm.linear_1), *_lin(opm.linear_2), *_lin(opm.linear_out))

    def opm_ref(s_, *p):
        return R.outer_product_update(s_, mask, *p)

    def opm_hip(s_, *p):
        bsz, n, _ = s_.shape
        dummy = torch.empty(bsz, n, n, opm.c_z, device=s_.device, dtype=torch.float32)
        return opm.run(s_.contiguous(), dummy, mask, residual=False, apply_mask=True, out=dummy)

    pair = pair + HipOp.apply(opm_hip, opm_ref, single, *opm_params)

    spa = den.SPAAttnBlock
    a = spa.mha
    spa_params = (spa.layer_norm_m.weight, spa.layer_norm_m.bias, spa.linear_z[0].weight, spa.linear_z[0].bias, spa.linear_z[1].weight,
                  a.linear_q.weight, a.linear_k.weight, a.linear_v.weight, a.linear_g.weight, a.linear_g.bias,
                  a.linear_o.weight, a.linear_o.bias)

    def spa_ref(s_, p_, *w):
        return R.single_pair_attention(s_, p_, *w, heads=H)

    def spa_hip(s_, p_, *w):
        s_, p_ = s_.contiguous(), p_.contiguous()
        mn_, qkvg_ = spa.project(s_)
        bias_ = spa.bias_from_pair(p_) if spa.pair_bias else torch.zeros(s_.shape[0], spa.no_heads, s_.shape[1], s_.shape[1], device=s_.device)
        return spa.attend(mn_, qkvg_, bias_, logits_fp32=True)

    if pair.is_cuda and LIBRARY_BWD and pair.numel() // pair.shape[-1] >= ops.WGRAD_MIN_ROWS and spa.pair_bias:
        # the pair bias as its own node with the hand-written backward (PairBiasFn, affine LayerNorm); the attention itself, whose
        # tensors are single-sized, keeps its torch restatement for the backward
        spa_bias = PairBiasFn.apply(pair, spa.linear_z[1].weight, None, spa.linear_z[0].weight, spa.linear_z[0].bias)
        att_params = (spa_params[0], spa_params[1], *spa_params[5:])

        def att_ref(s_, bias_, *w):
            return R.single_bias_attention(s_, bias_, *w, heads=H)

        def att_hip(s_, bias_, *w):
            mn, qkvg = spa.project(s_.contiguous())
            return spa.attend(mn, qkvg, bias_.contiguous(), logits_fp32=True)     # explicit: not inferred from the train / eval flag

        single = HipOp.apply(att_hip, att_ref, single, spa_bias, *att_params)
    else:
        single = HipOp.apply(spa_hip, spa_ref, single, pair, *spa_params)

    # ---- folding blocks, each under activation checkpointing like the reference (modules.py:399-401) ----
    for blk in den.folding_blocks:
        if use_checkpoint and pair.requires_grad:
            def run_block(s_, p_, blk=blk, mode=_forward_arith()):
                # the recompute inside the backward pass runs in the arithmetic of the forward (see _pin_arithmetic)
                with _lib.arithmetic(mode):
                    return folding_block(blk, s_, p_, mask)
            single, pair = checkpoint(run_block, single, pair, use_reentrant=False)
        else:
            single, pair = folding_block(blk, single, pair, mask)

    # ---- symmetrisation + heads (modules.py:403, model.py:364-374) ----
    wr, sm = model.weight_radial, model.seq_mlp
    head_params = (wr[1].weight, wr[1].bias, wr[3].weight, sm[1].weight, sm[1].bias, sm[3].weight)

    def heads_ref(s_, p_, *w):
        return R.heads(s_, p_, z, mask, *w)

    def heads_hip(s_, p_, *w):
        eps_raw = ops.coord_head(p_.contiguous(), z.contiguous(), mask, w[0], w[1], w[2])
        eps = ops.remove_mean(eps_raw, mask)
        h = ops.linear(s_.contiguous(), w[3], w[4], act=1, ln_a=True)
        return eps, ops.linear(h, w[5])

    P_h = pair.shape[-1]
    if (pair.is_cuda and not z.requires_grad and LIBRARY_BWD and HEADS_BWD and P_h % 64 == 0 and wr[1].weight.shape[0] % 64 == 0
            and wr[3].weight.shape[0] == 1):
        return HeadsFn.apply(heads_hip, z, mask, single, pair, *head_params)
    if z.requires_grad:             # the coordinates as an INPUT of the node: the restatement differentiates them too (r_ij and the distances)
        return HipOp.apply(lambda s_, p_, z_, *w: heads_hip(s_, p_, *w), lambda s_, p_, z_, *w: R.heads(s_, p_, z_, mask, *w),
                           single, pair, z, *head_params)
    return HipOp.apply(heads_hip, heads_ref, single, pair, *head_params)


# ---------------------------------------------------------------------------------------------------
# data-parallel gradient averaging (train.py:34-50: DDP over the GPUs of a node)
# ---------------------------------------------------------------------------------------------------

_FLAT_GRAD_BUFFERS = {}      # (device, numel) -> persistent flat fp32 buffer (no 65 MB allocation per step)


def _flat_views(params: Sequence[torch.nn.Parameter]):
    """(flat buffer, per-parameter views into it, the gradients as flat views).  view(-1), not reshape: a gradient in a
    non-viewable layout must raise here -- reshape would hand back a COPY and the averaged values would land in a temporary
    while p.grad stayed un-averaged."""
    grads = [p.grad.view(-1) for p in params]
    n = sum(g.numel() for g in grads)
    key = (grads[0].device, n)
    flat = _FLAT_GRAD_BUFFERS.get(key)
    if flat is None:
        flat = _FLAT_GRAD_BUFFERS[key] = torch.empty(n, device=grads[0].device, dtype=torch.float32)
    return flat, list(torch.split(flat, [g.numel() for g in grads])), grads


def all_reduce_gradients(params: Sequence[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None, slices: int = 1,
                         on_slice: Optional[Callable[[int, int], None]] = None) -> None:
    """Average the gradients of ``params`` over the ranks of ``group``: ONE all-reduce of the flattened gradient (16.3 M fp32 =
    65 MB for the reference configuration; RCCL over xGMI with backend "nccl" -- a single large message instead of DDP's 25 MB
    buckets: per-link bound rings want few, large collectives) in a persistent buffer.  Every trainable parameter must have a
    gradient on every rank -- the contract of the reference's ``strategy="ddp_find_unused_parameters_false"`` (train.py:38), under
    which DDP raises as well; materialising zeros instead would make Adam decay the moments of a parameter that took no part
    in the step.  A one-rank group still runs the collective (the code path is the same at every world size).  No-op without
    an initialised process group.

    ``slices`` > 1: the flat buffer goes out as that many consecutive all-reduces, all enqueued at once (asynchronously) on the
    communication stream; ``on_slice(first_param, end_param)`` is called on the compute stream as soon as slice k has come back
    and been unpacked, i.e. while slices k + 1 ... are still on the wire -- the hook for work that only needs the reduced gradients
    of those parameters (tools/train_bench.py --overlap measures an optimiser step per slice against the single message)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    params = [p for p in params if p.requires_grad]
    missing = [i for i, p in enumerate(params) if p.grad is None]
    if missing:
        raise RuntimeError(f"all_reduce_gradients: {len(missing)} trainable parameter(s) have no gradient (first index "
                           f"{missing[0]}); like DDP with find_unused_parameters=False this is an error")
    flat, views, grads = _flat_views(params)
    torch._foreach_copy_(views, grads)
    if slices <= 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if world > 1:
            flat.div_(world)
        torch._foreach_copy_(grads, views)
        if on_slice is not None:
            on_slice(0, len(params))
        return
    # parameter-aligned slices of about equal size
    sizes = [g.numel() for g in grads]
    target, bounds, acc = sum(sizes) / slices, [0], 0
    for i, n in enumerate(sizes):
        acc += n
        if acc >= target * len(bounds) and len(bounds) < slices:
            bounds.append(i + 1)
    if bounds[-1] != len(params):
        bounds.append(len(params))
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    works = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        works.append(dist.all_reduce(flat[offs[a]:offs[b]], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for (a, b), w in zip(zip(bounds[:-1], bounds[1:]), works):
        w.wait()                                            # stream-level wait for THIS slice only; the later ones stay in flight
        part = flat[offs[a]:offs[b]]
        if world > 1:
            part.div_(world)
        torch._foreach_copy_(grads[a:b], views[a:b])
        if on_slice is not None:
            on_slice(a, b)


def _grads_finite(params: Sequence[torch.nn.Parameter]) -> torch.Tensor:
    """0-dim bool on the device: every gradient finite (one multi-tensor norm, no host sync)."""
    grads = [p.grad for p in params if p.grad is not None]
    return torch.isfinite(torch.stack(torch._foreach_norm(grads)).sum())


class Fitter:
    """Optimisation driver the way ``train.py`` drives the model through Lightning's Trainer (train.py:34-57, model.py:203-217,
    528-549), without Lightning: per micro-batch ``training_step`` -> backward of loss / k; per k = ``accumulate_grad_batches``
    micro-batches (train.py:57; every published training command uses 8 or 10, README.md:136-169) ONE gradient average over the
    data-parallel ranks -> Adam step -> LinearLR step -> EMA update.  Gradients accumulate locally in ``p.grad`` in between:
    k times fewer collectives than a step per micro-batch, and the reference's optimiser trajectory.

    Non-finite guard without a host round trip (the model's ``nonfinite_policy``, diffusion_model.py): the optimiser step is
    SKIPPED ON THE DEVICE when the accumulated loss or any gradient is inf / NaN (fused Adam's ``found_inf``, the GradScaler
    protocol), the flag is read back one optimiser step LATER -- when it has long been written -- and then the process switches
    to PRD_ARITH_FP32 (policy "fp32", with a warning; the skipped step is lost, the parameters were never touched) or raises."""

    def __init__(self, model, optimizer, scheduler=None, group: Optional[dist.ProcessGroup] = None, accumulate_grad_batches: int = 1,
                 reduce_slices: int = 1):
        if accumulate_grad_batches < 1:
            raise ValueError("accumulate_grad_batches must be >= 1")
        self.model, self.optimizer, self.scheduler, self.group = model, optimizer, scheduler, group
        self.k = int(accumulate_grad_batches)
        self.reduce_slices = int(reduce_slices)
        self.micro = 0                      # micro-batches since the last optimiser step
        self.optimizer_steps = 0
        self.skipped_steps = 0
        self._pending = None                # found_inf flag of the previous optimiser step (device tensor), read lazily
        self._bad_loss = None               # device bool: some micro-batch loss of the running accumulation was non-finite
        self._params = [p for p in model.parameters() if p.requires_grad]
        self._device_guard = bool(self._params) and all(p.is_cuda for p in self._params) and \
            bool(getattr(optimizer, "defaults", {}).get("fused", False))

    def _settle(self):
        """Read the flag of the PREVIOUS optimiser step (no stall: that step finished long ago) and act on it."""
        if self._pending is None:
            return
        bad, self._pending = bool(self._pending.item() != 0), None
        if not bad:
            return
        self.skipped_steps += 1
        import warnings
        pol = getattr(self.model, "nonfinite_policy", "raise")
        cur = self.model._current_arith() if hasattr(self.model, "_current_arith") else _lib.arith()
        if pol == "fp32" and cur == 1:
            # (the flag is the all-reduced found_inf: every rank takes this branch together)
            warnings.warn("fit: an optimisation step produced non-finite loss / gradients under split-16 arithmetic (skipped on the "
                          "device, parameters untouched); pinning the model to PRD_ARITH_FP32", RuntimeWarning, stacklevel=3)
            self.model.arithmetic = "fp32"
            self.model.arith_fallbacks += 1
            return
        if pol != "off":
            raise _lib.NonFiniteError(f"fit: non-finite loss / gradients under {_lib.ARITH_NAMES[cur]} arithmetic; the optimiser "
                                      "step was skipped on the device (parameters untouched)")

    def step(self, batch, batch_idx: int, **step_kwargs) -> torch.Tensor:
        """One MICRO-batch; the optimiser moves on every k-th call.  Returns the detached loss of the micro-batch."""
        model, opt = self.model, self.optimizer
        if self.micro == 0:
            self._settle()
            opt.zero_grad(set_to_none=True)
            self._bad_loss = None
        guard = self._device_guard and getattr(model, "nonfinite_policy", "off") != "off"
        loss = model.training_step(batch, batch_idx, check_finite=not guard and getattr(model, "nonfinite_policy", "off") != "off",
                                   **step_kwargs)
        (loss / self.k if self.k > 1 else loss).backward()          # Lightning scales the loss by 1 / accumulate_grad_batches
        if guard:
            b = ~torch.isfinite(loss.detach())
            self._bad_loss = b if self._bad_loss is None else (self._bad_loss | b)
        self.micro += 1
        if self.micro == self.k:
            self.finish_accumulation()
        return loss.detach()

    def finish_accumulation(self) -> None:
        """Gradient average + optimiser step for the micro-batches accumulated so far (also the end-of-epoch flush of an
        incomplete group, as Lightning does)."""
        if self.micro == 0:
            return
        model, opt = self.model, self.optimizer
        all_reduce_gradients(self._params, self.group, slices=self.reduce_slices)
        guard = self._device_guard and getattr(model, "nonfinite_policy", "off") != "off"
        if guard:
            found = (self._bad_loss | ~_grads_finite(self._params)).to(torch.float32).reshape(())      # 0-dim, like GradScaler's
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
                dist.all_reduce(found, op=dist.ReduceOp.MAX, group=self.group)       # every rank skips or none does
            opt.grad_scale, opt.found_inf = None, found         # fused Adam: no update, no step count, when found_inf == 1
            self._pending = found
        opt.step()
        if guard:
            opt.found_inf = None
        if self.scheduler is not None:
            self.scheduler.step()
        model.ema.update(model.parameters())
        self.micro = 0
        self.optimizer_steps += 1

    def fit_epoch(self, loader, epoch: int = 0, max_steps: Optional[int] = None, to_device=None):
        """All micro-batches of ``loader`` (a PDBDataModule.train_dataloader()); returns the list of detached losses."""
        losses = []
        for i, batch in enumerate(loader):
            if max_steps is not None and self.optimizer_steps >= max_steps:
                break
            if to_device is not None:
                batch = {k: (v.to(to_device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            losses.append(self.step(batch, i))
        self.finish_accumulation()
        self._settle()
        return losses


def _cached_fitter(model, optimizer, scheduler, group, k: int) -> "Fitter":
    """The Fitter behind ``fit_step`` for this (model, optimizer): kept on the model object, rebuilt when the optimiser, the
    scheduler, the group or k changes (a pending incomplete group of the old one is flushed first)."""
    store = model.__dict__.setdefault("_prd_fitters", {})
    f = store.get(id(optimizer))
    if f is not None and (f.optimizer is not optimizer or f.scheduler is not scheduler or f.group is not group or f.k != k):
        if f.optimizer is optimizer:
            f.finish_accumulation()
        f = None
    if f is None:
        f = store[id(optimizer)] = Fitter(model, optimizer, scheduler, group, accumulate_grad_batches=k)
    return f


def fit_step(model, batch, batch_idx: int, optimizer, scheduler=None, group: Optional[dist.ProcessGroup] = None,
             accumulate_grad_batches: int = 1, **step_kwargs) -> torch.Tensor:
    """One MICRO-batch of the optimisation the way ``train.py`` drives it through Lightning: training_step -> backward of
    loss / k; on every k-th micro-batch (k = ``accumulate_grad_batches``, train.py:57) gradient average over the data-parallel
    ranks -> Adam step -> LinearLR step -> EMA update (model.py:203-217, 528-549).  Convenience form of ``Fitter``: the call goes
    to a Fitter cached on the model for this optimiser, so the group boundaries follow the COUNT of micro-batches seen (not
    ``batch_idx % k``: a loader may start anywhere), a non-finite step is skipped on the device without a host read-back per
    micro-batch, and ``fit_flush`` steps a trailing incomplete group at the end of an epoch as Lightning does.  With k = 1 every call
    is a full optimisation step."""
    k = int(accumulate_grad_batches)
    if k < 1:
        raise ValueError("accumulate_grad_batches must be >= 1")
    return _cached_fitter(model, optimizer, scheduler, group, k).step(batch, batch_idx, **step_kwargs)


def fit_flush(model, optimizer) -> int:
    """End of an epoch for the ``fit_step`` form: optimiser step for a trailing incomplete group of micro-batches (Lightning flushes
    it; nothing happens when none is pending) and read-back of the last step's non-finite flag.  Returns the number of optimiser
    steps taken so far."""
    f = model.__dict__.get("_prd_fitters", {}).get(id(optimizer))
    if f is None or f.optimizer is not optimizer:
        return 0
    f.finish_accumulation()
    f._settle()
    return f.optimizer_steps
