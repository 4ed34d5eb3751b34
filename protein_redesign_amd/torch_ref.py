"""Differentiable torch-op restatement of the hot-path operators, used ONLY inside backward passes.

The forward of every operator is a HIP kernel (ops.py).  For the first cut of the training step (SURVEY.md §8f "next" #1:
"HIP forward + PyTorch autograd over the restatement for backward, then hand-written backward kernels") the BACKWARD of an
operator recomputes it here with torch ops on the GPU under ``torch.enable_grad()`` and lets autograd produce the gradients;
operators with hand-written backward kernels (training.py lists them) do not come here at all.  Nothing in this file is ever used
to produce a forward value of the product path, and none of it runs on the CPU.

Every function takes explicit weight tensors (nn.Linear layout) and follows the reference lines it cites; shapes are
single [b,N,S], pair [b,N,N,P], mask [b,N].
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F


def ln(x, w=None, b=None):
    return F.layer_norm(x, x.shape[-1:], w, b, 1e-5)


class _PairLinear(torch.autograd.Function):
    """F.linear whose weight gradient over the b N N pair positions is the slab reduction of csrc/prd_bwd.hip
    (ops.linear_wgrad) instead of the library's long-K GEMM (450 us per call at N = 320, 8 workgroups)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = (dy2 @ w).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            got = ops.linear_wgrad(dy2, x.reshape(-1, x.shape[-1]), bias=want_db)
            dw, db = got if want_db else (got, None)
        elif want_db:
            db = dy2.sum(0)
        return dx, dw, db


def linear(x, w, b=None):
    """nn.Linear; at pair-position row counts with 64-multiple widths the backward takes the hand-written weight-gradient kernel."""
    from . import ops
    rows = x.numel() // x.shape[-1]
    if (x.is_cuda and rows >= ops.WGRAD_MIN_ROWS and (w.shape[0] % 64 == 0 or w.shape[0] <= 16) and w.shape[1] % 64 == 0
            and max(w.shape) <= 256 and torch.is_grad_enabled()):
        return _PairLinear.apply(x, w, b)
    return F.linear(x, w, b)


class _SmallTable(torch.autograd.Function):
    """F.embedding whose table gradient is the slab reduction of csrc/prd_bwd.hip (ops.embed_wgrad): torch's scatter-add has
    millions of collisions per table row (4 ms per table at N = 320), and the one-hot GEMM form is the BLAS's 8-workgroup long-K
    kernel (475 us)."""

    @staticmethod
    def forward(ctx, idx, table):
        ctx.save_for_backward(idx)
        ctx.card = table.shape[0]
        return F.embedding(idx, table)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        (idx,) = ctx.saved_tensors
        return None, ops.embed_wgrad(idx.reshape(-1).contiguous(), dy.reshape(-1, dy.shape[-1]).contiguous(), ctx.card)


def small_table_lookup(idx, table):
    """F.embedding for a table of a few rows over a huge index tensor: hand-written table gradient on the GPU at pair-position
    row counts; elsewhere as one-hot @ table (a dense GEMM backward instead of a scatter-add)."""
    if table.shape[0] > 128:
        return F.embedding(idx, table)
    if idx.is_cuda and idx.numel() >= 8192 and table.shape[1] <= 64 and torch.is_grad_enabled() and idx.dtype == torch.int64:
        return _SmallTable.apply(idx, table)
    return F.one_hot(idx, table.shape[0]).to(table.dtype) @ table


def gated_attention(x, mask, wq, wk, wv, wg, bg, wo, bo, heads: int, head_dim: int, bias: Optional[torch.Tensor] = None):
    """modules.py:185-225: LN, q/k/v (no bias), sigmoid gate, q pre-scaled by 1/sqrt(c), additive bias, key mask filled with
    -2**15, softmax, gate, output projection.  ``x`` [..., n, E], ``mask`` [..., n] (keys)."""
    x = ln(x)
    lead, n = x.shape[:-2], x.shape[-2]

    def split(t):
        return t.reshape(*lead, n, heads, head_dim).transpose(-2, -3)

    if x.is_cuda and torch.is_grad_enabled():
        # one GEMM for the four projections (and one for each of their gradients): on the GPU this restatement is the BACKWARD of
        # the single-track attention, a chain of ~35 launch-bound kernels at [b, N, 512] size
        hc = heads * head_dim
        qkvg = F.linear(x, torch.cat([wq, wk, wv, wg], dim=0))
        q, k, v = split(qkvg[..., :hc]), split(qkvg[..., hc:2 * hc]), split(qkvg[..., 2 * hc:3 * hc])
        g = split(torch.sigmoid(qkvg[..., 3 * hc:] + bg))
    else:
        q, k, v = split(linear(x, wq)), split(linear(x, wk)), split(linear(x, wv))
        g = split(torch.sigmoid(linear(x, wg, bg)))
    logits = torch.matmul((1.0 / math.sqrt(head_dim)) * q, k.transpose(-1, -2))
    if bias is not None:
        logits = logits + bias
    logits = logits.masked_fill(mask.unsqueeze(-2).unsqueeze(-2) < 0.5, -(2.0 ** 15))
    out = g * torch.matmul(torch.softmax(logits, dim=-1), v)
    out = out.transpose(-2, -3).reshape(*lead, n, heads * head_dim)
    return linear(out, wo, bo)


def pair_bias(pair, w, b=None, gamma=None, beta=None):
    """modules.py:300-304 (no LN affine, bias) / AF2_modules.py:454-459 (LN affine, no bias) -> [b,H,N,N]."""
    return linear(ln(pair, gamma, beta), w, b).permute(0, 3, 1, 2)


def triangle_attention(pair, mask, wq, wk, wv, wg, bg, wo, bo, heads: int, head_dim: int, ending: bool, row_chunk: int = 64):
    """modules.py:236-243; rows are processed ``row_chunk`` at a time so that the [rows,H,N,N] logits stay bounded."""
    m2 = mask.unsqueeze(-1) * mask.unsqueeze(-2)
    if ending:
        pair, m2 = pair.transpose(1, 2), m2.transpose(1, 2)
    outs = [gated_attention(pair[:, r:r + row_chunk], m2[:, r:r + row_chunk], wq, wk, wv, wg, bg, wo, bo, heads, head_dim)
            for r in range(0, pair.shape[1], row_chunk)]
    out = torch.cat(outs, dim=1)
    return out.transpose(1, 2) if ending else out


def triangle_multiplication(pair, mask, wp, bp, wg, bg, wo, bo, wog, bog, incoming: bool):
    """modules.py:262-274."""
    x = ln(pair)
    m2 = (mask.unsqueeze(-1) * mask.unsqueeze(-2)).unsqueeze(-1)
    ab = m2 * torch.sigmoid(linear(x, wg, bg)) * linear(x, wp, bp)
    a, b = torch.chunk(ab, 2, dim=-1)
    o = torch.einsum("bkid,bkjd->bijd" if incoming else "bikd,bjkd->bijd", a, b)
    return torch.sigmoid(linear(x, wog, bog)) * linear(ln(o), wo, bo)


class _OuterProduct(torch.autograd.Function):
    """prod[b,i,j,p] = sum_s x[b,i,s] x[b,j,s] w1[p,s] without the [b,N,N,S] intermediate, forward and backward as ONE batched
    GEMM each over the N P rows of a batch element:  prod[b,i,:,:]^T = (x_i * w1) x^T;  with T[b,i,p,s] = sum_j (dy[b,i,j,p] +
    dy[b,j,i,p]) x[b,j,s]:  dx[b,i,s] = sum_p w1[p,s] T[b,i,p,s],  dw1[p,s] = 1/2 sum_{b,i} x[b,i,s] T[b,i,p,s]  (symmetry of
    x_i x_j).  autograd of the einsum form took ~0.8 ms per call at N = 320."""

    @staticmethod
    def forward(ctx, x, w1):
        ctx.save_for_backward(x, w1)
        b, N, S = x.shape
        P = w1.shape[0]
        a = (x.unsqueeze(2) * w1).view(b, N * P, S)                     # [b, (i,p), s]
        return torch.bmm(a, x.transpose(1, 2)).view(b, N, P, N).permute(0, 1, 3, 2)

    @staticmethod
    def backward(ctx, dy):
        x, w1 = ctx.saved_tensors
        b, N, S = x.shape
        P = w1.shape[0]
        dsym = (dy + dy.transpose(1, 2)).permute(0, 1, 3, 2).reshape(b, N * P, N)
        T = torch.bmm(dsym, x).view(b, N, P, S)
        dx = (T * w1).sum(dim=2)
        dw1 = 0.5 * (T * x.unsqueeze(2)).sum(dim=(0, 1))
        return dx, dw1


class _OuterProductAB(torch.autograd.Function):
    """prod[b,i,j,p] = sum_s a[b,i,s] bb[b,j,s] wo[p,s] (the outer-product update's a_i (x) b_j -> Linear, AF2_modules.py:532-545)
    without the [b,N,N,S] intermediate: forward and backward as batched GEMMs over the N P rows of a batch element.
    T1[b,i,p,s] = sum_j dy[b,i,j,p] bb[b,j,s]: da = sum_p wo T1, dwo = sum_{b,i} a T1;  T2[b,j,p,s] = sum_i dy[b,i,j,p] a[b,i,s]:
    dbb = sum_p wo T2."""

    @staticmethod
    def forward(ctx, a, bb, wo):
        ctx.save_for_backward(a, bb, wo)
        b, N, S = a.shape
        P = wo.shape[0]
        aw = (a.unsqueeze(2) * wo).view(b, N * P, S)                    # [b, (i,p), s]
        return torch.bmm(aw, bb.transpose(1, 2)).view(b, N, P, N).permute(0, 1, 3, 2)

    @staticmethod
    def backward(ctx, dy):
        a, bb, wo = ctx.saved_tensors
        b, N, S = a.shape
        P = wo.shape[0]
        T1 = torch.bmm(dy.permute(0, 1, 3, 2).reshape(b, N * P, N), bb).view(b, N, P, S)
        T2 = torch.bmm(dy.permute(0, 2, 3, 1).reshape(b, N * P, N), a).view(b, N, P, S)
        return (T1 * wo).sum(dim=2), (T2 * wo).sum(dim=2), (T1 * a.unsqueeze(2)).sum(dim=(0, 1))


def outer_linear(single, w, b):
    """modules.py:283-287 in the split form W1 (x_i * x_j) + W2 x_i - W2 x_j + b (no [N,N,2S] concat)."""
    x = ln(single)
    S = x.shape[-1]
    w1, w2 = w[:, :S], w[:, S:]
    u = F.linear(x, w2)
    prod = _OuterProduct.apply(x, w1)
    return prod + u.unsqueeze(2) - u.unsqueeze(1) + b


def transition(x, w1, b1, w2, b2):
    """single_fc / pair_fc: LN -> Linear -> ReLU -> Linear (modules.py:306-311, 321-326)."""
    return linear(torch.relu(linear(ln(x), w1, b1)), w2, b2)


def outer_product_update(single, mask, g, bta, w1, b1, w2, b2, wo, bo):
    """AF2_modules.py:503-545 followed by the caller's mask (modules.py:395-397): m2 * upd."""
    x = ln(single, g, bta)
    m = mask.unsqueeze(-1)
    a, b = F.linear(x, w1, b1) * m, F.linear(x, w2, b2) * m
    outer = _OuterProductAB.apply(a, b, wo) + bo
    m2 = (mask.unsqueeze(-1) * mask.unsqueeze(-2)).unsqueeze(-1)
    return m2 * (outer / (m2 + 1e-3))


def single_pair_attention(single, pair, gm, bm, gz, bz, wz, wq, wk, wv, wg, bg, wo, bo, heads: int):
    """AF2_modules.py:421-473: unmasked, residual on the LayerNorm-ed input, head width = single_dim."""
    return single_bias_attention(single, pair_bias(pair, wz, None, gz, bz), gm, bm, wq, wk, wv, wg, bg, wo, bo, heads)


def single_bias_attention(single, bias, gm, bm, wq, wk, wv, wg, bg, wo, bo, heads: int):
    """single_pair_attention with the pair bias [b, H, N, N] given."""
    m = ln(single, gm, bm)
    b_, n, _ = m.shape

    def heads_of(t):
        return t.view(b_, n, heads, -1).transpose(-2, -3)

    if m.is_cuda and torch.is_grad_enabled():           # one GEMM for the four projections (see gated_attention)
        hc = wq.shape[0]
        qkvg = F.linear(m, torch.cat([wq, wk, wv, wg], dim=0))
        q, k, v = heads_of(qkvg[..., :hc]), heads_of(qkvg[..., hc:2 * hc]), heads_of(qkvg[..., 2 * hc:3 * hc])
        gate_pre = qkvg[..., 3 * hc:] + bg
    else:
        q, k, v = heads_of(F.linear(m, wq)), heads_of(F.linear(m, wk)), heads_of(F.linear(m, wv))
        gate_pre = F.linear(m, wg, bg)
    a = torch.softmax(torch.matmul(q / math.sqrt(q.shape[-1]), k.transpose(-1, -2)) + bias, dim=-1)
    o = torch.matmul(a, v).transpose(-2, -3)
    gate = torch.sigmoid(gate_pre).view(b_, n, heads, -1)
    return m + F.linear((o * gate).reshape(b_, n, -1), wo, bo)


def input_stage_single(batch, seq_t, atom_tabs: Sequence[torch.Tensor], w_rt, w_esm):
    """model.py:332-345: the single half of the input stage (atom feature tables, residue type and ESM projections)."""
    am, rm = batch["atom_mask"], batch["residue_mask"]
    sa = 1.0 / math.sqrt(len(atom_tabs))
    acc = 0.0
    for f, tab in enumerate(atom_tabs):
        idx = batch["atom_feats"][..., f]
        # on the GPU with a backward coming: one-hot @ table (its table gradient is a small dense GEMM; torch's embedding backward
        # sorts and scatter-adds: 58 us per table for 640 rows)
        acc = acc + sa * ((F.one_hot(idx, tab.shape[0]).to(tab.dtype) @ tab) if (idx.is_cuda and torch.is_grad_enabled() and tab.requires_grad)
                          else F.embedding(idx, tab))
    return am.unsqueeze(-1) * acc + rm.unsqueeze(-1) * (
        torch.relu(F.linear(ln(seq_t), w_rt)) + F.linear(ln(batch["residue_esm"]), w_esm))


def input_stage(batch, z, seq_t, mask, t, num_steps: int, max_bond_distance: int, max_relpos: int,
                atom_tabs: Sequence[torch.Tensor], bond_tabs: Sequence[torch.Tensor], bd_tab, rp_tab, w_rt, w_esm,
                centers, w_dist, freqs, w_beta):
    """model.py:332-361: single and pair inputs of the trunk."""
    am, rm = batch["atom_mask"], batch["residue_mask"]
    single = input_stage_single(batch, seq_t, atom_tabs, w_rt, w_esm)
    sb = 1.0 / math.sqrt(len(bond_tabs))
    bacc = 0.0
    for f, tab in enumerate(bond_tabs):
        bacc = bacc + sb * small_table_lookup(batch["bond_feats"][..., f], tab)
    am2 = (am.unsqueeze(-1) * am.unsqueeze(-2)).unsqueeze(-1)
    rm2 = (rm.unsqueeze(-1) * rm.unsqueeze(-2)).unsqueeze(-1)
    m2 = (mask.unsqueeze(-1) * mask.unsqueeze(-2)).unsqueeze(-1)
    ri, ci = batch["residue_index"], batch["residue_chain_index"]
    rel = (ri.unsqueeze(-1) - ri.unsqueeze(-2)).clamp(min=-max_relpos, max=max_relpos) + max_relpos
    chain = (ci.unsqueeze(-1) == ci.unsqueeze(-2)).float().unsqueeze(-1)
    pair = am2 * (batch["bond_mask"].unsqueeze(-1) * bacc + small_table_lookup(batch["bond_distance"].clamp(max=max_bond_distance), bd_tab))
    pair = pair + rm2 * (chain * small_table_lookup(rel, rp_tab))
    dist = torch.linalg.norm(z.unsqueeze(-2) - z.unsqueeze(-3), dim=-1)
    scale = (centers.numel() - 1) / 2.0
    rbf = torch.exp(-scale * torch.square(dist.unsqueeze(-1) - centers))
    wx = freqs * (t / num_steps)[:, None, None].unsqueeze(-1)
    sinus = torch.cat([torch.sin(wx), torch.cos(wx)], dim=-1)
    pair = pair + m2 * (linear(rbf, w_dist) + F.linear(sinus, w_beta))
    return single, pair


def seq_head(single, ws1, bs1, ws2):
    """model.py:371-374: sequence logits."""
    return F.linear(torch.relu(F.linear(ln(single), ws1, bs1)), ws2)


def heads(single, pair, z, mask, wr1, br1, wr2, ws1, bs1, ws2):
    """modules.py:403 (pair symmetrisation) + model.py:364-374: coordinate update and sequence logits."""
    pair = 0.5 * (pair + pair.transpose(1, 2))
    w = linear(torch.relu(linear(ln(pair), wr1, br1)), wr2)
    zij = z.unsqueeze(-2) - z.unsqueeze(-3)
    r = zij * torch.rsqrt(torch.sum(torch.square(zij), -1, keepdim=True) + 1e-4)
    m2 = (mask.unsqueeze(-1) * mask.unsqueeze(-2)).unsqueeze(-1)
    eps = (m2 * w * r).sum(dim=2)
    m = mask.unsqueeze(-1)
    eps = eps - m * (m * eps).sum(dim=1, keepdim=True) / m.sum(dim=1, keepdim=True)
    logits = F.linear(torch.relu(F.linear(ln(single), ws1, bs1)), ws2)
    return eps, logits
