"""Host-side mirror of the reference's trunk modules, with every ``forward`` running on HIP.

Class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys follow the
reference (ProteinReDiff/modules.py:129-404) so that these classes drop into code written against
it; the bodies only marshal tensors into the C ABI (ops.py -> libprd_hip.so).  Inputs must be CUDA
fp32 tensors -- there is no CPU path.  Inference only for now (no autograd through the kernels).

``mask_2d`` arguments: the reference always passes the outer product of a 0/1 node mask; the kernels
take the node mask itself, recovered here as the diagonal of ``mask_2d``.
"""
from __future__ import annotations

import os

import math
from argparse import Namespace
from typing import Mapping, Optional, Tuple

import torch
from torch import nn

from . import ops
from .af2_blocks import OuterProductUpdate, SPAttention

_VARIANCE_INITS = {            # init name -> (scale, fan mode, distribution)   (modules.py:142-157)
    "default": (1.0, "fan_in", "truncated_normal"),
    "relu": (2.0, "fan_in", "truncated_normal"),
    "glorot": (1.0, "fan_avg", "uniform"),
    "normal": (1.0, "fan_in", "normal"),
}
_TRUNC_STD_CORRECTION = 0.87962566103423978   # std of N(0,1) truncated to [-2, 2]


def _variance_scaling_(weight: torch.Tensor, scale: float, mode: str, distribution: str) -> None:
    fan_out, fan_in = weight.shape
    fan = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": 0.5 * (fan_in + fan_out)}[mode]
    var = scale / max(1.0, fan)
    if distribution == "truncated_normal":
        nn.init.trunc_normal_(weight, 0.0, math.sqrt(var) / _TRUNC_STD_CORRECTION)
    elif distribution == "normal":
        nn.init.normal_(weight, 0.0, math.sqrt(var))
    else:
        lim = math.sqrt(3.0 * var)
        nn.init.uniform_(weight, -lim, lim)


class Linear(nn.Linear):
    """nn.Linear with the named initialisers of reference modules.py:129-167."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, init: str = "default", init_fn=None):
        super().__init__(in_features, out_features, bias=bias)
        with torch.no_grad():
            if init_fn is not None:
                init_fn(self.weight, self.bias)
            elif init in _VARIANCE_INITS:
                _variance_scaling_(self.weight, *_VARIANCE_INITS[init])
                if bias:
                    self.bias.zero_()
            elif init in ("gating", "final"):
                self.weight.zero_()
                if bias:
                    self.bias.fill_(1.0 if init == "gating" else 0.0)
            else:
                raise ValueError(f"Invalid init: {init}")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear(x.contiguous(), self.weight, self.bias)


def _node_mask(mask_2d: torch.Tensor) -> torch.Tensor:
    return torch.diagonal(mask_2d, dim1=-2, dim2=-1).contiguous()


class Attention(nn.Module):
    """Gated multi-head attention (modules.py:170-225).  ``x`` is [b,N,E] (node axis) -- the
    row-batched use over the pair tensor goes through TriangleAttention's fused kernel instead."""

    def __init__(self, embed_dim: int, head_dim: int, num_heads: int):
        super().__init__()
        self.embed_dim, self.head_dim, self.num_heads = embed_dim, head_dim, num_heads
        self.scale = 1.0 / math.sqrt(head_dim)
        self.inf = 2.0 ** 15
        self.norm = nn.LayerNorm(embed_dim, elementwise_affine=False)
        self.q_proj = Linear(embed_dim, num_heads * head_dim, bias=False, init="glorot")
        self.k_proj = Linear(embed_dim, num_heads * head_dim, bias=False, init="glorot")
        self.v_proj = Linear(embed_dim, num_heads * head_dim, bias=False, init="glorot")
        self.gate_proj = Linear(embed_dim, num_heads * head_dim, init="gating")
        self.out_proj = Linear(num_heads * head_dim, embed_dim, init="final")

    def weights(self):
        return (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.gate_proj.weight,
                self.gate_proj.bias, self.out_proj.weight, self.out_proj.bias)

    def packed(self):
        ws = self.weights()[:5]
        return ops.cached_pack(self, "qkvg", ws, lambda: ops.pack_attention(*ws, self.scale))

    def run_single(self, x, mask, attn_bias, resid, ln_a=False, qkvg=None):
        """``x`` is the LayerNorm-ed input, or with ``ln_a`` the raw one (the norm is then fused into the projection).
        ``qkvg``: the q|k|v|gate projection if the caller already has it (the previous block's single track computes it together
        with its outer-linear term: both are linear in the same LN(single))."""
        return ops.gated_attention_single(x, mask, attn_bias, self.packed(), self.out_proj.weight, self.out_proj.bias,
                                          self.num_heads, self.head_dim, key_mask=True, resid=resid, ln_a=ln_a, qkvg=qkvg)

    def forward(self, x: torch.Tensor, mask: torch.Tensor, attn_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x.dim() != 3:
            raise RuntimeError("Attention.forward expects [b, N, E]; use TriangleAttention for pair rows")
        b, N, _ = x.shape
        if attn_bias is None:
            attn_bias = torch.zeros(b, self.num_heads, N, N, device=x.device, dtype=torch.float32)
        return self.run_single(x.contiguous(), mask.contiguous(), attn_bias.contiguous(), None, ln_a=True)


class TriangleAttention(nn.Module):
    """modules.py:228-243; one fused HIP operator (prd_tri_attn)."""

    def __init__(self, pair_dim: int, head_dim: int, num_heads: int, mode: str):
        super().__init__()
        if mode not in ("starting", "ending"):
            raise ValueError(f"Invalid mode: {mode}")
        self.attn = Attention(pair_dim, head_dim, num_heads)
        self.mode = mode

    def run(self, pair, mask, *, residual: bool, out=None, ws=None):
        a = self.attn
        return ops.tri_attn(pair, mask, a.weights(), a.num_heads, a.head_dim, ending=self.mode == "ending",
                            residual=residual, out=out, ws=ws)

    def forward(self, pair: torch.Tensor, mask_2d: torch.Tensor) -> torch.Tensor:
        return self.run(pair.contiguous(), _node_mask(mask_2d), residual=False)


class TriangleMultiplication(nn.Module):
    """modules.py:246-274; HIP operator prd_tri_mul (projection, batched contraction, gated output)."""

    def __init__(self, pair_dim: int, mode: str):
        super().__init__()
        if mode not in ("outgoing", "incoming"):
            raise ValueError(f"Invalid mode: {mode}")
        self.mode = mode
        self.norm = nn.LayerNorm(pair_dim, elementwise_affine=False)
        self.ab_proj = Linear(pair_dim, pair_dim * 2, init="default")
        self.ab_gate = Linear(pair_dim, pair_dim * 2, init="gating")
        self.ab_norm = nn.LayerNorm(pair_dim, elementwise_affine=False)
        self.out_proj = Linear(pair_dim, pair_dim, init="final")
        self.out_gate = Linear(pair_dim, pair_dim, init="gating")

    def weights(self):
        return (self.ab_proj.weight, self.ab_proj.bias, self.ab_gate.weight, self.ab_gate.bias,
                self.out_proj.weight, self.out_proj.bias, self.out_gate.weight, self.out_gate.bias)

    def run(self, pair, mask, *, residual: bool, out=None, ws=None):
        return ops.tri_mul(pair, mask, self.weights(), incoming=self.mode == "incoming", residual=residual, out=out, ws=ws)

    def forward(self, pair: torch.Tensor, mask_2d: torch.Tensor) -> torch.Tensor:
        return self.run(pair.contiguous(), _node_mask(mask_2d), residual=False)


_TRI_MUL_CHAIN = os.environ.get("PRD_TRI_MUL_CHAIN", "1") != "0"      # 0: two prd_tri_mul calls (A/B measurements)
# 1: the starting attention's output projection rides in the row load of the ending core (prd_tri_attn_core_fused).  Measured
# SLOWER (1.973 vs 1.931 ms per step, A/B in one run): each of the four head-workgroups of a row re-reads the previous og row
# (4 x 26 MB instead of one pass) and repeats the projection.  Off by default; kept as a tested opt-in.
_TRI_ATTN_FUSE = os.environ.get("PRD_TRI_ATTN_FUSE", "0") == "1"
_PAIR_HEAD = os.environ.get("PRD_PAIR_HEAD", "1") != "0"              # 0: pair_init, OPM tail and the first bias heads as three launches
_MERGE_HEAD = os.environ.get("PRD_MERGE_HEAD", "1") != "0"            # 0: OPM / SPA LayerNorms and projections as four launches
_HEAD_SLAB = os.environ.get("PRD_HEAD_SLAB", "0") == "1"              # 1: the merged head projection on the K-slab kernel + reduce launch (measured slower: 40.8 vs 31.4 us)
_MERGE_PROJ = os.environ.get("PRD_MERGE_PROJ", "1") != "0"            # 0: u and the next q|k|v|gate as two launches (A/B measurements)


class OuterLinear(nn.Module):
    """modules.py:277-287, without the [N,N,2S] concat: W1 (x_i*x_j) + W2 x_i - W2 x_j + b."""

    def __init__(self, single_dim: int, pair_dim: int):
        super().__init__()
        self.single_dim, self.pair_dim = single_dim, pair_dim
        self.norm = nn.LayerNorm(single_dim, elementwise_affine=False)
        self.linear = Linear(single_dim * 2, pair_dim, init="final")

    def run(self, single, pair, *, residual: bool, out=None):
        b, N, S = single.shape
        w = self.linear.weight
        u = torch.empty(b, N, self.pair_dim, device=single.device, dtype=torch.float32)
        if ops.ln_fusable(S):           # u = LN(single) W2^T with the LayerNorm inside the GEMM, which also writes x = LN(single)
            x = torch.empty_like(single)
            ops.gemm(single, w, u, b * N, self.pair_dim, S, S, 2 * S, self.pair_dim, b_off=S, a_ln=True, ln_out=x)
        else:
            x = ops.layer_norm(single)
            ops.gemm(x, w, u, b * N, self.pair_dim, S, S, 2 * S, self.pair_dim, b_off=S)
        return ops.outer_linear_pair(pair, x, u, w, self.linear.bias, residual=residual, out=out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, N, _ = x.shape
        dummy = torch.empty(b, N, N, self.pair_dim, device=x.device, dtype=torch.float32)
        return self.run(x.contiguous(), dummy, residual=False, out=dummy)


class _Rearrange(nn.Module):
    """Parameter-free stand-in for einops' Rearrange("... i j h -> ... h i j") in attn_bias (modules.py:303)."""

    def forward(self, x):
        return x.movedim(-1, -3)


class FoldingBlock(nn.Module):
    """modules.py:290-343: eight residual updates; here every pair update is applied in place."""

    def __init__(self, single_dim: int, pair_dim: int, head_dim: int, num_heads: int, transition_factor: int):
        super().__init__()
        self.attn_bias = nn.Sequential(
            nn.LayerNorm(pair_dim, elementwise_affine=False),
            Linear(pair_dim, num_heads, init="normal"),
            _Rearrange(),
        )
        self.single_attn = Attention(single_dim, head_dim, num_heads)
        self.single_fc = nn.Sequential(
            nn.LayerNorm(single_dim, elementwise_affine=False),
            Linear(single_dim, single_dim * transition_factor, init="relu"),
            nn.ReLU(),
            Linear(single_dim * transition_factor, single_dim, init="final"),
        )
        self.outer_linear = OuterLinear(single_dim, pair_dim)
        self.pair_mul_outgoing = TriangleMultiplication(pair_dim, "outgoing")
        self.pair_mul_incoming = TriangleMultiplication(pair_dim, "incoming")
        self.pair_attn_starting = TriangleAttention(pair_dim, head_dim, num_heads, "starting")
        self.pair_attn_ending = TriangleAttention(pair_dim, head_dim, num_heads, "ending")
        self.pair_fc = nn.Sequential(
            nn.LayerNorm(pair_dim, elementwise_affine=False),
            Linear(pair_dim, pair_dim * transition_factor, init="relu"),
            nn.ReLU(),
            Linear(pair_dim * transition_factor, pair_dim, init="final"),
        )

    def _after_transition_pack(self, next_block, tail):
        """Weights of the ONE projection that follows the transition: [OuterLinear's W2 | next consumer of LN(single)] -- the next
        block's q|k|v|gate (pack_attention layout) or ``tail`` = (weight, bias) of a ReLU layer (the sequence head's first)."""
        S, P = self.outer_linear.single_dim, self.outer_linear.pair_dim
        w_ol = self.outer_linear.linear.weight
        if next_block is not None:
            sa = next_block.single_attn
            ws = (w_ol, *sa.weights()[:5])

            def build():
                w, pb, cs = sa.packed()
                dev = w.device
                return (torch.cat([w_ol[:, S:], w]).contiguous(), torch.cat([torch.zeros(P, device=dev), pb]).contiguous(),
                        torch.cat([torch.ones(P, device=dev), cs]).contiguous())
            return ops.cached_pack(self, "after_transition_next", ws, build), 2, P + 3 * sa.num_heads * sa.head_dim
        if tail is not None:
            tw, tb = tail
            ws = (w_ol, tw, tb)

            def build():
                dev = tw.device
                return (torch.cat([w_ol[:, S:], tw]).contiguous(), torch.cat([torch.zeros(P, device=dev), tb]).contiguous(), None)
            return ops.cached_pack(self, "after_transition_tail", ws, build), 1, P
        return None, 0, 0

    def run_(self, single: torch.Tensor, pair: torch.Tensor, mask: torch.Tensor, ws=None, bias=None, next_block=None, spare_holder=None,
             qkvg=None, tail=None, extra=None):
        """In place on ``pair`` or -- fused attention form, gemm mode 1 -- ending in another buffer: use the RETURNED pair tensor.
        ``bias``: this block's attention bias if the previous block's fused tail
        already produced it; ``next_block``: the following FoldingBlock, whose attention bias is then computed
        by this block's fused tail.  ``qkvg``: this block's attention projection if the previous block already produced it.
        ``tail`` = (weight, bias): a ReLU layer applied to LN(single_out) in the same launch as the outer-linear term (last block:
        the sequence head's first layer).  Returns (single, pair, next_bias or None); with ``extra`` (a dict) the by-products go
        there: "qkvg" = the next block's projection, "tail" = the tail layer's output."""
        sa = self.single_attn
        b, N = mask.shape
        if ws is None:
            ws = torch.empty(max(ops.workspace_bytes("tri_mul", b, N, 0, pair.shape[-1]),
                                 ops.workspace_bytes("tri_attn", b, N, 0, pair.shape[-1])) // 4,
                             device=pair.device, dtype=torch.float32)
        if bias is None:
            bias = ops.pair_bias(pair, self.attn_bias[1].weight, self.attn_bias[1].bias)
        single = sa.run_single(single, mask, bias, single, ln_a=True, qkvg=qkvg)
        fc = self.single_fc
        wsum1 = ops.cached_pack(self, "fc1_rowsum", (fc[1].weight,), lambda: fc[1].weight.double().sum(1).float().contiguous())
        single, xhat = ops.transition_single(single, fc[1].weight, fc[1].bias, fc[3].weight, fc[3].bias, residual=True, wsum1=wsum1,
                                             want_ln=True)
        packed, act, act_from = self._after_transition_pack(next_block if extra is not None else None, tail if extra is not None else None)
        if packed is not None and _MERGE_PROJ:
            # everything that is linear in LN(single) goes through ONE launch: u of the outer-linear, and the next block's
            # q|k|v|gate or the tail layer (-1 launch of ~8 us per block; these launches are latency-, not work-bound)
            P = self.outer_linear.pair_dim
            x, cat = ops.project_many(single, packed, P, act=act, act_from=act_from, xhat=xhat)
            ol = self.outer_linear.linear
            ops.outer_linear_pair(pair, x, cat[..., :P], ol.weight, ol.bias, residual=True, out=pair)
            extra["qkvg" if next_block is not None else "tail"] = cat[..., P:]
        else:
            self.outer_linear.run(single, pair, residual=True, out=pair)
        if _TRI_MUL_CHAIN and ops.tri_mul_chain_supported(N, pair.shape[-1]):      # gemm mode 1: out-stage of the first + projection of the second fused
            ops.tri_mul_chain_(pair, mask, self.pair_mul_outgoing.weights(), self.pair_mul_incoming.weights(), ws=ws)
        else:
            self.pair_mul_outgoing.run(pair, mask, residual=True, out=pair, ws=ws)
            self.pair_mul_incoming.run(pair, mask, residual=True, out=pair, ws=ws)
        ta = self.pair_attn_ending.attn
        nog = b * N * N * 64
        if _TRI_ATTN_FUSE and ops.tri_attn_core_fused_supported(N, pair.shape[-1]) and ws.numel() >= 2 * nog:
            # starting attention: core only; its output projection + residual ride in the row load of the ending core, which
            # writes the updated pair tensor to the spare buffer (the residual stream changes buffers here)
            ts = self.pair_attn_starting.attn
            og_s = ops.tri_attn_core(pair, mask, ts.weights()[:5], ts.num_heads, ts.head_dim, ending=False,
                                     og=ws[:nog].view(b, N, N, 64))
            spare = spare_holder[0] if spare_holder else None
            if spare is None:
                spare = torch.empty_like(pair)
            og = ops.tri_attn_core_fused(pair, og_s, ts.out_proj.weight, ts.out_proj.bias, mask, ta.weights()[:5], ta.num_heads,
                                         ta.head_dim, ending=True, pair_out=spare, og=ws[nog: 2 * nog].view(b, N, N, 64))
            pair, spare = spare, pair
            if spare_holder is not None:
                spare_holder[0] = spare          # the buffer the residual stream just left: the next block's target
        elif ops.PERSISTENT_ATTN and ops.tri_attn_pair_supported(N, pair.shape[-1]):
            # (opt-in, PRD_PERSISTENT_ATTN=1) the starting attention and the ending core as ONE persistent launch
            ts = self.pair_attn_starting.attn
            og = ops.tri_attn_pair_(pair, mask, ts.weights(), ta.weights()[:5], ta.num_heads, ta.head_dim, og=ws[:nog].view(b, N, N, 64))
        else:
            self.pair_attn_starting.run(pair, mask, residual=True, out=pair, ws=ws)
            # ending triangle attention: core kernel, then ONE fused row pass = its output projection + the pair
            # transition + (if there is a next block) that block's attention bias
            og = ops.tri_attn_core(pair, mask, ta.weights()[:5], ta.num_heads, ta.head_dim, ending=True,
                                   og=ws[:nog].view(b, N, N, 64), stats=ws[nog:] if ws.numel() > nog else None)
        pf = self.pair_fc
        nb_w = next_block.attn_bias[1].weight if next_block is not None else None
        nb_b = next_block.attn_bias[1].bias if next_block is not None else None
        next_bias = ops.block_tail_(pair, og, ta.out_proj.weight, ta.out_proj.bias, pf[1].weight, pf[1].bias,
                                    pf[3].weight, pf[3].bias, nb_w, nb_b)
        return single, pair, next_bias

    def forward(self, single: torch.Tensor, pair: torch.Tensor, mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        single, pair, _ = self.run_(single.contiguous(), pair.contiguous().clone(), mask.contiguous())
        return single, pair


class Denoiser(nn.Module):
    """modules.py:346-404."""

    def __init__(self, args):
        super().__init__()
        if isinstance(args, Mapping):
            args = Namespace(**args)
        self.single_dim, self.esm_dim, self.pair_dim = args.single_dim, args.esm_dim, args.pair_dim
        self.head_dim, self.num_heads = args.head_dim, args.num_heads
        self.transition_factor, self.num_blocks = args.transition_factor, args.num_blocks
        self.n_recycles = args.n_recycles
        self.SPAAttnBlock = SPAttention(c_in=self.single_dim, c_hidden=self.single_dim, no_heads=self.num_heads,
                                        pair_bias=True, c_z=self.pair_dim)
        self.opm = OuterProductUpdate(c_m=self.single_dim, c_z=self.pair_dim, c_hidden=self.single_dim // 4)
        self.folding_blocks = nn.ModuleList(
            [FoldingBlock(self.single_dim, self.pair_dim, self.head_dim, self.num_heads, self.transition_factor)
             for _ in range(self.num_blocks)])

    def ws_floats(self, b: int, N: int) -> int:
        P = self.pair_dim
        return max(ops.workspace_bytes("tri_mul", b, N, 0, P), ops.workspace_bytes("tri_attn", b, N, 0, P)) // 4

    def project_single(self, single: torch.Tensor, mask: torch.Tensor):
        """Everything at the head of the trunk that depends on the single representation only (OPM's a | b projection, SPA's
        LayerNorm + q|k|v|g projection): independent of the pair input stage, so the caller may run it on a side stream."""
        if not _MERGE_HEAD or not ops.ln_fusable(single.shape[-1]):
            ab = self.opm.project(single, mask)
            mn, qkvg = self.SPAAttnBlock.project(single)
            return ab, mn, qkvg, False
        # Both projections are linear in the SAME normalised rows: the two affine LayerNorms differ only in (gamma, beta), which
        # fold into the weights -- LN_affine(x) W^T + b = LN(x) (W diag(gamma))^T + (b + W beta).  One launch (LayerNorm fused,
        # the plain normalised rows written on the side) instead of two LayerNorm + two GEMM launches.
        opm, spa = self.opm, self.SPAAttnBlock
        a = spa.mha
        params = (opm.layer_norm.weight, opm.layer_norm.bias, opm.linear_1.weight, opm.linear_1.bias, opm.linear_2.weight,
                  opm.linear_2.bias, spa.layer_norm_m.weight, spa.layer_norm_m.bias, a.linear_q.weight, a.linear_k.weight,
                  a.linear_v.weight, a.linear_g.weight, a.linear_g.bias)

        def build():
            go, bo, w1, b1, w2, b2, gs, bs, wq, wk, wv, wg, bg = [p.detach().double() for p in params]
            scale = 1.0 / math.sqrt(spa.c_hidden)
            w = torch.cat([w1 * go, w2 * go, wq * gs, wk * gs, wv * gs, wg * gs])
            bias = torch.cat([b1 + w1 @ bo, b2 + w2 @ bo, (wq @ bs) * scale, wk @ bs, wv @ bs, bg + wg @ bs])
            cs = torch.cat([torch.ones(2 * opm.c_hidden, dtype=torch.float64, device=w.device),
                            torch.full((wq.shape[0],), scale, dtype=torch.float64, device=w.device),
                            torch.ones(3 * wq.shape[0], dtype=torch.float64, device=w.device)])
            return w.float().contiguous(), bias.float().contiguous(), cs.float().contiguous(), w.sum(1).float().contiguous()
        w, bias, cs, wsum = ops.cached_pack(self, "head_proj", params, build)
        b, N, S = single.shape
        Ch2, HS = 2 * opm.c_hidden, a.linear_q.weight.shape[0]
        ab = torch.empty(b, N, Ch2, device=single.device, dtype=torch.float32)
        qkvg = torch.empty(b, N, 4 * HS, device=single.device, dtype=torch.float32)
        xhat = torch.empty_like(single)
        # _HEAD_SLAB (A/B): the K-slab kernel (every workgroup walks all four K slabs of its 160 x 64 tile) + its epilogue launch
        slab = _HEAD_SLAB and ops.slab_ok(b * N, Ch2 + 4 * HS, S)
        ops.gemm(single, w, ab, b * N, Ch2 + 4 * HS, S, S, S, Ch2, bias=bias, colscale=cs, act=2, act_from=Ch2 + 3 * HS,
                 rowmask=mask, rowmask_cols=Ch2, a_ln=True, ln_out=xhat, c2=qkvg, n_split=Ch2, slab=slab, wsum=wsum if slab else None)
        return ab, xhat, qkvg, True

    def run_(self, single: torch.Tensor, pair: torch.Tensor, mask: torch.Tensor, ws=None, pre=None, join=None, tail=None, pair_init=None):
        """OPM, SPA and the folding blocks, in place on ``pair``, WITHOUT the final symmetrisation
        (the fused coordinate head symmetrises on the fly, so the hot path never writes it back).
        ``pre`` = ``project_single(single, mask)`` if the caller already enqueued it; ``join()`` is then called before its
        results are consumed.  ``tail`` = (weight, bias) of a ReLU layer over LN(single_out) that the last block computes
        together with its outer-linear term; its output is then returned as a third value.  ``pair`` may be None when
        ``pair_init`` = (static_pair, z, centers, w_dist, ebeta) is given: the pair input stage then runs here, fused with the
        outer-product update and the first attention-bias heads where the library has that form (ops.pair_head)."""
        b, N = mask.shape
        if ws is None:
            ws = torch.empty(self.ws_floats(b, N), device=mask.device, dtype=torch.float32)
        if pre is None:
            pre = self.project_single(single, mask)
        ab, mn, qkvg, normed_only = pre
        if join is not None:
            join()
        spa = self.SPAAttnBlock
        blocks = list(self.folding_blocks)
        fused_head = False
        if pair is None:
            sp, z, centers, w_dist, eb = pair_init
            opm = self.opm
            if (_PAIR_HEAD and blocks and ops.pair_head_supported(self.pair_dim, w_dist.shape[1], opm.c_hidden)):
                ab0 = blocks[0].attn_bias[1]
                pair, spa_bias, bias = ops.pair_head(sp, z, mask, centers, w_dist, eb, ab, opm.linear_out.weight, opm.linear_out.bias,
                                                     apply_mask=True,
                                                     set_a=(spa.linear_z[1].weight, None, spa.linear_z[0].weight, spa.linear_z[0].bias),
                                                     set_b=(ab0.weight, ab0.bias, None, None))
                fused_head = True
            else:
                pair = ops.pair_init(sp, z, mask, centers, w_dist, eb)
        if not fused_head:
            self.opm.run(single, pair, mask, residual=True, apply_mask=True, out=pair, ab=ab)
            if blocks:      # SPAttention's pair bias and the first block's attention bias: one pass over the pair tensor
                ab0 = blocks[0].attn_bias[1]
                spa_bias, bias = ops.pair_bias2(pair, (spa.linear_z[1].weight, None, spa.linear_z[0].weight, spa.linear_z[0].bias),
                                                (ab0.weight, ab0.bias, None, None))
            else:
                spa_bias, bias = spa.bias_from_pair(pair), None
        # SPAttention's output projection (2048 -> 512) runs on the K-slab path where it qualifies; its reduce launch also writes
        # LN(single), with which the first block's q|k|v|gate projection starts (no LayerNorm prologue there)
        b_, N_, S_ = mn.shape
        xhat0 = None
        if blocks and _MERGE_PROJ and S_ <= 512 and ops.slab_ok(b_ * N_, spa.mha.linear_o.weight.shape[0], spa.no_heads * spa.c_hidden):
            xhat0 = torch.empty_like(mn)
        single = spa.attend(mn, qkvg, spa_bias, normed_only=normed_only, out_ln=xhat0)
        holder = [None]             # spare pair buffer of the fused attention form (the residual stream alternates between two)
        qkvg, extra = None, {}
        if xhat0 is not None:
            sa0 = blocks[0].single_attn
            qkvg = ops.project_qkvg(xhat0, sa0.packed(), sa0.num_heads * sa0.head_dim)
        for i, block in enumerate(blocks):
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            extra = {}
            single, pair, bias = block.run_(single, pair, mask, ws=ws, bias=bias, next_block=nxt, spare_holder=holder, qkvg=qkvg,
                                            tail=tail if nxt is None else None, extra=extra)
            qkvg = extra.get("qkvg")
        if tail is not None:
            return single, pair, extra.get("tail")
        return single, pair

    def forward(self, batch, z, t, single, pair, cache):
        """Like the reference, updates ``pair`` in place up to the final symmetrisation (modules.py:395)
        and ignores ``z`` / ``t``."""
        mask = batch["residue_and_atom_mask"].contiguous()
        single, pair = self.run_(single.contiguous(), pair, mask)
        pair = 0.5 * (pair + pair.transpose(1, 2))
        return single, pair, cache
