"""Host-side mirror of the AF2-derived blocks the denoiser uses (reference
ProteinReDiff/models/AF2_modules.py:94-545): same class names, constructor arguments, ``forward``
signatures and ``state_dict`` keys; the arithmetic runs in libprd_hip.so.

Quirks reproduced on purpose: SPAttention never applies its mask and adds the
attention update to the LayerNorm-ed input; its per-head width is ``c_hidden`` (= single_dim);
OuterProductUpdate is a per-channel product followed by Linear and a division by (m_i m_j + eps).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from . import ops


class Linear(nn.Linear):
    """AF2_modules.py:94-159 initialisers ("default" LeCun, "relu" He, "glorot", "gating", "normal", "final")."""

    def __init__(self, in_dim: int, out_dim: int, bias: bool = True, init: str = "default", init_fn=None):
        super().__init__(in_dim, out_dim, bias=bias)
        with torch.no_grad():
            if bias:
                self.bias.zero_()
            if init_fn is not None:
                init_fn(self.weight, self.bias)
            elif init in ("default", "relu"):
                var = (2.0 if init == "relu" else 1.0) / max(1, in_dim)
                nn.init.trunc_normal_(self.weight, 0.0, math.sqrt(var) / 0.87962566103423978,
                                      a=-2.0 * math.sqrt(var) / 0.87962566103423978,
                                      b=2.0 * math.sqrt(var) / 0.87962566103423978)
            elif init == "glorot":
                nn.init.xavier_uniform_(self.weight, gain=1)
            elif init == "gating":
                self.weight.zero_()
                if bias:
                    self.bias.fill_(1.0)
            elif init == "normal":
                nn.init.kaiming_normal_(self.weight, nonlinearity="linear")
            elif init == "final":
                self.weight.zero_()
            else:
                raise ValueError("Invalid init string.")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear(x.contiguous(), self.weight, self.bias)


class LayerNorm(nn.Module):
    """AF2_modules.py:161-182 (affine, eps 1e-5)."""

    def __init__(self, c_in: int, eps: float = 1e-5):
        super().__init__()
        self.c_in, self.eps = (c_in,), eps
        self.weight = nn.Parameter(torch.ones(c_in))
        self.bias = nn.Parameter(torch.zeros(c_in))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.layer_norm(x.contiguous(), self.weight, self.bias)


class Attention(nn.Module):
    """Parameter container of AF2_modules.py:189-249 (the arithmetic is driven by SPAttention)."""

    def __init__(self, c_q: int, c_k: int, c_v: int, c_hidden: int, no_heads: int, gating: bool = True):
        super().__init__()
        self.c_q, self.c_k, self.c_v, self.c_hidden, self.no_heads, self.gating = c_q, c_k, c_v, c_hidden, no_heads, gating
        self.linear_q = Linear(c_q, c_hidden * no_heads, bias=False, init="glorot")
        self.linear_k = Linear(c_k, c_hidden * no_heads, bias=False, init="glorot")
        self.linear_v = Linear(c_v, c_hidden * no_heads, bias=False, init="glorot")
        self.linear_o = Linear(c_hidden * no_heads, c_q, init="final")
        self.linear_g = Linear(c_q, c_hidden * no_heads, init="gating") if gating else None


class SPAttention(nn.Module):
    """Single-representation attention with pair bias (AF2_modules.py:369-473)."""

    def __init__(self, c_in, c_hidden, no_heads, pair_bias=False, c_z=None, inf=1e9):
        super().__init__()
        self.c_in, self.c_hidden, self.no_heads, self.pair_bias, self.c_z, self.inf = c_in, c_hidden, no_heads, pair_bias, c_z, inf
        self.layer_norm_m = LayerNorm(c_in)
        if pair_bias:
            self.linear_z = nn.Sequential(LayerNorm(c_z), Linear(c_z, no_heads, bias=False, init="normal"))
        self.mha = Attention(c_in, c_in, c_in, c_hidden, no_heads)

    def _packed(self):
        a = self.mha
        ws = (a.linear_q.weight, a.linear_k.weight, a.linear_v.weight, a.linear_g.weight, a.linear_g.bias)
        return ops.cached_pack(self, "qkvg", ws, lambda: ops.pack_attention(*ws, 1.0 / math.sqrt(self.c_hidden)))

    def project(self, m: torch.Tensor):
        """The part that only needs the single representation: LayerNorm + the packed q|k|v|gate projection."""
        mn = ops.layer_norm(m.contiguous(), self.layer_norm_m.weight, self.layer_norm_m.bias)
        return mn, ops.project_qkvg(mn, self._packed(), self.no_heads * self.c_hidden)

    def bias_from_pair(self, z: torch.Tensor) -> torch.Tensor:
        return ops.pair_bias(z.contiguous(), self.linear_z[1].weight, None, self.linear_z[0].weight, self.linear_z[0].bias)

    def attend(self, mn: torch.Tensor, qkvg: torch.Tensor, bias: torch.Tensor, normed_only: bool = False, out_ln=None,
               logits_fp32: Optional[bool] = None) -> torch.Tensor:
        """``normed_only``: ``mn`` is the PLAIN normalised input (no affine; Denoiser.project_single's merged projection); the
        residual LN_affine(m) = mn * gamma + beta is then formed in the output projection's epilogue (gamma as the residual's
        column factor, beta added to the bias).  ``out_ln``: buffer that receives LN(result) when the output projection runs on
        the K-slab path (ops.slab_ok): the first folding block's attention projection starts with it.  ``logits_fp32``: the
        logits on fp32 MFMA in either arithmetic (ops.gated_attention_single) -- what a DIFFERENTIABLE forward needs; the
        training path passes True explicitly (training.network), None = only while autograd is recording (whatever the
        module's train / eval flag says: a backward taken on an eval() model must see the same forward)."""
        if logits_fp32 is None:
            logits_fp32 = torch.is_grad_enabled()
        a = self.mha
        ln = self.layer_norm_m
        bo, rscale = a.linear_o.bias, None
        if normed_only:
            bo = ops.cached_pack(self, "bo_beta", (a.linear_o.bias, ln.bias), lambda: (a.linear_o.bias + ln.bias).contiguous())
            rscale = ln.weight
        # no mask: the reference builds a mask bias and drops it (AF2_modules.py:447 vs 461-463)
        return ops.gated_attention_single(mn, mn, bias, self._packed(), a.linear_o.weight, bo,
                                          self.no_heads, self.c_hidden, key_mask=False, resid=mn, qkvg=qkvg, rscale=rscale, out_ln=out_ln,
                                          logits_fp32=bool(logits_fp32))

    def forward(self, m: torch.Tensor, z: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, N, _ = m.shape
        if self.pair_bias and z is not None:
            bias = self.bias_from_pair(z)
        else:
            bias = torch.zeros(b, self.no_heads, N, N, device=m.device, dtype=torch.float32)
        mn, qkvg = self.project(m)
        return self.attend(mn, qkvg, bias)


class OuterProductUpdate(nn.Module):
    """AF2_modules.py:476-545."""

    def __init__(self, c_m, c_z, c_hidden, eps=1e-3):
        super().__init__()
        self.c_m, self.c_z, self.c_hidden, self.eps = c_m, c_z, c_hidden, eps
        self.layer_norm = nn.LayerNorm(c_m)
        self.linear_1 = Linear(c_m, c_hidden)
        self.linear_2 = Linear(c_m, c_hidden)
        self.linear_out = Linear(c_hidden, c_z, init="final")

    def project(self, m, mask):
        """The part that only needs the single representation: LayerNorm (affine) + the packed a | b projection, masked."""
        b, N, S = m.shape
        Ch = self.c_hidden
        x = ops.layer_norm(m, self.layer_norm.weight, self.layer_norm.bias)
        ab = torch.empty(b, N, 2 * Ch, device=m.device, dtype=torch.float32)
        ws = (self.linear_1.weight, self.linear_2.weight, self.linear_1.bias, self.linear_2.bias)
        w12, b12 = ops.cached_pack(self, "ab", ws, lambda: (torch.cat(ws[:2]).contiguous(), torch.cat(ws[2:]).contiguous()))
        ops.gemm(x, w12, ab, b * N, 2 * Ch, S, S, S, 2 * Ch, bias=b12, rowmask=mask)
        return ab

    def run(self, m, pair, mask, *, residual: bool, apply_mask: bool, out=None, ab=None):
        if ab is None:
            ab = self.project(m, mask)
        return ops.opm_pair(pair, ab, mask, self.linear_out.weight, self.linear_out.bias,
                            residual=residual, apply_mask=apply_mask, out=out)

    def forward(self, m: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, N, _ = m.shape
        if mask is None:
            mask = m.new_ones(b, N)
        dummy = torch.empty(b, N, N, self.c_z, device=m.device, dtype=torch.float32)
        return self.run(m.contiguous(), dummy, mask.contiguous(), residual=False, apply_mask=False, out=dummy)
