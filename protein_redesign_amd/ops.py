"""Python-side operator wrappers over the C ABI (include/prd_hip.h).

Each function takes CUDA fp32 tensors, allocates outputs / scratch with PyTorch's caching allocator,
and enqueues HIP kernels on the current torch stream.  Weight tensors are passed as they sit in the
``state_dict`` (nn.Linear layout).  Nothing here has a CPU implementation.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch

from ._lib import PrdGemm, check, dptr, lib, stream

F32 = torch.float32


_QUEUES = {}


def task_queue(device) -> int:
    """Device pointer of the per-device task-queue counters (zero-initialised int32s, see prd_hip.h), or 0 = static
    wave-major task order.  The static order measured faster for every row kernel at the bench shape,
    so the queue is opt-in: PRD_TASK_QUEUE=1."""
    import os
    if not os.environ.get("PRD_TASK_QUEUE"):
        return 0
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    q = _QUEUES.get(key)
    if q is None:
        q = _QUEUES[key] = torch.zeros(256, dtype=torch.int32, device=f"cuda:{key}")
    return q.data_ptr()


class SideStream:
    """Fork / join of a second HIP stream around kernels that do not depend on each other (works eagerly and inside
    ``torch.cuda.graph`` capture, where the branches become parallel paths of the hipGraph): the single track's small kernels at
    the head and tail of a step are independent of the pair-track kernels next to them.  MEASURED SLOWER on MI355X (2.388 vs
    2.350 ms per step, two A/B pairs on one box: the cross-branch dependencies of the replayed graph cost more than the ~70 us
    of small kernels they take off the critical path), so it is OFF unless ``PRD_SIDE_STREAM=1``."""

    def __init__(self, device):
        import os
        self.enabled = os.environ.get("PRD_SIDE_STREAM", "0") == "1"
        self.stream = torch.cuda.Stream(device=device) if self.enabled else None

    def fork(self):
        """Context manager: launches inside run on the side stream, ordered after everything enqueued so far on the current one."""
        import contextlib
        if not self.enabled:
            return contextlib.nullcontext()
        self.stream.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.stream)

    def join(self):
        """The current stream waits for everything enqueued on the side stream."""
        if self.enabled:
            torch.cuda.current_stream().wait_stream(self.stream)


def _off(t, elem_offset: int = 0) -> int:
    """device address of element ``elem_offset`` of a contiguous tensor (or of a raw device address, for column blocks)"""
    return (t if isinstance(t, int) else dptr(t)) + 4 * elem_offset


def round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, *, a_off=0, b_off=0, c_off=0, G1=1, G2=1,
         sa=(0, 0), sb=(0, 0), sc=(0, 0), b_kn=False, alpha=1.0, bias=None, act=0, act_from=0,
         addmat=None, sad=(0, 0), ldadd=0, colmask=None, scm1=0, fill=0.0, rowmask=None, srm1=0,
         mulmat=None, mul_off=0, smu=(0, 0), ldmul=0, resid=None, res_off=0, sr=(0, 0), ldr=0,
         colscale=None, tile_hint=0, a_ln=False, ln_out=None, c2=None, n_split=0, rowmask_cols=0, rscale=None,
         slab=False, wsum=None, out_ln=None, a_scale=0.0, mul_pos=False, unsupported_ok=False):
    """Raw batched GEMM + epilogue (see PrdGemm in include/prd_hip.h).  ``a_ln`` may be 2 (row softmax of A, see the header);
    with ``unsupported_ok`` a PRD_ERR_UNSUPPORTED shape returns None instead of raising (the caller falls back)."""
    g = PrdGemm()
    g.A, g.B, g.C = _off(A, a_off), _off(B, b_off), _off(Cout, c_off)
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.G1, g.G2 = M, N, K, lda, ldb, ldc, G1, G2
    g.sa1, g.sa2, g.sb1, g.sb2, g.sc1, g.sc2 = sa[0], sa[1], sb[0], sb[1], sc[0], sc[1]
    g.b_kn, g.alpha = int(b_kn), float(alpha)
    g.bias, g.act, g.act_from = dptr(bias), act, act_from
    g.addmat, g.sad1, g.sad2, g.ldadd = dptr(addmat), sad[0], sad[1], ldadd
    g.colmask, g.scm1, g.fill = dptr(colmask), scm1, float(fill)
    g.rowmask, g.srm1 = dptr(rowmask), srm1
    g.mulmat = (_off(mulmat, mul_off) if mulmat is not None else None)
    g.smu1, g.smu2, g.ldmul = smu[0], smu[1], ldmul
    g.resid = (_off(resid, res_off) if resid is not None else None)
    g.sr1, g.sr2, g.ldr = sr[0], sr[1], ldr
    g.colscale, g.tile_hint = dptr(colscale), tile_hint
    g.a_ln = int(a_ln) if a_ln in (1, 2) else (1 if a_ln else 0)
    g.ln_out, g.ldlo = dptr(ln_out), (ln_out.shape[-1] if ln_out is not None else 0)
    g.C2, g.ldc2, g.n_split = dptr(c2), (c2.shape[-1] if c2 is not None else 0), n_split
    g.rowmask_cols, g.rscale = rowmask_cols, dptr(rscale)
    if slab:                        # K split across workgroups (prd_hip.h: PrdGemm.ws); the library falls back by itself if the shape does not qualify
        wsb = gemm_workspace(Cout.device, int(lib().prd_gemm_slab_workspace(M, N, K)))
        g.ws, g.ws_bytes = dptr(wsb), wsb.numel() * 4
    g.wsum = dptr(wsum)
    g.out_ln, g.ldol = dptr(out_ln), (out_ln.shape[-1] if out_ln is not None else 0)
    g.a_scale = float(a_scale)
    g.mul_pos = int(mul_pos)
    g.arith = lib().prd_get_gemm_mode() | (lib().prd_get_tune() << 8)
    import ctypes
    code = lib().prd_gemm(ctypes.byref(g), stream())
    if unsupported_ok and code == -3:
        return None
    check(code, "prd_gemm")
    return Cout


_GEMM_WS = {}


def gemm_workspace(device, nbytes: int) -> torch.Tensor:
    """Scratch of the K-slab GEMM path (partial tiles; at most a few tens of MB).  Eager launches: one persistent buffer per
    (device, stream) -- the launches that use it are ordered on that stream.  Under graph capture: a fresh tensor per call from the
    capturing graph's own memory pool (the allocator reuses it between the GEMMs of the captured step, which are ordered on the
    capture stream, and the graph keeps the addresses alive exactly as long as it exists; a cached buffer would outlive the graph
    it was carved from)."""
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    n = max(1, (nbytes + 3) // 4)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n, device=f"cuda:{dev}", dtype=F32)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)     # per stream: launches on different streams must not share partials
    buf = _GEMM_WS.get(key)
    if buf is None or buf.numel() < n:
        buf = _GEMM_WS[key] = torch.empty(n, device=f"cuda:{dev}", dtype=F32)
    return buf


def split16_gemm_ok(M: int, N: int, K: int) -> bool:
    """True when an unbatched [M, K] x [N, K]^T GEMM lands on the split-16 tile kernel (gemm_h2: split-16 mode, K a multiple of 32,
    at least 512 tiles of 64 x 64) -- otherwise the fp32 tile kernel would run, and for large shapes the BLAS is the better fallback."""
    return lib().prd_get_gemm_mode() == 1 and K % 32 == 0 and ((M + 63) // 64) * ((N + 63) // 64) >= 512


def slab_ok(M: int, N: int, K: int) -> bool:
    """True when a linear of this shape takes the K-slab path (prd_hip.h: prd_gemm_slab_ok) in the current arithmetic."""
    return lib().prd_gemm_slab_ok(M, N, K) == 1


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = 0,
           alpha: float = 1.0, resid: Optional[torch.Tensor] = None, rowmask: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, ln_a: bool = False, rscale: Optional[torch.Tensor] = None,
           slab: bool = False, wsum: Optional[torch.Tensor] = None, out_ln: Optional[torch.Tensor] = None,
           ln_a_out: Optional[torch.Tensor] = None, relu_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(alpha * x W^T + bias) [* rowmask] [+ resid] for x [..., K], W [N, K]; ``ln_a``: x is LayerNorm-ed
    (no affine) inside the GEMM instead of by a separate launch, and with ``ln_a_out`` [rows, K] the normalised rows are
    written there as well; ``relu_mask`` [rows, N]: y is zeroed where relu_mask <= 0 (the ReLU backward from recomputed activations)."""
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    if ln_a and not ln_fusable(K):
        x, ln_a = layer_norm(x.contiguous()), False
        if ln_a_out is not None:
            ln_a_out.copy_(x.view_as(ln_a_out))
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=F32)
    a, lda = x, K
    if not x.is_contiguous():          # a column block of a wider GEMM output
        a, lda = row_block(x)
    gemm(a, w, out, M, N, K, lda, w.stride(0), N, alpha=alpha, bias=bias, act=act,
         rowmask=rowmask, resid=resid, ldr=N, a_ln=ln_a, rscale=rscale, slab=slab, wsum=wsum, out_ln=out_ln,
         ln_out=ln_a_out if ln_a else None, mulmat=relu_mask, ldmul=N, mul_pos=relu_mask is not None)
    return out


def ln_fusable(K: int) -> bool:
    """PrdGemm.a_ln keeps a lane's K slice in registers: K <= 512 and a multiple of 4 (prd_hip.h)."""
    return K <= 512 and K % 4 == 0


def layer_norm(x: torch.Tensor, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None) -> torch.Tensor:
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    y = torch.empty_like(x)
    check(lib().prd_ln_rows(dptr(x), dptr(y), dptr(gamma), dptr(beta), rows, Cn, Cn, Cn, stream()), "prd_ln_rows")
    return y


def softmax_rows_(x: torch.Tensor, n: int) -> torch.Tensor:
    ld = x.shape[-1]
    check(lib().prd_softmax_rows(dptr(x), x.numel() // ld, n, ld, stream()), "prd_softmax_rows")
    return x


# ---------------------------------------------------------------------------------------------------
# input stage
# ---------------------------------------------------------------------------------------------------

def static_pair(batch, tabs, max_bond_distance: int, max_relpos: int, P: int) -> torch.Tensor:
    am = batch["atom_mask"]
    b, N = am.shape
    out = torch.empty(b, N, N, P, device=am.device, dtype=F32)
    i64 = torch.int64
    check(lib().prd_static_pair(
        dptr(out), dptr(am), dptr(batch["residue_mask"]), dptr(batch["bond_mask"]),
        dptr(batch["bond_feats"], i64), dptr(batch["bond_distance"], i64),
        dptr(batch["residue_index"], i64), dptr(batch["residue_chain_index"], i64),
        dptr(tabs[0]), dptr(tabs[1]), dptr(tabs[2]), dptr(tabs[3]), dptr(tabs[4]),
        max_bond_distance, max_relpos, b, N, P, stream()), "prd_static_pair")
    return out


def atom_embed(atom_feats, atom_mask, tables_cat, offsets, S: int) -> torch.Tensor:
    b, N, nf = atom_feats.shape
    out = torch.empty(b, N, S, device=atom_mask.device, dtype=F32)
    check(lib().prd_atom_embed(dptr(out), dptr(atom_feats, torch.int64), dptr(atom_mask), dptr(tables_cat),
                               dptr(offsets, torch.int32), nf, b, N, S, stream()), "prd_atom_embed")
    return out


def single_init(static_single, seq_t, residue_mask, w_rt, out=None) -> torch.Tensor:
    b, N, S = static_single.shape
    if out is None:
        out = torch.empty_like(static_single)
    check(lib().prd_single_init(dptr(out), dptr(static_single), dptr(seq_t), dptr(residue_mask), dptr(w_rt),
                                b * N, S, seq_t.shape[-1], stream()), "prd_single_init")
    return out


def time_embed(t, freqs, w_beta, num_steps: int, out=None) -> torch.Tensor:
    b = t.shape[0]
    P, TD = w_beta.shape
    if out is None:
        out = torch.empty(b, P, device=t.device, dtype=F32)
    check(lib().prd_time_embed(dptr(out), dptr(t, torch.int64), dptr(freqs), dptr(w_beta), num_steps, b, P, TD,
                               stream()), "prd_time_embed")
    return out


def pair_init(static_pair_t, z, mask, centers, w_dist, ebeta, out=None) -> torch.Tensor:
    b, N, _, P = static_pair_t.shape
    if out is None:
        out = torch.empty_like(static_pair_t)
    check(lib().prd_pair_init(dptr(out), dptr(static_pair_t), dptr(z), dptr(mask), dptr(centers), dptr(w_dist),
                              dptr(ebeta), b, N, P, w_dist.shape[1], stream()), "prd_pair_init")
    return out


# ---------------------------------------------------------------------------------------------------
# pair-track operators
# ---------------------------------------------------------------------------------------------------

def pair_bias(pair, w, bvec=None, gamma=None, beta=None) -> torch.Tensor:
    b, N, _, P = pair.shape
    H = w.shape[0]
    out = torch.empty(b, H, N, N, device=pair.device, dtype=F32)
    check(lib().prd_pair_bias(dptr(out), dptr(pair), dptr(gamma), dptr(beta), dptr(w), dptr(bvec), b, N, P, H,
                              stream()), "prd_pair_bias")
    return out


def pair_bias2(pair, set_a, set_b):
    """Two pair_bias results from one pass over ``pair``; each set = (w, bvec, gamma, beta) as in pair_bias."""
    b, N, _, P = pair.shape
    (wa, ba, ga, bea), (wb, bb, gb, beb) = set_a, set_b
    Ha, Hb = wa.shape[0], wb.shape[0]
    oa = torch.empty(b, Ha, N, N, device=pair.device, dtype=F32)
    ob = torch.empty(b, Hb, N, N, device=pair.device, dtype=F32)
    check(lib().prd_pair_bias2(dptr(oa), dptr(pair), dptr(ga), dptr(bea), dptr(wa), dptr(ba), Ha,
                               dptr(ob), dptr(gb), dptr(beb), dptr(wb), dptr(bb), Hb, b, N, P, stream()), "prd_pair_bias2")
    return oa, ob


def opm_pair(pair, ab, mask, w_out, b_out, *, residual: bool, apply_mask: bool, out=None) -> torch.Tensor:
    b, N, _, P = pair.shape
    Cc = ab.shape[-1] // 2
    if out is None:
        out = torch.empty_like(pair)
    flags = (1 if residual else 0) | (2 if apply_mask else 0)
    check(lib().prd_opm_pair(dptr(out), dptr(pair), dptr(ab), dptr(mask), dptr(w_out), dptr(b_out), flags,
                             b, N, P, Cc, stream()), "prd_opm_pair")
    return out


def row_block(t: torch.Tensor):
    """(device pointer, row pitch in floats) of ``t`` [b, N, C]: a contiguous tensor or a COLUMN BLOCK of one (``big[..., c0:c1]``:
    unit column stride, rows ``pitch`` floats apart, batches N rows apart) -- how the kernels take an operand that sits inside a
    wider GEMM output."""
    if t.dim() != 3 or t.stride(-1) != 1 or t.stride(0) != t.shape[1] * t.stride(1) or not t.is_cuda or t.dtype != F32:
        raise RuntimeError("expected a CUDA fp32 [b, N, C] tensor or a column block of one")
    if (t.storage_offset() | t.stride(1)) & 3:
        raise RuntimeError("column block must start on a 16-byte boundary and have a row pitch that is a multiple of 4 floats")
    return t.data_ptr(), t.stride(1)


def pair_head_supported(P: int, dist_dim: int, C: int) -> bool:
    """True when the fused head of the pair track (prd_pair_head) exists for these widths in the current arithmetic."""
    return lib().prd_pair_head_supported(P, dist_dim, C) == 1


def pair_head(static_pair_t, z, mask, centers, w_dist, ebeta, ab, w_out, b_out, *, apply_mask: bool, set_a, set_b, out=None):
    """pair_init -> OuterProductUpdate tail (residual) -> two attention-bias heads, one row pass (prd_pair_head).
    set = (w, bvec, gamma, beta) as in pair_bias2.  Returns (pair, bias_a [b,Ha,N,N], bias_b [b,Hb,N,N])."""
    b, N, _, P = static_pair_t.shape
    (wa, ba, ga, bea), (wb, bb_, gb, beb) = set_a, set_b
    Ha, Hb = wa.shape[0], wb.shape[0]
    if out is None:
        out = torch.empty_like(static_pair_t)
    oa = torch.empty(b, Ha, N, N, device=out.device, dtype=F32)
    ob = torch.empty(b, Hb, N, N, device=out.device, dtype=F32)
    check(lib().prd_pair_head(dptr(out), dptr(static_pair_t), dptr(z), dptr(mask), dptr(centers), dptr(w_dist), dptr(ebeta),
                              w_dist.shape[1], dptr(ab), dptr(w_out), dptr(b_out), ab.shape[-1] // 2, int(apply_mask),
                              dptr(oa), dptr(ga), dptr(bea), dptr(wa), dptr(ba), Ha, dptr(ob), dptr(gb), dptr(beb), dptr(wb), dptr(bb_), Hb,
                              b, N, P, stream()), "prd_pair_head")
    return out, oa, ob


def outer_linear_pair(pair, x, u, w, bias, *, residual: bool, out=None) -> torch.Tensor:
    """``u`` [b,N,P]: contiguous, or a column block of a wider GEMM output (row_block)."""
    b, N, _, P = pair.shape
    if out is None:
        out = torch.empty_like(pair)
    up, ldu = row_block(u)
    check(lib().prd_outer_linear(dptr(out), dptr(pair), dptr(x), up, ldu, dptr(w), dptr(bias), int(residual),
                                 b, N, P, x.shape[-1], task_queue(pair.device), stream()), "prd_outer_linear")
    return out


def workspace_bytes(op: str, b: int, N: int, S: int, P: int) -> int:
    return int(lib().prd_workspace_bytes(op.encode(), b, N, S, P))


def tri_mul(pair, mask, wts, *, incoming: bool, residual: bool, out=None, ws=None) -> torch.Tensor:
    """wts = (ab_proj.w, ab_proj.b, ab_gate.w, ab_gate.b, out_proj.w, out_proj.b, out_gate.w, out_gate.b)"""
    b, N, _, P = pair.shape
    if out is None:
        out = torch.empty_like(pair)
    nbytes = workspace_bytes("tri_mul", b, N, 0, P)
    if ws is None:
        ws = torch.empty(nbytes // 4, device=pair.device, dtype=F32)
    check(lib().prd_tri_mul(dptr(out), dptr(pair), dptr(mask), *[dptr(w) for w in wts], int(incoming), int(residual),
                            b, N, P, dptr(ws), ws.numel() * 4, task_queue(pair.device), stream()), "prd_tri_mul")
    return out


def tri_mul_chain_supported(N: int, P: int) -> bool:
    """True when the fused outgoing -> incoming chain (prd_tri_mul_chain) exists for this shape in the current arithmetic mode."""
    return lib().prd_tri_mul_chain_supported(N, P) == 1


def tri_mul_chain_(pair, mask, wts_outgoing, wts_incoming, ws=None) -> torch.Tensor:
    """pair += TriMul_outgoing(pair); pair += TriMul_incoming(pair), in place (modules.py:336-337), with the output stage of the
    first and the projection stage of the second fused into one row pass (prd_tri_mul_chain; gemm mode 1)."""
    import ctypes
    b, N, _, P = pair.shape
    nbytes = workspace_bytes("tri_mul", b, N, 0, P)
    if ws is None:
        ws = torch.empty(nbytes // 4, device=pair.device, dtype=F32)
    arr = ctypes.c_void_p * 8
    wa = arr(*[dptr(w) for w in wts_outgoing])
    wb = arr(*[dptr(w) for w in wts_incoming])
    check(lib().prd_tri_mul_chain(dptr(pair), dptr(mask), ctypes.cast(wa, ctypes.c_void_p), ctypes.cast(wb, ctypes.c_void_p),
                                  b, N, P, dptr(ws), ws.numel() * 4, stream()), "prd_tri_mul_chain")
    return pair


# 1: the starting triangle attention (core + output projection + residual) and the ending attention's core as ONE persistent launch with
# two in-kernel grid barriers (prd_tri_attn_pair) instead of three launches.  Bit-identical; measured in DESIGN.md 4.3.
PERSISTENT_ATTN = os.environ.get("PRD_PERSISTENT_ATTN", "0") == "1"
_PAIR_BARS = {}


def tri_attn_pair_supported(N: int, P: int) -> bool:
    return lib().prd_tri_attn_pair_supported(N, P) == 1


def tri_attn_pair_bar(device) -> torch.Tensor:
    """The 32 uint32 (counters, memberships, [1] = timeout flag) of prd_tri_attn_pair on ``device``: zeroed by every call, stream-ordered."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    bar = _PAIR_BARS.get(key)
    if bar is None:
        bar = _PAIR_BARS[key] = torch.zeros(32, dtype=torch.int32, device=f"cuda:{key}")
    return bar


def tri_attn_pair_(pair, mask, wts_start, wts_end, H: int, c: int, og=None) -> torch.Tensor:
    """pair += TriangleAttention_starting(pair) in place, then og [b,N,N,64] = the gated head outputs of the ENDING attention on the
    updated pair (its output projection rides in block_tail_): modules.py:338-339 as ONE persistent launch (prd_tri_attn_pair).
    wts_start = (q.w, k.w, v.w, gate.w, gate.b, out.w, out.b), wts_end = (q.w, k.w, v.w, gate.w, gate.b)."""
    import ctypes
    b, N, _, P = pair.shape
    if og is None:
        og = torch.empty(b, N, N, 64, device=pair.device, dtype=F32)
    wa = (ctypes.c_void_p * 7)(*[dptr(w) for w in wts_start])
    wb = (ctypes.c_void_p * 5)(*[dptr(w) for w in wts_end])
    bar = tri_attn_pair_bar(pair.device)
    check(lib().prd_tri_attn_pair(dptr(og), dptr(pair), dptr(mask), ctypes.cast(wa, ctypes.c_void_p), ctypes.cast(wb, ctypes.c_void_p),
                                  b, N, P, H, c, bar.data_ptr(), stream()), "prd_tri_attn_pair")
    return og


def tri_attn_pair_timed_out(device) -> bool:
    """True when a grid barrier of the last prd_tri_attn_pair on ``device`` gave up (synchronises; tests and the sampler's final check)."""
    return int(tri_attn_pair_bar(device)[1].item()) != 0


def tri_mul_backward(dy, pair, mask, wts, *, incoming: bool, ws=None):
    """Gradients of the TriangleMultiplication update (ops.tri_mul with residual=False) with respect to ``pair`` and its eight
    weight tensors, on the hand-written backward kernels (csrc/prd_bwd.hip): forward recompute (projection, contraction) ->
    output-stage backward -> the two gradient contractions on the forward contraction kernel -> projection-stage backward.
    The weight gradients are tall-skinny GEMMs over all N^2 rows and go through torch (rocBLAS)."""
    wp, bp, wg, bg, wo, bo, wog, bog = wts
    b, N, _, P = pair.shape
    ldn = round_up(N, 32)
    dev = pair.device
    unit = b * P * N * ldn
    if ws is None:                                       # forward recompute of the operands and the contraction output
        ws = torch.empty(3 * unit, device=dev, dtype=F32)
        scratch = torch.empty_like(pair)
        check(lib().prd_tri_mul(dptr(scratch), dptr(pair), dptr(mask), *[dptr(w) for w in wts], int(incoming), 0, b, N, P,
                                dptr(ws), ws.numel() * 4, 0, stream()), "prd_tri_mul (recompute)")
    AB = ws[:2 * unit].view(b, 2 * P, N, ldn)
    O = ws[2 * unit:].view(b, P, N, ldn)
    dy = dy.contiguous()
    dz, dgp, dx1 = torch.empty_like(pair), torch.empty_like(pair), torch.empty_like(pair)
    x, lo = torch.empty_like(pair), torch.empty_like(pair)       # LN(pair), LN(O) by pair position: inputs of the weight gradients
    # dA[i][k] = sum_j dO[i][j] B^T[k][j];  dB[j][k] = sum_i dO^T[j][i] A^T[k][i]: one contraction over the stacked operands
    # dO | dO^T | B^T | A^T (dO written in place by the output-stage backward, the transposes by prd_tri_mul_bwd_operands)
    ops4 = torch.empty(b, 4 * P, N, ldn, device=dev, dtype=F32)
    # (the transposed weight images are staged from wo / wog / wp / wg read column-wise: no transposed copies)
    check(lib().prd_tri_mul_out_bwd(dptr(dz), dptr(dgp), dptr(ops4), dptr(dx1), dptr(dy), dptr(pair), dptr(O), dptr(wo), dptr(bo),
                                    dptr(wog), dptr(bog), None, None, dptr(x), dptr(lo), 4 * P, b, N, P, stream()),
          "prd_tri_mul_out_bwd")
    check(lib().prd_tri_mul_bwd_operands(dptr(ops4), dptr(AB), b, N, P, stream()), "prd_tri_mul_bwd_operands")
    dAB = torch.empty(b, 2 * P, N, ldn, device=dev, dtype=F32)
    check(lib().prd_tri_mul_contract(dptr(dAB), dptr(ops4), b, N, 2 * P, stream()), "prd_tri_mul_contract")
    del ops4
    dpair = torch.empty_like(pair)
    dpp = torch.empty(b, N, N, 2 * P, device=dev, dtype=F32)
    dpg = torch.empty(b, N, N, 2 * P, device=dev, dtype=F32)
    check(lib().prd_tri_mul_proj_bwd(dptr(dpair), dptr(dpp), dptr(dpg), dptr(dAB), dptr(dx1), dptr(pair), dptr(mask), dptr(wp),
                                     dptr(bp), dptr(wg), dptr(bg), None, None, int(incoming), b, N, P, stream()),
          "prd_tri_mul_proj_bwd")
    # weight gradients: dW = dOut^T In over all rows (linear_wgrad)
    x, lo = x.view(-1, P), lo.view(-1, P)
    dz2, dgp2, dpp2, dpg2 = dz.view(-1, P), dgp.view(-1, P), dpp.view(-1, 2 * P), dpg.view(-1, 2 * P)
    grads = (*linear_wgrad(dpp2, x, bias=True), *linear_wgrad(dpg2, x, bias=True), *linear_wgrad(dz2, lo, bias=True),
             *linear_wgrad(dgp2, x, bias=True))
    return dpair, grads


TRI_ATTN_BWD_V2 = os.environ.get("PRD_TRI_ATTN_BWD_V2", "1") != "0"      # 0: the fp32-MFMA backward core in split-16 mode too (A/B measurements)


def tri_attn_backward(dy, pair, mask, wts, H: int, c: int, *, ending: bool, og=None, lse=None, residual: bool = False):
    """Gradients of the TriangleAttention update (ops.tri_attn with residual=False; ``residual``: of pair + update, i.e. dy is added
    to the pair gradient) with respect to ``pair`` and its seven weight
    tensors on the hand-written backward (csrc/prd_bwd.hip): out-projection backward (row GEMM) -> attention core backward per
    (row, head) -> projections backward (row GEMM) -> LayerNorm backward.  Weight gradients: slab reductions over all N^2 rows
    (linear_wgrad)."""
    wq, wk, wv, wg, bg, wo, bo = wts
    b, N, _, P = pair.shape
    HC = H * c
    dev = pair.device
    dy = dy.contiguous()
    if og is None:                                                                              # forward recompute: gated head outputs
        lse = torch.empty(b * N, H, N, 2, device=dev, dtype=F32) if tri_attn_lse_supported(N, P) else None
        og = tri_attn_core(pair, mask, (wq, wk, wv, wg, bg), H, c, ending=ending, lse=lse)
    dog = pair_linear(dy.view(-1, P), wo.t())                                                     # d og = dy W_o
    if dog is None:
        dog = linear(dy, wo.t().contiguous())
    dqkvg = torch.empty(b, N, N, 4, HC, device=dev, dtype=F32)
    x = None
    if lib().prd_get_gemm_mode() == 1 and TRI_ATTN_BWD_V2 and lib().prd_tri_attn_bwd_core_v2_supported(N, P) == 1:
        x = torch.empty_like(pair)                                                                # LN(pair), left by the core
        check(lib().prd_tri_attn_bwd_core_v2(dptr(dqkvg), dptr(dog), dptr(og), dptr(pair), dptr(mask), dptr(wq), dptr(wk), dptr(wv), dptr(wg),
                                             dptr(bg), dptr(lse) if lse is not None else None, dptr(x), int(ending), b, N, P, H, c, stream()),
              "prd_tri_attn_bwd_core_v2")
    else:
        check(lib().prd_tri_attn_bwd_core(dptr(dqkvg), dptr(dog), dptr(pair), dptr(mask), dptr(wq), dptr(wk), dptr(wv), dptr(wg), dptr(bg),
                                          int(ending), b, N, P, H, c, stream()), "prd_tri_attn_bwd_core")
    wcat = torch.cat([wq, wk, wv, wg], dim=0)                                                     # [4 HC, P]
    dxn = pair_linear(dqkvg.view(-1, 4 * HC), wcat.t())                                          # gradient of LN(pair)
    if dxn is None:
        dxn = linear(dqkvg.view(b, N, N, 4 * HC), wcat.t().contiguous())
    dpair = torch.empty_like(pair)
    check(lib().prd_ln_rows_bwd(dptr(dpair), dptr(dxn), dptr(pair), dptr(dy) if residual else None, b * N * N, P, stream()), "prd_ln_rows_bwd")
    x = (x if x is not None else layer_norm(pair.contiguous())).view(-1, P)
    d2 = dqkvg.view(-1, 4, HC)
    dy2 = dy.view(-1, P)
    dw4, db4 = linear_wgrad(dqkvg.view(-1, 4 * HC), x, bias=True)                                # d W_q | W_k | W_v | W_g stacked [4 HC, P]
    grads = (dw4[:HC], dw4[HC:2 * HC], dw4[2 * HC:3 * HC], dw4[3 * HC:], db4[3 * HC:], *linear_wgrad(dy2, og.view(-1, HC), bias=True))
    return dpair, grads


WGRAD_MIN_ROWS = 8192


def outer_linear_bwd_reduce(T: torch.Tensor, w1: torch.Tensor, x: torch.Tensor, chunks: int = 4):
    """(dx [R, S] = sum_p T w1,  dw1 [P, S] = sum_r T x) for T [R, P, S], w1 [P, S], x [R, S] (prd_outer_linear_bwd_reduce)."""
    R_, P, S = T.shape
    dx = torch.empty(R_, S, device=T.device, dtype=F32)
    part = torch.empty(chunks, P, S, device=T.device, dtype=F32)
    check(lib().prd_outer_linear_bwd_reduce(dptr(dx), dptr(part), chunks, dptr(T), dptr(w1.contiguous()), dptr(x.contiguous()), R_, P, S, stream()),
          "prd_outer_linear_bwd_reduce")
    return dx, part.sum(0)


def sym_rows(x: torch.Tensor, scale: float = 0.5) -> torch.Tensor:
    """scale * (x + x.transpose(1, 2)) for x [b, N, N, P] (prd_sym_rows)."""
    b, N, _, P = x.shape
    out = torch.empty_like(x)
    check(lib().prd_sym_rows(dptr(out), dptr(x), float(scale), b, N, P, stream()), "prd_sym_rows")
    return out


def sym_transpose(dy: torch.Tensor) -> torch.Tensor:
    """[b, N, N, P] -> [b, N, P, N]: out[b,i,p,j] = dy[b,i,j,p] + dy[b,j,i,p] (prd_sym_transpose)."""
    b, N, _, P = dy.shape
    out = torch.empty(b, N, P, N, device=dy.device, dtype=F32)
    check(lib().prd_sym_transpose(dptr(out), dptr(dy), b, N, P, stream()), "prd_sym_transpose")
    return out


PAIR_LINEAR = os.environ.get("PRD_PAIR_LINEAR", "1") != "0"      # 0: the activation-gradient GEMMs of the backward through prd_gemm (A/B)


def pair_linear(x2: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, ln_in: bool = False,
                xn_out: Optional[torch.Tensor] = None, act: int = 0, relu_mask: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """act(LN?(x2) w^T + bias) over the rows of x2 [rows, K] on the row kernel prd_pair_linear (weights resident in LDS, rows
    streamed), with ``relu_mask`` zeroed where relu_mask <= 0.  Returns None where the row kernel does not serve the call (shape,
    arithmetic, few rows, strided input): the caller then uses ``linear``."""
    rows, K = x2.shape
    OUT = w.shape[0]
    if not (PAIR_LINEAR and x2.is_cuda and x2.is_contiguous() and rows >= WGRAD_MIN_ROWS and lib().prd_pair_linear_supported(K, OUT) == 1):
        return None
    out = torch.empty(rows, OUT, device=x2.device, dtype=F32)
    kn = (not w.is_contiguous()) and w.t().is_contiguous()      # a transposed view of a [K, OUT] matrix: staged as it lies in memory
    wsrc = w.t() if kn else w.contiguous()
    check(lib().prd_pair_linear(dptr(out), dptr(x2), dptr(wsrc), dptr(bias), rows, K, OUT, int(ln_in), dptr(xn_out), act,
                                dptr(relu_mask), int(kn), stream()), "prd_pair_linear")
    return out


def ln_rows_bwd(dy2: torch.Tensor, x2: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx of nn.LayerNorm(C, elementwise_affine=False) over the rows of x2 [rows, C] (prd_ln_rows_bwd), plus ``res`` if given."""
    rows, Cn = x2.shape
    dx = torch.empty_like(x2)
    check(lib().prd_ln_rows_bwd(dptr(dx), dptr(dy2), dptr(x2), dptr(res), rows, Cn, stream()), "prd_ln_rows_bwd")
    return dx


def pair_bias_bwd(dbias: torch.Tensor, wf: torch.Tensor, pair: torch.Tensor):
    """Backward of an attention-bias head in one pass over the pair rows (prd_pair_bias_bwd): dbias [b, H, N, N] (any layout: made
    contiguous), wf [H, P] (W diag(gamma) for the affine form), pair [b, N, N, P] -> (dx [b N N, P], xn = LN(pair) rows, d2 [b N N, H]);
    None where the kernel does not cover the shape (P = 64, H in (4, 8))."""
    b, N, _, P = pair.shape
    H = wf.shape[0]
    if P != 64 or H not in (4, 8) or not pair.is_cuda:
        return None
    x2 = pair.contiguous().view(-1, P)
    dx, xn = torch.empty_like(x2), torch.empty_like(x2)
    d2 = torch.empty(x2.shape[0], H, device=pair.device, dtype=F32)
    check(lib().prd_pair_bias_bwd(dptr(dx), dptr(xn), dptr(d2), dptr(dbias.contiguous()), dptr(wf.contiguous()), dptr(x2), b, N * N, H, P,
                                  stream()), "prd_pair_bias_bwd")
    return dx, xn, d2


def linear_wgrad(dy2: torch.Tensor, x2: torch.Tensor, bias: bool = False):
    """dW [O, I] = dy2^T x2 for row-major 2-D views dy2 [rows, O] and x2 [rows, I] (row stride = their stride(0), unit column
    stride): the weight gradient of a linear applied at every pair position; with ``bias`` also db [O] = column sums of dy2 from the
    same pass -> (dW, db).  Hand-written slab reduction (prd_linear_wgrad) for the shapes it covers -- rows >= 8192, I a multiple of
    64, O a multiple of 64 or at most 16, both up to 256 -- else a library GEMM."""
    rows, O = dy2.shape
    I = x2.shape[1]
    wide = (O % 64 == 0 and dy2.stride(0) % 2 == 0 and x2.stride(0) % 2 == 0 and dy2.storage_offset() % 2 == 0
            and x2.storage_offset() % 2 == 0)
    if (rows >= WGRAD_MIN_ROWS and (wide or O <= 16) and I % 64 == 0 and O <= 256 and I <= 256 and dy2.stride(1) == 1
            and x2.stride(1) == 1):
        dw = torch.empty(O, I, device=dy2.device, dtype=F32)
        db = torch.empty(O, device=dy2.device, dtype=F32) if bias else None
        nbytes = lib().prd_linear_wgrad_workspace(rows, O, I)
        ws = torch.empty(nbytes // 4, device=dy2.device, dtype=F32)
        check(lib().prd_linear_wgrad(dptr(dw), dptr(db), dy2.data_ptr(), x2.data_ptr(), rows, O, I, dy2.stride(0), x2.stride(0),
                                     dptr(ws), nbytes, stream()), "prd_linear_wgrad")
        return (dw, db) if bias else dw
    dw = dy2.t() @ x2
    return (dw, dy2.sum(0)) if bias else dw


def rbf_rows(z: torch.Tensor, centers: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """[b, N, N, R] radial-basis rows of the pair distances times mask_i mask_j (prd_rbf_rows)."""
    b, N, _ = z.shape
    R_ = centers.numel()
    out = torch.empty(b, N, N, R_, device=z.device, dtype=F32)
    check(lib().prd_rbf_rows(dptr(out), dptr(z.contiguous()), dptr(centers), dptr(mask.contiguous()), b, N, R_, stream()), "prd_rbf_rows")
    return out


def embed_wgrad(idx: torch.Tensor, dy2: torch.Tensor, card: int, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dtable [card, C] = scatter-sum of the rows of dy2 [rows, C] (times ``scale`` [rows] if given) by idx [rows] (int64): the
    gradient of a small embedding table looked up at every pair position (prd_embed_wgrad; card <= 128, C <= 64, contiguous inputs)."""
    rows, Cn = dy2.shape
    dt = torch.empty(card, Cn, device=dy2.device, dtype=F32)
    nbytes = lib().prd_embed_wgrad_workspace(rows, card, Cn)
    ws = torch.empty(nbytes // 4, device=dy2.device, dtype=F32)
    check(lib().prd_embed_wgrad(dptr(dt), dptr(idx, torch.int64), dptr(dy2), dptr(scale), rows, card, Cn, Cn, dptr(ws), nbytes, stream()),
          "prd_embed_wgrad")
    return dt


def embed_wgrad_multi(idxs, dy2: torch.Tensor, cards, scales=None):
    """The gradients of several small tables looked up at the same rows, in one pass over dy2 [rows, C] (prd_embed_wgrad_multi):
    ``idxs`` int64 [rows] each, ``cards`` their table sizes, ``scales`` optional per-set row factors [rows] (entries may be None).
    Returns the list of [card_k, C] gradients."""
    import ctypes
    K = len(idxs)
    rows, Cn = dy2.shape
    total = int(sum(cards))
    dt = torch.empty(total, Cn, device=dy2.device, dtype=F32)
    nbytes = lib().prd_embed_wgrad_workspace(rows, total, Cn)
    ws = torch.empty(nbytes // 4, device=dy2.device, dtype=F32)
    ip = (ctypes.c_void_p * K)(*[dptr(i, torch.int64) for i in idxs])
    sp = (ctypes.c_void_p * K)(*[dptr(s) for s in (scales if scales is not None else [None] * K)])
    cp = (ctypes.c_int * K)(*[int(c) for c in cards])
    check(lib().prd_embed_wgrad_multi(dptr(dt), ip, sp, cp, K, dptr(dy2), rows, Cn, Cn, dptr(ws), nbytes, stream()), "prd_embed_wgrad_multi")
    return list(torch.split(dt, [int(c) for c in cards], dim=0))


def tri_attn_uses_long_rows(N: int, P: int) -> bool:
    """True when rows of N positions take the re-projecting long-row core kernel (prd_hip.h: prd_tri_attn_variant)."""
    return tri_attn_variant(N, P) >= 1


def tri_attn_variant(N: int, P: int) -> int:
    """0 short rows, 1 long rows (fp32 kernel), 2 long rows (split-operand kernel), 3 key-chunked rows (N > 960) --
    prd_hip.h: prd_tri_attn_variant."""
    v = lib().prd_tri_attn_variant(N, P)
    if v < 0:
        check(v, "prd_tri_attn_variant")
    return v


def tri_attn(pair, mask, wts, H: int, c: int, *, ending: bool, residual: bool, out=None, ws=None) -> torch.Tensor:
    """wts = (q.w, k.w, v.w, gate.w, gate.b, out.w, out.b)"""
    b, N, _, P = pair.shape
    if out is None:
        out = torch.empty_like(pair)
    nbytes = workspace_bytes("tri_attn", b, N, 0, P)
    if ws is None:
        ws = torch.empty(nbytes // 4, device=pair.device, dtype=F32)
    check(lib().prd_tri_attn(dptr(out), dptr(pair), dptr(mask), *[dptr(w) for w in wts], int(ending), int(residual),
                             b, N, P, H, c, dptr(ws), ws.numel() * 4, task_queue(pair.device), stream()), "prd_tri_attn")
    return out


def pair_transition(pair, w1, b1, w2, b2, *, residual: bool, out=None) -> torch.Tensor:
    b, N, _, P = pair.shape
    if out is None:
        out = torch.empty_like(pair)
    check(lib().prd_pair_transition(dptr(out), dptr(pair), dptr(w1), dptr(b1), dptr(w2), dptr(b2), int(residual),
                                    b, N, P, task_queue(pair.device), stream()), "prd_pair_transition")
    return out


def block_tail_(pair, og, wo, bo, w1, b1, w2, b2, bias_w=None, bias_b=None) -> Optional[torch.Tensor]:
    """In place: pair += W_o og + b_o; pair += transition(pair); returns the next block's attention bias
    [b,H,N,N] = Linear(LN(pair)) when ``bias_w`` is given (else None)."""
    b, N, _, P = pair.shape
    bias_out = None
    H = 0
    if bias_w is not None:
        H = bias_w.shape[0]
        bias_out = torch.empty(b, H, N, N, device=pair.device, dtype=F32)
    check(lib().prd_block_tail(dptr(pair), dptr(og), dptr(wo), dptr(bo), dptr(w1), dptr(b1), dptr(w2), dptr(b2),
                               dptr(bias_w), dptr(bias_b), dptr(bias_out), b, N, P, H, task_queue(pair.device), stream()),
          "prd_block_tail")
    return bias_out


def coord_head(pair, z, mask, w1, b1, w2) -> torch.Tensor:
    b, N, _, P = pair.shape
    out = torch.empty(b, N, 3, device=pair.device, dtype=F32)
    check(lib().prd_coord_head(dptr(out), dptr(pair), dptr(z), dptr(mask), dptr(w1), dptr(b1), dptr(w2), b, N, P,
                               stream()), "prd_coord_head")
    return out


def remove_mean(x, mask) -> torch.Tensor:
    b, N, D = x.shape
    out = torch.empty_like(x)
    check(lib().prd_remove_mean(dptr(out), dptr(x), dptr(mask), b, N, D, stream()), "prd_remove_mean")
    return out


def reverse_update_(z, seq_t, t, noise_pred, seq_pred, noise, mask, coef, num_steps: int):
    """noise: the whole table [T-1, b, N, 3]; the kernel picks row T-1-t and decrements t."""
    b, N, _ = z.shape
    check(lib().prd_reverse_update(dptr(z), dptr(seq_t), dptr(t, torch.int64), dptr(noise_pred), dptr(seq_pred),
                                   dptr(noise), dptr(mask), dptr(coef), b, N, seq_pred.shape[-1], num_steps, stream()),
          "prd_reverse_update")


def step_boundary_(z, seq_t, t, eps_raw, seq_pred, noise, mask, coef, num_steps: int, single_next, static_single, residue_mask,
                   w_rt, ebeta_next, freqs, w_beta, sync, seq_h=None, w_seq=None):
    """Reverse update + the next step's single / time-embedding inputs in one launch (prd_hip.h: prd_step_boundary).
    ``seq_h`` (+ ``w_seq``): the sequence head's hidden units -- its last layer then runs inside this launch and ``seq_pred`` is
    written instead of read."""
    b, N, _ = z.shape
    P, TD = w_beta.shape
    hp, ldh = row_block(seq_h) if seq_h is not None else (None, 0)
    check(lib().prd_step_boundary(dptr(z), dptr(seq_t), dptr(t, torch.int64), dptr(eps_raw), dptr(seq_pred), dptr(noise), dptr(mask),
                                  dptr(coef), dptr(single_next), dptr(static_single), dptr(residue_mask), dptr(w_rt),
                                  dptr(ebeta_next), dptr(freqs), dptr(w_beta), dptr(sync, torch.int32), b, N, seq_pred.shape[-1],
                                  num_steps, static_single.shape[-1], P, TD, hp, ldh, dptr(w_seq),
                                  (seq_h.shape[-1] if seq_h is not None else 0), stream()), "prd_step_boundary")


def tri_attn_stats_floats(b: int, N: int, P: int, H: int = 4) -> int:
    """floats of softmax statistics the key-chunked core needs (0 unless rows exceed the LDS: prd_tri_attn_stats_bytes)"""
    return lib().prd_tri_attn_stats_bytes(b, N, P, H) // 4


def tri_attn_lse_supported(N: int, P: int) -> bool:
    """True when the forward core can keep the softmax statistics of its queries for prd_tri_attn_bwd_core_v2 (split-16 mode, rows
    of up to 384 positions: the second-generation short-row kernels)."""
    return (lib().prd_get_gemm_mode() == 1 and lib().prd_tri_attn_v2_form(N, P) in (1, 2)
            and lib().prd_tri_attn_bwd_core_v2_supported(N, P) == 1 and TRI_ATTN_BWD_V2)


def tri_attn_core(pair, mask, wts, H: int, c: int, *, ending: bool, og=None, stats=None, lse=None) -> torch.Tensor:
    """First launch of tri_attn alone: og[b,N,N,64]; wts = (q.w, k.w, v.w, gate.w, gate.b).  Rows beyond the LDS (N > 960) run
    key-chunked and need ``stats`` (tri_attn_stats_floats; allocated here if not given).  ``lse`` [b*N, H, N, 2] (only where
    tri_attn_lse_supported): receives (m, log2 l) of every query, which the split-16 backward core then does not recompute."""
    b, N, _, P = pair.shape
    if og is None:
        og = torch.empty(b, N, N, 64, device=pair.device, dtype=F32)
    if lse is not None:
        if not tri_attn_lse_supported(N, P) or lse.numel() != b * N * H * N * 2:
            raise ValueError("tri_attn_core: lse is kept by the split-16 short-row kernels only, as [b*N, H, N, 2]")
        check(lib().prd_tri_attn_core_v2_lse(dptr(og), dptr(lse), dptr(pair), dptr(mask), *[dptr(w) for w in wts], int(ending),
                                             b, N, P, H, c, stream()), "prd_tri_attn_core_v2_lse")
        return og
    nst = tri_attn_stats_floats(b, N, P, H)
    if nst:
        if stats is None:
            stats = torch.empty(nst, device=pair.device, dtype=F32)
        if stats.numel() < nst:
            raise ValueError("tri_attn_core: stats buffer too small")
        check(lib().prd_tri_attn_core_chunked(dptr(og), dptr(pair), dptr(mask), *[dptr(w) for w in wts], int(ending),
                                              b, N, P, H, c, dptr(stats), stats.numel() * 4, stream()), "prd_tri_attn_core_chunked")
        return og
    check(lib().prd_tri_attn_core(dptr(og), dptr(pair), dptr(mask), *[dptr(w) for w in wts], int(ending),
                                  b, N, P, H, c, stream()), "prd_tri_attn_core")
    return og


def tri_attn_v2_supported(N: int, P: int) -> bool:
    """True when the second-generation core (csrc/prd_tri2.hip) serves rows of N positions."""
    return lib().prd_tri_attn_v2_supported(N, P) == 1


def tri_attn_core_v2(pair, mask, wts, H: int, c: int, *, ending: bool, og=None) -> torch.Tensor:
    """The second-generation core called directly (split-16 arithmetic whatever the gemm mode); arguments as tri_attn_core."""
    b, N, _, P = pair.shape
    if og is None:
        og = torch.empty(b, N, N, 64, device=pair.device, dtype=F32)
    check(lib().prd_tri_attn_core_v2(dptr(og), dptr(pair), dptr(mask), *[dptr(w) for w in wts], int(ending),
                                     b, N, P, H, c, stream()), "prd_tri_attn_core_v2")
    return og


def tri_attn_core_v2_lse(pair, mask, wts, H: int, c: int, *, ending: bool, lse, og=None) -> torch.Tensor:
    """prd_tri_attn_core_v2_lse called directly (split-16 arithmetic whatever the gemm mode): og, and (m, log2 l) of every query
    into ``lse`` [b*N, H, N, 2]."""
    b, N, _, P = pair.shape
    if og is None:
        og = torch.empty(b, N, N, 64, device=pair.device, dtype=F32)
    check(lib().prd_tri_attn_core_v2_lse(dptr(og), dptr(lse), dptr(pair), dptr(mask), *[dptr(w) for w in wts], int(ending),
                                         b, N, P, H, c, stream()), "prd_tri_attn_core_v2_lse")
    return og


def tri_attn_core_fused_supported(N: int, P: int) -> bool:
    """True when prd_tri_attn_core_fused exists for this shape in the current arithmetic mode."""
    return lib().prd_tri_attn_core_fused_supported(N, P) == 1


def tri_attn_core_fused(pair, og_in, wo_in, bo_in, mask, wts, H: int, c: int, *, ending: bool, pair_out, og=None) -> torch.Tensor:
    """og of a triangle attention over the rows of ``pair + og_in W_o^T + b_o`` (the previous attention's residual update,
    applied on the fly and written to ``pair_out``, a different buffer); wts = (q.w, k.w, v.w, gate.w, gate.b)."""
    b, N, _, P = pair.shape
    if og is None:
        og = torch.empty(b, N, N, 64, device=pair.device, dtype=F32)
    check(lib().prd_tri_attn_core_fused(dptr(og), dptr(pair_out), dptr(pair), dptr(og_in), dptr(wo_in), dptr(bo_in), dptr(mask),
                                        *[dptr(w) for w in wts], int(ending), b, N, P, H, c, stream()), "prd_tri_attn_core_fused")
    return og


def tri_attn_out(pair, og, wo, bo, *, residual: bool, out=None) -> torch.Tensor:
    """Second launch of tri_attn alone: (residual ? pair : 0) + og W_o^T + b_o."""
    b, N, _, P = pair.shape
    if out is None:
        out = torch.empty_like(pair)
    check(lib().prd_tri_attn_out(dptr(out), dptr(pair), dptr(og), dptr(wo), dptr(bo), int(residual), b, N, P,
                                 task_queue(pair.device), stream()), "prd_tri_attn_out")
    return out


# ---------------------------------------------------------------------------------------------------
# single-track composites (LayerNorm + GEMMs + softmax)
# ---------------------------------------------------------------------------------------------------

def cached_pack(module, name: str, params, build):
    """Memoise a packed copy of several parameters on ``module`` until any of them changes in place."""
    key = tuple((p.data_ptr(), p._version) for p in params)
    cache = module.__dict__.setdefault("_prd_pack_cache", {})
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            cache[name] = (key, build())
    return cache[name][1]


SPA_CORE = os.environ.get("PRD_SPA_CORE", "1") != "0"       # 0: SPAttention's logits / softmax / P V as three GEMM-path launches (A/B)


def pack_attention(wq, wk, wv, wg, bg, q_scale: float):
    """[q | k | v | gate] projections as ONE GEMM: weight [4HC, S], bias (gate only), per-column scale (q only)."""
    HC = wq.shape[0]
    w = torch.cat([wq, wk, wv, wg], dim=0).contiguous()
    bias = torch.cat([torch.zeros(3 * HC, device=w.device, dtype=F32), bg]).contiguous()
    colscale = torch.cat([torch.full((HC,), q_scale, device=w.device, dtype=F32),
                          torch.ones(3 * HC, device=w.device, dtype=F32)]).contiguous()
    return w, bias, colscale


def project_qkvg(x_normed, packed, HC: int, ln_a: bool = False) -> torch.Tensor:
    """[q * scale | k | v | sigmoid(gate)] = one packed GEMM over the node rows (``packed`` from ``pack_attention``)."""
    b, N, S = x_normed.shape
    w, pbias, colscale = packed
    L = 4 * HC
    qkvg = torch.empty(b, N, L, device=x_normed.device, dtype=F32)
    gemm(x_normed, w, qkvg, b * N, L, S, S, S, L, bias=pbias, colscale=colscale, act=2, act_from=3 * HC, a_ln=ln_a)
    return qkvg


def gated_attention_single(x_normed, mask, bias, packed, wo, bo, H: int, c: int, *,
                           key_mask: bool, resid: Optional[torch.Tensor], ln_a: bool = False,
                           qkvg: Optional[torch.Tensor] = None, rscale: Optional[torch.Tensor] = None,
                           out_ln: Optional[torch.Tensor] = None, logits_fp32: bool = True) -> torch.Tensor:
    """Multi-head gated attention over the node axis with an additive [b,H,N,N] bias.

    Covers reference modules.py:185-225 (c = head_dim, key mask filled with -2**15, q pre-scaled by
    1/sqrt(c)) and models/AF2_modules.py:251-293,613-628 (c = single_dim, no mask).  ``packed`` comes
    from ``pack_attention``.  Returns ``resid + out_proj(...)`` (or the bare update when ``resid`` is None).
    ``ln_a``: ``x_normed`` is the raw input and its (affine-free) LayerNorm is fused into the q|k|v|g projection.
    ``qkvg``: the projection, if the caller already computed it (``project_qkvg``, e.g. on a side stream).
    ``rscale``: per-column factor of ``resid`` (an affine-LayerNorm-ed residual given as the plain normalised rows x gamma, with
    beta folded into ``bo`` by the caller).  ``out_ln``: buffer for LN(result) (no affine) -- only with an output projection that takes
    the K-slab path (``slab_ok(b N, S_out, H c)``), whose reduce launch holds whole output rows.  ``logits_fp32`` (wide heads only):
    the logits on fp32 MFMA in either arithmetic -- what a training forward needs (below); sampling passes False."""
    b, N, S = x_normed.shape
    HC = H * c
    w, pbias, colscale = packed
    L = 4 * HC
    if qkvg is None:
        if ln_a and not ln_fusable(S):
            x_normed, ln_a = layer_norm(x_normed.contiguous()), False
        qkvg = project_qkvg(x_normed, packed, HC, ln_a)
    o = torch.empty(b, N, HC, device=x_normed.device, dtype=F32)
    if c == 16 and HC == 64:
        # heads of width 16 (FoldingBlock.single_attn): fused logits + bias + mask + softmax + PV + gate
        qp, ldq = row_block(qkvg)
        check(lib().prd_single_attn_core(dptr(o), qp, ldq, dptr(bias), dptr(mask) if key_mask else None,
                                         b, N, H, c, stream()), "prd_single_attn_core")
        return linear(o, wo, bo, resid=resid, rscale=rscale)
    if SPA_CORE and not logits_fp32 and lib().prd_spa_attn_core_supported(N, c) == 1:
        # logits + softmax + P V in one launch (prd_spa_attn_core: split-16 arithmetic; a training forward keeps the fp32-MFMA logits)
        qp, ldq = row_block(qkvg)
        nws = int(lib().prd_spa_attn_core_workspace(b, N, H, c))
        wsb = torch.empty(max(nws // 4, 4), device=o.device, dtype=F32)
        check(lib().prd_spa_attn_core(dptr(o), qp, ldq, dptr(bias), dptr(mask) if key_mask else None, b, N, H, c, dptr(wsb), nws, stream()),
              "prd_spa_attn_core")
        return linear(o, wo, bo, resid=resid, rscale=rscale, slab=slab_ok(b * N, wo.shape[0], HC), out_ln=out_ln)
    ldp = round_up(N, 4)
    logits = torch.empty(b, H, N, ldp, device=x_normed.device, dtype=F32)
    gemm(qkvg, qkvg, logits, N, N, c, L, L, ldp, b_off=HC, G1=b, G2=H, sa=(N * L, c), sb=(N * L, c),
         sc=(H * N * ldp, N * ldp), addmat=bias, sad=(H * N * N, N * N), ldadd=N,
         colmask=(mask if key_mask else None), scm1=N, fill=-(2.0 ** 15),
         tile_hint=32 if logits_fp32 else 0)
    # logits_fp32: fp32 MFMA in either arithmetic -- a softmax follows, and with full-strength weights (|logit| ~ 50) the split
    # operands' 2^-22 shows in the GRADIENTS of q / k at the 1e-4 level (tests/test_training_gpu.py: the backward differentiates an
    # fp32 restatement); a sampling step has no such consumer and stays inside the 1e-5 parity bound with the split operands
    pv = dict(b_off=2 * HC, G1=b, G2=H, sa=(H * N * ldp, N * ldp), sb=(N * L, c), sc=(N * HC, c), b_kn=True, mulmat=qkvg, mul_off=3 * HC,
              smu=(N * L, c), ldmul=L, a_scale=1024.0)                                                        # probabilities: see PrdGemm.a_scale
    # the row softmax rides in the P V product where the library has that form (a_ln = 2); else its own launch
    if ldp != N or gemm(logits, qkvg, o, N, c, N, ldp, L, HC, a_ln=2, unsupported_ok=True, **pv) is None:
        softmax_rows_(logits, N)
        gemm(logits, qkvg, o, N, c, N, ldp, L, HC, **pv)
    return linear(o, wo, bo, resid=resid, rscale=rscale, slab=slab_ok(b * N, wo.shape[0], HC), out_ln=out_ln)


def project_many(single, packed, P_first: int, *, act: int, act_from: int, xhat=None):
    """ONE GEMM over LN(single) (LayerNorm without affine, fused) for several consumers of the same normalised rows:
    ``packed`` = (W [Ncat, S], bias [Ncat] or None, colscale [Ncat] or None).  Returns (x = LN(single) [b,N,S], C [b,N,Ncat]);
    the consumers take column blocks of C (row_block).  ``xhat``: LN(single) if the producer of ``single`` already wrote it."""
    b, N, S = single.shape
    w, pbias, colscale = packed
    Ncat = w.shape[0]
    out = torch.empty(b, N, Ncat, device=single.device, dtype=F32)
    if xhat is not None:
        gemm(xhat, w, out, b * N, Ncat, S, S, S, Ncat, bias=pbias, colscale=colscale, act=act, act_from=act_from)
        return xhat, out
    if ln_fusable(S):
        x = torch.empty_like(single)
        gemm(single, w, out, b * N, Ncat, S, S, S, Ncat, bias=pbias, colscale=colscale, act=act, act_from=act_from, a_ln=True, ln_out=x)
    else:
        x = layer_norm(single)
        gemm(x, w, out, b * N, Ncat, S, S, S, Ncat, bias=pbias, colscale=colscale, act=act, act_from=act_from)
    return x, out


def transition_single(single, w1, b1, w2, b2, *, residual: bool, wsum1=None, want_ln: bool = False):
    """single + W2 relu(W1 LN(single) + b1) + b2 (modules.py:306-311).  ``wsum1`` = row sums of W1: lets the first linear take
    the K-slab path, where the LayerNorm is applied by linearity (PrdGemm.wsum).  ``want_ln``: also return
    LN(result) -- the next linears of the block start with it -- when the second linear runs on the K-slab path, whose
    reduce launch holds whole output rows; returns (result, LN(result) or None) then."""
    b, N, S = single.shape
    Hd = w1.shape[0]
    slab1 = wsum1 is not None and slab_ok(b * N, Hd, S)
    slab2 = slab_ok(b * N, S, Hd)
    h = linear(single, w1, b1, act=1, ln_a=True, slab=slab1, wsum=wsum1 if slab1 else None)   # LayerNorm (no affine) fused into the first linear
    xhat = torch.empty_like(single) if (want_ln and slab2 and S <= 512) else None
    out = linear(h, w2, b2, resid=single if residual else None, slab=slab2, out_ln=xhat)
    return (out, xhat) if want_ln else out
