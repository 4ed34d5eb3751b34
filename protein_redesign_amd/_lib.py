"""ctypes binding of libprd_hip.so (include/prd_hip.h).  The product path is HIP only: importing an
operator without the built library, or calling it on a CPU tensor, raises -- there is no fallback."""
from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRD_LIB", os.path.join(HERE, "libprd_hip.so"))   # PRD_LIB: experiment builds

vp, ci, cf, cz, cll = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong


class PrdGemm(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("M", ci), ("N", ci), ("K", ci),
        ("lda", ci), ("ldb", ci), ("ldc", ci),
        ("G1", ci), ("G2", ci),
        ("sa1", cll), ("sa2", cll), ("sb1", cll), ("sb2", cll), ("sc1", cll), ("sc2", cll),
        ("b_kn", ci), ("alpha", cf), ("colscale", vp),
        ("bias", vp), ("act", ci), ("act_from", ci),
        ("addmat", vp), ("sad1", cll), ("sad2", cll), ("ldadd", ci),
        ("colmask", vp), ("scm1", cll), ("fill", cf),
        ("rowmask", vp), ("srm1", cll),
        ("mulmat", vp), ("smu1", cll), ("smu2", cll), ("ldmul", ci),
        ("resid", vp), ("sr1", cll), ("sr2", cll), ("ldr", ci),
        ("tile_hint", ci),
        ("a_ln", ci),
        ("ln_out", vp), ("ldlo", ci),
        ("arith", ci),
        ("C2", vp), ("ldc2", ci), ("n_split", ci),
        ("rowmask_cols", ci),
        ("rscale", vp),
        ("ws", vp), ("ws_bytes", cz),
        ("wsum", vp),
        ("out_ln", vp), ("ldol", ci),
        ("a_scale", cf),
        ("mul_pos", ci),
    ]


# name -> argtypes (every entry point of include/prd_hip.h; tests check the export list against the header)
SIGNATURES = {
    "prd_version": [],
    "prd_tri_attn_variant": [ci, ci, ci],
    "prd_gemm": [C.POINTER(PrdGemm), vp],
    "prd_gemm_slab_workspace": [ci, ci, ci],
    "prd_gemm_slab_ok": [ci, ci, ci, ci],
    "prd_ln_rows": [vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "prd_softmax_rows": [vp, ci, ci, ci, vp],
    "prd_static_pair": [vp] * 13 + [ci] * 5 + [vp],
    "prd_atom_embed": [vp] * 5 + [ci] * 4 + [vp],
    "prd_single_init": [vp] * 5 + [ci] * 3 + [vp],
    "prd_time_embed": [vp] * 4 + [ci] * 4 + [vp],
    "prd_pair_init": [vp] * 7 + [ci] * 5 + [vp],
    "prd_pair_bias": [vp] * 6 + [ci] * 4 + [vp],
    "prd_pair_bias2": [vp] * 6 + [ci] + [vp] * 5 + [ci] * 4 + [vp],
    "prd_opm_pair": [vp] * 6 + [ci] * 6 + [vp],
    "prd_pair_head_supported": [ci, ci, ci, ci],
    "prd_pair_head": [vp] * 7 + [ci] + [vp] * 3 + [ci, ci] + [vp] * 5 + [ci] + [vp] * 5 + [ci] * 5 + [vp],
    "prd_outer_linear": [vp] * 4 + [ci] + [vp] * 2 + [ci] * 5 + [vp, ci, vp],
    "prd_tri_mul": [vp] * 11 + [ci] * 5 + [vp, cz, vp, ci, vp],
    "prd_tri_mul_contract": [vp, vp, ci, ci, ci, ci, vp],
    "prd_tri_mul_chain_supported": [ci, ci, ci],
    "prd_tri_attn_core_fused_supported": [ci, ci, ci],
    "prd_tri_attn_core_fused": [vp] * 12 + [ci] * 6 + [vp],
    "prd_tri_mul_chain": [vp, vp, vp, vp, ci, ci, ci, vp, cz, ci, vp],
    "prd_tri_mul_out_bwd": [vp] * 15 + [ci] * 4 + [vp],
    "prd_tri_mul_bwd_operands": [vp, vp, ci, ci, ci, vp],
    "prd_tri_mul_proj_bwd": [vp] * 13 + [ci] * 5 + [vp],
    "prd_tri_attn_bwd_core": [vp] * 9 + [ci] * 6 + [vp],
    "prd_tri_attn_bwd_core_v2_supported": [ci, ci],
    "prd_tri_attn_bwd_core_v2": [vp] * 12 + [ci] * 6 + [vp],
    "prd_ln_rows_bwd": [vp, vp, vp, vp, cll, ci, vp],
    "prd_pair_bias_bwd": [vp, vp, vp, vp, vp, vp, ci, cll, ci, ci, vp],
    "prd_sym_transpose": [vp, vp, ci, ci, ci, vp],
    "prd_sym_rows": [vp, vp, cf, ci, ci, ci, vp],
    "prd_outer_linear_bwd_reduce": [vp, vp, ci, vp, vp, vp, cll, ci, ci, vp],
    "prd_pair_linear_supported": [ci, ci, ci],
    "prd_pair_linear": [vp, vp, vp, vp, cll, ci, ci, ci, vp, ci, vp, ci, ci, vp],
    "prd_linear_wgrad_workspace": [cll, ci, ci],
    "prd_linear_wgrad": [vp, vp, vp, vp, cll, ci, ci, ci, ci, vp, cz, ci, vp],
    "prd_embed_wgrad_workspace": [cll, ci, ci],
    "prd_embed_wgrad": [vp, vp, vp, vp, cll, ci, ci, ci, vp, cz, vp],
    "prd_rbf_rows": [vp, vp, vp, vp, ci, ci, ci, vp],
    "prd_embed_wgrad_multi": [vp, vp, vp, vp, ci, vp, cll, ci, ci, vp, cz, vp],
    "prd_tri_attn": [vp] * 10 + [ci] * 7 + [vp, cz, vp, ci, vp],
    "prd_pair_transition": [vp] * 6 + [ci] * 4 + [vp, ci, vp],
    "prd_block_tail": [vp] * 11 + [ci] * 4 + [vp, ci, vp],
    "prd_single_attn_core": [vp, vp, ci, vp, vp] + [ci] * 4 + [vp],
    "prd_coord_head": [vp] * 7 + [ci] * 4 + [vp],
    "prd_remove_mean": [vp] * 3 + [ci] * 3 + [vp],
    "prd_reverse_update": [vp] * 8 + [ci] * 4 + [vp],
    "prd_step_boundary": [vp] * 16 + [ci] * 7 + [vp, ci, vp, ci] + [vp],
    "prd_tri_attn_core": [vp] * 8 + [ci] * 7 + [vp],
    "prd_tri_attn_core_v2": [vp] * 8 + [ci] * 7 + [vp],
    "prd_tri_attn_core_v2_lse": [vp] * 9 + [ci] * 7 + [vp],
    "prd_tri_attn_v2_form": [ci, ci, ci],
    "prd_tri_attn_core_chunked": [vp] * 8 + [ci] * 6 + [vp, cz, vp],
    "prd_tri_attn_stats_bytes": [ci] * 5,
    "prd_tri_attn_v2_supported": [ci, ci, ci],
    "prd_tri_attn_out": [vp] * 5 + [ci] * 4 + [vp, ci, vp],
    "prd_spa_attn_core_supported": [ci, ci, ci],
    "prd_spa_attn_core_workspace": [ci, ci, ci, ci],
    "prd_spa_attn_core": [vp, vp, ci, vp, vp, ci, ci, ci, ci, vp, cz, ci, vp],
    "prd_workspace_bytes": [C.c_char_p, ci, ci, ci, ci],
    "prd_tri_attn_pair_supported": [ci, ci, ci],
    "prd_tri_attn_pair": [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp],
}

GEMM_MODES = {"fp32": 0, "split16": 1, "bf16x3": 1}      # "bf16x3": earlier name of the split-operand mode
DEFAULT_GEMM_MODE = "split16"       # process default of the Python host side (env PRD_GEMM_MODE overrides)

# entry points that take the arithmetic as their last argument before the stream ...
_ARITH_BEFORE_STREAM = ("prd_coord_head", "prd_pair_head", "prd_pair_init", "prd_opm_pair", "prd_outer_linear", "prd_tri_mul", "prd_tri_mul_contract", "prd_tri_mul_proj_bwd",
                        "prd_tri_attn", "prd_tri_attn_core", "prd_tri_attn_out", "prd_pair_transition", "prd_block_tail", "prd_tri_mul_chain",
                        "prd_linear_wgrad", "prd_pair_linear", "prd_spa_attn_core", "prd_tri_attn_pair")
# ... and the queries that take it as their last argument
_ARITH_LAST = ("prd_tri_attn_variant", "prd_tri_mul_chain_supported", "prd_tri_attn_core_fused_supported", "prd_tri_attn_stats_bytes",
               "prd_gemm_slab_ok", "prd_pair_head_supported", "prd_pair_linear_supported", "prd_spa_attn_core_supported",
               "prd_tri_attn_pair_supported")
# entry points without an arithmetic that still dispatch between kernel generations: the PRD_TUNE_* switch word alone
_TUNE_BEFORE_STREAM = ("prd_tri_attn_core_v2", "prd_tri_attn_core_v2_lse")
_TUNE_LAST = ("prd_tri_attn_v2_supported", "prd_tri_attn_v2_form")


def tune_from_env(env=None) -> int:
    """The PRD_TUNE_* switch word (prd_hip.h) from the A/B environment variables -- read HERE, on the host side: the library
    itself never looks at the environment."""
    env = os.environ if env is None else env

    def geti(name, default):
        v = env.get(name)
        return default if v is None or v == "" else int(v)

    t = geti("PRD_TA_VARIANT", 0) & 15
    if geti("PRD_TA2_V3", 1) == 0:
        t |= 1 << 4
    if geti("PRD_TA2_LONG", 1) == 0:
        t |= 1 << 5
    f = geti("PRD_TA2_FLAGS", -1)
    if f >= 0:
        t |= (1 << 6) | ((f & 31) << 7)
    if geti("PRD_OL_VARIANT", 0) == 1:
        t |= 1 << 12
    nw = geti("PRD_TMS_NW", 8)
    t |= (1 << 13) if nw == 12 else (2 << 13) if nw == 16 else 0
    if geti("PRD_GEMM_KG", 1) == 0:
        t |= 1 << 15
    if geti("PRD_GEMM_SLAB", 1) == 0:
        t |= 1 << 16
    if geti("PRD_GEMM_BRING", 1) == 0:
        t |= 1 << 17
    if geti("PRD_GEMM_XCDCOLS", 0) == 1:
        t |= 1 << 18
    if geti("PRD_TA2_TAIL", 1) == 0:
        t |= 1 << 19
    if geti("PRD_TA2_XCD8", 1) == 0:
        t |= 1 << 20
    if geti("PRD_TA2_GV", 1) == 0:
        t |= 1 << 21
    if geti("PRD_TMP_NW", 12) == 16:
        t |= 1 << 22
    if geti("PRD_TMS_DEPTH", 2) == 3 and nw == 8:
        t |= 3 << 13
    return t


ABI_VERSION = 101          # include/prd_hip.h PRD_VERSION this binding is written against (101: prd_step_boundary's sync = 2 int32)


class _Library:
    """The loaded C library plus the ONE piece of state of the host side: the arithmetic (prd_hip.h: PRD_ARITH_*) that the
    Python operators pass to every call.  The C ABI itself is stateless; ``prd_set_gemm_mode`` / ``prd_get_gemm_mode`` live
    here, in the host language, with the names round 2 used so that callers and tests read the same."""

    def __init__(self, cdll, mode: int, tune: int = 0):
        self._cdll = cdll
        self._mode = mode
        self._tune = tune               # PRD_TUNE_* switches (A/B measurements), injected with the arithmetic
        self._wrapped = {}

    def prd_set_tune(self, tune: int) -> int:
        if tune < 0 or tune >= (1 << 23):
            return -1
        self._tune = int(tune)
        return 0

    def prd_get_tune(self) -> int:
        return self._tune

    def prd_set_gemm_mode(self, mode: int) -> int:
        if mode not in (0, 1):
            return -1
        self._mode = int(mode)
        return 0

    def prd_get_gemm_mode(self) -> int:
        return self._mode

    def __getattr__(self, name):
        fn = self._wrapped.get(name)
        if fn is None:
            raw = getattr(self._cdll, name)
            if name in _ARITH_BEFORE_STREAM:
                def fn(*args, _raw=raw):
                    return _raw(*args[:-1], self._mode | (self._tune << 8), args[-1])
            elif name in _ARITH_LAST:
                def fn(*args, _raw=raw):
                    return _raw(*args, self._mode | (self._tune << 8))
            elif name in _TUNE_BEFORE_STREAM:
                def fn(*args, _raw=raw):
                    return _raw(*args[:-1], self._tune, args[-1])
            elif name in _TUNE_LAST:
                def fn(*args, _raw=raw):
                    return _raw(*args, self._tune)
            else:
                fn = raw
            self._wrapped[name] = fn
        return fn


_lib = None


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m protein_redesign_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP hot path.")
        cdll = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(cdll, name)
            fn.argtypes = argtypes
            fn.restype = cz if name in ("prd_workspace_bytes", "prd_linear_wgrad_workspace", "prd_embed_wgrad_workspace", "prd_tri_attn_stats_bytes",
                                        "prd_gemm_slab_workspace", "prd_spa_attn_core_workspace") else ci
        if cdll.prd_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports PRD_VERSION {cdll.prd_version()}, this binding was written against {ABI_VERSION} "
                               "(include/prd_hip.h lists what changed): rebuild with `python -m protein_redesign_amd.build`")
        mode = os.environ.get("PRD_GEMM_MODE", DEFAULT_GEMM_MODE)
        if os.environ.get("PRD_BF16X3"):                               # older spelling of PRD_GEMM_MODE=bf16x3
            mode = "bf16x3"
        if mode not in GEMM_MODES:
            raise RuntimeError(f"PRD_GEMM_MODE must be one of {sorted(GEMM_MODES)}, got {mode!r}")
        _lib = _Library(cdll, GEMM_MODES[mode], tune_from_env())
    return _lib


def arith() -> int:
    """The arithmetic the operators pass to the library (PRD_ARITH_FP32 = 0 / PRD_ARITH_SPLIT16 = 1)."""
    return lib().prd_get_gemm_mode()


ARITH_NAMES = {0: "fp32 (PRD_ARITH_FP32)", 1: "split16 (PRD_ARITH_SPLIT16)"}


class arithmetic:
    """``with arithmetic("fp32"): ...`` -- the operators called inside pass that arithmetic to the library; the previous default
    comes back on exit (also on an exception).  ``None`` leaves the default alone.  The library itself is stateless (prd_hip.h):
    this only changes what the Python binding injects, so it is a fresh CALL in another arithmetic, never a re-exec."""

    def __init__(self, mode):
        if isinstance(mode, str):
            if mode not in GEMM_MODES:
                raise ValueError(f"arithmetic must be one of {sorted(GEMM_MODES)}, got {mode!r}")
            mode = GEMM_MODES[mode]
        self.mode = mode
        self.prev = None

    def __enter__(self):
        if self.mode is not None:
            self.prev = lib().prd_get_gemm_mode()
            if lib().prd_set_gemm_mode(self.mode) != 0:
                raise ValueError(f"invalid arithmetic {self.mode!r}")
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            lib().prd_set_gemm_mode(self.prev)
        return False


class NonFiniteError(RuntimeError):
    """A sampling loop or an optimisation step produced inf / NaN (see ProteinReDiffModel.nonfinite_policy)."""


def row_gemm_description(b3: bool) -> str:
    """What the GEMMs of the pair kernels compute in, for bench.py's JSON line (stated truthfully, not as a precision claim)."""
    if b3:
        return ("split16: fp32 operands split into fp16 hi + lo (both rounded to nearest: 24 bits; 3 products hi*hi + hi*lo + lo*hi) "
                "and multiplied on the fp16 MFMA pipe with fp32 accumulation -- the row GEMMs of the pair track, the node-row "
                "linears of the single track (incl. SPAttention's logits / P*V), the coordinate head, the triangle-multiplication contraction, "
                "Q*K^T and P*V of the triangle attention; "
                "the single-track attention core and pair_bias run fp32 MFMA / FMA.  "
                "Parity tolerances are the same as in fp32 mode (PRD_GEMM_MODE=fp32)")
    return "fp32-mfma"


def check(code: int, what: str):
    if code != 0:
        names = {-1: "PRD_ERR_ARG", -2: "PRD_ERR_ALIGN", -3: "PRD_ERR_UNSUPPORTED", -4: "PRD_ERR_WORKSPACE"}
        raise RuntimeError(f"{what} failed: {names.get(code, 'hipError_t ' + str(code))}")


def dptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("protein_redesign_amd operators run on the GPU only (got a CPU tensor); "
                           "there is no CPU fallback")
    if t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("expected a contiguous tensor")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
