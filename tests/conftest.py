import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def mismatch_report(got, want, rerun=None, limit=8):
    """Assertion message for a failed whole-tensor comparison: HOW MANY elements are off by more than 1e-3 of the rms of ``want``, WHERE
    the first of them sit, and -- with ``rerun`` (a callable that evaluates the GPU operator again) -- whether a second launch
    reproduces ``got`` bit for bit (a deterministic error of the kernel) or not (a race, or a box that drops bits).  The kernels are
    deterministic by construction (static task order, no atomics on the data path), so the second case never is "noise"."""
    got, want = torch.as_tensor(got), torch.as_tensor(want)
    d = (got.double() - want.double()).abs()
    rms = float(want.double().pow(2).mean().sqrt())
    bad = torch.nonzero(d > 1e-3 * rms)
    msg = f"rel-L2 {rel_l2(got, want):.3e}; {bad.shape[0]} of {d.numel()} elements off by > 1e-3 rms; first at {bad[:limit].tolist()}"
    if not torch.isfinite(got).all():
        msg += f"; {int((~torch.isfinite(got)).sum())} non-finite"
    if rerun is not None:
        again = torch.as_tensor(rerun())
        same = torch.equal(again, got)
        msg += ("; a second launch reproduces it bit for bit" if same else
                f"; a second launch DIFFERS (rel-L2 of the second one against the oracle {rel_l2(again, want):.3e}): race or faulty box")
    return msg


@pytest.fixture(scope="session")
def golden():
    def load(name):
        path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
        z = np.load(path, allow_pickle=False)
        case = json.loads(str(z["case"]))
        return case, z
    return load


# ---------------------------------------------------------------------------------------------------
# PRD_LDS_POISON=1 (debug runs of the GPU suite): fill the LDS of every CU with NaN patterns before EVERY operator call, so that a
# kernel that reads LDS it never wrote fails deterministically instead of depending on what ran before on the box (round 5: the
# merge of tri_attn_core_v2 / v3 read the partial slot of a helper wave without tiles for rows shorter than 97 positions).
# Needs tools/ubench/liblds_poison.so (build line in tools/ubench/lds_poison.hip).
# ---------------------------------------------------------------------------------------------------
def _install_lds_poison():
    import ctypes
    import types
    from protein_redesign_amd import ops
    so = os.path.join(ROOT, "tools", "ubench", "liblds_poison.so")
    if not os.path.exists(so):
        raise RuntimeError(f"PRD_LDS_POISON is set but {so} is missing: build it (see tools/ubench/lds_poison.hip)")
    pois = ctypes.CDLL(so)
    pois.prd_dbg_poison_lds.argtypes = [ctypes.c_uint, ctypes.c_void_p]
    skip = {"task_queue", "round_up", "_off", "cached_pack", "row_block", "workspace_bytes", "gemm_workspace", "slab_ok", "ln_fusable",
            "split16_gemm_ok"}

    def wrap(fn):
        def w(*a, **k):
            if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                assert pois.prd_dbg_poison_lds(0x7fc00000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            return fn(*a, **k)
        w.__wrapped__ = fn
        return w
    for name in dir(ops):
        fn = getattr(ops, name)
        if isinstance(fn, types.FunctionType) and fn.__module__ == ops.__name__ and name not in skip and not hasattr(fn, "__wrapped__"):
            setattr(ops, name, wrap(fn))


if os.environ.get("PRD_LDS_POISON"):
    _install_lds_poison()
