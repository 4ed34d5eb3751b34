#!/usr/bin/env python
"""Per-phase cycle totals of a row kernel from in-kernel stamps (library built with -DPRD_TIMING, PRD_LIB=...).
usage: phase_timing.py tri_mul_proj|tri_mul_out|tri_mul_contract [N [gemm_mode]]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_redesign_amd import _lib, ops  # noqa: E402
from protein_redesign_amd.constants import make_args  # noqa: E402
from protein_redesign_amd.diffusion_model import ProteinReDiffModel  # noqa: E402
from protein_redesign_amd.synthetic import deterministic_state_dict  # noqa: E402
from protein_redesign_amd.weights import spec_tensors  # noqa: E402

PHASES = {"tri_mul_contract": ["load issue + LDS reads + MFMA", "wait next chunk + split + LDS writes", "barrier", "first chunk (exposed)",
                               "output stores", "-", "tile decode"],
          "tri_mul_proj": ["fetch+prefetch issue", "wait for row", "layernorm", "mfma", "epilogue+stores", "exit", "prologue"],
          "outer_linear": ["wait x_i + products + MFMA issue + next row's requests", "partial stores (wait MFMAs)", "barrier", "reduction + epilogue of the previous row",
                           "task decode + x_j slice through LDS", "-", "prologue (W1 slice -> registers)", "exit"],
          "outer_linear_v1": ["task decode", "K loop", "u / pair rows + store", "mirrored rows + store", "-", "-", "prologue", "exit"],
          "pair_tail": ["decode + load issue", "wait rows + out-projection", "LayerNorm + split", "transition GEMMs", "epilogue + stores",
                        "next attention bias", "prologue", "exit"],
          "tri_mul_out": ["decode + load issue", "wait row + LN", "gate GEMM + sigmoid", "wait O + LN", "projection GEMM",
                          "epilogue + stores", "prologue", "exit"]}


def main():
    which = sys.argv[1]
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    args = make_args(single_dim=512, pair_dim=64, num_blocks=2, num_steps=1000)
    m = ProteinReDiffModel(args)
    m.load_state_dict(deterministic_state_dict(spec_tensors(args), seed=1))
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(0)
    pair = torch.randn(1, N, N, 64, generator=g).cuda()
    mask = torch.ones(1, N).cuda()
    blk = m.Denoiser.folding_blocks[0]
    ws = torch.empty(m.Denoiser.ws_floats(1, N), device="cuda")
    L = _lib.lib()
    L.prd_debug_read.argtypes = [ctypes.c_void_p]
    L.prd_debug_select.argtypes = [ctypes.c_int]
    assert L.prd_debug_select({"tri_mul_proj": 1, "tri_mul_out": 2, "tri_mul_contract": 3}.get(which, 0)) == 0
    if len(sys.argv) > 3:
        L.prd_set_gemm_mode(int(sys.argv[3]))
    buf = np.zeros(256 * 16 * 8 * 4, dtype=np.uint64)
    with torch.inference_mode():
        if which == "outer_linear":
            single = torch.randn(1, N, 512, generator=g).cuda()
            for _ in range(3):
                blk.outer_linear.run(single, pair.clone(), residual=True, out=pair.clone())
            torch.cuda.synchronize()
            L.prd_debug_read_pair.argtypes = [ctypes.c_void_p]
            assert L.prd_debug_read_pair(buf.ctypes.data) == 0
        elif which == "pair_tail":
            nxt, ta, pf = m.Denoiser.folding_blocks[1], blk.pair_attn_ending.attn, blk.pair_fc
            og = torch.randn(1, N, N, 64, generator=g).cuda()
            for _ in range(3):
                ops.block_tail_(pair.clone(), og, ta.out_proj.weight, ta.out_proj.bias, pf[1].weight, pf[1].bias, pf[3].weight, pf[3].bias,
                                nxt.attn_bias[1].weight, nxt.attn_bias[1].bias)
            torch.cuda.synchronize()
            L.prd_debug_read_pair.argtypes = [ctypes.c_void_p]
            assert L.prd_debug_read_pair(buf.ctypes.data) == 0
        else:
            for _ in range(3):
                blk.pair_mul_outgoing.run(pair, mask, residual=True, out=pair.clone(), ws=ws)
            torch.cuda.synchronize()
            assert L.prd_debug_read(buf.ctypes.data) == 0
    t = buf[: 256 * 16 * 8].reshape(256, 16, 8).astype(np.float64)
    if which == "tri_mul_contract":
        t = buf[: 256 * 16 * 8].reshape(256, 16, 8).astype(np.float64)
    nw = int((t.sum(axis=(0, 2)) > 0).sum())
    t = t[:, :nw, : len(PHASES[which])]
    tot = t.sum(axis=2)
    print(f"{which}: N={N}, {nw} waves/WG; per-wave total cycles mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
    for k, name in enumerate(PHASES[which]):
        print(f"  {name:20s} mean {t[:, :, k].mean():9.0f}  ({100 * t[:, :, k].mean() / tot.mean():5.1f} %)   max over waves {t[:, :, k].max():9.0f}")


if __name__ == "__main__":
    main()
