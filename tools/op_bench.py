#!/usr/bin/env python
"""Time single operators / kernels of the hot path at the bench shape with HIP events, in both row-GEMM modes (optionally with an
experiment build: PRD_LIB=/path/to/lib.so).  usage: op_bench.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from protein_redesign_amd import _lib, ops  # noqa: E402
from protein_redesign_amd.constants import make_args  # noqa: E402
from protein_redesign_amd.diffusion_model import ProteinReDiffModel  # noqa: E402
from protein_redesign_amd.synthetic import deterministic_state_dict  # noqa: E402
from protein_redesign_amd.weights import spec_tensors  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    args = make_args(single_dim=512, pair_dim=64, num_blocks=2, num_steps=1000)
    m = ProteinReDiffModel(args)
    m.load_state_dict(deterministic_state_dict(spec_tensors(args), seed=1))
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(0)
    pair = torch.randn(1, N, N, 64, generator=g).cuda()
    single = torch.randn(1, N, 512, generator=g).cuda()
    mask = torch.ones(1, N).cuda()
    blk, nxt = m.Denoiser.folding_blocks[0], m.Denoiser.folding_blocks[1]
    ws = torch.empty(m.Denoiser.ws_floats(1, N), device="cuda")
    og = torch.empty(1, N, N, 64, device="cuda")
    ta = blk.pair_attn_ending.attn
    pf = blk.pair_fc
    rows = {}
    for mode in ("fp32", "split16"):
        _lib.lib().prd_set_gemm_mode(_lib.GEMM_MODES[mode])
        with torch.inference_mode():
            res = {
                "tri_mul(out: proj+contract+out)": timeit(lambda: blk.pair_mul_outgoing.run(pair, mask, residual=True, out=pair, ws=ws)),
                "tri_mul(in)": timeit(lambda: blk.pair_mul_incoming.run(pair, mask, residual=True, out=pair, ws=ws)),
                "tri_attn_core(start)": timeit(lambda: ops.tri_attn_core(pair, mask, ta.weights()[:5], 4, 16, ending=False, og=og)),
                "tri_attn_core(end)": timeit(lambda: ops.tri_attn_core(pair, mask, ta.weights()[:5], 4, 16, ending=True, og=og)),
                "outer_linear(+ln+u)": timeit(lambda: blk.outer_linear.run(single, pair, residual=True, out=pair)),
                "block_tail": timeit(lambda: ops.block_tail_(pair, og, ta.out_proj.weight, ta.out_proj.bias, pf[1].weight, pf[1].bias,
                                                             pf[3].weight, pf[3].bias, nxt.attn_bias[1].weight, nxt.attn_bias[1].bias)),
                "tri_attn(start: core+out)": timeit(lambda: blk.pair_attn_starting.run(pair, mask, residual=True, out=pair, ws=ws)),
                "pair_transition": timeit(lambda: ops.pair_transition(pair, pf[1].weight, pf[1].bias, pf[3].weight, pf[3].bias, residual=True, out=pair)),
            }
        for k, v in res.items():
            rows.setdefault(k, {})[mode] = v
        pair.copy_(torch.randn(1, N, N, 64, generator=g))       # in-place residual updates drift: fresh values per mode
    print(f"# N = {N}, us per call (HIP events, back to back)")
    print(f"{'operator':36s} {'fp32':>9s} {'split16':>9s}")
    for k, v in rows.items():
        print(f"{k:36s} {v['fp32']:9.1f} {v['split16']:9.1f}")
    _lib.lib().prd_set_gemm_mode(0)


if __name__ == "__main__":
    main()
