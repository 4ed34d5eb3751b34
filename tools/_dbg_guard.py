import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import torch
from protein_redesign_amd import _lib, ops
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.diffusion_model import ProteinReDiffModel, ReverseDiffusion
from protein_redesign_amd.synthetic import NoiseSource, batch_to, clone_batch, deterministic_state_dict, synthetic_batch
from protein_redesign_amd.weights import spec_tensors
DEV="cuda"
PFX = "Denoiser.folding_blocks.0.pair_mul_outgoing.ab_proj"
for scale in (1e5, 1e7, 1e9):
    args = make_args(single_dim=64, pair_dim=64, num_blocks=2, esm_dim=16, num_steps=6, mask_prob=0.4)
    params = dict(deterministic_state_dict(spec_tensors(args), seed=31))
    print("ab_proj weight std", float(params[PFX + ".weight"].std()), "bias std", float(params[PFX + ".bias"].std()))
    params[PFX + ".weight"] = params[PFX + ".weight"] * scale
    params[PFX + ".bias"] = params[PFX + ".bias"] * scale
    model = ProteinReDiffModel(args); model.load_state_dict(params); model = model.to(DEV).eval()
    model.nonfinite_policy = "off"
    for mode in (1, 0):
        _lib.lib().prd_set_gemm_mode(mode)
        one = batch_to(synthetic_batch([(5, 27)], esm_dim=16, seed=3), DEV)
        loop = ReverseDiffusion(model, one, [NoiseSource(9, 0)])
        loop.step()
        torch.cuda.synchronize()
        print("scale", scale, "mode", mode, "after 1 step: z finite", bool(torch.isfinite(loop.z).all()), "sync", loop.sync.tolist(), "zmax", float(loop.z.abs().max()))
        blk = model.Denoiser.folding_blocks[0].pair_mul_outgoing
        g = torch.Generator().manual_seed(1)
        pair = torch.randn(1, 32, 32, 64, generator=g).to(DEV)
        mask = torch.ones(1, 32, device=DEV)
        out = ops.tri_mul(pair, mask, blk.weights(), incoming=False, residual=False)
        print("   direct tri_mul finite:", bool(torch.isfinite(out).all()), float(out.abs().max()))
    _lib.lib().prd_set_gemm_mode(1)
