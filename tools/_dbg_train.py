import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import torch
import prd_oracle as O
from protein_redesign_amd import _lib, ops, training
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.diffusion_model import ProteinReDiffModel
from protein_redesign_amd.synthetic import NoiseSource, batch_to, deterministic_state_dict, synthetic_batch
from protein_redesign_amd.weights import spec_tensors
DEV = "cuda"
args = make_args(single_dim=64, pair_dim=32, num_blocks=2, esm_dim=16, num_steps=50, mask_prob=0.3, learning_rate=1e-3, warmup_steps=2)
params = deterministic_state_dict(spec_tensors(args), seed=5, style="near_init")
model = ProteinReDiffModel(args); model.load_state_dict(params); model = model.to(DEV).train()
model.run_setup_schedule(); model.setup_schedule = True
model.nonfinite_policy = sys.argv[1] if len(sys.argv) > 1 else "off"
cfg = model.configure_optimizers()
opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
batch = batch_to(synthetic_batch([(4, 18), (3, 14)], esm_dim=16, seed=6, n_total=24), DEV)
g = torch.Generator().manual_seed(3)
t = torch.tensor([11, 30], device=DEV)
nz = O.remove_mean(torch.randn(2, 24, 3, generator=g), (batch["atom_mask"] + batch["residue_mask"]).cpu()).to(DEV)
ns = O.remove_mean(torch.randn(2, 24, 21, generator=g), batch["residue_mask"].cpu()).to(DEV)
import warnings
for step in range(4):
    src = [NoiseSource(1, k) for k in range(2)]
    try:
        l = training.fit_step(model, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, step, opt, sched, t=t, noise_z=nz, noise_seq=ns, sources=src)
        print("step", step, "loss", float(l), "mode", _lib.arith())
    except Exception as e:
        print("step", step, "EXC", type(e).__name__, str(e)[:200])
        break
    badp = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    print("   non-finite params:", len(badp), badp[:5])
