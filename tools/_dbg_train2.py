import sys, os, types
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import torch
import prd_oracle as O
from protein_redesign_amd import _lib, ops, training
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.diffusion_model import ProteinReDiffModel
from protein_redesign_amd.synthetic import NoiseSource, batch_to, deterministic_state_dict, synthetic_batch
from protein_redesign_amd.weights import spec_tensors
DEV = "cuda"
MODE = sys.argv[1] if len(sys.argv) > 1 else "a"
import ctypes
POIS = ctypes.CDLL(os.path.join(os.getcwd(), "tools/ubench/liblds_poison.so"))
POIS.prd_dbg_poison_lds.argtypes = [ctypes.c_uint, ctypes.c_void_p]
def poison():
    assert POIS.prd_dbg_poison_lds(0x7fc07fc0 if "h" in MODE else 0x7fc00000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
def wrap(name, fn):
    def w(*a, **k):
        if "p" in MODE:
            poison()
        out = fn(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, o in enumerate(outs):
            if torch.is_tensor(o) and o.is_floating_point() and not torch.isfinite(o).all():
                bad = (~torch.isfinite(o)).nonzero()
                print(f"non-finite: ops.{name} output {i} shape {tuple(o.shape)} count {bad.shape[0]} first idx {bad[0].tolist()} last idx {bad[-1].tolist()}")
                for j, x in enumerate(a):
                    if torch.is_tensor(x) and x.is_floating_point():
                        print(f"    arg {j} shape {tuple(x.shape)} finite {bool(torch.isfinite(x).all())}")
                for kk, x in k.items():
                    if torch.is_tensor(x) and x.is_floating_point():
                        print(f"    kw {kk} shape {tuple(x.shape)} finite {bool(torch.isfinite(x).all())}")
        return out
    return w
if "w" in MODE:
    for name in dir(ops):
        fn = getattr(ops, name)
        if isinstance(fn, types.FunctionType) and fn.__module__ == ops.__name__ and name not in ("task_queue", "round_up", "_off", "cached_pack", "row_block", "dptr", "stream", "check", "lib", "gemm_workspace"):
            setattr(ops, name, wrap(name, fn))
args = make_args(single_dim=64, pair_dim=32, num_blocks=2, esm_dim=16, num_steps=50, mask_prob=0.3, learning_rate=1e-3, warmup_steps=2)
params = deterministic_state_dict(spec_tensors(args), seed=5, style="near_init")
model = ProteinReDiffModel(args); model.load_state_dict(params); model = model.to(DEV).train()
model.run_setup_schedule(); model.setup_schedule = True
model.nonfinite_policy = "off"
if "o" in MODE:
    cfg = model.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
batch = batch_to(synthetic_batch([(4, 18), (3, 14)], esm_dim=16, seed=6, n_total=24), DEV)
g = torch.Generator().manual_seed(3)
t = torch.tensor([11, 30], device=DEV)
nz = O.remove_mean(torch.randn(2, 24, 3, generator=g), (batch["atom_mask"] + batch["residue_mask"]).cpu()).to(DEV)
ns = O.remove_mean(torch.randn(2, 24, 21, generator=g), batch["residue_mask"].cpu()).to(DEV)
src = [NoiseSource(1, k) for k in range(2)]
loss = model.training_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0, t=t, noise_z=nz, noise_seq=ns, sources=src)
print(MODE, "train loss", float(loss.detach()))
