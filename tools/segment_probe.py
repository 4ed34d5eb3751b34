#!/usr/bin/env python
"""Per-segment deviations of the N = 320 / T = 1000 loop (tests/golden/cfg2_t1000.npz): every 25-step segment restarted from the
reference's own state, in both arithmetics, against the fp32 reference's next stored state, and the two arithmetics against each
other.  GPU box:  python tools/segment_probe.py   (the bound of tests/test_hip_parity.py::test_trajectory_segments_vs_reference_golden
comes from oracle/gen_yardstick.py --segments: the fp64 reference over the same segments)."""
import sys, os, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
from protein_redesign_amd import _lib
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.diffusion_model import ProteinReDiffModel, ReverseDiffusion
from protein_redesign_amd.synthetic import NoiseSource, batch_to, deterministic_state_dict, synthetic_batch
from protein_redesign_amd.weights import spec_tensors
z = np.load('tests/golden/cfg2_t1000.npz'); case = json.loads(str(z['case']))
def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64); return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
args = make_args(**case['args'])
params = deterministic_state_dict(spec_tensors(args), seed=case['weight_seed'], style=case.get('weight_style', 'random'), scales=case.get('weight_scales'))
model = ProteinReDiffModel(args); model.load_state_dict(params); model = model.to('cuda').eval()
one = batch_to(synthetic_batch([tuple(case['traj_sample'])], esm_dim=args['esm_dim'], seed=case['batch_seed'] + 500), 'cuda')
steps = [int(v) for v in z['seg_step']]
seg_z, seg_s = torch.from_numpy(z['seg_z']), torch.from_numpy(z['seg_seq_t'])
ends = {}
for mode in (0, 1):
    _lib.lib().prd_set_gemm_mode(mode)
    loop = ReverseDiffusion(model, one, [NoiseSource(7, 0)])
    with torch.inference_mode():
        for k, start in enumerate(steps[:-1]):
            loop.restart(start, seg_z[k:k+1], seg_s[k:k+1])
            while loop.steps_done < steps[k+1]: loop.step()
            ends[(mode, k)] = (loop.z.cpu().numpy().copy(), loop.seq_t.cpu().numpy().copy())
for k, start in enumerate(steps[:-1]):
    a, b = ends[(0, k)], ends[(1, k)]
    print(f"{start:4d}  fp32: z {rel(a[0], seg_z[k+1:k+2].numpy()):.2e} seq {rel(a[1], seg_s[k+1:k+2].numpy()):.2e} | split16: z {rel(b[0], seg_z[k+1:k+2].numpy()):.2e} seq {rel(b[1], seg_s[k+1:k+2].numpy()):.2e} | modes: z {rel(a[0], b[0]):.2e} seq {rel(a[1], b[1]):.2e}  |seq| {np.linalg.norm(seg_s[k+1].numpy()):.3e}")
