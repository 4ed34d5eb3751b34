#!/usr/bin/env python
"""Free-running reverse-diffusion loops against the reference-derived yardstick (GPU box; no /root/reference needed).

For every long-trajectory fixture (tests/golden/<case>.npz = the imported reference in fp32, <case>_f64.npz = the same run
with model.double(), both written by oracle/gen_golden.py / oracle/gen_yardstick.py) and both arithmetic modes of the
library, run the HIP path FREE (no restarts) and print, at every stored step,

    delta_ref   = rel-L2(reference fp32, reference fp64)     -- the reference's own sensitivity to fp32 round-off
    hip_vs_f64  = rel-L2(HIP, reference fp64)
    hip_vs_f32  = rel-L2(HIP, reference fp32)

of the coordinates z entering that step, and of the final positions / logits.  Usage:
    python tools/trajectory_conditioning.py [case ...] > profiles/r03_trajectory.txt
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from protein_redesign_amd import _lib  # noqa: E402
from protein_redesign_amd.constants import make_args  # noqa: E402
from protein_redesign_amd.diffusion_model import ProteinReDiffModel, ReverseDiffusion  # noqa: E402
from protein_redesign_amd.synthetic import NoiseSource, batch_to, deterministic_state_dict, synthetic_batch  # noqa: E402
from protein_redesign_amd.weights import spec_tensors  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
NOISE_SEED = 7


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def free_run(case, mode):
    """states entering the stored steps, final (pos, logits)"""
    lib = _lib.lib()
    prev = lib.prd_get_gemm_mode()
    assert lib.prd_set_gemm_mode(1 if mode == "split16" else 0) == 0
    try:
        args = make_args(**case["args"])
        params = deterministic_state_dict(spec_tensors(args), seed=case["weight_seed"], style=case.get("weight_style", "random"), scales=case.get("weight_scales"))
        model = ProteinReDiffModel(args)
        model.load_state_dict(params)
        model = model.to("cuda").eval()
        one = batch_to(synthetic_batch([tuple(case["traj_sample"])], esm_dim=args["esm_dim"], seed=case["batch_seed"] + 500), "cuda")
        loop = ReverseDiffusion(model, one, [NoiseSource(NOISE_SEED, 0)])
        every = case["traj_every"]
        zs = []
        with torch.inference_mode():
            while loop.steps_done < args["num_steps"]:
                if loop.steps_done % every == 0:
                    zs.append(loop.z.cpu().numpy().astype(np.float64))
                loop.step()
            pos, logits = loop.result()
        return zs, pos.cpu().numpy(), logits.cpu().numpy()
    finally:
        assert lib.prd_set_gemm_mode(prev) == 0


def report(name, out=sys.stdout):
    f32 = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = json.loads(str(f32["case"]))
    p64 = os.path.join(GOLDEN, name + "_f64.npz")
    if not os.path.exists(p64):
        return report_f32_only(name, f32, case, out)
    f64 = np.load(p64)
    steps = [int(v) for v in f64["seg_step"]]
    n = len(steps)
    ref32 = [f32["seg_z"][k].astype(np.float64) for k in range(n)]
    ref64 = [f64["seg_z_f64"][k] for k in range(n)]
    dref = [rel(ref32[k], ref64[k]) for k in range(n)]
    res = {"delta_ref": dref, "steps": steps}
    print(f"== {name}: N = {sum(case['traj_sample'])}, T = {case['args']['num_steps']}, weights '{case.get('weight_style', 'random')}', "
          f"state compared every {case['traj_every']} steps ==", file=out)
    runs = {}
    for mode in ("fp32", "split16"):
        zs, pos, logits = free_run(case, mode)
        runs[mode] = (zs, pos, logits)
        res[mode + "_vs_f64"] = [rel(zs[k][0], ref64[k]) for k in range(n)]
        res[mode + "_vs_f32"] = [rel(zs[k][0], ref32[k]) for k in range(n)]
    print(f"{'step':>6s} {'delta_ref':>10s} | {'fp32 vs f64':>11s} {'ratio':>6s} {'fp32 vs f32':>11s} | {'split16 vs f64':>14s} {'ratio':>6s} {'split16 vs f32':>14s}", file=out)
    for k in range(n):
        d = max(dref[k], 1e-30)
        print(f"{steps[k]:6d} {dref[k]:10.2e} | {res['fp32_vs_f64'][k]:11.2e} {res['fp32_vs_f64'][k] / d:6.1f} {res['fp32_vs_f32'][k]:11.2e} | "
              f"{res['split16_vs_f64'][k]:14.2e} {res['split16_vs_f64'][k] / d:6.1f} {res['split16_vs_f32'][k]:14.2e}", file=out)
    if "traj_pos_f64" in f64:
        dp, dl = rel(f32["traj_pos"], f64["traj_pos_f64"]), rel(f32["traj_logits"], f64["traj_logits_f64"])
        print(f" final positions: delta_ref {dp:.2e}; " + "; ".join(
            f"{m} vs f64 {rel(runs[m][1], f64['traj_pos_f64']):.2e} (x{rel(runs[m][1], f64['traj_pos_f64']) / dp:.1f}), vs f32 {rel(runs[m][1], f32['traj_pos']):.2e}"
            for m in runs), file=out)
        print(f" final logits:    delta_ref {dl:.2e}; " + "; ".join(
            f"{m} vs f64 {rel(runs[m][2], f64['traj_logits_f64']):.2e} (x{rel(runs[m][2], f64['traj_logits_f64']) / dl:.1f}), vs f32 {rel(runs[m][2], f32['traj_logits']):.2e}"
            for m in runs), file=out)
        res["final"] = {"delta_ref_pos": dp, "delta_ref_logits": dl,
                        **{m + "_pos_vs_f64": rel(runs[m][1], f64["traj_pos_f64"]) for m in runs},
                        **{m + "_logits_vs_f64": rel(runs[m][2], f64["traj_logits_f64"]) for m in runs}}
    print(file=out)
    return res


def report_f32_only(name, f32, case, out=sys.stdout):
    """no fp64 twin of this fixture (yet): the free run against the fp32 reference alone"""
    n = len(f32["seg_step"]) if "seg_step" in f32 else f32["seg_z"].shape[0]
    steps = [int(v) for v in f32["seg_step"]] if "seg_step" in f32 else [k * case["traj_every"] for k in range(n)]
    ref32 = [f32["seg_z"][k].astype(np.float64) for k in range(n)]
    print(f"== {name}: N = {sum(case['traj_sample'])}, T = {case['args']['num_steps']}, weights '{case.get('weight_style', 'random')}', "
          f"state compared every {case['traj_every']} steps; NO fp64 twin: HIP vs the fp32 reference only ==", file=out)
    runs = {m: free_run(case, m) for m in ("fp32", "split16")}
    print(f"{'step':>6s} | {'fp32 vs f32':>11s} | {'split16 vs f32':>14s} | {'split16 vs fp32 (HIP)':>22s}", file=out)
    for k in range(n):
        print(f"{steps[k]:6d} | {rel(runs['fp32'][0][k][0], ref32[k]):11.2e} | {rel(runs['split16'][0][k][0], ref32[k]):14.2e} | "
              f"{rel(runs['split16'][0][k][0], runs['fp32'][0][k][0]):22.2e}", file=out)
    if "traj_pos" in f32:
        print(" final positions: " + "; ".join(f"{m} vs f32 {rel(runs[m][1], f32['traj_pos']):.2e}" for m in runs), file=out)
        print(" final logits:    " + "; ".join(f"{m} vs f32 {rel(runs[m][2], f32['traj_logits']):.2e}" for m in runs), file=out)
    print(file=out)


if __name__ == "__main__":
    names = sys.argv[1:] or [n for n in ("cfg1_t200", "cfg1_t200_random", "cfg1_t1000", "cfg2_t1000")
                             if os.path.exists(os.path.join(GOLDEN, n + ".npz"))]
    print("Free-running reverse-diffusion loops: HIP path vs the imported reference in fp32 and in fp64 (tools/trajectory_conditioning.py)")
    print("delta_ref = rel-L2(reference fp32, reference fp64); ratio = (HIP vs reference fp64) / delta_ref\n")
    for nm in names:
        report(nm)
