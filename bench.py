#!/usr/bin/env python
"""Headline benchmark: denoising steps/s (network forward + reverse update) per complex on the
BASELINE.json configs[1] workload -- synthetic 256-residue + 64-atom-ligand complex (N = 320),
single_dim 512, pair_dim 64, 4 folding blocks, T = 1000 -- on N GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Samples of ``sample()`` never interact, so the path shards by sample index
with no collective in the data path (weak scaling: every rank runs ``--samples-per-gpu`` complexes);
RCCL is used only for the barrier and the max-over-ranks of the elapsed time.  A "step" is one
reverse-diffusion step of every complex on the rank, replayed from one captured hipGraph with all
inputs (weights, static embeddings, noise table) resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4 at the fp32 vector rate
HBM_PEAK_GBS = 8000.0


def step_flops(N, S, P, H=4, c=16, nb=4):
    """Algorithmic flops of one denoising step of one complex (contractions only)."""
    Hc = H * c
    pre = 2 * N * N * 256 * P + 2 * N * 21 * S
    opm = 4 * N * S * (S // 4) + 2 * N * N * (S // 4) * P
    spa = 2 * N * N * P * H + 8 * N * S * H * S + 4 * H * N * N * S + 2 * N * H * S * S
    blk = (2 * N * N * P * H + 10 * N * S * Hc + 4 * H * N * N * c + 16 * N * S * S + 2 * N * N * S * P + 4 * N * S * P
           + 16 * N * N * P * P + 4 * P * N ** 3 + 8 * N * N * P * P + 20 * N * N * P * Hc + 8 * Hc * N ** 3 + 16 * N * N * P * P)
    heads = 2 * N * N * P * P + 2 * N * N * P + 2 * N * S * S + 42 * N * S
    return pre + opm + spa + nb * blk + heads


def tri_attn_core_flops(b, N, P, H=4, c=16):
    """q,k,v,g projections (8 N^2 P Hc) + QK^T and PV (4 Hc N^3) of ONE launch of tri_attn_core_kernel."""
    return b * (8 * N * N * P * H * c + 4 * H * c * N ** 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--samples-per-gpu", type=int, default=1)
    ap.add_argument("--residues", type=int, default=256)
    ap.add_argument("--atoms", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm

    from protein_redesign_amd import ops
    from protein_redesign_amd.constants import make_args
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel, ReverseDiffusion
    from protein_redesign_amd.synthetic import NoiseSource, batch_to, deterministic_state_dict, synthetic_batch
    from protein_redesign_amd.weights import spec_tensors

    S, P, NB, T = 512, 64, 4, 1000
    margs = make_args(single_dim=S, pair_dim=P, num_blocks=NB, num_steps=T, mask_prob=0.3)
    params = deterministic_state_dict(spec_tensors(margs), seed=1)
    model = ProteinReDiffModel(margs)
    model.load_state_dict(params)
    model = model.to(dev).eval()
    model.use_hip_graph = not a.no_graph
    bpg = a.samples_per_gpu
    N = a.atoms + a.residues
    batch = synthetic_batch([(a.atoms, a.residues)] * bpg, seed=0)
    sources = [NoiseSource(0, rank * bpg + k) for k in range(bpg)]      # keyed by GLOBAL sample index
    loop = ReverseDiffusion(model, batch_to(batch, dev), sources)

    def advance(n):
        for _ in range(n):
            if loop.steps_done >= loop.T:
                loop.reset()
            loop.step()

    advance(2)                         # eager step + graph capture (setup, not warm-up)
    advance(a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    advance(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    pos, logits = loop.result()
    finite = bool(torch.isfinite(pos).all() and torch.isfinite(logits).all())

    # ---- dominant kernel, measured live with HIP events on the launch stream (rank 0) ----
    roofline = None
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        pair = torch.randn(bpg, N, N, P, generator=g).to(dev)
        ta = model.Denoiser.folding_blocks[0].pair_attn_starting.attn
        wts = ta.weights()[:5]
        og = torch.empty(bpg, N, N, 64, device=dev)
        for i in range(4):
            ops.tri_attn_core(pair, loop.mask, wts, 4, 16, ending=bool(i & 1), og=og)
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):                      # starting / ending modes alternate, as inside a step
            ops.tri_attn_core(pair, loop.mask, wts, 4, 16, ending=bool(i & 1), og=og)
        e1.record()
        torch.cuda.synchronize()
        kus = e0.elapsed_time(e1) * 1e3 / reps
        kfl = tri_attn_core_flops(bpg, N, P)
        ach = kfl / (kus * 1e-6) / 1e12
        traffic = None          # HBM-side bytes per launch from the separate rocprofv3 --pmc passes (profiles/README.md)
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and bpg == 1 and N == 320:
            traffic = json.load(open(tpath))["tri_attn_core_kernel"]["traffic_bytes_per_launch"]
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                    "kernel": "tri_attn_core_kernel", "launches_per_step": 2 * NB,
                    "flops_per_launch": kfl, "avg_launch_us": round(kus, 2)}

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on the host cores, bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import prd_oracle as O
        pb = O.prepare_batch(synthetic_batch([(a.atoms, a.residues)], seed=0), 0.3, [NoiseSource(0, 0).randperm(a.residues)])
        g = torch.Generator().manual_seed(0)
        z, seq_t, t = torch.randn(1, N, 3, generator=g), torch.randn(1, N, 21, generator=g), torch.tensor([T // 2])
        times = []
        with torch.inference_mode():
            for i in range(3):
                c0 = time.perf_counter()
                eps, lg = O.network_step(params, margs, pb, z, seq_t, pb["residue_and_atom_mask"], t)
                z = (z - 0.01 * eps)                       # reverse update is negligible next to the network
                seq_t = torch.softmax(lg, -1) * 2 - 1
                times.append(time.perf_counter() - c0)
        best = min(times[1:])
        cpu = {"value": round(1.0 / best, 4), "unit": "denoising-steps/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": f"oracle/prd_oracle.py network_step, same N={N} complex, 1 warm-up + min of 2 steps"}

    if rank == 0:
        total_steps = world * bpg * a.steps
        value = total_steps / dt
        flops = step_flops(N, S, P, nb=NB)
        out = {
            "metric": "denoising-steps/sec (fwd+rev) per complex, N=256 res", "value": round(value, 3),
            "unit": "denoising-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic {a.residues}-residue + {a.atoms}-atom ligand (N={N}), "
                                   f"single_dim={S} pair_dim={P} num_blocks={NB} num_steps={T}, "
                                   f"{bpg} complex/GPU, sharded by sample index (no data-path collective)",
                       "samples_per_gpu": bpg, "hip_graph": not a.no_graph, "outputs_finite": finite,
                       # default: fp32 MFMA everywhere.  PRD_BF16X3=1 opts in to the experimental split-bf16 row GEMMs
                       # (fp32-accurate); such a run says so here and is not the headline number
                       "row_gemm": "bf16x3-split (opt-in, experimental)" if os.environ.get("PRD_BF16X3") else "fp32-mfma"},
            "step_gflop": round(flops / 1e9, 1),
            "step_tflops": round(flops * bpg / (dt / a.steps) / 1e12, 2),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
