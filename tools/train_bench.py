#!/usr/bin/env python
"""Optimisation micro-steps/s of the training path (BASELINE configs[3] per-GPU share: batch_size 2, single_dim 512, pair_dim 64,
4 blocks) on synthetic PDBbind-like complexes: training_step (HIP forward, per-operator backward, per-block recompute) +
backward + Adam + LinearLR + EMA.  usage: train_bench.py [--residues 256 --atoms 64 --batch 2 --steps 5 --accumulate 1 --reduce-slices 1]
Under torch.distributed (torchrun) every rank runs its own batch and the gradients are averaged with one flat RCCL all-reduce per
OPTIMISER step (every --accumulate micro-batches; --reduce-slices > 1: that many asynchronous pieces, see training.all_reduce_gradients).
The line carries ms per MICRO-batch, GPU kernel launches per micro-batch (torch profiler, one extra step after the timed region)
and how many of them are ATen / Tensile / copy kernels rather than kernels of libprd_hip.so."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--residues", type=int, default=256)
    ap.add_argument("--atoms", type=int, default=64)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--accumulate", type=int, default=1, help="accumulate_grad_batches (train.py:57): optimiser step every k micro-batches")
    ap.add_argument("--reduce-slices", type=int, default=1, help="gradient all-reduce as this many asynchronous pieces")
    ap.add_argument("--no-launch-count", action="store_true")
    a = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # under torch.distributed.run, also with one rank
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)                  # "nccl" is RCCL on ROCm
    from protein_redesign_amd import training
    from protein_redesign_amd.constants import make_args
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel
    from protein_redesign_amd.synthetic import batch_to, deterministic_state_dict, synthetic_batch
    from protein_redesign_amd.weights import spec_tensors
    args = make_args(single_dim=512, pair_dim=64, num_blocks=4, num_steps=1000, mask_prob=0.3)
    model = ProteinReDiffModel(args)
    model.load_state_dict(deterministic_state_dict(spec_tensors(args), seed=1, style="near_init"))
    model = model.to(dev).train()
    cfg = model.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    batch = batch_to(synthetic_batch([(a.atoms, a.residues)] * a.batch, seed=rank), dev)

    fitter = training.Fitter(model, opt, sched, accumulate_grad_batches=a.accumulate, reduce_slices=a.reduce_slices)

    def step(i):
        return fitter.step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, i)

    for i in range(a.warmup * a.accumulate):
        step(i)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    nmicro = a.steps * a.accumulate
    losses = [step(a.warmup * a.accumulate + i) for i in range(nmicro)]   # loss tensors: read back after the timed region, as a training
    torch.cuda.synchronize()                                    # loop that logs every n-th step does (no host round trip per step)
    dt = (time.perf_counter() - t0) / nmicro
    losses = [float(x) for x in losses]
    launches = None
    if not a.no_launch_count:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(a.accumulate):
                step((a.warmup + a.steps) * a.accumulate + i)
            torch.cuda.synchronize()
        names = [e.name for e in prof.events() if e.device_type is not None and "cuda" in str(e.device_type).lower()]
        foreign = [n for n in names if n.startswith(("void at::", "at::", "Cijk_", "__amd_rocclr", "void (anonymous namespace)::elementwise", "void rocprim"))
                   or "at::native" in n]
        launches = {"per_micro_batch": round(len(names) / a.accumulate, 1), "aten_tensile_copy": round(len(foreign) / a.accumulate, 1)}
    if rank == 0:
        print(json.dumps({"metric": "optimisation micro-steps/s per complex (q-noising + fwd + bwd + Adam + EMA)", "value": round(world * a.batch / dt, 3),
                          "ms_per_step": round(dt * 1e3, 2), "n_gpus": world, "batch_per_gpu": a.batch, "N": a.atoms + a.residues,
                          "accumulate_grad_batches": a.accumulate, "reduce_slices": a.reduce_slices, "launches": launches,
                          "optimizer_steps": fitter.optimizer_steps, "skipped_steps": fitter.skipped_steps,
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "losses": [round(x, 4) for x in losses],
                          "backend": "nccl (RCCL), one flat gradient all-reduce per optimiser step" if (world > 1 or launched) else "single process"}))
    if world > 1 or launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
