#!/usr/bin/env python
"""Optimisation micro-steps/s of the training path (BASELINE configs[3] per-GPU share: batch_size 2, single_dim 512, pair_dim 64,
4 blocks) on synthetic PDBbind-like complexes: training_step (HIP forward, per-operator backward, per-block recompute) +
backward + Adam + LinearLR + EMA.  usage: train_bench.py [--residues 256 --atoms 64 --batch 2 --steps 5]
Under torch.distributed (torchrun) every rank runs its own batch and the gradients are averaged with one flat RCCL all-reduce."""
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

    def step(i):
        return training.fit_step(model, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, i, opt, sched)

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    losses = [step(a.warmup + i) for i in range(a.steps)]      # loss tensors: read back after the timed region, as a training loop
    torch.cuda.synchronize()                                    # that logs every n-th step does (no host round trip per step)
    dt = (time.perf_counter() - t0) / a.steps
    losses = [float(x) for x in losses]
    if rank == 0:
        print(json.dumps({"metric": "optimisation micro-steps/s per complex (q-noising + fwd + bwd + Adam + EMA)", "value": round(world * a.batch / dt, 3),
                          "ms_per_step": round(dt * 1e3, 2), "n_gpus": world, "batch_per_gpu": a.batch, "N": a.atoms + a.residues,
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "losses": [round(x, 4) for x in losses],
                          "backend": "nccl (RCCL), one flat gradient all-reduce per step" if (world > 1 or launched) else "single process"}))
    if world > 1 or launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
