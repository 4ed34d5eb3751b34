#!/usr/bin/env python
"""Which aten operators (outside the hand-written kernels) the optimisation step spends GPU time in, by operator and input shape
(torch.profiler; 2 complexes of N = 320).  usage: train_aten_profile.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    from protein_redesign_amd.constants import make_args
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel
    from protein_redesign_amd.synthetic import batch_to, deterministic_state_dict, synthetic_batch
    from protein_redesign_amd.weights import spec_tensors
    args = make_args(single_dim=512, pair_dim=64, num_blocks=4, num_steps=1000, mask_prob=0.3)
    model = ProteinReDiffModel(args)
    model.load_state_dict(deterministic_state_dict(spec_tensors(args), seed=1, style="near_init"))
    model = model.to(dev).train()
    cfg = model.configure_optimizers()
    opt = cfg["optimizer"]
    batch = batch_to(synthetic_batch([(64, 256)] * 2, seed=0), dev)

    def step(i):
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(batch, i)
        loss.backward()
        opt.step()

    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step(2)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=4):
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = e.self_cuda_time_total
        if t > 0:
            rows.append((t, e.count, e.key, str(e.input_shapes)[:70], [s for s in e.stack if "protein_redesign_amd" in s or "tools/" in s][:2]))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    print(f"total self device time {tot / 1e3:.2f} ms")
    rows = [r for r in rows if r[2].startswith("aten::") or "Backward" in r[2] or "Fn" in r[2]]
    if "--fills" in sys.argv:           # where the small fills / copies of a step come from: count by operator and shape
        cnt = {}
        for e in prof.key_averages(group_by_input_shape=True):
            if e.key in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::clone", "aten::ones_like", "aten::full",
                         "aten::masked_fill", "aten::one_hot", "aten::contiguous", "aten::cat", "aten::add_", "aten::add", "aten::mul", "aten::sum"):
                cnt[(e.key, str(e.input_shapes)[:80])] = cnt.get((e.key, str(e.input_shapes)[:80]), 0) + e.count
        for (k, sh), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:50]:
            print(f"{n:5d}  {k:22s} {sh}")
        return
    for t, n, k, sh, st in rows[:60]:
        print(f"{t / 1e3:8.3f} ms {n:4d}  {k[:40]:40s} {sh:70s} {' <- '.join(x.split('/')[-1][:60] for x in st)}")


if __name__ == "__main__":
    main()
