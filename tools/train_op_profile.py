#!/usr/bin/env python
"""Where the optimisation step spends its time, by trunk operator (forward / backward of every autograd node of training.py),
HIP events around each node.  usage: train_op_profile.py [--steps 3]   (2 complexes of N = 320, the per-GPU share of configs[3])"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
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
    batch = batch_to(synthetic_batch([(64, 256)] * 2, seed=0), dev)
    events = collections.defaultdict(list)

    def timed(kind, name, fn):
        def wrapper(*args_, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*args_, **kw)
            e1.record()
            events[(name, kind)].append((e0, e1))
            return out
        return wrapper

    def label(ref):
        q = getattr(ref, "__qualname__", str(ref))
        parts = q.split(".<locals>.")
        return (parts[0] + ("." + parts[-1] if len(parts) > 1 else "")).replace("_update", "")

    for cls in (training.HipOp, training.TriMulFn, training.TriAttnFn, training.PairTransitionFn, training.PairBiasFn, training.OuterLinearFn,
                training.InputStageFn, training.HeadsFn):
        of, ob = cls.forward, cls.backward

        def fwd(ctx, *t, _of=of, _cls=cls):
            name = label(t[1]) if _cls is training.HipOp else _cls.__name__
            ctx._prof_name = name
            return timed("fwd", name, _of)(ctx, *t)

        def bwd(ctx, *g, _ob=ob):
            return timed("bwd", ctx._prof_name, _ob)(ctx, *g)

        cls.forward, cls.backward = staticmethod(fwd), staticmethod(bwd)

    def step(i):
        return training.fit_step(model, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, i, opt, sched)

    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    events.clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.steps):
        step(2 + i)
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / a.steps
    print(f"optimisation step: {total:.2f} ms (2 complexes of N = 320; event overhead included)")
    rows = []
    for (name, kind), evs in events.items():
        ms = sum(x.elapsed_time(y) for x, y in evs) / a.steps
        rows.append((ms, name, kind, len(evs) // a.steps))
    acc = 0.0
    print(f"{'operator':28s} {'pass':4s} {'calls':>5s} {'ms/step':>8s} {'share':>6s}")
    for ms, name, kind, n in sorted(rows, reverse=True):
        acc += ms
        print(f"{name:28s} {kind:4s} {n:5d} {ms:8.3f} {100 * ms / total:6.1f}")
    print(f"{'(outside the trunk nodes: input stage, loss, optimiser, EMA)':60s} {total - acc:8.3f} {100 * (total - acc) / total:6.1f}")


if __name__ == "__main__":
    main()
