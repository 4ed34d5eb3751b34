#!/usr/bin/env python
"""Which operators of the training path launch ATen / Tensile / copy kernels, and what they cost: one optimisation step of
tools/train_bench.py's workload under the torch profiler, device time of every non-libprd kernel attributed to the autograd node
(forward / backward of training.py's Functions, as tools/train_op_profile.py labels them) whose CPU range contains the launching
ATen operator.  usage: train_aten_sites.py [--top 60]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
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
    fitter = training.Fitter(model, cfg["optimizer"], cfg["lr_scheduler"]["scheduler"])
    batch = batch_to(synthetic_batch([(64, 256)] * 2, seed=0), dev)

    from torch.profiler import record_function
    from torch.utils._python_dispatch import TorchDispatchMode
    import traceback
    line_counts = collections.Counter()
    counting = [False]

    class LineLog(TorchDispatchMode):
        """every ATen operator dispatched inside a trunk node, keyed by the innermost frame of this package"""
        def __init__(self, scope):
            super().__init__()
            self.scope = scope

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            if counting[0]:
                site = "?"
                for fr in reversed(traceback.extract_stack()):
                    if "protein_redesign_amd/" in fr.filename:
                        site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                        break
                line_counts[(self.scope, site, str(func.__name__ if hasattr(func, "__name__") else func))] += 1
            return func(*args, **(kwargs or {}))

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
            with record_function("prd:" + name + " fwd"), LineLog(name + " fwd"):
                return _of(ctx, *t)

        def bwd(ctx, *g, _ob=ob):
            with record_function("prd:" + ctx._prof_name + " bwd"), LineLog(ctx._prof_name + " bwd"):
                return _ob(ctx, *g)

        cls.forward, cls.backward = staticmethod(fwd), staticmethod(bwd)

    def step(i):
        return fitter.step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, i)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(3)
        torch.cuda.synchronize()
    counting[0] = True
    step(4)
    torch.cuda.synchronize()
    counting[0] = False
    evs = prof.events()
    scopes = sorted(((e.time_range.start, e.time_range.end, e.name[4:]) for e in evs if e.name.startswith("prd:")), key=lambda t: t[0])

    def scope_of(t):
        best = "(outside the trunk nodes)"
        for s0, s1, nm in scopes:                     # innermost = the latest start that still contains t
            if s0 > t:
                break
            if t <= s1:
                best = nm
        return best

    sites = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
    by_scope = collections.defaultdict(lambda: [0.0, 0])
    own_us = foreign_us = 0.0
    own_n = foreign_n = 0
    for e in evs:
        if not e.kernels or e.name.startswith("prd:"):
            continue
        if e.cpu_children and any(c.kernels for c in e.cpu_children):
            continue                                  # attribute at the innermost operator that owns the kernels
        for k in e.kernels:
            nm = k.name
            foreign = nm.startswith(("void at::", "at::", "Cijk_", "__amd_rocclr", "void rocprim")) or "at::native" in nm
            if not foreign:
                own_us += k.duration
                own_n += 1
                continue
            foreign_us += k.duration
            foreign_n += 1
            site = scope_of(e.time_range.start)
            rec = sites[(site, e.name)]
            rec[0] += k.duration
            rec[1] += 1
            by_scope[site][0] += k.duration
            by_scope[site][1] += 1
    print(f"one optimisation step: own kernels {own_n} launches {own_us:.0f} us; ATen / Tensile / copy {foreign_n} launches {foreign_us:.0f} us")
    for site, (us, n) in sorted(by_scope.items(), key=lambda kv: -kv[1][0]):
        print(f"{us:8.1f} us {n:4d} x  {site}")
    print()
    rows = sorted(sites.items(), key=lambda kv: -kv[1][0])
    for (site, op), (us, n, names) in rows[:a.top]:
        print(f"{us:8.1f} us {n:4d} x  {op:28s} {site}")
    print()
    print("ATen operators dispatched inside the trunk nodes (one step), by source line:")
    skip = ("empty", "view", "detach", "t.default", "transpose", "permute", "reshape", "_unsafe_view", "alias", "as_strided", "slice", "select",
            "unsqueeze", "squeeze", "expand", "split", "unbind", "is_", "size", "stride", "empty_like", "_local_scalar")
    agg = collections.Counter()
    for (scope, site, fn), n in line_counts.items():
        if any(fn.startswith(k) for k in skip):
            continue
        agg[(scope, site, fn)] += n
    for (scope, site, fn), n in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        print(f"{n:4d} x  {scope:28s} {site:24s} {fn}")


if __name__ == "__main__":
    main()



