#!/usr/bin/env python
"""Experiment (not the product path): the optimisation step of tools/train_bench.py captured ONCE as a HIP graph and replayed --
how far the step is from its launch floor.  Fixed complex, fixed noise / t / redesign mask (a real loop would capture one graph per
bucket of the bucket sampler and copy each batch into the static inputs).  usage: train_graph_bench.py [--steps 20]"""
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
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from protein_redesign_amd import ops
    from protein_redesign_amd.constants import make_args
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel
    from protein_redesign_amd.synthetic import batch_to, deterministic_state_dict, synthetic_batch
    from protein_redesign_amd.weights import spec_tensors
    args = make_args(single_dim=512, pair_dim=64, num_blocks=4, num_steps=1000, mask_prob=0.3)
    model = ProteinReDiffModel(args)
    model.load_state_dict(deterministic_state_dict(spec_tensors(args), seed=1, style="near_init"))
    model = model.to(dev).train()
    model.run_setup_schedule()
    model.setup_schedule = True
    opt = torch.optim.Adam(model.parameters(), lr=model.learning_rate, fused=True, capturable=True)
    batch = batch_to(synthetic_batch([(a.atoms, a.residues)] * a.batch, seed=0), dev)
    pb = model.prepare_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, 0)
    x, mask = pb["x"], pb["residue_and_atom_mask"]
    num_nodes = (mask > 0.5).sum(-1)
    g = torch.Generator(device=dev).manual_seed(0)
    t = torch.randint(0, model.num_steps, (x.size(0),), device=dev, generator=g)
    noise_z = ops.remove_mean(torch.randn(x.shape, device=dev, generator=g), mask.contiguous())
    seq = pb["residue_one_hot"]
    noise_seq = ops.remove_mean(torch.randn(seq.shape, device=dev, generator=g), pb["residue_mask"].contiguous())

    def body():
        loss = torch.mean(model.diffusion_loss(pb, x, mask, t, noise_z, noise_seq) / num_nodes)
        loss.backward()
        opt.step()
        return loss.detach()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def eager():
        opt.zero_grad(set_to_none=True)
        return body()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            l0 = eager()
    torch.cuda.current_stream().wait_stream(s)
    ms_eager = timed(eager, a.steps)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        lg = body()
    ms_graph = timed(graph.replay, a.steps)
    print(json.dumps({"eager_ms_per_step": round(ms_eager, 2), "graph_replay_ms_per_step": round(ms_graph, 2), "loss_eager": float(l0),
                      "loss_graph": float(lg), "batch": a.batch, "N": a.atoms + a.residues,
                      "note": "fixed inputs; fwd + bwd + fused Adam in one captured graph (no LR schedule / EMA / data loading)"}))


if __name__ == "__main__":
    main()
