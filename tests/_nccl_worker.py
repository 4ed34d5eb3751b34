"""One RCCL rank of tests/test_multi_gpu.py (started by torch.distributed.run, one process per GPU): the PRODUCT sampler
(ProteinReDiffModel.sample on the HIP path) under distributed.sample_sharded with backend "nccl", compared on every rank with
single-rank runs of samples another rank drew.  Exit code 0 = identical."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from protein_redesign_amd.constants import make_args
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel
    from protein_redesign_amd.distributed import sample_sharded, shard_range
    from protein_redesign_amd.synthetic import NoiseSource, batch_to, clone_batch, deterministic_state_dict, synthetic_batch
    from protein_redesign_amd.weights import spec_tensors
    args = make_args(single_dim=64, pair_dim=32, num_blocks=1, esm_dim=16, num_steps=5, mask_prob=0.3)
    model = ProteinReDiffModel(args)
    model.load_state_dict(deterministic_state_dict(spec_tensors(args), seed=3))
    model = model.to(dev).eval()
    one = batch_to(synthetic_batch([(4, 20)], esm_dim=16, seed=4), dev)
    num_samples = 2 * world + 1                                   # uneven shards
    pos, logits = sample_sharded(lambda b, src: model.sample(b, sources=src), one, num_samples, seed=5, batch_size=2)
    assert pos.shape == (num_samples, 24, 3) and logits.shape == (num_samples, 24, 21)
    ok = True
    for k in list(shard_range(num_samples, world, (rank + 1) % world))[:2]:        # samples of the NEXT rank, recomputed alone
        p1, l1 = model.sample(clone_batch(one), sources=[NoiseSource(5, k)])
        ok &= bool(torch.equal(pos[k], p1[0]) and torch.equal(logits[k], l1[0]))
    ok &= not torch.allclose(pos[0], pos[1])
    flag = torch.tensor([int(ok)], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1 else 1)


if __name__ == "__main__":
    main()
