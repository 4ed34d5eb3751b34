"""world_size-2 gloo test of the sample-sharding path (CPU): the shard / gather logic is exercised with
the CPU oracle standing in for the per-GPU sampler, and must reproduce the single-process result
bit for bit, in global sample order."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from protein_redesign_amd.constants import make_args
from protein_redesign_amd.distributed import sample_sharded, shard_range
from protein_redesign_amd.synthetic import deterministic_state_dict, synthetic_batch
from protein_redesign_amd.weights import spec_tensors

ARGS = dict(single_dim=32, pair_dim=8, head_dim=4, num_heads=2, num_blocks=1, esm_dim=16, dist_dim=16, time_dim=16,
            num_steps=3, mask_prob=0.3)


def oracle_sampler():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import prd_oracle as O
    args = make_args(**ARGS)
    params = deterministic_state_dict(spec_tensors(args), seed=3)
    return lambda batch, sources: O.sample(params, args, batch, sources)


def run_single(num_samples, batch_size):
    torch.set_num_threads(1)
    batch = synthetic_batch([(3, 8)], esm_dim=16, seed=4)
    return sample_sharded(oracle_sampler(), batch, num_samples, seed=5, batch_size=batch_size)


def _worker(rank, world, port, num_samples, batch_size, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = synthetic_batch([(3, 8)], esm_dim=16, seed=4)
    pos, logits = sample_sharded(oracle_sampler(), batch, num_samples, seed=5, batch_size=batch_size)
    torch.save((pos, logits), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    for n in (1, 5, 8, 64):
        for w in (1, 2, 3, 8):
            idx = [i for r in range(w) for i in shard_range(n, w, r)]
            assert idx == list(range(n))
            sizes = [len(shard_range(n, w, r)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("num_samples,batch_size", [(5, 2), (4, 1)])
def test_two_ranks_match_one_rank(tmp_path, num_samples, batch_size):
    want_pos, want_logits = run_single(num_samples, batch_size=1)
    port = 29500 + (os.getpid() % 1000) + num_samples
    mp.spawn(_worker, args=(2, port, num_samples, batch_size, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        pos, logits = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert pos.shape == (num_samples, 11, 3) and logits.shape == (num_samples, 11, 21)
        # per-sample results do not depend on sharding or on the per-rank batch size
        assert torch.allclose(pos, want_pos, rtol=0, atol=1e-5)
        assert torch.allclose(logits, want_logits, rtol=0, atol=1e-5)
    # samples are distinct (keyed noise), not the duplicates of the reference's identically seeded ranks
    assert not torch.allclose(want_pos[0], want_pos[1])


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[2]: 64 samples over 8 ranks, as one batch of 8 or two batches of 4 per rank
# ---------------------------------------------------------------------------------------------------

def _stub_sampler(calls):
    """A sampler that costs nothing and is keyed on the NoiseSource alone: sample k is the first draws of NoiseSource(seed, k),
    whatever rank, shard or batch it is drawn in.  Records the batch sizes it was called with."""
    def sampler(batch, sources):
        b, N = batch["atom_mask"].shape
        assert b == len(sources)
        calls.append(b)
        return torch.stack([s.randn(N, 3) for s in sources]), torch.stack([s.randn(N, 21) for s in sources])
    return sampler


def _worker_stub(rank, world, port, num_samples, batch_size, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = synthetic_batch([(3, 8)], esm_dim=16, seed=4)
    calls = []
    pos, logits = sample_sharded(_stub_sampler(calls), batch, num_samples, seed=5, batch_size=batch_size)
    torch.save((pos, logits, calls), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,num_samples,batch_size", [(8, 64, 8), (8, 64, 4), (2, 64, 8), (8, 61, 8)])
def test_configs2_split_over_ranks(tmp_path, world, num_samples, batch_size):
    """The product's sample_sharded + gather_samples under a real multi-rank process group (gloo): 64 samples on 8 ranks as
    1 x 8 or 2 x 4 per rank (BASELINE configs[2]), on 2 ranks, and an uneven 61: every rank ends with all samples in global
    index order, equal to what one process draws, and each rank ran exactly its contiguous block in batches of batch_size."""
    from protein_redesign_amd.synthetic import NoiseSource
    batch = synthetic_batch([(3, 8)], esm_dim=16, seed=4)
    want_pos = torch.stack([NoiseSource(5, k).randn(11, 3) for k in range(num_samples)])
    port = 29700 + (os.getpid() % 200) + world + batch_size
    mp.spawn(_worker_stub, args=(world, port, num_samples, batch_size, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        pos, logits, calls = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert pos.shape == (num_samples, 11, 3) and logits.shape == (num_samples, 11, 21)
        assert torch.equal(pos, want_pos)
        mine = len(shard_range(num_samples, world, r))
        full, rest = divmod(mine, batch_size)
        assert calls == [batch_size] * full + ([rest] if rest else [])
    del batch
