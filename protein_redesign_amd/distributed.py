"""Sharding ``num_samples`` reverse-diffusion samples over the GPUs of one node.

Samples never interact, so the path shards by GLOBAL sample index with no collective
inside the loop: rank r runs a contiguous block of indices through the full T-step loop on its own
GPU, then ONE all_gather (RCCL over xGMI when the backend is "nccl"; ~2 MB for 64 x 320-node samples)
returns every sample to every rank in index order.  Noise and the redesign mask of sample k are keyed
by (seed, k) (synthetic.NoiseSource), so 1-, 2-, 4- and 8-GPU runs give identical per-sample results --
unlike the reference's DDP scripts, which reseed every rank identically (generate.py:95).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .synthetic import NoiseSource, clone_batch


def shard_range(num_samples: int, world_size: int, rank: int) -> range:
    """Contiguous block of global sample indices owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(num_samples, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def repeat_batch(batch: Dict[str, torch.Tensor], n: int) -> Dict[str, torch.Tensor]:
    """RepeatDataset + collate of the reference (data.py:145-155, generate.py:138-153): n copies of one complex."""
    out = {}
    for k, v in batch.items():
        out[k] = v.repeat(n, *([1] * (v.dim() - 1))) if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 1 else v
    return out


def gather_samples(pos: torch.Tensor, logits: torch.Tensor, num_samples: int,
                   group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The one collective of the sampling job: every rank contributes the samples of its shard (``pos [n_r,N,3]``,
    ``logits [n_r,N,21]``, n_r = len(shard_range(num_samples, world, rank))) and receives all ``num_samples`` in global
    index order.  RCCL all_gather when the backend is "nccl"; shards are padded to the largest one.  World size 1 (or no
    process group): returns the inputs."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    mine = len(shard_range(num_samples, world, rank))
    if pos.shape[0] != mine:
        raise ValueError(f"rank {rank} holds {pos.shape[0]} samples, its shard of {num_samples} over {world} ranks is {mine}")
    if not distributed:
        return pos.contiguous(), logits.contiguous()
    cap = len(shard_range(num_samples, world, 0))              # largest shard (a one-rank group still runs the collective)
    packed = torch.zeros(cap, pos.shape[1], 3 + logits.shape[-1], device=pos.device, dtype=torch.float32)
    packed[:mine] = torch.cat([pos, logits], dim=-1)
    gathered = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(gathered, packed, group=group)
    allr = torch.cat([gathered[r][:len(shard_range(num_samples, world, r))] for r in range(world)])
    return allr[..., :3].contiguous(), allr[..., 3:].contiguous()


def sample_sharded(sampler: Callable[[Dict[str, torch.Tensor], Sequence[NoiseSource]], Tuple[torch.Tensor, torch.Tensor]],
                   complex_batch: Dict[str, torch.Tensor], num_samples: int, seed: int = 0, batch_size: int = 1,
                   group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Draw ``num_samples`` samples of ONE complex (a batch dict with leading dim 1).

    ``sampler(batch, sources) -> (pos [b,N,3], logits [b,N,21])`` is ``ProteinReDiffModel.sample`` on a
    GPU rank (tests inject a CPU stand-in).  Works without an initialised process group (world size 1).
    Returns all samples, ordered by global index, on every rank."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    mine = list(shard_range(num_samples, world, rank))
    pos_l: List[torch.Tensor] = []
    log_l: List[torch.Tensor] = []
    for s in range(0, len(mine), batch_size):
        idx = mine[s:s + batch_size]
        sources = [NoiseSource(seed, k) for k in idx]
        pos, logits = sampler(repeat_batch(clone_batch(complex_batch), len(idx)), sources)
        pos_l.append(pos)
        log_l.append(logits)
    N = complex_batch["atom_mask"].shape[1]
    device = pos_l[0].device if pos_l else complex_batch["atom_mask"].device
    pos = torch.cat(pos_l) if pos_l else torch.zeros(0, N, 3, device=device)
    logits = torch.cat(log_l) if log_l else torch.zeros(0, N, 21, device=device)
    return gather_samples(pos, logits, num_samples, group=group)
