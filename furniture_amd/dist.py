"""Multi-GPU plumbing: environments shard trivially (no cross-env state); the only exchange is
the per-step all-gather of the observation slab (+ reward, done) to the learner.  ``torch.distributed`` backend "nccl"
is RCCL on ROCm; on a fully connected 8-GPU xGMI node this ~1 MiB-per-rank gather is latency-bound, so one
``all_gather_into_tensor`` per field per step is issued (no bucketing, no ring tuning)."""

import torch
import torch.distributed as dist


def shard_range(rank, world_size, envs_per_rank):
    """Global env indices [lo, hi) owned by ``rank``: env i lives on rank i // envs_per_rank and is seeded seed + i, so
    results do not depend on the number of GPUs."""
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def gather_observations(obs, reward, done):
    """All ranks contribute [n, d] / [n] / [n]; every rank gets the [world*n, ...] tensors in global env order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return obs, reward, done
    w = dist.get_world_size()
    out = []
    for t in (obs, reward, done):
        t = t.contiguous()
        g = torch.empty((w * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(g, t)
        out.append(g)
    return tuple(out)
