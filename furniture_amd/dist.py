"""Multi-GPU plumbing: environments shard trivially (no cross-env state); the only exchange is
the per-step all-gather of the observation slab (+ reward, done) to the learner.  ``torch.distributed`` backend "nccl"
is RCCL on ROCm; on a fully connected 8-GPU xGMI node this ~1 MiB-per-rank gather is latency-bound (xGMI is point to point,
7 links per GPU), so the three fields travel in ONE ``all_gather_into_tensor`` per step: they are packed into a single
[n, d + 2] slab (reward and done as two extra float32 columns -- done is 0 / 1, exact) and split again on arrival.  No
bucketing, no ring tuning: a 1 MiB message does not reach the bandwidth regime."""

import torch
import torch.distributed as dist

_BUF = {}


def shard_range(rank, world_size, envs_per_rank):
    """Global env indices [lo, hi) owned by ``rank``: env i lives on rank i // envs_per_rank and is seeded seed + i, so
    results do not depend on the number of GPUs."""
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def gather_observations(obs, reward, done, tag=0, stream=None, group=None):
    """All ranks contribute [n, d] / [n] / [n]; every rank gets the [world*n, ...] tensors in global env order.  The returned
    tensors are views of a persistent receive buffer (one per ``tag``): valid until the next call with the same tag and shapes.

    stream: the HIP stream the producing fsim_step was enqueued on (``FSim.torch_stream``).  The packing copies and the
    collective are then enqueued BEHIND the step kernel on that stream -- no host synchronisation between step and gather
    (SURVEY.md section 8e); the caller synchronises the stream once, when it needs the result.  group: the process group to
    gather over -- RCCL runs all collectives of one communicator on one internal stream, so env slabs that are stepped
    pipelined on separate streams need a communicator each (``dist.new_group()``), or slab B's gather queues behind slab A's
    step kernel.  Without a process group (single process) the inputs are returned as they are."""
    if not (dist.is_available() and dist.is_initialized()):
        return obs, reward, done
    if stream is not None and obs.is_cuda:
        with torch.cuda.stream(stream):
            return gather_observations(obs, reward, done, tag=tag, group=group)
    w = dist.get_world_size(group)
    n, d = obs.shape
    if obs.dtype == torch.bfloat16 and reward.dtype == torch.float32 and obs.is_contiguous():
        # bf16 observation slab (BASELINE config 2's narrow slab): still ONE collective and persistent buffers -- the row is packed
        # byte-wise as [obs bf16 x d | reward fp32 | done u8], half the bytes of the fp32 slab
        key = (obs.device, n, d, w, tag, "bf16")
        row = 2 * d + 5
        if key not in _BUF:
            _BUF[key] = (torch.empty((n, row), dtype=torch.uint8, device=obs.device), torch.empty((w * n, row), dtype=torch.uint8, device=obs.device),
                         torch.empty((w * n, 2 * d), dtype=torch.uint8, device=obs.device), torch.empty((w * n, 4), dtype=torch.uint8, device=obs.device))
        send, recv, o8, r8 = _BUF[key]
        send[:, :2 * d] = obs.view(torch.uint8)
        send[:, 2 * d:2 * d + 4] = reward.contiguous().view(torch.uint8).view(n, 4)
        send[:, 2 * d + 4] = done.to(torch.uint8)
        dist.all_gather_into_tensor(recv, send, group=group)
        o8.copy_(recv[:, :2 * d])
        r8.copy_(recv[:, 2 * d:2 * d + 4])
        return o8.view(torch.bfloat16), r8.view(torch.float32).view(w * n), recv[:, 2 * d + 4].to(done.dtype)
    if obs.dtype != torch.float32 or reward.dtype != torch.float32:
        out = []
        for t in (obs, reward, done):  # generic path: one collective per field
            t = t.contiguous()
            g = torch.empty((w * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(g, t, group=group)
            out.append(g)
        return tuple(out)
    key = (obs.device, n, d, w, tag)
    if key not in _BUF:  # persistent staging buffers: no allocation inside the step loop
        _BUF[key] = (torch.empty((n, d + 2), dtype=torch.float32, device=obs.device),
                     torch.empty((w * n, d + 2), dtype=torch.float32, device=obs.device))
    send, recv = _BUF[key]
    send[:, :d] = obs
    send[:, d] = reward
    send[:, d + 1] = done
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv[:, :d], recv[:, d], recv[:, d + 1].to(done.dtype)


def step_wait_and_gather(sim, obs, reward, done, tag=0, stream=None, group=None):
    """What a learner rank does at the end of a slab-step under RCCL: FIRST ``sim.sync()`` -- which, besides waiting for the step, may
    re-step an env whose contacts did not fit the kernel's slots and rewrite its rows (include/fsim.h fsim_overflow_resteps) --, THEN the
    all-gather, so that every rank receives the rows that are final.  A gather enqueued behind the step kernel instead would carry the
    first pass's rows of such an env, and repeating it only on the rank that saw the re-step is not possible (a collective is entered by
    every rank).  One collective per slab-step on every rank, whatever happened.  ``sim``: anything with ``sync()`` (furniture_amd.sim.FSim)."""
    sim.sync()
    out = gather_observations(obs, reward, done, tag=tag, stream=stream, group=group)
    if stream is not None and obs.is_cuda:
        stream.synchronize()
    return out
