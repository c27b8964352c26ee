"""Multi-process path on CPU (gloo, world_size 2): env sharding, seed invariance and the observation all-gather that
bench.py performs per step.  No GPU: the shards carry synthetic observation slabs."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from furniture_amd.dist import shard_range, gather_observations
from furniture_amd.envs import ResetTableSampler
from furniture_amd.mjcf.model import load_compiled


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = 6
    lo, hi = shard_range(rank, world, per)
    assert (lo, hi) == (rank * per, (rank + 1) * per)
    m = load_compiled("Sawyer", "table_lack_0825")
    cfg = SimpleNamespace(furn_xyz_rand=0.02, furn_rot_rand=3, agent_xyz_rand=0.001)
    parts, _ = ResetTableSampler(m, cfg, 123, lo, per).draw()
    obs = torch.as_tensor(parts[:, :8]).float() + 0.0  # stand-in observation slab: depends only on the GLOBAL env index
    rew = torch.arange(lo, hi).float()
    done = (torch.arange(lo, hi) % 2).to(torch.uint8)
    g_obs, g_rew, g_done = gather_observations(obs, rew, done)
    assert g_obs.shape == (world * per, 8) and g_rew.tolist() == list(range(world * per))
    assert g_done.tolist() == [i % 2 for i in range(world * per)]
    whole, _ = ResetTableSampler(m, cfg, 123, 0, world * per).draw()
    assert np.array_equal(g_obs.numpy(), whole[:, :8])  # 1-process and 2-process batches are identical per env
    # the timing reduction bench.py uses: max over ranks
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(world)
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_two_process_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]
