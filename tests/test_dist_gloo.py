"""Multi-process path on CPU (gloo, world_size 2): env sharding, seed invariance and the observation all-gather that
bench.py performs per step.  No GPU: the shards carry synthetic observation slabs."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from furniture_amd.dist import shard_range, gather_observations, step_wait_and_gather
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
    # bf16 observation slab: one byte-packed collective; the bf16 values, the fp32 rewards and the done flags arrive unchanged
    b_obs, b_rew, b_done = gather_observations(obs.bfloat16(), rew + 0.125, done, tag=1)
    assert b_obs.dtype == torch.bfloat16 and torch.equal(b_obs, torch.as_tensor(whole[:, :8]).float().bfloat16())
    assert b_rew.tolist() == [i + 0.125 for i in range(world * per)] and b_done.tolist() == g_done.tolist()
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


def _env_worker(rank, world, port, q):
    """real env observations: the CPU oracle env stands in for the device (same seeding rule: global env i <- seed + i)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Sawyer", "table_lack_0825")
    per = 2

    def run(i):
        env = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=20, seed=123 + i))
        env.reset()
        ob, rew, done, _ = env.step(np.random.RandomState(1000 + i).uniform(-1, 1, 9))
        return env.flat_obs(ob), rew, done

    lo, hi = shard_range(rank, world, per)
    mine = [run(i) for i in range(lo, hi)]
    obs = torch.as_tensor(np.stack([x[0] for x in mine])).float()
    rew = torch.as_tensor([x[1] for x in mine]).float()
    done = torch.as_tensor([int(x[2]) for x in mine]).to(torch.uint8)
    g_obs, g_rew, g_done = gather_observations(obs, rew, done)
    if rank == 0:  # the one-process batch of the same global envs: bit-identical per env
        whole = [run(i) for i in range(world * per)]
        assert np.array_equal(g_obs.numpy(), np.stack([x[0] for x in whole]).astype(np.float32))
        assert np.array_equal(g_rew.numpy(), np.asarray([x[1] for x in whole], dtype=np.float32))
        assert g_done.tolist() == [int(x[2]) for x in whole]
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_one_rank_and_two_rank_batches_are_identical_per_env():
    """env i -> rank i // envs_per_rank, seed 123 + i, so the gathered observation / reward / done slab does not depend
    on the number of ranks: 2 ranks x 2 oracle envs (reset + one random-action step) against the same 4 envs in one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_env_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]


def _restep_worker(rank, world, port, q):
    """the handle's sync() rewrites one env's rows on rank 1 only (what fsim_sync does when it re-steps an env that dropped contacts);
    the slab every rank receives must carry the rewritten rows"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per, d = 4, 5
    obs = torch.full((per, d), float(rank))
    rew = torch.full((per,), 10.0 + rank)
    done = torch.zeros(per, dtype=torch.uint8)

    class Handle:
        synced = 0

        def sync(self):  # (first pass's rows are in obs / rew / done when this is called)
            self.synced += 1
            if rank == 1:
                obs[2] = 99.0
                rew[2] = -1.0
                done[2] = 1

    h = Handle()
    g_obs, g_rew, g_done = step_wait_and_gather(h, obs, rew, done)
    assert h.synced == 1
    want_obs = torch.cat([torch.full((per, d), 0.0), torch.full((per, d), 1.0)])
    want_obs[per + 2] = 99.0
    want_rew = torch.cat([torch.full((per,), 10.0), torch.full((per,), 11.0)])
    want_rew[per + 2] = -1.0
    assert torch.equal(g_obs, want_obs) and torch.equal(g_rew, want_rew), (rank, g_obs, g_rew)
    assert g_done.tolist() == [0] * (per + 2) + [1, 0]
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_gathered_slab_carries_the_rows_of_an_overflow_restep():
    """a re-step inside fsim_sync on ONE rank must not leave the learner a stale row.  bench.py's RCCL path enters
    the all-gather after the sync on every rank (furniture_amd/dist.py step_wait_and_gather)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_restep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]
