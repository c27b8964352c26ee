"""Multi-GPU path as far as ONE GPU allows (reference fan-out furniture/env/base.py:74-80): two
PROCESSES share the one device (both ranks on cuda:0), each steps its shard of the global env range on the device, the observation slabs
are gathered over a process group -- and every env's observation, reward and done flag is bit-identical to the same global envs
stepped by ONE process in one batch of twice the size.  (RCCL refuses two ranks on one device, so the group is gloo and the slabs
travel as host tensors; what is under test is the sharding rule -- global env i lives on rank i // envs_per_rank, seeded seed + i,
first_env_index -- with REAL device observations, and that an env's bits do not depend on the batch it is stepped in.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

PER, STEPS, SEED = 96, 4, 41


def _run(first, n):
    """reset + STEPS steps of global envs [first, first + n) on cuda:0; actions are a function of the GLOBAL env index"""
    from furniture_amd.envs import make_vec_env
    from tests.scenarios import counter_actions
    env = make_vec_env("Sawyer", n, furniture_name="table_lack_0825", max_episode_steps=3, seed=SEED, record_vid=False, unity=False,
                       control_type="impedance", first_env_index=first, multi_wave="rule")
    ob = env.reset()
    out = [torch.cat([ob["object_ob"], ob["robot_ob"]], dim=1).cpu()]
    rews, dones = [], []
    for t in range(STEPS):  # (crosses the in-kernel auto-reset at max_episode_steps = 3: the reset tables are sharded the same way)
        a = torch.as_tensor(np.stack([counter_actions(7, first + i, t, 9) for i in range(n)])).float().to(env.sim.device)
        ob, rew, done, info = env.step(a)
        out.append(torch.cat([ob["object_ob"], ob["robot_ob"]], dim=1).cpu())
        rews.append(rew.cpu()), dones.append(done.cpu().to(torch.uint8))
    env.close()
    return torch.stack(out), torch.stack(rews), torch.stack(dones)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from furniture_amd.dist import shard_range, gather_observations
    lo, hi = shard_range(rank, world, PER)
    obs, rew, done = _run(lo, hi - lo)  # [STEPS + 1, PER, d], [STEPS, PER], [STEPS, PER]
    gathered = []
    for t in range(STEPS):
        g_obs, g_rew, g_done = gather_observations(obs[t + 1].contiguous(), rew[t].contiguous(), done[t].contiguous(), tag=t)
        gathered.append((g_obs.clone(), g_rew.clone(), g_done.clone()))
    g0 = [torch.empty_like(obs[0]) for _ in range(world)]
    dist.all_gather(g0, obs[0].contiguous())
    if rank == 0:  # the one-process batch of the same global envs
        w_obs, w_rew, w_done = _run(0, world * PER)
        assert torch.equal(torch.cat(g0), w_obs[0]), "reset observations differ between 2 x %d and 1 x %d envs" % (PER, world * PER)
        for t in range(STEPS):
            assert torch.equal(gathered[t][0], w_obs[t + 1]), t
            assert torch.equal(gathered[t][1], w_rew[t]) and torch.equal(gathered[t][2], w_done[t]), t
        assert bool(w_done[2].all())  # (the auto-reset was inside the window)
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_two_processes_on_one_gpu_reproduce_the_one_process_batch_per_env():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]
