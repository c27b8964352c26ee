"""Asynchronous stepping (fsim_step_subset, furniture_amd/async_env.py): envs are independent, so stepping them in dynamically
formed subsets on several streams must reproduce the synchronous per-env trajectories bit for bit -- including in-kernel
auto-resets, whose reset tables are uploaded while other envs' kernels are running."""
import numpy as np
import pytest
import torch

from furniture_amd.async_env import FurnitureAsyncBatchEnv
from furniture_amd.envs import FurnitureBatchEnv, make_config
from tests.scenarios import counter_actions

pytestmark = pytest.mark.gpu


def _cfg():
    return make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=4, seed=11)


def test_async_trajectories_equal_sync():
    n, K = 96, 7
    sync = FurnitureBatchEnv("Sawyer", n, config=_cfg())
    ob = sync.reset()
    want = np.zeros((K, n, sync.sim.obs_dim), np.float32)
    wrew, wdone = np.zeros((K, n), np.float32), np.zeros((K, n), bool)
    for t in range(K):
        a = np.stack([counter_actions(3, i, t, sync.dof) for i in range(n)])
        ob, rew, done, _ = sync.step(a)
        want[t] = torch.cat([ob["object_ob"], ob["robot_ob"]], 1).cpu().numpy()
        wrew[t], wdone[t] = rew.cpu().numpy(), done.cpu().numpy()
    sync.close()
    assert wdone[3].all() and wdone[:3].sum() == 0  # a time-limit reset happens inside the run

    env = FurnitureAsyncBatchEnv("Sawyer", n, config=_cfg(), cheap_min_fraction=0.3)
    env.reset()
    steps = np.zeros(n, dtype=int)
    # mark a third of the envs "expensive" by hand so that both kinds of batches and all queues are exercised from the start
    env._expensive[::3] = True
    ids = np.arange(n)
    env.send(np.stack([counter_actions(3, i, 0, env.dof) for i in ids]), ids)
    got = 0
    while got < n * K:
        ids, obs, rew, done, info = env.recv()
        o, r, d = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        for j, i in enumerate(ids):
            t = steps[i]
            assert bool(d[j]) == wdone[t, i] and r[j] == wrew[t, i], (i, t, "reward/done", r[j], wrew[t, i])
            assert np.array_equal(o[j], want[t, i]), (i, t, float(np.abs(o[j] - want[t, i]).max()), int(np.abs(o[j] - want[t, i]).argmax()))
        steps[ids] += 1
        got += len(ids)
        nxt = ids[steps[ids] < K]
        if len(nxt):
            env.send(np.stack([counter_actions(3, i, steps[i], env.dof) for i in nxt]), nxt)
    assert (steps == K).all() and env.stats["expensive_batches"] > 0 and env.stats["cheap_batches"] > 0
    env.close()


def test_subset_step_leaves_other_rows_untouched():
    n = 16
    b = FurnitureBatchEnv("Sawyer", n, config=_cfg())
    b.reset()
    sim, dev = b.sim, b.sim.device
    before = b._obs.clone()
    ids = torch.tensor([3, 7, 8], dtype=torch.int32, device=dev)
    b._act.zero_()
    torch.cuda.synchronize()
    sim.step_subset(1, ids, 3, b._act, b._obs, b._rew, b._done, b._info)
    assert sim.queue_busy(1) in (True, False)
    sim.queue_sync(1)
    assert not sim.queue_busy(1)
    changed = (b._obs != before).any(dim=1).cpu().numpy()
    assert changed[[3, 7, 8]].all() and not np.delete(changed, [3, 7, 8]).any()
    assert b._info.cpu().numpy()[[3, 7, 8], 5].tolist() == [1, 1, 1] and b._info.cpu().numpy()[0, 5] == 0  # episode_length
    b.close()
