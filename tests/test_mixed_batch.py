"""BASELINE config 5 (mixed-furniture batch: lanes cycle table_lack_0825 / chair_agne_0007 / shelf_ivar_0678).
CPU: lane assignment, padded slab layout and its world_size-2 gloo all-gather.  GPU: every lane of the mixed batch is
bit-identical to the same global env index stepped inside a homogeneous batch, and one lane per furniture matches the
fp64 oracle env."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from furniture_amd.mixed import lane_assignment, padded_layout

NAMES = ["table_lack_0825", "chair_agne_0007", "shelf_ivar_0678"]


def test_lane_assignment_is_invariant_to_sharding():
    whole = lane_assignment(12, 3, 0)
    assert [r.tolist() for r in whole] == [[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8, 11]]
    # rank 1 of 2 (lanes 6..11): the same global lane keeps the same model
    part = lane_assignment(6, 3, 6)
    for j in range(3):
        assert (6 + part[j]).tolist() == [g for g in whole[j].tolist() if g >= 6]
    assert padded_layout([5, 3, 7], 29) == (49, 78)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    from furniture_amd.dist import gather_observations
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per, k, nparts, robot = 6, 3, [5, 3, 7], 4
    oc, od = padded_layout(nparts, robot)
    rows = lane_assignment(per, k, rank * per)
    slab = torch.zeros((per, od))
    for j in range(k):
        for r in rows[j]:
            g = rank * per + int(r)
            slab[r, :7 * nparts[j]] = g + 1          # object_ob of a model with nparts[j] parts
            slab[r, oc:] = -(g + 1)                  # robot_ob
    g_obs, g_rew, _ = gather_observations(slab, torch.arange(rank * per, (rank + 1) * per).float(), torch.zeros(per, dtype=torch.uint8))
    assert g_obs.shape == (world * per, od)
    for g in range(world * per):
        j = g % k
        assert (g_obs[g, :7 * nparts[j]] == g + 1).all() and (g_obs[g, 7 * nparts[j]:oc] == 0).all() and (g_obs[g, oc:] == -(g + 1)).all()
    assert g_rew.tolist() == list(range(world * per))
    dist.barrier()
    dist.destroy_process_group()
    q.put(rank)


def test_padded_slab_gather_two_processes():
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


@pytest.mark.gpu
def test_mixed_batch_lanes_equal_homogeneous_batches():
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    from furniture_amd.mixed import FurnitureMixedBatchEnv
    from tests.scenarios import counter_actions
    n, k = 9, len(NAMES)
    kw = dict(unity=False, record_vid=False, control_type="impedance", max_episode_steps=150)
    mix = FurnitureMixedBatchEnv("Sawyer", NAMES, n, config=make_config(**kw), auto_reset=False)
    from furniture_amd.mjcf.model import load_compiled
    assert mix.obs_dim == 7 * max(load_compiled("Sawyer", nm).nparts for nm in NAMES) + 29  # object_ob padded to the largest part count
    ob = mix.reset()
    acts = [np.stack([counter_actions(77, i, t, mix.dof) for i in range(n)]) for t in range(3)]
    traj = [torch.cat([ob["object_ob"], ob["robot_ob"]], 1).cpu().numpy().copy()]
    rews = []
    for a in acts:
        ob, rew, done, info = mix.step(a)
        traj.append(torch.cat([ob["object_ob"], ob["robot_ob"]], 1).cpu().numpy().copy())
        rews.append(rew.cpu().numpy().copy())
    assert info["model_id"].tolist() == [i % k for i in range(n)]
    mix.close()
    for j, name in enumerate(NAMES):
        rows = list(range(j, n, k))
        hom = FurnitureBatchEnv("Sawyer", len(rows), config=make_config(furniture_name=name, **kw), auto_reset=False, env_indices=rows)
        ko = 7 * hom.n_obj
        o = hom.reset()
        seq = [torch.cat([o["object_ob"], o["robot_ob"]], 1).cpu().numpy().copy()]
        for t, a in enumerate(acts):
            o, rew, _, _ = hom.step(a[rows])
            seq.append(torch.cat([o["object_ob"], o["robot_ob"]], 1).cpu().numpy().copy())
            assert np.array_equal(rew.cpu().numpy(), rews[t][rows])
        for t in range(len(seq)):
            assert np.array_equal(traj[t][rows][:, :ko], seq[t][:, :ko])           # object_ob, bit-exact
            assert np.all(traj[t][rows][:, ko:mix.obj_cols] == 0)                   # padding
            assert np.array_equal(traj[t][rows][:, mix.obj_cols:], seq[t][:, ko:])  # robot_ob
        hom.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["chair_agne_0007", "shelf_ivar_0678"])
def test_config5_models_match_oracle(name):
    """The two config-5 models that have no other parity test: in-kernel reset + random steps vs the fp64 oracle env."""
    from furniture_amd.mjcf.model import load_compiled
    from furniture_amd.sim import FSim, INFO_DIM, default_config
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    from tests.scenarios import counter_actions
    m = load_compiled("Sawyer", name)
    n = 2
    cfg = default_config()
    cfg.max_episode_steps = 150
    cfg.auto_reset = 0
    sim = FSim(m, n, config=cfg)
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10)) for i in range(n)]
    obs_o = [e.reset() for e in envs]
    sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]),
                         np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    for e in range(n):
        assert np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(obs_o[e])).max() < 1e-4
    dof = sim.dof_action
    act = torch.zeros((n, dof), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    for t in range(3):
        a = np.stack([counter_actions(321, i, t, dof) for i in range(n)])
        act.copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        for e in range(n):
            ob, r, d, _ = envs[e].step(a[e])
            assert np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob)).max() < 5e-4
            assert abs(float(rew[e]) - r) < 1e-4 and bool(done[e]) == d
    sim.close()
