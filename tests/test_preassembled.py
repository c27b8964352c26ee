"""Pre-assembled starts (FurnitureEnv.set_subtask / config.preassembled / config.num_connects, furniture.py:163, 204-207,
1476-1503, 1542-1566): the CPU restatement here, the device against it in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import preassembled_rows
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig


def test_recipe_rows_resolve_to_connectors_of_the_right_parts():
    m = load_compiled("Sawyer", "table_lack_0825")
    ids, pairs, angles = preassembled_rows(m, [0, 1, 2, 3])
    assert list(ids) == [0, 1, 2, 3] and list(angles) == [90, 90, 270, 270]
    top = m.meta["part_names"].index("4_part4")
    legs = set()
    for k_table, k_leg in pairs:  # _connect(site2_id, site1_id): recipe site2 is on the table top, site1 on a leg
        assert int(m.conn_partid[k_table]) == top and int(m.conn_partid[k_leg]) != top
        legs.add(int(m.conn_partid[k_leg]))
    assert len(legs) == 4
    m2 = load_compiled("Sawyer", "swivel_chair_0700")  # no recipe file: the list holds weld ids
    ids, pairs, angles = preassembled_rows(m2, [1])
    assert list(ids) == [1] and pairs is None and angles is None


def test_oracle_reset_with_recipe_steps_preassembled():
    m = load_compiled("Sawyer", "table_lack_0825")
    env = FurnitureEnvOracle(m, OracleConfig(seed=7, preassembled=[0, 1], max_episode_steps=50))
    ob = env.reset()
    assert env._num_connected == 2 and env._prev_num_connected == 0 and not env._connected and env._connected_body1 is None
    act = np.asarray(env.sim.model.eq_active).astype(int)
    assert act.sum() == 2
    roots = {env._find_group(i) for i in range(env.nparts)}
    assert len(roots) == env.nparts - 2  # two legs joined the table top's group
    assert (env._subtask_part1, env._subtask_part2) != (-1, -1)
    # the welded legs stand on their connectors of the (upside-down) table top: their sites coincide with the table's
    sites, conn = list(m.meta["site_names"]), [int(x) for x in m.conn_siteid]
    for i in (0, 1):
        row = m.meta["site_recipe"][i]
        p1 = env.sim.data.site_xpos[sites.index(row[0])]
        p2 = env.sim.data.site_xpos[sites.index(row[1])]
        assert np.linalg.norm(p1 - p2) < 5e-3, (i, p1, p2)
    assert np.isfinite(env.flat_obs(ob)).all()
    # the reference's counters: the first step pays success_reward for the pre-assembled connects (prev_num_connected = 0)
    ob, rew, done, info = env.step(np.zeros(9))
    assert rew > 190 and not done
    # set_subtask(3, num_connects=1): success after ONE more connect
    env.set_subtask(3, num_connects=1)
    env.reset()
    assert env._num_connected == 3 and env._success_num_conn == 4


def test_oracle_reset_with_welds_preassembled_without_a_recipe():
    m = load_compiled("Sawyer", "swivel_chair_0700")
    env = FurnitureEnvOracle(m, OracleConfig(seed=3, preassembled=[0]))
    env.reset()
    assert int(np.asarray(env.sim.model.eq_active)[0]) == 1 and env._num_connected == 0
    p1, p2 = int(m.eq_part1[0]), int(m.eq_part2[0])
    assert env._find_group(p1) == env._find_group(p2)
    # the active weld pulled the two parts into their assembled relative pose during the reset's 400 substeps
    from furniture_amd import transform_utils as T
    rel = T.rel_pose(env._part_qpos(p1), env._part_qpos(p2))
    assert np.abs(rel[:3] - m.eq_data0[0][:3]).max() < 5e-3


def test_oracle_assembled_and_fix_init():
    """config.assembled (furniture.py:1502-1503, 1526-1530): every weld on, one group, the parts stay at the XML's assembled poses;
    config.fix_init (furniture.py:1518-1525): the first placement is kept, later resets take no placement draw"""
    m = load_compiled("Sawyer", "table_lack_0825")
    env = FurnitureEnvOracle(m, OracleConfig(seed=4, assembled=True))
    ob = env.reset()
    assert np.asarray(env.sim.model.eq_active).astype(int).tolist() == [1] * m.neq
    assert len({env._find_group(i) for i in range(env.nparts)}) == 1 and (env._subtask_part1, env._subtask_part2) == (-1, -1)
    # (the XML lays the parts out side by side: the active welds pull them into the assembled relative poses during the reset)
    from furniture_amd import transform_utils as T
    for e in range(m.neq):
        rel = T.rel_pose(env._part_qpos(int(m.eq_part1[e])), env._part_qpos(int(m.eq_part2[e])))
        assert np.abs(rel[:3] - m.eq_data0[e][:3]).max() < 2e-2, (e, rel[:3], m.eq_data0[e][:3])
    assert np.isfinite(env.flat_obs(ob)).all()
    a = FurnitureEnvOracle(m, OracleConfig(seed=4, fix_init=True))
    b = FurnitureEnvOracle(m, OracleConfig(seed=4))
    a.reset(); b.reset()
    pa0, pb0 = a.reset_draws["part_qpos"].copy(), b.reset_draws["part_qpos"].copy()
    assert np.array_equal(pa0, pb0)
    a.reset(); b.reset()
    assert np.array_equal(a.reset_draws["part_qpos"], pa0) and not np.array_equal(b.reset_draws["part_qpos"], pb0)
    # ... and the robot noise of the second reset comes earlier in the stream (no placement draws in between)
    assert not np.array_equal(np.stack(a.reset_draws["noise"]), np.stack(b.reset_draws["noise"]))


def test_host_sampler_follows_fix_init_and_assembled():
    from furniture_amd.envs import ResetTableSampler, make_config
    m = load_compiled("Sawyer", "table_lack_0825")
    for kw in (dict(fix_init=True), dict(assembled=True), dict()):
        s = ResetTableSampler(m, make_config(**kw), 4, 0, 1)
        o = FurnitureEnvOracle(m, OracleConfig(seed=4, **kw))
        for rep in range(3):
            p, nz = s.draw()
            o.reset()
            assert np.abs(p[0] - o.reset_draws["part_qpos"].reshape(-1)).max() < 1e-6, (kw, rep)
            assert np.abs(nz[0] - np.stack(o.reset_draws["noise"]).reshape(-1)).max() < 1e-7, (kw, rep)
