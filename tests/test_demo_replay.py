"""Replay of the reference's MuJoCo-recorded Cursor demo (tests/golden/demo_cursor7.npz, from demos/Cursor_7.pkl via
scripts/make_golden_demo.py): the only trajectory in the reference tree that came out of MuJoCo itself.  62 frames:
cursors move, each selects a chair part by contact and carries it, ten approach steps, the connect.

The recording predates today's assets (the seat connector sits 8 mm lower there and the seat's yaw is snapped to the
column's), so the seat's z / yaw carry that documented offset; everything else -- cursor paths, which part each cursor
picks up and when, x / y of carried parts, resting parts not moving at all, the lift of the column at the connect, the
connect happening on the eleventh request -- must match the recording."""
import os

import numpy as np
import pytest

from furniture_amd.mjcf.model import load_compiled

D = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_cursor7.npz"))


def _check(frames_parts, frames_cursor, connected_at):
    rec_p, rec_c = D["parts"], np.concatenate([D["cursor0"], D["cursor1"]], axis=1)
    n = len(rec_p)
    assert np.abs(frames_cursor - rec_c[1:n]).max() < 1e-6                       # cursor paths
    dp = frames_parts[:, :, :3] - rec_p[1:n, :, :3]
    assert np.abs(dp[:, :, :2]).max() < 1.5e-3                                    # x, y of every part, every frame
    assert np.abs(dp[:, 0, 2]).max() < 2e-4                                       # the base never moves
    assert np.abs(dp[:58, 1, 2]).max() < 5e-4 and abs(dp[-1, 1, 2]) < 5e-4         # column: resting, then lifted by _connect
    assert np.abs(dp[:, 2, 2]).max() < 1.2e-2                                     # seat height (asset revision offset)
    dq = np.minimum(np.abs(frames_parts[:, :, 3:] - rec_p[1:n, :, 3:]), np.abs(frames_parts[:, :, 3:] + rec_p[1:n, :, 3:]))
    assert dq.max() < 6e-3                                                        # orientations (incl. the yaw snap of the seat)
    assert connected_at == 60
    # the column is lifted off the floor by the connect in the recording too (0.0691 -> 0.0720)
    assert abs(rec_p[61, 1, 2] - 0.0720) < 2e-4 and abs(frames_parts[-1, 1, 2] - 0.0720) < 5e-4


def _check_ext(frames_parts, frames_cursor):
    """frames 62-91: one cursor carries the welded column + seat about, the base stays where it is"""
    rec_p, rec_c = D["parts_ext"], np.concatenate([D["cursor0_ext"], D["cursor1_ext"]], axis=1)
    n = len(rec_p)
    assert np.abs(frames_cursor - rec_c[1:n]).max() < 1e-6
    dp = frames_parts[61:, :, :3] - rec_p[62:n, :, :3]
    assert np.abs(dp[:, 0]).max() < 5e-4                                          # the base
    assert np.abs(dp[:, 1]).max() < 2.5e-3                                        # the column, carried (x, y, z)
    assert np.abs(dp[:, 2, :2]).max() < 2.5e-3 and np.abs(dp[:, 2, 2]).max() < 1.2e-2  # the seat on it (height: asset revision offset)
    assert np.abs(rec_p[91, 1, :3] - rec_p[62, 1, :3]).max() > 0.1                # (the recording: it is carried more than 10 cm)
    dq = np.minimum(np.abs(frames_parts[61:, :, 3:] - rec_p[62:n, :, 3:]), np.abs(frames_parts[61:, :, 3:] + rec_p[62:n, :, 3:]))
    assert dq.max() < 6e-3


def _replay_oracle(actions):
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Cursor", "swivel_chair_0700")
    assert list(m.meta["part_names"]) == [str(x) for x in D["part_names"]]
    env = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=10000, move_speed=0.025, rotate_speed=22.5))
    env.reset()
    for i in range(m.nparts):
        env._set_part_qpos(i, D["parts"][0, i, :3], D["parts"][0, i, 3:])
    env.sim.data.qvel[:] = 0
    env.sim.model.body_pos[m.cursor_bodyid[0]] = D["cursor0"][0]
    env.sim.model.body_pos[m.cursor_bodyid[1]] = D["cursor1"][0]
    env.sim.forward()
    env._cursor_selected = [None, None]
    P, C, connected_at = [], [], None
    for t, a in enumerate(actions):
        ob, r, done, info = env.step(a)
        P.append([env._part_qpos(i) for i in range(m.nparts)])
        C.append(np.concatenate([env._cursor_pos(0), env._cursor_pos(1)]))
        if info["connected_this_step"] and connected_at is None:
            connected_at = t
    return np.array(P), np.array(C), connected_at


def test_oracle_replays_mujoco_recorded_cursor_demo():
    _check(*_replay_oracle(D["actions"]))


def test_oracle_carries_the_connected_parts_as_recorded():
    P, C, connected_at = _replay_oracle(D["actions_ext"])
    assert connected_at == 60
    _check_ext(P, C)


@pytest.mark.gpu
def test_device_replays_mujoco_recorded_cursor_demo():
    import torch
    from furniture_amd.sim import FSim, INFO_DIM, default_config
    m = load_compiled("Cursor", "swivel_chair_0700")
    cfg = default_config()
    cfg.max_episode_steps = 10000
    cfg.auto_reset = 0
    cfg.move_speed = 0.025
    n = 2
    sim = FSim(m, n, config=cfg)
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    o = FurnitureEnvOracle(m, OracleConfig(seed=123))
    o.reset()
    sim.set_reset_tables(np.tile(o.reset_draws["part_qpos"].reshape(1, -1), (n, 1)), None)
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    q = np.asarray(m.qpos0, dtype=np.float64).copy()
    for i in range(m.nparts):
        a0 = int(m.part_qposadr[i])
        q[a0:a0 + 7] = D["parts"][0, i]
    cur = np.concatenate([D["cursor0"][0], D["cursor1"][0], [0, 0]])
    sim.set_state(qpos=q[None], qvel=np.zeros((1, m.nv)), qacc_warmstart=np.zeros((1, m.nv)), xfrc_applied=np.zeros((1, 6 * m.nparts)),
                  cursor=cur[None])
    sim.physics_forward()
    act = torch.zeros((n, 15), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    P, C, connected_at = [], [], None
    for t, a in enumerate(D["actions_ext"]):
        act.copy_(torch.as_tensor(np.tile(a.astype(np.float32), (n, 1))))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        ob = obs[0].cpu().numpy().astype(np.float64)
        P.append(ob[:7 * m.nparts].reshape(m.nparts, 7))
        C.append(ob[7 * m.nparts:7 * m.nparts + 6])
        if int(info[0, 6]) and connected_at is None:
            connected_at = t
    _check(np.array(P)[:61], np.array(C)[:61], connected_at)
    _check_ext(np.array(P), np.array(C))  # frames 62-91: the welded column + seat carried about
    assert torch.equal(obs[0], obs[1])
    sim.close()
