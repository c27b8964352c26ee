"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs,
plus size-independent properties at BASELINE size (4096 envs).  Tolerances: the device integrates in fp32, the oracle
in fp64; connector/attach indices, collision masks and weld activity are integers and must be bit-exact."""
import numpy as np
import pytest
import torch

from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
from oracle.oracle_sim import OracleSim
from tests.scenarios import counter_actions, cursor_attach_state, pinch_attach_state

pytestmark = pytest.mark.gpu


def _initial(m, n, rng, arm_noise=0.3, vel=0.2):
    q = np.tile(m.qpos0, (n, 1))
    q[:, m.arm_qposadr] = m.arm_initqpos + rng.uniform(-arm_noise, arm_noise, (n, len(m.arm_qposadr)))
    q[:, m.grip_qposadr] = m.grip_initqpos
    for i in range(m.nparts):
        a = m.part_qposadr[i]
        q[:, a:a + 7] = m.part_initqpos[i]
        q[:, a:a + 2] += rng.uniform(-0.02, 0.02, (n, 2))
        q[:, a + 2] += 0.01
    return q, rng.uniform(-vel, vel, (n, m.nv))


@pytest.mark.parametrize("key", [("Sawyer", "table_lack_0825"), ("Sawyer", "swivel_chair_0700"), ("Baxter", "desk_mikael_1064")])
def test_forward_dynamics_match_oracle(key):
    """kinematics, RNE bias and accelerations of random states (free space + velocities)."""
    m = load_compiled(*key)
    n = 32
    q, v = _initial(m, n, np.random.RandomState(0), arm_noise=0.2)
    sim = FSim(m, n)
    sim.set_state(qpos=q, qvel=v)
    sim.physics_forward()
    st = {k: t.cpu().numpy() for k, t in sim.get_state("qacc", "xpos", "xquat", "qfrc_bias", "ncon").items()}
    orc = OracleSim(m)
    orc.set_solver(100, 1e-10, "newton")
    checked = 0
    for e in range(n):
        orc.reset()
        orc.data.qpos[:], orc.data.qvel[:] = q[e], v[e]
        orc.forward()
        assert np.abs(st["xpos"][e].reshape(-1, 3) - orc.data.xpos).max() < 2e-6
        assert np.abs(st["xquat"][e].reshape(-1, 4) - orc.data.xquat).max() < 2e-6
        assert np.abs(st["qfrc_bias"][e] - orc.data.qfrc_bias).max() < 2e-4 * (1 + np.abs(orc.data.qfrc_bias).max())
        if orc.ncon == 0 and st["ncon"][e, 0] <= 1:  # (the l0/base sphere-cylinder pair touches at exactly 0 distance)
            assert np.abs(st["qacc"][e] - orc.data.qacc).max() < 1e-5 * (1 + np.abs(orc.data.qacc).max())
            checked += 1
    assert checked >= n // 2
    sim.close()


def test_contact_trajectory_matches_oracle(sawyer_lack):
    """parts dropped on the floor, arm gravity-compensated: 400 substeps stay within 1e-5 of the fp64 oracle."""
    m = sawyer_lack
    n = 16
    q, _ = _initial(m, n, np.random.RandomState(1), arm_noise=0.0)
    sim = FSim(m, n)
    sim.set_state(qpos=q, qvel=np.zeros((n, m.nv)), qacc_warmstart=np.zeros((n, m.nv)))
    sim.physics_forward()
    bias = sim.get_state("qfrc_bias")["qfrc_bias"].cpu().numpy()
    rd = np.concatenate([m.arm_dofadr, m.grip_dofadr])
    app = np.zeros((n, m.nv))
    app[:, rd] = bias[:, rd]
    sim.set_state(qfrc_applied=app)
    sim.physics_step(400)
    st = sim.get_state("qpos", "qvel", "contact_geoms", "ncon")
    for e in range(3):
        o = OracleSim(m)
        o.set_solver(100, 1e-10, "newton")
        o.reset()
        o.data.qpos[:] = q[e]
        o.forward()
        o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]
        for _ in range(400):
            o.step()
        assert np.abs(st["qpos"][e].cpu().numpy() - o.data.qpos).max() < 1e-5
        assert np.abs(st["qvel"][e].cpu().numpy() - o.data.qvel).max() < 1e-4
        cg = st["contact_geoms"][e].cpu().numpy().reshape(-1, 2)
        gpu = sorted(tuple(int(x) for x in r) for r in cg if r[0] >= 0)
        floor = m.floor_geomid[0]
        assert [c for c in gpu if c[0] == floor] == sorted(c for c in o.contacts() if c[0] == floor)  # 5 parts x 4 corners
    sim.close()


def test_thousand_substeps_within_1e4_of_oracle(sawyer_lack):
    """north_star's bar: qpos/qvel within 1e-4 of the CPU path over 1000 (physics) steps on fixed seeds -- parts settling
    on the floor with the arm gravity-compensated and a constant gripper command (fp32 device vs fp64 oracle)."""
    m = sawyer_lack
    n = 4
    q, _ = _initial(m, n, np.random.RandomState(7), arm_noise=0.05)
    sim = FSim(m, n)
    sim.set_state(qpos=q, qvel=np.zeros((n, m.nv)), qacc_warmstart=np.zeros((n, m.nv)))
    sim.physics_forward()
    bias = sim.get_state("qfrc_bias")["qfrc_bias"].cpu().numpy()
    rd = np.concatenate([m.arm_dofadr, m.grip_dofadr])
    app = np.zeros((n, m.nv))
    app[:, rd] = bias[:, rd]
    sim.set_state(qfrc_applied=app)
    sim.physics_step(1000)
    st = sim.get_state("qpos", "qvel")
    for e in range(2):
        o = OracleSim(m)
        o.set_solver(100, 1e-10, "newton")
        o.reset()
        o.data.qpos[:] = q[e]
        o.forward()
        o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]
        for _ in range(1000):
            o.step()
        assert np.abs(st["qpos"][e].cpu().numpy() - o.data.qpos).max() < 1e-4
        assert np.abs(st["qvel"][e].cpu().numpy() - o.data.qvel).max() < 1e-4
    sim.close()


@pytest.mark.parametrize("fsim_mw", ["0", "1", "all"], indirect=True)
def test_env_reset_steps_and_attach_match_oracle(sawyer_lack, fsim_mw):
    """(every kernel path: one wave per env, the rule's k_env_step_x, four waves for every env)"""
    m = sawyer_lack
    n = 4
    cfg = default_config()
    cfg.max_episode_steps = 150
    cfg.auto_reset = 0
    sim = FSim(m, n, config=cfg)
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10)) for i in range(n)]
    obs_o = [e.reset() for e in envs]
    parts = np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs])
    noise = np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs])
    sim.set_reset_tables(parts, noise)
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    for e in range(n):
        assert np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(obs_o[e])).max() < 5e-5
    act = torch.zeros((n, 9), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    for t in range(5):
        a = np.stack([counter_actions(123, i, t, 9) for i in range(n)])
        act.copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        for e in range(n):
            ob, r, d, inf = envs[e].step(a[e])
            assert np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob)).max() < 2e-4
            assert abs(float(rew[e]) - r) < 1e-5 and bool(done[e]) == d
    # scripted pinch + connect: integer results bit-exact
    o = envs[0]
    q, xfrc, masks = pinch_attach_state(m, o.sim.data.qpos.copy(), o.sim.data.xpos.copy(), o.sim.data.xquat.copy())
    o.sim.data.qpos[:], o.sim.data.qvel[:], o.sim.data.qacc_warmstart[:] = q, 0, 0
    for i in range(m.nparts):
        o.sim.data.xfrc_applied[m.part_bodyid[i]] = xfrc.reshape(-1, 6)[i]
    gm = sim.get_state("geom_contype", "geom_conaffinity")
    for g, (ct, ca) in masks.items():
        o.sim.model.geom_contype[g], o.sim.model.geom_conaffinity[g] = ct, ca
        gm["geom_contype"][:, g], gm["geom_conaffinity"][:, g] = ct, ca
    sim.set_state(qpos=q[None], qvel=np.zeros((1, m.nv)), qacc_warmstart=np.zeros((1, m.nv)), xfrc_applied=xfrc[None],
                  geom_contype=gm["geom_contype"], geom_conaffinity=gm["geom_conaffinity"])
    a = np.zeros(9, dtype=np.float32)
    a[7] = a[8] = 1.0
    act.copy_(torch.as_tensor(np.tile(a, (n, 1))))
    torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info)
    sim.sync()
    ob, r, d, inf = o.step(a)
    gi = info[0].cpu().numpy()
    assert inf["num_connected"] == 1
    assert (gi[0], gi[3], gi[4], gi[6]) == (inf["num_connected"], inf["site1"], inf["site2"], inf["connected_this_step"])
    assert abs(float(rew[0]) - r) < 1e-4
    st = sim.get_state("eq_active", "eq_data", "geom_contype", "geom_conaffinity", "group")
    assert np.array_equal(st["eq_active"][0].cpu().numpy(), o.sim.model.eq_active)
    assert np.array_equal(st["geom_contype"][0].cpu().numpy(), o.sim.model.geom_contype)
    assert np.array_equal(st["geom_conaffinity"][0].cpu().numpy(), o.sim.model.geom_conaffinity)
    # union-find (furniture.py:2738-2759): ONE reading of the parent array -- the root every part resolves to (the same representative
    # on both sides: the merge direction is part of the reference's behaviour; path compression may differ and does not matter)
    g = [int(x) for x in st["group"][0].cpu().numpy()]

    def root(i):
        while g[i] != i:
            i = g[i]
        return i
    assert [root(i) for i in range(m.nparts)] == [o._find_group(i) for i in range(m.nparts)]
    assert np.abs(st["eq_data"][0].cpu().numpy().reshape(-1, 7) - o.sim.model.eq_data).max() < 1e-5
    # all n device envs saw the same state and action: identical integer outcomes
    assert torch.equal(info[:, [0, 3, 4, 6]], info[0:1, [0, 3, 4, 6]].expand(n, 4))
    sim.close()


@pytest.mark.parametrize("key", [("Sawyer", "swivel_chair_0700"), ("Baxter", "desk_mikael_1064")])
def test_env_reset_and_steps_match_oracle_other_models(key):
    """BASELINE configs 3 and 4 (Sawyer + swivel chair, bimanual Baxter + desk): in-kernel reset and random steps against
    the oracle env, same reset tables and actions."""
    m = load_compiled(*key)
    n = 2
    cfg = default_config()
    cfg.max_episode_steps = 150
    cfg.auto_reset = 0
    sim = FSim(m, n, config=cfg)
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10)) for i in range(n)]
    obs_o = [e.reset() for e in envs]
    parts = np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs])
    noise = np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs])
    sim.set_reset_tables(parts, noise)
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
            ob, r, d, inf = envs[e].step(a[e])
            assert np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob)).max() < 5e-4
            assert abs(float(rew[e]) - r) < 1e-4 and bool(done[e]) == d
    sim.close()


@pytest.mark.parametrize("fsim_mw", ["0", "all"], indirect=True)  # (fsim_physics_step: the one-wave and the four-wave physics kernel)
def test_welded_assembly_in_the_gripper_matches_oracle(sawyer_lack, fsim_mw):
    """All four welds active + the gripper pinching a leg: one 39-dof island (robot + 5 welded parts), i.e. the
    large-island Cholesky path and the weld/contact cross blocks of the Hessian, against the fp64 oracle."""
    from furniture_amd import transform_utils as T
    m = sawyer_lack
    o = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150))
    o.reset()
    q, xfrc, masks = pinch_attach_state(m, o.sim.data.qpos.copy(), o.sim.data.xpos.copy(), o.sim.data.xquat.copy())
    pq = lambda i: q[m.part_qposadr[i]:m.part_qposadr[i] + 7]
    eq_data = np.stack([T.rel_pose(pq(int(m.eq_part1[e])), pq(int(m.eq_part2[e]))) for e in range(len(m.eq_part1))])
    n = 4
    sim = FSim(m, n)
    gm = sim.get_state("geom_contype", "geom_conaffinity")
    osim = OracleSim(m)
    osim.set_solver(100, 1e-10, "newton")
    osim.reset()
    for g, (ct, ca) in masks.items():
        osim.model.geom_contype[g], osim.model.geom_conaffinity[g] = ct, ca
        gm["geom_contype"][:, g], gm["geom_conaffinity"][:, g] = ct, ca
    osim.data.qpos[:], osim.data.qvel[:], osim.data.qacc_warmstart[:] = q, 0, 0
    osim.model.eq_active[:] = 1
    osim.model.eq_data[:] = eq_data
    for i in range(m.nparts):
        osim.data.xfrc_applied[m.part_bodyid[i]] = xfrc.reshape(-1, 6)[i]
    sim.set_state(qpos=np.tile(q, (n, 1)), qvel=np.zeros((n, m.nv)), qacc_warmstart=np.zeros((n, m.nv)),
                  xfrc_applied=np.tile(xfrc, (n, 1)), geom_contype=gm["geom_contype"], geom_conaffinity=gm["geom_conaffinity"],
                  eq_active=np.ones((n, len(m.eq_part1)), dtype=np.int32), eq_data=np.tile(eq_data.reshape(-1), (n, 1)))
    # close the fingers on the leg with the arm gravity-compensated
    sim.physics_forward()
    osim.forward()
    bias = sim.get_state("qfrc_bias")["qfrc_bias"].cpu().numpy()
    app = np.zeros((n, m.nv))
    app[:, m.arm_dofadr] = bias[:, m.arm_dofadr]
    ctrl = np.zeros(m.nu)
    ctrl[-2:] = (m.ctrl_bias + m.ctrl_weight)[-2:]  # gripper actuators: action +1 = close
    sim.set_state(qfrc_applied=app, ctrl=np.tile(ctrl, (n, 1)))
    osim.data.qfrc_applied[m.arm_dofadr] = osim.data.qfrc_bias[m.arm_dofadr]
    osim.data.ctrl[:] = ctrl
    sim.physics_step(60)
    for _ in range(60):
        osim.step()
    st = sim.get_state("qpos", "qvel", "ncon")
    assert int(st["ncon"][0, 0]) >= 2  # the pads do hold the leg
    assert np.abs(st["qpos"][0].cpu().numpy() - osim.data.qpos).max() < 2e-4
    assert np.abs(st["qvel"][0].cpu().numpy() - osim.data.qvel).max() < 5e-3
    assert torch.equal(st["qpos"][0], st["qpos"][n - 1])
    sim.close()


@pytest.mark.parametrize("after_attach", [False, True])
def test_cursor_agent_matches_oracle(after_attach):
    """FurnitureCursorEnv on the device: reset, random 15-dof steps (cursor moves, selection by contact), then
    a scripted attach -- both cursors hold an aligned leg / table pair and ask to connect: ten approach steps
    (slerp / lerp of the held group) and the connect itself, all against the oracle env.  after_attach: with
    config.reset_robot_after_attach the connect sends both cursors back to their start positions (furniture.py:919-925, 1763-1768)."""
    m = load_compiled("Cursor", "table_lack_0825")
    n = 2
    cfg = default_config()
    cfg.max_episode_steps = 150
    cfg.auto_reset = 0
    cfg.reset_robot_after_attach = 1 if after_attach else 0
    sim = FSim(m, n, config=cfg)
    assert sim.dof_action == 15 and sim.obs_dim == 7 * m.nparts + 8
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10, reset_robot_after_attach=after_attach)) for i in range(n)]
    obs_o = [e.reset() for e in envs]
    parts = np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs])
    sim.set_reset_tables(parts, None)
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    for e in range(n):
        assert np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(obs_o[e])).max() < 1e-4
    act = torch.zeros((n, 15), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)

    def step_both(a, tol):
        act.copy_(torch.as_tensor(np.asarray(a, dtype=np.float32)))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        out = []
        for e in range(n):
            ob, r, d, inf = envs[e].step(np.asarray(a[e], dtype=np.float64))
            assert np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob)).max() < tol, (e, np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(ob)).max())
            assert abs(float(rew[e]) - r) < 1e-4 and bool(done[e]) == d
            out.append(inf)
        return out

    for t in range(6):  # cursors wander; select flags are random, selection needs a contact
        step_both(np.stack([counter_actions(77, i, t, 15) for i in range(n)]), 5e-4)
    # scripted attach
    cur = sim.get_state("cursor")["cursor"].cpu().numpy()
    qs = []
    for e in range(n):
        o = envs[e]
        q, tpart = cursor_attach_state(m, o.sim.data.qpos.copy())
        o.sim.data.qpos[:], o.sim.data.qvel[:], o.sim.data.qacc_warmstart[:] = q, 0, 0
        o._cursor_selected = [0, tpart]
        o._connect_step = 0
        o.sim.forward()
        qs.append(q)
        cur[e, 6], cur[e, 7] = 0 + 1, tpart + 1
    sim.set_state(qpos=np.stack(qs), qvel=np.zeros((n, m.nv)), qacc_warmstart=np.zeros((n, m.nv)), cursor=cur)
    sim.physics_forward()
    a = np.zeros((n, 15), dtype=np.float32)
    a[:, 6] = a[:, 13] = 1.0   # keep both selections
    a[:, 14] = 1.0             # connect
    connected_at = None
    for t in range(13):
        infs = step_both(a, 2e-3)
        gi = info.cpu().numpy()
        for e in range(n):
            assert (gi[e, 0], gi[e, 6]) == (infs[e]["num_connected"], infs[e]["connected_this_step"])
        if infs[0]["connected_this_step"] and connected_at is None:
            connected_at = t
    assert connected_at == 10  # ten approach steps (_num_connect_steps = 10), then _connect
    st = sim.get_state("eq_active", "eq_data", "cursor")
    assert np.array_equal(st["eq_active"][0].cpu().numpy(), envs[0].sim.model.eq_active)
    assert np.abs(st["eq_data"][0].cpu().numpy().reshape(-1, 7) - envs[0].sim.model.eq_data).max() < 2e-3
    assert st["cursor"][0, 7].item() == 0  # _connect drops cursor 1's selection (furniture.py:914-915)
    if after_attach:  # (at the connect step; the two steps after it moved them by nothing: the action's move entries are zero)
        assert np.abs(st["cursor"][0, :6].cpu().numpy() - np.array([-0.2, 0.0, 0.05, 0.2, 0.0, 0.05])).max() < 1e-6
        assert np.abs(np.asarray(envs[0].sim.model.body_pos)[m.cursor_bodyid].reshape(-1) - np.array([-0.2, 0.0, 0.05, 0.2, 0.0, 0.05])).max() < 1e-12
    sim.close()


def test_determinism_and_batch_invariance(sawyer_lack):
    """Same inputs -> bit-identical state, run to run and independent of batch size / env position in the batch."""
    m = sawyer_lack
    q, v = _initial(m, 64, np.random.RandomState(3), arm_noise=0.2)

    def run(idx):
        sim = FSim(m, len(idx))
        sim.set_state(qpos=q[idx], qvel=v[idx])
        sim.physics_step(100)
        out = sim.get_state("qpos", "qvel")
        sim.close()
        return out["qpos"].cpu().numpy(), out["qvel"].cpu().numpy()

    a = run(np.arange(64))
    b = run(np.arange(64))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    c = run(np.arange(17, 30))
    assert np.array_equal(a[0][17:30], c[0]) and np.array_equal(a[1][17:30], c[1])


def test_baseline_size_properties(sawyer_lack):
    """BASELINE config 2 size (4096 envs): random-action steps stay finite, episodes terminate at the time limit with
    auto-reset, parts rest on the floor, state round-trips through get/set."""
    m = sawyer_lack
    n = 4096
    cfg = default_config()
    cfg.max_episode_steps = 3
    sim = FSim(m, n, config=cfg)
    from furniture_amd.envs import ResetTableSampler
    from types import SimpleNamespace
    parts, noise = ResetTableSampler(m, SimpleNamespace(furn_xyz_rand=0.02, furn_rot_rand=3, agent_xyz_rand=0.001), 123, 0, 64).draw()
    sim.set_reset_tables(np.tile(parts, (n // 64, 1)), np.tile(noise, (n // 64, 1)))
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    act = torch.empty((n, 9), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    dones = []
    for t in range(4):
        act.uniform_(-1, 1, generator=g)
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert int((info[:, 12] != 0).sum()) <= n // 100  # contact-slot / survivor-list overflow is reported and rare
        dones.append(int(done.sum()))
    assert dones[:2] == [0, 0] and dones[2] == n and dones[3] == 0  # equality time limit (Q9) + auto-reset
    assert int(info[:, 5].max()) == 1  # episode_length restarted
    z = obs[:, 2:35:7]  # part heights from object_ob
    assert float(z.min()) > 0.005 and float(z.max()) < 0.05
    # state round trip
    st = sim.get_state("qpos", "qvel", "eq_active", "geom_contype")
    sim.set_state(qpos=st["qpos"], qvel=st["qvel"], eq_active=st["eq_active"], geom_contype=st["geom_contype"])
    st2 = sim.get_state("qpos", "qvel", "eq_active", "geom_contype")
    for k in st:
        assert torch.equal(st[k], st2[k])
    sim.close()


def test_batched_env_surface():
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    env = FurnitureBatchEnv("Sawyer", 8, config=make_config(unity=False, record_vid=False, control_type="impedance",
                                                            furniture_name="table_lack_0825", max_episode_steps=150))
    ob = env.reset()
    assert ob["object_ob"].shape == (8, 35) and ob["robot_ob"].shape == (8, 29)
    assert env.action_space.spaces["default"].shape == (9,) and env.dof == 9
    ob, rew, done, info = env.step(np.zeros((8, 9), dtype=np.float32))
    assert rew.shape == (8,) and done.dtype == torch.bool and int(info["episode_length"][0]) == 1
    assert torch.allclose(rew, torch.zeros_like(rew))  # zero action: no ctrl penalty, nothing touched
    env.close()


def test_single_env_classes_incl_cursor():
    """The reference-shaped single-env classes (gym surface B1): Sawyer and the Cursor agent (BASELINE config 1's env)."""
    from furniture_amd.envs import FurnitureCursorEnv, FurnitureSawyerEnv, make_config
    env = FurnitureCursorEnv(make_config(unity=False, record_vid=False, furniture_name="toy_table", max_episode_steps=20))
    ob = env.reset()
    assert ob["object_ob"].shape == (35,) and ob["robot_ob"].shape == (8,) and env.dof == 15
    assert np.allclose(ob["robot_ob"][:6], [-0.2, 0, 0.05, 0.2, 0, 0.05], atol=1e-6)
    rng = np.random.RandomState(123)
    for _ in range(5):
        ob, rew, done, info = env.step(rng.uniform(-1, 1, 15).astype(np.float32))
        assert rew in (0.0, 100.0) and np.isfinite(ob["object_ob"]).all()
        assert np.all(np.abs(ob["robot_ob"][:6]) < 1.5) and ob["robot_ob"][2] >= 0.045
    env.close()
    env = FurnitureSawyerEnv(make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825",
                                         max_episode_steps=20))
    ob = env.reset()
    ob, rew, done, info = env.step(np.zeros(9, dtype=np.float32))
    assert ob["robot_ob"].shape == (29,) and not done
    env.close()


def test_thousand_env_steps_within_1e4_of_oracle(sawyer_lack):
    """north_star's bar taken literally: 1000 env steps (50 000 physics substeps) on a fixed seed -- the whole env (reset,
    _setup_action with its stale gravity compensation, 50 substeps per step, observation) under a smooth joint-velocity command
    that keeps the arm clear of the parts, which rest on the floor in contact the whole time.  fp32 device vs fp64 oracle:
    observation (part poses, joint positions / velocities, end-effector pose and velocity) within 1e-4 at every 50th step and
    at the end.  (With contact-rich random actions two correct integrators decorrelate long before that: covered by the
    400-substep contact trajectories and the scripted attach above.)"""
    m = sawyer_lack
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset = 5000, 0
    sim = FSim(m, 1, config=cfg)
    env = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=5000, seed=123, solver_tolerance=1e-10))
    ob_o = env.reset()
    sim.set_reset_tables(env.reset_draws["part_qpos"].reshape(1, -1), np.stack(env.reset_draws["noise"]).reshape(1, -1))
    dev = sim.device
    obs = torch.zeros((1, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    assert np.abs(obs[0].cpu().numpy() - env.flat_obs(ob_o)).max() < 1e-4
    act = torch.zeros((1, 9), device=dev)
    rew = torch.zeros(1, device=dev)
    done = torch.zeros(1, dtype=torch.uint8, device=dev)
    info = torch.zeros((1, INFO_DIM), dtype=torch.int32, device=dev)
    phase = np.arange(7) * 0.9
    worst = 0.0
    for t in range(1000):
        a = np.zeros(9, dtype=np.float32)
        a[:7] = 0.05 * np.sin(0.2 * t + phase)  # joint position amplitude ~0.04 rad: no robot contact, no joint limit
        a[7], a[8] = -1.0, -1.0
        act.copy_(torch.as_tensor(a[None]))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        ob, r, d, _ = env.step(a.astype(np.float64))
        if t % 50 == 49:
            sim.sync()
            dvec = np.abs(obs[0].cpu().numpy() - env.flat_obs(ob))
            err = float(dvec.max())
            worst = max(worst, err)
            assert err < 1e-4, (t, err, int(dvec.argmax()))
            assert all(int(m.geom_bodyid[g1]) == 0 for g1, _ in env.sim.contacts())  # only part-floor contacts: the arm stays clear
    sim.sync()
    assert not bool(done[0]) and not d
    print("1000 env steps: worst observation error %.2e" % worst)
    sim.close()


def test_second_reset_continues_the_reference_rng_stream(sawyer_lack):
    """Every reset() consumes exactly one pass of the env's reset-time RNG stream (seed + i, furniture/env/base.py:77): a table
    uploaded ahead for an auto-reset that has not happened yet is the NEXT draw and must not be dropped.  Single-env class
    (auto_reset off) and batched env (auto_reset on), three resets each, against the oracle env's consecutive resets."""
    from furniture_amd.envs import FurnitureBatchEnv, FurnitureSawyerEnv, make_config
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=50, seed=321)
    orc = FurnitureEnvOracle(sawyer_lack, OracleConfig(max_episode_steps=50, seed=321, solver_tolerance=1e-10))
    want = [orc.flat_obs(orc.reset())[:35] for _ in range(3)]
    assert np.abs(want[0] - want[1]).max() > 1e-3  # the placements differ from draw to draw
    env = FurnitureSawyerEnv(make_config(**kw))
    for k in range(3):
        ob = env.reset()
        assert np.abs(ob["object_ob"] - want[k]).max() < 2e-4, k
        env.step(np.zeros(9, dtype=np.float32))
    env.close()
    benv = FurnitureBatchEnv("Sawyer", 2, config=make_config(**kw))
    for k in range(3):
        ob = benv.reset()
        assert np.abs(ob["object_ob"][0].cpu().numpy() - want[k]).max() < 2e-4, k
        benv.step(np.zeros((2, 9), dtype=np.float32))
    benv.close()


@pytest.mark.parametrize("auto_reset", [0, 1])
def test_unstable_simulation_fails_the_step_and_resets(sawyer_lack, auto_reset):
    """_do_simulation's exception path (furniture.py:2889-2897) + _after_step (furniture.py:463-467): a state that blows the solver
    up (huge velocity) => fail flag, -unstable_penalty_coef reward, terminal step, and an observation of a freshly reset episode.
    With auto_reset the in-step reset is folded into the worker's reset and the host is told to drop one RNG draw (needs_table = 2)."""
    from furniture_amd.sim import INFO_EPISODE_LENGTH, INFO_FAIL, INFO_NEEDS_TABLE
    m = sawyer_lack
    n = 4
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset = 100, auto_reset
    sim = FSim(m, n, config=cfg)
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=100, seed=50 + i, solver_tolerance=1e-10)) for i in range(n)]
    for e in envs:
        e.reset()
    tabs = (np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]), np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    sim.set_reset_tables(*tabs)
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    sim.reset(None, obs)
    sim.sync()
    obs0 = obs.clone()
    sim.set_reset_tables(*tabs)  # the same placement again: the post-failure observation is then known
    st = sim.get_state("qvel")
    qv = st["qvel"].clone()
    qv[1] = float("nan")            # env 1: NaN velocity
    qv[2, m.part_dofadr[0]] = 1e12  # env 2: a part at 1e12 m/s
    sim.set_state(qvel=qv)
    act = torch.zeros((n, 9), device=dev)
    sim.step(act, obs, rew, done, info)
    sim.sync()
    inf = info.cpu().numpy()
    assert list(inf[:, INFO_FAIL]) == [0, 1, 1, 0]
    assert list(done.cpu().numpy()) == [0, 1, 1, 0]
    assert np.allclose(rew.cpu().numpy(), [0, -100, -100, 0], atol=1e-6)
    assert torch.isfinite(obs).all()
    for e in (1, 2):  # observation of a fresh episode from the same table = the first reset's observation ...
        d = np.abs(obs[e].cpu().numpy() - obs0[e].cpu().numpy())
        assert d[:35].max() < 2e-4
        # ... except that the reference's in-step reset is followed by one more forward pass before _get_obs (the step goes on),
        # so the end-effector pose is one integration step newer than in reset()'s own observation
        assert d[35:].max() < (2e-4 if auto_reset else 5e-3)
    assert list(inf[:, INFO_NEEDS_TABLE]) == ([0, 2, 2, 0] if auto_reset else [0, 0, 0, 0])
    assert list(inf[:, INFO_EPISODE_LENGTH]) == [1, 1, 1, 1]
    # the failed envs carry on: next step is an ordinary first step of the new episode
    sim.step(act, obs, rew, done, info)
    sim.sync()
    assert torch.isfinite(obs).all() and list(done.cpu().numpy()) == [0, 0, 0, 0] and list(info[:, INFO_FAIL].cpu().numpy()) == [0, 0, 0, 0]
    sim.close()


def test_set_init_qpos_resets_from_the_given_state(sawyer_lack):
    """set_init_qpos (furniture.py:315-316, used inside _reset :1505-1519, 1568, 1617): the reset starts from a given {qpos, qvel}
    instead of a sampled placement and consumes no RNG draw; set_init_qpos(None) goes back to the sampled stream where it was."""
    from furniture_amd.envs import FurnitureSawyerEnv, make_config
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=50, seed=77)
    orc = FurnitureEnvOracle(sawyer_lack, OracleConfig(max_episode_steps=50, seed=77, solver_tolerance=1e-10))
    env = FurnitureSawyerEnv(make_config(**kw))
    o0 = orc.flat_obs(orc.reset())
    d0 = env.reset()
    assert np.abs(np.concatenate([d0["object_ob"], d0["robot_ob"]]) - o0).max() < 2e-4
    # a state to start from: the oracle's post-reset state with two parts moved and the arm bent
    init = dict(qpos=orc.sim.data.qpos.copy(), qvel=np.zeros(sawyer_lack.nv))
    init["qpos"][sawyer_lack.part_qposadr[1]:sawyer_lack.part_qposadr[1] + 2] += [0.07, -0.05]
    init["qpos"][sawyer_lack.part_qposadr[3] + 2] += 0.05
    init["qpos"][sawyer_lack.arm_qposadr[1]] -= 0.2
    orc.set_init_qpos(init)
    env.set_init_qpos(init)
    for _ in range(2):  # the same state every time
        o1 = orc.flat_obs(orc.reset())
        d1 = env.reset()
        assert np.abs(np.concatenate([d1["object_ob"], d1["robot_ob"]]) - o1).max() < 2e-4
    assert np.abs(o1[:35] - o0[:35]).max() > 0.03
    orc.set_init_qpos(None)
    env.set_init_qpos(None)
    o2 = orc.flat_obs(orc.reset())  # the second draw of the stream (the init-state resets took none)
    d2 = env.reset()
    assert np.abs(np.concatenate([d2["object_ob"], d2["robot_ob"]]) - o2).max() < 2e-4
    # the reference's policy sequencing alternates set_init_qpos and set_subtask on one env: once the init state is cleared,
    # pre-assembled starts must be accepted again (they used to be refused for the life of the handle)
    env.set_subtask(1)
    orc.set_subtask(1)
    o3 = orc.flat_obs(orc.reset())
    d3 = env.reset()
    assert np.abs(np.concatenate([d3["object_ob"], d3["robot_ob"]])[:35:7] - o3[:35:7]).max() < 5e-3
    env.close()


def test_reset_with_another_furniture_id_swaps_the_model_and_keeps_the_rng_stream(sawyer_lack):
    """reset(furniture_id) (furniture.py:318-334): the env rebuilds for the other furniture; the reset-time RNG stream is the
    env's one self._rng and carries on across the switch."""
    from furniture_amd.envs import FurnitureSawyerEnv, furniture_names, make_config
    names = furniture_names()
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=50, seed=9)
    env = FurnitureSawyerEnv(make_config(**kw))
    orc = FurnitureEnvOracle(sawyer_lack, OracleConfig(max_episode_steps=50, seed=9, solver_tolerance=1e-10))
    o = orc.flat_obs(orc.reset())
    d = env.reset()
    assert np.abs(np.concatenate([d["object_ob"], d["robot_ob"]]) - o).max() < 2e-4
    m2 = load_compiled("Sawyer", "swivel_chair_0700")
    orc2 = FurnitureEnvOracle(m2, OracleConfig(max_episode_steps=50, seed=9, solver_tolerance=1e-10))
    orc2._rng = orc._rng  # one stream
    o2 = orc2.flat_obs(orc2.reset())
    d2 = env.reset(furniture_id=names.index("swivel_chair_0700"))
    assert d2["object_ob"].shape == (7 * m2.nparts,)
    assert np.abs(np.concatenate([d2["object_ob"], d2["robot_ob"]]) - o2).max() < 2e-4
    ob, rew, done, info = env.step(np.zeros(9, dtype=np.float32))
    assert np.isfinite(ob["robot_ob"]).all() and "touch_reward" in info and info["ctrl_penalty"] == 0.0
    env.close()


def test_preassembled_starts_match_the_oracle_env(sawyer_lack):
    """config.preassembled / set_subtask / num_connects (furniture.py:163, 204-207, 1476-1503, 1542-1566).  With a recipe the reset
    connects the listed recipe steps (_project_connector_quat + _connect(site2, site1)) between its two settling phases; without one
    it switches the listed welds on before the parts are placed.  Device vs oracle env: observation, weld activity, groups,
    counters, the first step's reward (which pays the pre-assembled connects: _prev_num_connected starts at 0)."""
    from furniture_amd.envs import FurnitureSawyerEnv, make_config

    def err(d, o, nparts):
        """observation error with each part's quaternion compared up to its sign: the recipe's 90 / 270 degree targets put
        lookat_to_quat exactly on its m00 == m11 branch tie (x = -y), which fp32 and fp64 rounding break differently -- q or -q,
        the same rotation (the reference's own sign there is decided by fp64 rounding noise)"""
        x = np.concatenate([d["object_ob"], d["robot_ob"]]).copy()
        for i in range(nparts):
            if np.dot(x[7 * i + 3:7 * i + 7], o[7 * i + 3:7 * i + 7]) < 0:
                x[7 * i + 3:7 * i + 7] *= -1
        return np.abs(x - o).max()

    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=50, seed=31)
    env = FurnitureSawyerEnv(make_config(preassembled=[0, 1], **kw))
    orc = FurnitureEnvOracle(sawyer_lack, OracleConfig(max_episode_steps=50, seed=31, solver_tolerance=1e-10, preassembled=[0, 1]))
    o = orc.flat_obs(orc.reset())
    d = env.reset()
    assert err(d, o, 5) < 5e-4
    st = env._b.sim.get_state("eq_active", "env_block")
    assert st["eq_active"][0].cpu().numpy().astype(int).tolist() == np.asarray(orc.sim.model.eq_active).astype(int).tolist()
    rng = np.random.RandomState(5)
    for t in range(3):
        a = rng.uniform(-1, 1, 9)
        ob, r, done, info = env.step(a)
        ob_o, r_o, done_o, _ = orc.step(a)
        assert err(ob, orc.flat_obs(ob_o), 5) < 1e-3, t
        assert abs(r - r_o) < 1e-3 and done == done_o, (t, r, r_o)
        assert info["num_connected"] == orc._num_connected == 2
        if t == 0:
            assert r > 190  # 2 x success_reward
    # set_subtask(3, num_connects=1): three legs on, success after one more connect
    env.set_subtask(3, num_connects=1)
    orc.set_subtask(3, num_connects=1)
    o = orc.flat_obs(orc.reset())
    d = env.reset()
    assert err(d, o, 5) < 5e-4
    assert env.num_subtask() == 1
    env.close()
    # a furniture without a recipe file: the list holds weld ids
    m2 = load_compiled("Sawyer", "swivel_chair_0700")
    kw["furniture_name"] = "swivel_chair_0700"
    env = FurnitureSawyerEnv(make_config(preassembled=[0], **kw))
    orc = FurnitureEnvOracle(m2, OracleConfig(max_episode_steps=50, seed=31, solver_tolerance=1e-10, preassembled=[0]))
    o = orc.flat_obs(orc.reset())
    d = env.reset()
    assert err(d, o, m2.nparts) < 5e-3  # (the active weld yanks the two parts together across the floor during the reset: measured 1.0e-3)
    st = env._b.sim.get_state("eq_active")
    assert st["eq_active"][0].cpu().numpy().astype(int).tolist() == np.asarray(orc.sim.model.eq_active).astype(int).tolist() == [1, 0]
    env.close()


def test_furniture_gym_wrapper_takes_the_reference_kwargs():
    """furniture/env/furniture_gym.py:16-46: FurnitureGym(id=, name=, **overrides) -> parser defaults, overrides, make_env"""
    import sys
    from furniture_amd.config import FurnitureGym
    argv, sys.argv = sys.argv, ["prog"]
    try:
        env = FurnitureGym(id="IKEASawyer-v0", name="FurnitureSawyerEnv", unity=False, record_vid=False, max_episode_steps=4, control_type="impedance")
    finally:
        sys.argv = argv
    assert env._max_episode_steps == 4 and env.num_subtask() == 2  # swivel_chair_0700: the id's furniture
    ob = env.reset()
    assert ob["object_ob"].shape == (21,)
    for t in range(4):
        ob, r, d, info = env.step(np.zeros(9))
    assert d and np.isfinite(ob["robot_ob"]).all()
    env.close()


def test_assembled_and_fix_init_match_the_oracle_env(sawyer_lack):
    """config.assembled (every weld on from the start, one group, the parts start from the XML layout and are pulled together) and
    config.fix_init (the first placement is kept; later resets take no placement draw), furniture.py:1502-1503, 1518-1530"""
    from furniture_amd.envs import FurnitureSawyerEnv, make_config
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=50, seed=21)
    cat = lambda d: np.concatenate([d["object_ob"], d["robot_ob"]])
    env = FurnitureSawyerEnv(make_config(fix_init=True, **kw))
    orc = FurnitureEnvOracle(sawyer_lack, OracleConfig(max_episode_steps=50, seed=21, solver_tolerance=1e-10, fix_init=True))
    first = None
    for rep in range(3):
        o = orc.flat_obs(orc.reset())
        d = cat(env.reset())
        assert np.abs(d - o).max() < 2e-4, rep
        if first is None:
            first = d[:35].copy()
        else:
            assert np.abs(d[:35] - first).max() < 1e-4  # the same placement every time (the robot's joint noise differs)
    env.close()
    env = FurnitureSawyerEnv(make_config(assembled=True, **kw))
    orc = FurnitureEnvOracle(sawyer_lack, OracleConfig(max_episode_steps=50, seed=21, solver_tolerance=1e-10, assembled=True))
    o = orc.flat_obs(orc.reset())
    d = cat(env.reset())
    st = env._b.sim.get_state("eq_active")
    assert st["eq_active"][0].cpu().numpy().astype(int).tolist() == [1] * sawyer_lack.neq
    # five parts yanked together over ~1 m by four welds inside the reset -- a violent transient in which the two integrators part
    # company; what both must reach is the assembled configuration: every weld's relative pose as the model prescribes it
    from furniture_amd import transform_utils as T
    for dev_or_orc in (d, o):
        parts = dev_or_orc[:35].reshape(5, 7)
        assert np.isfinite(dev_or_orc).all()
        for e in range(sawyer_lack.neq):
            rel = T.rel_pose(parts[int(sawyer_lack.eq_part1[e])], parts[int(sawyer_lack.eq_part2[e])])
            assert np.abs(rel[:3] - sawyer_lack.eq_data0[e][:3]).max() < 2e-2, (e, rel[:3], sawyer_lack.eq_data0[e][:3])
    ob, r, done, info = env.step(np.zeros(9))
    assert info["num_connected"] == 0 and not done and np.isfinite(cat(ob)).all()
    env.close()


@pytest.mark.parametrize("fsim_mw", ["0", "1", "all"], indirect=True)
def test_specialised_and_generic_kernels_agree(sawyer_lack, fsim_mw):
    """The benchmark model runs on kernels whose layout offsets and sizes are compile-time constants (fsim_spec.hpp); FSIM_GENERIC=1
    forces the run-time-layout kernels every other model uses.  Same templates, same algorithm; not bit-identical (the compiler
    contracts a * b + c into an FMA where the expression shape allows it, and that shape differs once offsets and sizes are
    constants), so the two are compared like two correct fp32 implementations: 1e-5 on the observation over a reset (400 substeps)
    and 2e-4 over the 300 substeps of random actions that follow; done flags and integer state exact."""
    import os
    import torch
    from furniture_amd.envs import make_vec_env

    def run(generic):
        before = os.environ.get("FSIM_GENERIC")  # (restored afterwards: the suite's FSIM_GENERIC=1 variant run must stay generic for the tests that follow)
        if generic:
            os.environ["FSIM_GENERIC"] = "1"
        try:
            env = make_vec_env("Sawyer", 96, furniture_name="table_lack_0825", max_episode_steps=4, seed=5, record_vid=False, unity=False, control_type="impedance")
        finally:
            if before is None:
                os.environ.pop("FSIM_GENERIC", None)
            else:
                os.environ["FSIM_GENERIC"] = before
        assert env.sim.kernel_variant == ("generic" if generic else "sawyer_table_lack_0825")
        out = [env.reset()]
        g = torch.Generator(device=env.sim.device)
        g.manual_seed(3)
        rews = []
        for t in range(6):
            ob, rew, done, info = env.step(torch.empty((96, 9), device=env.sim.device).uniform_(-1, 1, generator=g))
            out.append({k: v.clone() for k, v in ob.items()})
            rews.append((rew.clone(), done.clone()))
        st = {k: v.clone() for k, v in env.sim.get_state("qpos", "qvel", "qacc_warmstart", "eq_active", "geom_contype").items()}
        env.close()
        return out, rews, st

    a, ra, sa = run(False)
    b, rb, sb = run(True)
    for t, (x, y) in enumerate(zip(a, b)):
        for k in x:
            err = float((x[k] - y[k]).abs().max())
            assert err < (1e-5 if t == 0 else 2e-4), (t, k, err)
    for (r1, d1), (r2, d2) in zip(ra, rb):
        assert float((r1 - r2).abs().max()) < 1e-5 and torch.equal(d1, d2)
    for k in ("eq_active", "geom_contype"):
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("slots,variant", [(128, "generic2"), (512, "generic8")])
def test_the_kernels_with_several_slot_sets_agree_with_the_specialised_one(slots, variant, monkeypatch):
    """The kernels that carry two / eight contact-slot sets per lane in the Newton solve (`generic2`: furniture with ten parts and more; `generic8`: the
    re-step ladder's last rung) as the BASE kernel of the benchmark model (FSIM_NCON_MAX), where the specialised 48-slot kernel is the yardstick: the
    further sets are empty here, every loop over them runs, and the result must be the specialised kernel's -- compared like two correct fp32
    implementations (test_specialised_and_generic_kernels_agree): 1e-5 over the reset, 2e-4 over 300 substeps of random actions; integer state exact."""
    import torch
    from furniture_amd.envs import make_vec_env

    def run(n_slots):
        if n_slots:
            monkeypatch.setenv("FSIM_NCON_MAX", str(n_slots))
        try:
            env = make_vec_env("Sawyer", 32, furniture_name="table_lack_0825", max_episode_steps=4, seed=5, record_vid=False, unity=False, control_type="impedance")
        finally:
            monkeypatch.delenv("FSIM_NCON_MAX", raising=False)
        assert env.sim.kernel_variant == (variant if n_slots else "sawyer_table_lack_0825") and env.sim.max_contacts == (n_slots or 48)
        out = [env.reset()]
        g = torch.Generator(device=env.sim.device)
        g.manual_seed(3)
        rews = []
        for t in range(6):
            ob, rew, done, info = env.step(torch.empty((32, 9), device=env.sim.device).uniform_(-1, 1, generator=g))
            out.append({k: v.clone() for k, v in ob.items()})
            rews.append((rew.clone(), done.clone()))
        st = {k: v.clone() for k, v in env.sim.get_state("eq_active", "geom_contype").items()}
        env.close()
        return out, rews, st

    a, ra, sa = run(0)
    b, rb, sb = run(slots)
    for t, (x, y) in enumerate(zip(a, b)):
        for k in x:
            err = float((x[k] - y[k]).abs().max())
            assert err < (1e-5 if t == 0 else 2e-4), (t, k, err)
    for (r1, d1), (r2, d2) in zip(ra, rb):
        assert float((r1 - r2).abs().max()) < 1e-5 and torch.equal(d1, d2)
    for k in ("eq_active", "geom_contype"):
        assert torch.equal(sa[k], sb[k]), k
