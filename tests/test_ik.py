"""control_type "ik" with a batched damped-least-squares solver in place of pybullet (PARITY UNPINNED for the solver:
oracle/ik.py header).  CPU: the URDF chain the reference's IK runs on against the MJCF kinematics, solver properties, the env
flow on the oracle.  GPU: the device IK stage + 3 x 50 substeps against the oracle env."""
import numpy as np
import pytest
import torch

from furniture_amd import transform_utils as T
from furniture_amd.mjcf.model import load_compiled
from oracle import ik as IK
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
from oracle.oracle_sim import OracleSim


@pytest.fixture(scope="module")
def model():
    return load_compiled("Sawyer", "table_lack_0825")


def test_urdf_chain_equals_mjcf_kinematics(model):
    """pybullet's sawyer_arm.urdf (link 6 = right_l6) and the MJCF arm are the same chain: the reference relies on it when it turns
    the commanded right_hand orientation by Rz(-90 deg) into a link-6 target (sawyer_ik_controller.py:248-258; MJCF right_hand =
    right_l6 . Trans(0, 0, 0.0245) . Rz(+90 deg), robots/sawyer/robot.xml:113-119)."""
    m = model
    sim = OracleSim(m)
    rng = np.random.RandomState(0)
    Rb, pb, hb = IK.q2m(m.ik_base_quat), m.ik_base_pos, int(m.hand_bodyid[0])
    for _ in range(5):
        q = m.arm_initqpos + rng.uniform(-0.7, 0.7, 7)
        sim.reset()
        sim.data.qpos[m.arm_qposadr] = q
        sim.forward()
        p, R, _, _ = IK.fk(m, q)
        hand_R = Rb.T @ sim.data.xmat[hb].reshape(3, 3)
        l6_R = hand_R @ IK.rot_z(-np.pi / 2)
        l6_p = Rb.T @ (sim.data.xpos[hb] - pb) - l6_R @ np.array([0, 0, 0.0245])
        assert np.abs(l6_p + l6_R @ m.ik_eef_pos[0] - p).max() < 1e-6   # end effector = CoM frame of right_l6 (getLinkState()[0])
        assert np.abs(l6_R - R).max() < 1e-5                          # (the MJCF quaternion of right_hand is rounded to 6 digits)


def test_solver_reaches_the_target_and_respects_limits(model):
    m = model
    rng = np.random.RandomState(1)
    for _ in range(20):
        q0 = m.arm_initqpos + rng.uniform(-0.3, 0.3, 7)
        p0, R0, _, _ = IK.fk(m, q0)
        tp = p0 + rng.uniform(-0.03, 0.03, 3)          # one env step moves the target by at most move_speed * 0.3
        w = rng.uniform(-0.2, 0.2, 3)
        tR = IK.q2m(np.concatenate([[1.0], 0.5 * w])) @ R0
        q = IK.solve(m, q0, tp, tR)
        p, R, _, _ = IK.fk(m, q)
        assert np.abs(p - tp).max() < 1e-6 and np.abs(IK.rotvec(tR @ R.T)).max() < 1e-6
        assert (q >= IK.IK_LOWER).all() and (q <= IK.IK_UPPER).all()
        assert np.abs(q - q0).max() < 0.5               # a nearby solution, not a flip of the redundant arm
    assert np.allclose(IK.velocities([0.0, 0.5, -0.5], [0.1, 0.0, 0.0]), [0.5, -1.0, 1.0])  # -5 (q - q_cmd), clipped to [-1, 1]


def test_quaternion_helpers_doctests():
    """transform_utils.py:112-119 (quat_inverse doctest) and quat2mat on a known rotation."""
    rng = np.random.RandomState(2)
    for _ in range(10):
        q = rng.randn(4)
        q /= np.linalg.norm(q)   # the doctest draws a unit quaternion; both functions round through float32
        assert np.allclose(T.quat_multiply(q, T.quat_inverse(q)), [0, 0, 0, 1], atol=1e-6)
    assert np.allclose(T.quat2mat([0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)]), IK.rot_z(np.pi / 2), atol=1e-6)


def test_oracle_env_ik_step_moves_the_hand_where_commanded(model):
    m = model
    e = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123, control_type="ik"))
    ob = e.reset()
    assert ob["robot_ob"].shape == (15,)   # gripper_qpos 2, eef pos 3, quat 4, velp 3, velr 3 (furniture_sawyer.py:125-153)
    hb = int(m.hand_bodyid[0])
    p0 = e.sim.data.xpos[hb].copy()
    quat0 = ob["robot_ob"][5:9].copy()
    for _ in range(4):
        a = np.zeros(8)
        a[0], a[6] = 1.0, -1.0   # action x -> d_pos = move_speed * [-a1, a0, a2] in the base frame = +x in the world
        ob, _, _, _ = e.step(a)
    moved = e.sim.data.xpos[hb] - p0
    assert 0.08 < moved[0] < 0.16 and abs(moved[1]) < 0.02 and abs(moved[2]) < 0.02  # 4 steps x 0.1 x user_sensitivity 0.3
    assert min(np.abs(ob["robot_ob"][5:9] - quat0).max(), np.abs(ob["robot_ob"][5:9] + quat0).max()) < 0.1  # orientation held


@pytest.mark.gpu
@pytest.mark.parametrize("ctype", ["ik", "ik_quaternion"])
def test_device_ik_env_matches_oracle(model, ctype):
    from furniture_amd.sim import FSim, INFO_DIM, default_config
    m, n = model, 2
    dof = 8 if ctype == "ik" else 9
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset, cfg.control_type = 150, 0, (7 if ctype == "ik" else 8)
    sim = FSim(m, n, config=cfg)
    assert sim.dof_action == dof and sim.obs_dim == 7 * m.nparts + 15
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10, control_type=ctype)) for i in range(n)]
    obs_o = [e.reset() for e in envs]
    sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]),
                         np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    blk = sim.get_state("env_block")["env_block"][:, -26:].cpu().numpy().view(np.float32)  # fsim_ik.hpp EI_*
    for i, e in enumerate(envs):
        assert np.abs(obs[i].cpu().numpy() - e.flat_obs(obs_o[i])).max() < 1e-4
        assert np.abs(blk[i, :3] - e._ik_target_pos).max() < 1e-5 and np.abs(blk[i, 3:7] - e._initial_right_hand_quat).max() < 1e-5
    act = torch.zeros((n, dof), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    rng = np.random.RandomState(5)
    for t in range(4):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
        if ctype == "ik_quaternion":   # a small rotation relative to the current hand orientation (wxyz, not normalised)
            a[:, 3] = 1.0
            a[:, 4:7] *= 0.05
        elif t == 0:
            a[:, 3:6] = 0
        act.copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)   # one launch: IK solve + 3 x (P controller, 50 substeps)
        sim.sync()
        blk = sim.get_state("env_block")["env_block"][:, -26:].cpu().numpy().view(np.float32)
        for i, e in enumerate(envs):
            ob, r, d_, _ = e.step(a[i].astype(np.float64))
            assert np.abs(obs[i].cpu().numpy() - e.flat_obs(ob)).max() < 2e-4
            assert np.abs(blk[i, 7:14] - e._ik_q_cmd).max() < 1e-4        # commanded joint positions
            assert np.abs(blk[i, :3] - e._ik_target_pos).max() < 1e-5
            if ctype == "ik":
                assert np.abs(blk[i, 3:7] - e._initial_right_hand_quat).max() < 1e-5
            assert abs(float(rew[i]) - r) < 1e-5 and bool(done[i]) == d_
    sim.close()


@pytest.mark.gpu
def test_default_gym_id_runs_with_ik():
    """IKEASawyer-v0's defaults (control_type 'ik', swivel_chair_0700; furniture/env/__init__.py:33-43) now construct and step."""
    from furniture_amd.envs import make
    env = make("IKEASawyer-v0", unity=False, record_vid=False)
    assert env.dof == 8
    ob = env.reset()
    assert ob["robot_ob"].shape == (15,)
    ob, r, d, info = env.step(np.array([0.5, 0, 0, 0, 0, 0, -1, 0], dtype=np.float32))
    assert np.isfinite(ob["robot_ob"]).all() and np.isfinite(ob["object_ob"]).all()
    env.close()


@pytest.mark.gpu
def test_device_ik_tracks_an_end_effector_path(model):
    """Task-level validation of the stand-in solver ('validate by task success, not trajectories'): 64 envs follow a
    commanded square path (+x, +z, -x, -z in the world, 8 steps a side) under control_type 'ik'.  The hand must end each side
    within 1.5 cm of where the accumulated command puts it (move_speed 0.1 x user_sensitivity 0.3 = 3 cm per step), keep its
    orientation within 6 degrees, and return to the start within 2 cm -- for every env of the batch."""
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    n = 64
    env = FurnitureBatchEnv("Sawyer", n, config=make_config(unity=False, record_vid=False, control_type="ik", furniture_name="table_lack_0825",
                                                            max_episode_steps=500, seed=7), auto_reset=False)
    ob = env.reset()
    eef0 = ob["robot_ob"][:, 2:5].clone()
    quat0 = ob["robot_ob"][:, 5:9].clone()
    a = torch.zeros((n, 8), device=env.sim.device)
    a[:, 6], a[:, 7] = -1.0, -1.0
    # action axes -> world: d_pos(base) = move_speed * [-a1, a0, a2], and the Sawyer base is yawed -90 deg: world x <- a0, world z <- a2
    sides = [(0, +1.0, 0), (2, +1.0, 2), (0, -1.0, 0), (2, -1.0, 2)]
    expect = eef0.clone()
    for comp, sign, world_axis in sides:
        for _ in range(8):
            a[:, :6] = 0
            a[:, comp] = sign
            ob, _, done, _ = env.step(a)
        a[:, :6] = 0
        for _ in range(3):   # let the P controller settle on the last target
            ob, _, _, _ = env.step(a)
        expect[:, world_axis] += sign * 8 * 0.1 * 0.3
        eef = ob["robot_ob"][:, 2:5]
        assert float((eef - expect).abs().max()) < 0.015, (comp, sign, float((eef - expect).abs().max()))
        q = ob["robot_ob"][:, 5:9]
        cosang = (q * quat0).sum(dim=1).abs().clamp(max=1.0)
        assert float(torch.rad2deg(2 * torch.acos(cosang)).max()) < 6.0
    assert float((ob["robot_ob"][:, 2:5] - eef0).abs().max()) < 0.02 and not bool(done.any())
    env.close()


def _scripted_pick(step_fn, get_obs, n, n_obj, model=None, attach=False, legs=None):
    """furniture_amd.scripted.PickAndAttach driven through (step_fn, get_obs) adapters: -> (leg 0 height, summed reward, num_connected)."""
    from furniture_amd.mjcf.model import load_compiled
    from furniture_amd.scripted import PickAndAttach
    pol = PickAndAttach(model if model is not None else load_compiled("Sawyer", "table_lack_0825"), n, attach=attach)

    def step(a):
        r, nc = step_fn(a)
        obj, rob = get_obs()
        return {"object_ob": obj, "robot_ob": rob}, r, None, {"num_connected": nc}
    obj, rob = get_obs()
    total, ncon, ob = pol.run(step, {"object_ob": obj, "robot_ob": rob}, legs=legs)
    return np.asarray(ob["object_ob"])[:, 2], total, ncon


def test_scripted_pick_on_the_oracle_env(model):
    """The same policy on the fp64 oracle env (CPU): leg 0 ends ~12 cm above the floor, touch + pick rewards are paid."""
    e = FurnitureEnvOracle(model, OracleConfig(max_episode_steps=500, seed=123, control_type="ik_quaternion"))
    last = {"ob": e.reset()}

    def step_fn(a):
        ob, r, d, info = e.step(a[0].astype(np.float64))
        last["ob"] = ob
        return np.array([r]), np.array([info["num_connected"]])
    get = lambda: (last["ob"]["object_ob"][None], last["ob"]["robot_ob"][None])
    z, rew, _ = _scripted_pick(step_fn, get, 1, model.nparts)
    assert z[0] > 0.08 and rew[0] > 100.0   # touch_reward 10 + pick_reward 100 - control penalties


def test_scripted_attach_on_the_oracle_env(model):
    """Pick leg 0, hang it vertically, carry it over a table connector, connect: the weld of the first subtask becomes active, the
    leg and the table top share a group, success_reward is paid (furniture.py:847-924, 926-1042)."""
    e = FurnitureEnvOracle(model, OracleConfig(max_episode_steps=1000, seed=123, control_type="ik_quaternion"))
    last = {"ob": e.reset()}

    def step_fn(a):
        ob, r, d, info = e.step(a[0].astype(np.float64))
        last["ob"] = ob
        return np.array([r]), np.array([info["num_connected"]])
    z, rew, ncon = _scripted_pick(step_fn, lambda: (last["ob"]["object_ob"][None], last["ob"]["robot_ob"][None]), 1, model.nparts, model, attach=True)
    assert ncon[0] == 1 and rew[0] > 200.0 and e.sim.model.eq_active[0] == 1 and e._find_group(0) == e._find_group(4)


def test_scripted_two_leg_assembly_on_the_oracle_env(model):
    """Two subtasks in one episode: leg 0, release, back off, leg 3 -- each to the free table connector nearest to it.  Both welds
    active, both legs in the table top's group, success_reward paid twice."""
    e = FurnitureEnvOracle(model, OracleConfig(max_episode_steps=2000, seed=123, control_type="ik_quaternion"))
    last = {"ob": e.reset()}

    def step_fn(a):
        ob, r, d, info = e.step(a[0].astype(np.float64))
        last["ob"] = ob
        return np.array([r]), np.array([info["num_connected"]])
    z, rew, ncon = _scripted_pick(step_fn, lambda: (last["ob"]["object_ob"][None], last["ob"]["robot_ob"][None]), 1, model.nparts, model,
                                  attach=True, legs=(0, 3))
    assert ncon[0] == 2 and rew[0] > 400.0 and e._find_group(0) == e._find_group(4) == e._find_group(3)
    assert int(np.sum(e.sim.model.eq_active)) == 2


def test_scripted_full_assembly_on_the_oracle_env(model):
    """All four legs of table_lack_0825, one episode, observation-only policy under ik_quaternion: num_connected reaches 4, every part
    is in one group, the episode ends with success (furniture.py:470-480: done when all parts are connected)."""
    from furniture_amd.scripted import FULL_TABLE
    e = FurnitureEnvOracle(model, OracleConfig(max_episode_steps=4000, seed=123, control_type="ik_quaternion"))
    last = {"ob": e.reset(), "done": False}

    def step_fn(a):
        ob, r, d, info = e.step(a[0].astype(np.float64))
        last["ob"], last["done"] = ob, d
        return np.array([r]), np.array([info["num_connected"]])
    z, rew, ncon = _scripted_pick(step_fn, lambda: (last["ob"]["object_ob"][None], last["ob"]["robot_ob"][None]), 1, model.nparts, model,
                                  attach=True, legs=FULL_TABLE)
    assert ncon[0] == 4 and rew[0] > 800.0 and len({e._find_group(i) for i in range(5)}) == 1 and last["done"]
    assert int(np.sum(e.sim.model.eq_active)) == 4


@pytest.mark.gpu
def test_scripted_full_assembly_on_the_device(model):
    """The whole table on the HIP path, 8 placements, no retries in the script: at least half of the envs finish all four subtasks
    (5 of 6 on the fp64 oracle env when written), every env at least two; an env that finishes reports done, and num_connected is
    what the welds say."""
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    from furniture_amd.scripted import FULL_TABLE
    n = 8
    env = FurnitureBatchEnv("Sawyer", n, config=make_config(unity=False, record_vid=False, control_type="ik_quaternion",
                                                            furniture_name="table_lack_0825", max_episode_steps=6000, seed=123), auto_reset=False)
    last = {"ob": env.reset(), "done": None}

    def step_fn(a):
        ob, r, d, info = env.step(a)
        last["ob"] = ob
        last["done"] = d.cpu().numpy().astype(bool) if last["done"] is None else (last["done"] | d.cpu().numpy().astype(bool))
        return r.cpu().numpy(), info["num_connected"].cpu().numpy()
    get = lambda: (last["ob"]["object_ob"].cpu().numpy(), last["ob"]["robot_ob"].cpu().numpy())
    z, rew, ncon = _scripted_pick(step_fn, get, n, model.nparts, model, attach=True, legs=FULL_TABLE)
    act = env.sim.get_state("eq_active")["eq_active"].cpu().numpy()
    print("full scripted assembly on the device: num_connected", ncon.tolist(), "reward", rew.round(0).tolist(), "done", last["done"].tolist())
    assert (ncon >= 2).all() and (ncon == 4).sum() >= n // 2, (ncon, rew.round(1))
    assert ((ncon == 4) == last["done"]).all() and (act.sum(axis=1) >= np.minimum(ncon, 4)).all()
    env.close()


@pytest.mark.gpu
def test_scripted_two_leg_assembly_on_the_device(model):
    """The two-leg episode on the HIP path, 8 placements: at least five finish both subtasks (no retries in the script), every env
    attaches at least the first leg or fails it cleanly (num_connected is what the welds and groups say)."""
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    n = 8
    env = FurnitureBatchEnv("Sawyer", n, config=make_config(unity=False, record_vid=False, control_type="ik_quaternion",
                                                            furniture_name="table_lack_0825", max_episode_steps=3000, seed=123), auto_reset=False)
    last = {"ob": env.reset()}

    def step_fn(a):
        ob, r, d, info = env.step(a)
        last["ob"] = ob
        return r.cpu().numpy(), info["num_connected"].cpu().numpy()
    get = lambda: (last["ob"]["object_ob"].cpu().numpy(), last["ob"]["robot_ob"].cpu().numpy())
    z, rew, ncon = _scripted_pick(step_fn, get, n, model.nparts, model, attach=True, legs=(0, 3))
    st = env.sim.get_state("eq_active", "group")
    act = st["eq_active"].cpu().numpy()
    print("two-leg scripted assembly on the device: num_connected", ncon.tolist(), "reward", rew.round(0).tolist())
    assert (act.sum(axis=1) == ncon).all()
    assert (ncon == 2).sum() >= 5 and ((ncon == 2) <= (rew > 400.0)).all(), (ncon, rew.round(1))
    env.close()


@pytest.mark.gpu
def test_scripted_pick_on_the_device(model):
    """Task success on the HIP path: 16 envs (different placements), observation-only scripted pick under ik_quaternion.
    At least three quarters of the envs (the open-loop script has no re-grasp; 14 of 16 when written) pay the touch + pick
    rewards and hold leg 0 ~12 cm above the floor at the end; env 0 (the oracle test's seed) is one of them."""
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    n = 16
    env = FurnitureBatchEnv("Sawyer", n, config=make_config(unity=False, record_vid=False, control_type="ik_quaternion",
                                                            furniture_name="table_lack_0825", max_episode_steps=500, seed=123), auto_reset=False)
    last = {"ob": env.reset()}

    def step_fn(a):
        ob, r, d, info = env.step(a)
        last["ob"] = ob
        return r.cpu().numpy(), info["num_connected"].cpu().numpy()
    get = lambda: (last["ob"]["object_ob"].cpu().numpy(), last["ob"]["robot_ob"].cpu().numpy())
    z, rew, _ = _scripted_pick(step_fn, get, n, model.nparts)
    ok = (z > 0.08) & (rew > 100.0)
    assert ok.sum() >= (3 * n) // 4 and ok[0], (z.round(3), rew.round(1))
    env.close()


@pytest.mark.gpu
def test_scripted_attach_on_the_device(model):
    """The whole first subtask on the HIP path under IK control: pick leg 0, hang it vertically, carry it over a connector of
    the table top, connect -- num_connected becomes 1, the weld is active and the two parts share a group (bit-exact integers),
    success_reward is paid.  8 placements; the open-loop script has no retries, so three quarters must succeed, env 0 (the
    oracle test's seed) among them."""
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    n = 8
    env = FurnitureBatchEnv("Sawyer", n, config=make_config(unity=False, record_vid=False, control_type="ik_quaternion",
                                                            furniture_name="table_lack_0825", max_episode_steps=1000, seed=123), auto_reset=False)
    last = {"ob": env.reset()}

    def step_fn(a):
        ob, r, d, info = env.step(a)
        last["ob"] = ob
        return r.cpu().numpy(), info["num_connected"].cpu().numpy()
    get = lambda: (last["ob"]["object_ob"].cpu().numpy(), last["ob"]["robot_ob"].cpu().numpy())
    z, rew, ncon = _scripted_pick(step_fn, get, n, model.nparts, model, attach=True)
    st = env.sim.get_state("eq_active", "group")
    ok = (ncon == 1) & (rew > 200.0) & (st["eq_active"][:, 0].cpu().numpy() == 1)
    grp = st["group"].cpu().numpy()
    print("scripted attach on the device: %d of %d placements connected" % (int(ok.sum()), n))
    assert ok.sum() >= (3 * n) // 4 and ok[0], (ncon, rew.round(1))
    for i in np.nonzero(ok)[0]:
        assert grp[i, 0] == 4 or grp[i, 4] == grp[i, 0] or grp[i, grp[i, 0]] == grp[i, 4]  # leg 0 and the table top in one group
    env.close()


def test_pybullet_joint_order_reproduces_the_reference_constants():
    """baxter_ik_controller.py:124-137 hard-codes pybullet indices (effectors 27 / 45, arm joints 13-17, 19, 20 / 31-35, 37, 38): the
    depth-first joint order of the URDF tree gives exactly those, and the resulting chains equal the MJCF kinematics (the URDF
    gripper frame is the MJCF hand frame moved 2.5 cm along its z axis, both arms, any configuration)."""
    m = load_compiled("Baxter", "desk_mikael_1064")
    assert m.ik_joint_pos.shape == (14, 3) and m.ik_params.tolist() == [1.0, 2.0, 1.0, 0.0]
    sim = OracleSim(m)
    rng = np.random.RandomState(3)
    Rb, pb = IK.q2m(m.ik_base_quat), m.ik_base_pos
    for _ in range(3):
        q = m.arm_initqpos + rng.uniform(-0.5, 0.5, 14)
        sim.reset()
        sim.data.qpos[m.arm_qposadr] = q
        sim.forward()
        for arm in range(2):
            p, R, _, _ = IK.fk(m, q[7 * arm:7 * arm + 7], arm)
            hb = int(m.hand_bodyid[arm])
            hand_R = Rb.T @ sim.data.xmat[hb].reshape(3, 3)
            hand_p = Rb.T @ (sim.data.xpos[hb] - pb)
            assert np.abs(hand_R.T @ R - np.eye(3)).max() < 1e-5
            assert np.abs(hand_R.T @ (p - hand_p) - np.array([0, 0, 0.025])).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("ctype", ["ik", "ik_quaternion"])
def test_device_baxter_ik_matches_oracle(ctype):
    """Bimanual Baxter under IK control (IKEABaxter-v0's default control type): two chains, user_sensitivity 1, P gain 2, rest pose =
    current joints, no Rz convention -- device vs the fp64 oracle env."""
    from furniture_amd.sim import FSim, INFO_DIM, default_config
    m, n = load_compiled("Baxter", "desk_mikael_1064"), 2
    dof = 15 if ctype == "ik" else 17
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset, cfg.control_type = 150, 0, (7 if ctype == "ik" else 8)
    sim = FSim(m, n, config=cfg)
    assert sim.dof_action == dof and sim.obs_dim == 7 * m.nparts + 30
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10, control_type=ctype)) for i in range(n)]
    obs_o = [e.reset() for e in envs]
    sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]),
                         np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    for i, e in enumerate(envs):
        assert np.abs(obs[i].cpu().numpy() - e.flat_obs(obs_o[i])).max() < 1e-4
    act = torch.zeros((n, dof), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    rng = np.random.RandomState(8)
    for t in range(3):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
        if ctype == "ik_quaternion":
            for arm in range(2):
                a[:, 7 * arm + 3] = 1.0
                a[:, 7 * arm + 4:7 * arm + 7] *= 0.05
        else:  # moderate rotation commands: a target a quarter turn away per step leaves the damped solve ill-conditioned in fp32
            a[:, 3:6] *= 0.0 if t == 0 else 0.25
            a[:, 9:12] *= 0.0 if t == 0 else 0.25
        act.copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        blk = sim.get_state("env_block")["env_block"][:, -52:].cpu().numpy().view(np.float32)   # two EI blocks of 26 words
        for i, e in enumerate(envs):
            ob, r, d_, _ = e.step(a[i].astype(np.float64))
            assert np.abs(obs[i].cpu().numpy() - e.flat_obs(ob)).max() < 3e-4
            for arm in range(2):
                assert np.abs(blk[i, 26 * arm + 7:26 * arm + 14] - e._ik_q_cmd[7 * arm:7 * arm + 7]).max() < 1e-4
                assert np.abs(blk[i, 26 * arm:26 * arm + 3] - e._ik_tp[arm]).max() < 1e-5
            assert abs(float(rew[i]) - r) < 1e-5 and bool(done[i]) == d_
    sim.close()


@pytest.mark.gpu
def test_default_baxter_gym_id_runs_with_ik():
    """IKEABaxter-v0's defaults (control_type 'ik', furniture_id 1; furniture/env/__init__.py:47-57)."""
    from furniture_amd.envs import furniture_names, make
    from furniture_amd.mjcf.model import _COMPILED_DIR
    import os
    name = furniture_names()[1]
    if not os.path.exists(os.path.join(_COMPILED_DIR, "Baxter__%s__vel.npz" % name)):
        pytest.skip("the default Baxter furniture (%s) is not among the shipped compiled models" % name)
    env = make("IKEABaxter-v0", unity=False, record_vid=False)
    assert env.dof == 15
    ob = env.reset()
    ob, r, d, info = env.step(np.zeros(15, dtype=np.float32))
    assert ob["robot_ob"].shape == (30,) and np.isfinite(ob["robot_ob"]).all()
    env.close()


@pytest.mark.gpu
def test_default_cursor_gym_id_runs():
    """IKEACursor-v0's defaults (furniture_id 0 = bed_dalselv_0270; furniture/env/__init__.py:19-29)."""
    from furniture_amd.envs import make
    env = make("IKEACursor-v0", unity=False, record_vid=False)
    ob = env.reset()
    ob, r, d, info = env.step(np.zeros(15, dtype=np.float32))
    assert np.isfinite(ob["object_ob"]).all() and ob["robot_ob"].shape == (8,)
    env.close()
