"""SURVEY f2 on the device: the torque-level arm controllers (control_type position_orientation / position / joint_impedance /
joint_velocity / joint_torque) run as a per-substep stage of the fused step kernel, against the fp64 oracle env whose controller
restatement is pinned to the reference's own classes (tests/test_controllers_golden.py).

The reference's reset flow leaves the motor-actuated arm spinning (no velocity actuators to damp it, gravity compensation taken
from a stale pass, qvel never cleared: furniture.py:1572-1640), which makes trajectories chaotic; the parity scenario therefore
starts both sides from the oracle's post-reset poses with the arm at rest.  Two variants: qfrc_applied as the reset leaves it
(the double gravity compensation the reference applies: qfrc_applied + the qfrc_bias inside ctrl) and qfrc_applied = 0."""
import numpy as np
import pytest
import torch

from furniture_amd.envs import CONTROLLER_CODES, FurnitureBatchEnv, make_config
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, FsimError, INFO_DIM, default_config
from oracle import controllers as C
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", C.TYPES)
@pytest.mark.parametrize("keep_applied", [True, False])
def test_controller_steps_match_oracle(kind, keep_applied):
    m = load_compiled("Sawyer", "table_lack_0825", kind)
    n = 2
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset, cfg.control_type = 150, 0, CONTROLLER_CODES[kind]
    sim = FSim(m, n, config=cfg)
    assert sim.dof_action == C.control_dim(kind) + 2 and sim.obs_dim == 7 * m.nparts + 15
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10, control_type=kind)) for i in range(n)]
    for e in envs:
        e.reset()
    sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]),
                         np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    sim.reset(None, obs)
    sim.sync()
    for e in envs:  # clean start: the oracle's part poses, arm at its initial pose and at rest
        d = e.sim.data
        d.qvel[:] = 0
        d.qacc_warmstart[:] = 0
        d.qpos[m.arm_qposadr] = m.arm_initqpos
        e.sim.forward()
        d.qfrc_applied[:] = 0
        if keep_applied:
            e._gravity_comp()
    sim.set_state(qpos=np.stack([e.sim.data.qpos for e in envs]), qvel=np.zeros((n, m.nv)), qacc_warmstart=np.zeros((n, m.nv)),
                  qfrc_applied=np.stack([e.sim.data.qfrc_applied for e in envs]))
    dof = sim.dof_action
    act = torch.zeros((n, dof), device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.uint8, device=dev)
    info = torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    rng = np.random.RandomState(5)
    for t in range(3):
        a = rng.uniform(-1.2, 1.2, (n, dof)).astype(np.float32)  # beyond [-1, 1]: transform_action clips
        act.copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        st = sim.get_state("ctrl", "env_block")
        for i, e in enumerate(envs):
            ob, r, d_, _ = e.step(a[i].astype(np.float64))
            assert np.abs(obs[i].cpu().numpy() - e.flat_obs(ob)).max() < 2e-4
            want = e.sim.data.ctrl
            assert np.abs(st["ctrl"][i].cpu().numpy() - want).max() < 5e-4 * (1 + np.abs(want).max())  # last substep's ctrl
            assert abs(float(rew[i]) - r) < 1e-5 and bool(done[i]) == d_
            ek = st["env_block"][i].cpu().numpy()[-52:]  # controller block (fsim_ctrl.hpp EK_*): kind, ramp step
            assert ek[0] == CONTROLLER_CODES[kind] - 1 and ek[1] == e._ctrl[0]["step"] == 50
    sim.close()


def test_reference_flow_runs_and_controller_state_survives_reset():
    """The reference's own flow (reset, then steps) with in-kernel auto-reset: finite observations, and the controller block is
    NOT cleared by a reset (controller.reset() runs only in _reset_internal, i.e. on the first reset: furniture.py:1885-1887)."""
    env = FurnitureBatchEnv("Sawyer", 8, config=make_config(unity=False, record_vid=False, control_type="joint_velocity",
                                                             furniture_name="table_lack_0825", max_episode_steps=3))
    assert env.dof == 9 and env.observation_space.spaces["robot_ob"].shape == (15,)
    env.reset()
    g = torch.Generator(device=env.sim.device)
    g.manual_seed(1)
    a = torch.empty((8, env.dof), device=env.sim.device)
    for t in range(4):
        ob, rew, done, info = env.step(a.uniform_(-1, 1, generator=g))
        assert bool(torch.isfinite(ob["robot_ob"]).all()) and bool(torch.isfinite(ob["object_ob"]).all())
        if t == 2:
            assert bool(done.all())  # time limit -> auto-reset inside the launch
            last = env.sim.get_state("env_block")["env_block"][:, -52:].cpu().numpy().view(np.float32)[:, 12:19].copy()
    blk = env.sim.get_state("env_block")["env_block"][:, -52:].cpu().numpy()
    assert (blk[:, 1] == 50).all()
    assert np.abs(last).max() > 0  # last_goal of the joint ramp carried across the reset (it seeds the next ramp)
    env.close()


def test_rejections():
    m_vel = load_compiled("Sawyer", "table_lack_0825")
    cfg = default_config()
    cfg.control_type = CONTROLLER_CODES["position"]
    with pytest.raises(FsimError):  # velocity-actuated model: the controllers need the motor model (robot_torque.xml)
        FSim(m_vel, 1, config=cfg)
    cfg.control_type = 1
    with pytest.raises(FsimError):  # the reference's 'torque' path is broken (8-vector into 9 actuators)
        FSim(load_compiled("Sawyer", "table_lack_0825", "joint_torque"), 1, config=cfg)
