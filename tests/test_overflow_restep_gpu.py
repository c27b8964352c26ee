"""Contact-overflow re-step (include/fsim.h fsim_overflow_resteps).  The benchmark model runs on 48 contact slots -- what lets eight envs
share a CU's LDS -- and about 1.6 times per million env-steps an env needs more (scripts/dev/overflow_census.py).  Round 3 dropped the
surplus contacts for that substep and said so in a flag; now the step launch keeps every env's pre-step record, lists the envs that
overflowed, and fsim_sync steps those again from the kept record on a four-wave team with a 64-slot layout.  Tested on the benchmark's
own first overflow (4096 envs of the benchmark seed: env 3707 at step 5): with the re-step the env matches the fp64 oracle env (MuJoCo's
contact arrays are never short: nconmax = 5000), no sticky report remains, and every other env of the batch has the bits it has without it."""
import numpy as np
import pytest
import torch

from furniture_amd.envs import ResetTableSampler, make_config
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import E_OVERFLOW, FSim, INFO_DIM, default_config

pytestmark = pytest.mark.gpu
N, STEPS, SEED = 4096, 6, 123


def _run(m):
    cfg = default_config()
    cfg.max_episode_steps = 150
    sim = FSim(m, N, config=cfg)
    sim.set_reset_tables(*ResetTableSampler(m, make_config(), SEED, 0, N).draw())
    dev = sim.device
    obs = torch.zeros((N, sim.obs_dim), device=dev)
    rew, done = torch.zeros(N, device=dev), torch.zeros(N, dtype=torch.uint8, device=dev)
    info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
    act = torch.empty((N, 9), device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    sim.reset(None, obs)
    sim.sync()
    trace, acts = [obs.clone()], []
    for t in range(STEPS):
        act.uniform_(-1, 1, generator=g)
        torch.cuda.synchronize()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        acts.append(act.clone())
        trace.append(obs.clone())
    sticky = sim.get_state("env_block")["env_block"].view(torch.int32)[:, E_OVERFLOW].cpu().numpy()
    out = dict(trace=trace, acts=acts, sticky=sticky, resteps=sim.overflow_resteps(), qpos=sim.get_state("qpos")["qpos"].clone(), rew=rew.clone())
    sim.close()
    return out


def test_an_env_that_overflows_48_contact_slots_is_stepped_again_with_64(monkeypatch):
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Sawyer", "table_lack_0825")
    monkeypatch.setenv("FSIM_NO_OVERFLOW_REDO", "1")
    off = _run(m)
    flagged = np.nonzero(off["sticky"])[0]
    assert off["resteps"] == 0 and 1 <= len(flagged) <= 3, flagged  # (the census: env 3707 at step 5)
    monkeypatch.delenv("FSIM_NO_OVERFLOW_REDO")
    on = _run(m)
    assert on["resteps"] >= len(flagged) and not on["sticky"].any(), (on["resteps"], np.nonzero(on["sticky"])[0])
    # everybody else: the same bits with and without
    others = np.ones(N, dtype=bool)
    others[flagged] = False
    for a, b in zip(on["trace"], off["trace"]):
        assert torch.equal(a[torch.as_tensor(others)], b[torch.as_tensor(others)])
    assert torch.equal(on["qpos"][torch.as_tensor(others)], off["qpos"][torch.as_tensor(others)])
    # the env itself, against the fp64 oracle env of the same seed and actions -- with and without the re-step
    e = int(flagged[0])
    orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=SEED + e, solver_tolerance=1e-10))
    ref = [orc.flat_obs(orc.reset())]
    for t in range(STEPS):
        ref.append(orc.flat_obs(orc.step(on["acts"][t][e].cpu().numpy().astype(np.float64))[0]))
    err_on = [float(np.abs(on["trace"][t][e].cpu().numpy() - ref[t]).max()) for t in range(STEPS + 1)]
    err_off = [float(np.abs(off["trace"][t][e].cpu().numpy() - ref[t]).max()) for t in range(STEPS + 1)]
    print("env %d: obs error vs oracle per step, with the re-step %s, without %s" % (e, np.round(err_on, 5), np.round(err_off, 5)))
    assert max(err_on) < 2e-3, err_on


def test_a_reset_that_overflows_64_slots_is_repeated_with_128(monkeypatch):
    """The same mechanism for RESET launches and for models on 64 slots (re-step kernel: the generic one-wave kernel with two slots per
    lane).  table_bjorkudden_0207 under config.assembled: the welds yank the parts together through more simultaneous contacts than 64
    slots hold -- reset() used to raise."""
    from furniture_amd.envs import ContactOverflowError, make_vec_env
    kw = dict(furniture_name="table_bjorkudden_0207", max_episode_steps=5, seed=3, record_vid=False, unity=False, control_type="impedance", assembled=True)
    monkeypatch.setenv("FSIM_NO_OVERFLOW_REDO", "1")
    env = make_vec_env("Sawyer", 2, **kw)
    assert env.sim.max_contacts == 64
    with pytest.raises(ContactOverflowError):
        env.reset()
    env.close()
    monkeypatch.delenv("FSIM_NO_OVERFLOW_REDO")
    env = make_vec_env("Sawyer", 2, **kw)
    ob = env.reset()
    assert env.sim.overflow_resteps() >= 1
    assert not bool((env.sim.get_state("env_block")["env_block"].view(torch.int32)[:, E_OVERFLOW] != 0).any())
    assert bool(torch.isfinite(ob["object_ob"]).all())
    for t in range(3):
        ob, rew, done, info = env.step(torch.zeros((2, env.dof), device=env.sim.device))
    assert bool(torch.isfinite(ob["object_ob"]).all()) and not bool((info["contact_overflow"] != 0).any())
    env.close()


@pytest.mark.parametrize("furniture", ["three_blocks_peg", "table_torsby_1549"])
def test_a_reset_beyond_64_slots_takes_the_second_rung_to_128(furniture):
    """Round 6: the re-step is a ladder.  Cursor + three_blocks_peg / table_torsby_1549 run on 48 contact slots and pass through 80 / 72
    simultaneous contacts while their reset settles (counted with the oracle): the 64-slot rung overflows too, the env is repeated once more on
    the generic one-wave kernel with 128 slots, and the reset that comes out is the fp64 oracle env's (which holds 256 contacts) -- no sticky
    report, both rungs counted."""
    from furniture_amd.envs import FurnitureCursorEnv
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Cursor", furniture)
    kw = dict(unity=False, record_vid=False, furniture_name=furniture, max_episode_steps=50, seed=11)
    env = FurnitureCursorEnv(make_config(**kw))
    assert env._b.sim.max_contacts == 48
    orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=11, solver_tolerance=1e-10))
    peak = [0]
    step0 = orc.sim.step

    def counted():
        step0()
        peak[0] = max(peak[0], orc.sim.ncon)
    orc.sim.step = counted
    o = orc.flat_obs(orc.reset())
    assert peak[0] > 64, peak[0]
    d = env.reset()  # (raises ContactOverflowError if a sticky report is left)
    assert env._b.sim.overflow_resteps() >= 2  # the 64-slot rung and the 128-slot rung
    got = np.concatenate([d["object_ob"], d["robot_ob"]])
    assert np.abs(got - o).max() < 2e-3, float(np.abs(got - o).max())
    sticky = env._b.sim.get_state("env_block")["env_block"].view(torch.int32)[:, E_OVERFLOW].cpu().numpy()
    assert not sticky.any()
    env.close()


def _sampler_start(m, name, seed):
    """the state the reset's first substep sees: robot at its initial pose, the parts where the reference's sampler puts them"""
    parts, _ = ResetTableSampler(m, make_config(furniture_name=name), seed, 0, 1).draw()
    q = np.array(m.qpos0, dtype=np.float64)
    q[m.arm_qposadr], q[m.grip_qposadr] = m.arm_initqpos, m.grip_initqpos
    pq = np.asarray(parts).reshape(-1, 7)
    for i in range(m.nparts):
        q[m.part_qposadr[i]:m.part_qposadr[i] + 7] = pq[i]
    return q


@pytest.mark.parametrize("furniture,tol", [("table_liden_0921", 1e-6), ("bookcase_grevback_0484", 5e-4)])
def test_the_512_slot_kernel_against_the_oracle_on_planks_inside_each_other(furniture, tol, monkeypatch):
    """The last rung's kernel by itself (`generic8`: eight contact slots per lane in the Newton solve; every island factored in LDS by
    fs_chol_all_lds because one island holds more than 64 dofs), as the base kernel of a one-env batch (FSIM_NCON_MAX=512), over the first
    substeps of the start the reference's sampler gives these furniture: the planks INSIDE each other -- table_liden_0921 268 contacts and
    bookcase_grevback_0484 370 (the fourteen planks stacked at one point: coincident boxes), all parts in ONE island of 72 / 84 dofs, thrown
    apart at 80 - 480 m/s.  Against OracleSim (fp64; its capacity is 1024 contacts and it raises when that is short): the same contact count
    (+ 1: the robot's l0 / base pair at exactly zero distance, which fp32 lists), qpos within 1e-6 for table_liden's first two substeps (measured
    1e-7) and 5e-4 for the coincident planks of grevback (measured 6e-5 / 1.1e-4: which face pair a box-box contact of two coincident boxes
    picks is a tie that fp32 breaks differently)."""
    from oracle.oracle_sim import OracleSim
    monkeypatch.setenv("FSIM_NCON_MAX", "512")
    m = load_compiled("Sawyer", furniture)
    q = _sampler_start(m, furniture, 3)
    sim = FSim(m, 1)
    assert sim.kernel_variant == "generic8" and sim.max_contacts == 512
    sim.set_state(qpos=q[None], qvel=np.zeros((1, m.nv)), qacc_warmstart=np.zeros((1, m.nv)))
    o = OracleSim(m)
    o.set_solver(100, 1e-10, "newton")
    o.reset()
    o.data.qpos[:] = q
    o.forward()
    peak = 0
    for t in range(2):
        sim.physics_step(1)
        o.step()
        st = sim.get_state("qpos", "qvel", "ncon", "solver_iters")
        peak = max(peak, o.ncon)
        assert int(st["ncon"][0]) in (o.ncon, o.ncon + 1), (t, int(st["ncon"][0]), o.ncon)
        assert abs(int(st["solver_iters"][0]) - o.last_solver_iters) <= 3
        assert np.abs(st["qpos"][0].cpu().numpy() - o.data.qpos).max() < tol, t
        assert np.abs(st["qvel"][0].cpu().numpy() - o.data.qvel).max() < 1e-3 * np.abs(o.data.qvel).max(), t
    assert peak > 256 and np.abs(o.data.qvel).max() > 30.0
    sim.close()


@pytest.mark.parametrize("furniture,tol", [("table_liden_0921", 2e-3), ("bookcase_billy_0191", None), ("bookcase_grevback_0484", None)])
def test_a_reset_with_the_planks_inside_each_other_takes_the_last_rung_to_512(furniture, tol, monkeypatch):
    """the three Sawyer furniture whose reset() raised.  They run on 128 slots (two per lane); the reset the reference's
    sampler gives them starts with the planks inside each other -- 240 to 370 contacts and every part in one island of 66 to 84 dofs for the
    first substeps -- and is repeated on the ladder's last rung: 512 slots on the one-wave kernel `generic8`, islands of more than 64 dofs
    through the LDS-resident factorisation.  reset() returns (no sticky report), the rung is counted, and the observation is the fp64 oracle env's:
    within 2e-3 for table_liden (measured 3e-4) and, for the two bookcases, in the median (the planks start COINCIDENT, fly apart at up to
    480 m/s and are still tumbling at 10 - 16 m/s when the reset's 401 substeps end: a part's pose is then sensitive to the last bit, in the
    oracle's own fp32 build too)."""
    from furniture_amd.envs import ContactOverflowError, FurnitureSawyerEnv
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Sawyer", furniture)
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name=furniture, max_episode_steps=50, seed=3)
    monkeypatch.setenv("FSIM_NO_OVERFLOW_REDO", "1")
    env = FurnitureSawyerEnv(make_config(**kw))
    assert env._b.sim.max_contacts == 128 and env._b.sim.kernel_variant == "generic2"
    with pytest.raises(ContactOverflowError):  # without the ladder the reset says that it dropped contacts (sticky report), as in round 5
        env.reset()
    env.close()
    monkeypatch.delenv("FSIM_NO_OVERFLOW_REDO")
    env = FurnitureSawyerEnv(make_config(**kw))
    orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=3, solver_tolerance=1e-10))
    peak = [0]
    step0 = orc.sim.step

    def counted():
        step0()
        peak[0] = max(peak[0], orc.sim.ncon)
    orc.sim.step = counted
    o = orc.flat_obs(orc.reset())
    assert peak[0] > 128, peak[0]
    d = env.reset()
    assert env._b.sim.overflow_resteps() >= 1
    got = np.concatenate([d["object_ob"], d["robot_ob"]])
    dd = np.abs(got - o)
    assert np.isfinite(got).all() and np.median(dd) < 5e-4, float(np.median(dd))
    if tol is not None:
        assert dd.max() < tol, float(dd.max())
    assert np.abs(d["robot_ob"] - o[7 * m.nparts:]).max() < 5e-4  # the robot is not part of the pile
    sticky = env._b.sim.get_state("env_block")["env_block"].view(torch.int32)[:, E_OVERFLOW].cpu().numpy()
    assert not sticky.any()
    rng = np.random.RandomState(2)
    for t in range(3):  # the steps that follow (parts still settling: some overflow 128 slots again and are re-stepped)
        a = rng.uniform(-1, 1, 9)
        ob, r, done, info = env.step(a)
        ob_o, r_o, done_o, _ = orc.step(a)
        g = np.concatenate([ob["object_ob"], ob["robot_ob"]])
        assert np.isfinite(g).all() and int(info["contact_overflow"]) == 0 and done == done_o
        assert np.median(np.abs(g - orc.flat_obs(ob_o))) < 1e-3, t
    env.close()


def test_a_pile_of_eleven_planks_is_one_island_of_66_dofs_and_goes_to_the_last_rung(monkeypatch):
    """The other capacity of a kernel: its island map holds islands of up to 64 dofs (one lane per dof).  bookcase_billy_0191's eleven planks
    stacked in ONE pile (set_init_qpos, tests/scenarios.py stacked_layout) are one island of 66 dofs with about 45 contacts -- the 128 slots hold
    them, the lane map does not.  Through round 5 (and here with the ladder off) such a step failed like an unstable simulation; now the kernel
    raises the capacity report as well, the env is listed for the re-step ladder and its last rung (LDS-resident factorisation) solves it:
    reset and steps against the fp64 oracle env."""
    from furniture_amd.envs import ContactOverflowError, FurnitureSawyerEnv
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    from tests.scenarios import stacked_layout
    name = "bookcase_billy_0191"
    m = load_compiled("Sawyer", name)
    lay = stacked_layout(m)
    q = np.array(m.qpos0, dtype=float)
    q[m.arm_qposadr], q[m.grip_qposadr] = m.arm_initqpos, m.grip_initqpos
    for p in range(m.nparts):
        q[m.part_qposadr[p]:m.part_qposadr[p] + 7] = lay[p]
    init = {"qpos": q, "qvel": np.zeros(m.nv)}
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name=name, max_episode_steps=50, seed=3)
    orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=3, solver_tolerance=1e-10))
    orc.set_init_qpos(init)
    o = orc.flat_obs(orc.reset())
    pairs = {tuple(sorted((int(m.body_partid[m.geom_bodyid[a]]), int(m.body_partid[m.geom_bodyid[b]])))) for a, b in orc.sim.contacts()}
    linked = {p for pr in pairs for p in pr if pr[0] >= 0 and pr[1] >= 0}
    assert len(linked) == m.nparts and orc.sim.ncon <= 100, (sorted(linked), orc.sim.ncon)  # every plank touches another one: one island; the slots suffice
    monkeypatch.setenv("FSIM_NO_OVERFLOW_REDO", "1")
    env = FurnitureSawyerEnv(make_config(**kw))
    env.set_init_qpos(init)
    with pytest.raises((ContactOverflowError, RuntimeError)):  # without the ladder: reported, not integrated wrongly
        env.reset()
    env.close()
    monkeypatch.delenv("FSIM_NO_OVERFLOW_REDO")
    env = FurnitureSawyerEnv(make_config(**kw))
    env.set_init_qpos(init)
    d = env.reset()
    assert env._b.sim.overflow_resteps() >= 1
    got = np.concatenate([d["object_ob"], d["robot_ob"]])
    dd = np.abs(got - o)
    # (measured: median 1e-7, one plank of the pile 1.5e-3 -- eleven planks held on each other by friction alone over 401 substeps)
    assert np.median(dd) < 1e-5 and dd.max() < 5e-3, (float(np.median(dd)), float(dd.max()))
    rng = np.random.RandomState(2)
    for t in range(3):
        a = rng.uniform(-1, 1, 9)
        ob, r, done, info = env.step(a)
        ob_o, r_o, done_o, _ = orc.step(a)
        dd = np.abs(np.concatenate([ob["object_ob"], ob["robot_ob"]]) - orc.flat_obs(ob_o))
        assert np.median(dd) < 1e-4 and dd.max() < 1e-2 and int(info["contact_overflow"]) == 0 and int(info["fail"]) == 0 and done == done_o, (t, float(dd.max()))
    assert env._b.sim.overflow_resteps() >= 4  # the reset and each of the three steps
    env.close()
