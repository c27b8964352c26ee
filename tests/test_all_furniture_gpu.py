"""Every furniture shipped compiled for the Sawyer agent runs reset + random steps on the device, including those with more than 64 dofs (up to 93: fourteen parts --
the island solver fills its four 16-lane rows twice) and those with ten parts and more, whose contacts at rest need more than 64
slots (128: two contact slots per lane in the Newton solve); nothing is refused at fsim_create, and a step that drops contacts raises.  The parity tests cover the BASELINE configs' models; this one is
breadth: the generic kernels, the model compiler's tables and the host-side samplers on models nobody looked at individually;
plus one device-vs-oracle reset on the largest model."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_every_compiled_sawyer_furniture_resets_and_steps():
    import torch
    from furniture_amd.envs import ContactOverflowError, make_vec_env
    from furniture_amd.mjcf.model import _COMPILED_DIR, load_compiled
    from furniture_amd.sim import FsimError

    names = sorted(os.path.basename(p)[len("Sawyer__"):-len("__vel.npz")] for p in glob.glob(os.path.join(_COMPILED_DIR, "Sawyer__*__vel.npz")))
    assert len(names) >= 60
    ran, refused, unplaceable, troubled = [], [], [], []
    for name in names:
        m = load_compiled("Sawyer", name)
        try:
            env = make_vec_env("Sawyer", 4, furniture_name=name, max_episode_steps=3, seed=11, record_vid=False, unity=False, control_type="impedance")
        except FsimError as e:
            refused.append((name, str(e)))
            continue
        try:
            ob = env.reset()
        except ContactOverflowError:  # (the reset itself dropped contacts: reported by reset() since round 4 -- the flag is sticky on the device)
            troubled.append((name, m.nparts, 2))
            env.close()
            continue
        except RuntimeError as e:
            # the reference's UniformRandomSampler raises RandomizationError for the same furniture and seeds (checked by running
            # it: cabinet_akurum_0021, table_hemnes_0539 never place with the default jitter; bookcase_grevback_0484's fourteen
            # planks all start at the origin with 5 mm placement radii and do not place for these seeds either) -- not a device matter
            assert "Cannot place all objects" in str(e), (name, str(e))
            unplaceable.append(name)
            env.close()
            continue
        assert ob["object_ob"].shape == (4, 7 * m.nparts)
        g = torch.Generator(device=env.sim.device)
        g.manual_seed(1)
        trouble = 0
        for t in range(4):  # crosses an in-kernel auto-reset (max_episode_steps = 3)
            a = torch.empty((4, 9), device=env.sim.device).uniform_(-1, 1, generator=g)
            try:
                ob, rew, done, info = env.step(a)
            except ContactOverflowError:
                trouble = 2
                break
            assert bool(torch.isfinite(ob["object_ob"]).all()) and bool(torch.isfinite(ob["robot_ob"]).all()) and bool(torch.isfinite(rew).all()), (name, t)
            trouble |= int(info["fail"].max()) | (int(info["contact_overflow"].max()) << 1)
            if not trouble:
                assert bool(done.all()) == (t == 2), (name, t)
        if trouble:  # contact slots / survivor list overflowed or the simulation was flagged unstable: reported, counted, bounded below
            troubled.append((name, m.nparts, trouble))
            env.close()
            continue
        # the parts rest on the floor after the reset: no part centre below it, none gone for good.  (Not "within a metre or two": the
        # placement sampler spaces the parts by bounding radii and chair_agam_0005's meshes still overlap -- the constraint solver
        # pushes them apart at 10 m/s, in the fp64 oracle too (1.1 m in one step, scripts/dev/agam_check.py), and how far such a part has
        # slid after 4 steps depends on the last bit of the solver's summation order: 1.4 m with one build, 3.1 m with the next.)
        z = ob["object_ob"].reshape(4, m.nparts, 7)[:, :, 2]
        assert float(z.min()) > -0.01 and float(ob["object_ob"].reshape(4, m.nparts, 7)[:, :, :3].abs().max()) < 10.0, name
        env.close()
        ran.append(name)
    print("ran %d furniture models (%d of them with more than 64 dofs), refused %d: %s; placement sampler gives up (as the reference's does) on %s" % (
        len(ran), sum(load_compiled("Sawyer", x).nv > 64 for x in ran), len(refused), refused, unplaceable))
    print("overflowed or failed (name, parts, fail | overflow << 1):", troubled)
    # (through round 5 two models raised here: bookcase_billy_0191 (11 planks) and table_liden_0921 (12 parts) pass through 240-270 simultaneous
    #  contacts, all parts in one island, while the reset throws the planks apart; round 6: the re-step ladder's last rung takes them --
    #  512 slots, islands of more than 64 dofs: tests/test_overflow_restep_gpu.py)
    assert refused == [] and troubled == [], (refused, troubled)
    assert sorted(unplaceable) == ["bookcase_grevback_0484", "cabinet_akurum_0021", "table_hemnes_0539"], unplaceable
    assert len(ran) == len(names) - 3
    # since round 5 all 64 furniture of the reference compile (furniture/tests/test_furniture_init.py:14-55 resets them all): the three
    # that collide mesh geoms run like the others (their hulls: tests/test_mesh.py)
    assert len(names) == 64 and {"chair_agne_0010", "chair_bertil_0148", "shelf_liden_0922"} <= set(ran)


def test_a_model_with_more_than_64_dofs_matches_the_oracle_env():
    """Sawyer + bed_dalselv_0270 (69 dofs, 10 parts): reset (about 400 substeps with the parts settling) and random steps, device vs fp64 oracle env"""
    from furniture_amd.envs import FurnitureSawyerEnv, make_config
    from furniture_amd.mjcf.model import load_compiled
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Sawyer", "bed_dalselv_0270")
    assert m.nv == 69
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="bed_dalselv_0270", max_episode_steps=50, seed=3)
    env = FurnitureSawyerEnv(make_config(**kw))
    orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=3, solver_tolerance=1e-10))
    o = orc.flat_obs(orc.reset())
    d = env.reset()
    assert np.abs(np.concatenate([d["object_ob"], d["robot_ob"]]) - o).max() < 5e-4
    rng = np.random.RandomState(2)
    for t in range(3):
        a = rng.uniform(-1, 1, 9)
        ob, r, done, info = env.step(a)
        ob_o, r_o, done_o, _ = orc.step(a)
        assert np.abs(np.concatenate([ob["object_ob"], ob["robot_ob"]]) - orc.flat_obs(ob_o)).max() < 1e-3, t
        assert abs(r - r_o) < 1e-4 and done == done_o
    env.close()


def test_config_assembled_constructs_for_furniture_with_more_welds_than_recipe_steps(monkeypatch):
    """config.assembled switches every weld on (furniture.py:1502-1503): the weld ids go to the device as they are -- furniture whose
    recipe has fewer steps than the model has welds (bench_bjursta_0210: 8 welds, 4 steps) used to die in the constructor.
    (The welds pull the parts -- started at the XML's poses -- together in a violent transient that runs through more simultaneous
    contacts than the slots hold on some models: reported by reset() since the overflow flag is sticky; downgraded to a warning here,
    the test is about construction and weld activity.)"""
    from furniture_amd.envs import make_vec_env
    monkeypatch.setenv("FSIM_ALLOW_OVERFLOW", "1")
    from furniture_amd.mjcf.model import load_compiled
    for name in ("bench_bjursta_0210", "chair_ingolf_0650", "table_bjorkudden_0207", "table_lack_0825"):
        m = load_compiled("Sawyer", name)
        env = make_vec_env("Sawyer", 2, furniture_name=name, max_episode_steps=5, seed=3, record_vid=False, unity=False, control_type="impedance", assembled=True)
        ob = env.reset()
        assert ob["object_ob"].shape == (2, 7 * m.nparts)
        act = env.sim.get_state("eq_active")["eq_active"]
        assert int(act.sum()) == 2 * m.neq, name
        env.close()


def test_the_fourteen_part_bookcase_matches_the_oracle_env_from_a_laid_out_start():
    """The largest model of SURVEY section 8's size table stepped on the device: bookcase_grevback_0484 (93 dofs: the island solver's four
    16-lane rows filled twice; 128 contact slots: two per lane, kernels `generic2`) from a start in which the fourteen planks lie flat
    next to each other (set_init_qpos, furniture.py:315-316, 1505-1519 -- tests/scenarios.py spread_layout), reset and random-action
    steps against the fp64 oracle env: 56 plank-floor contacts throughout."""
    from furniture_amd.envs import FurnitureSawyerEnv, make_config
    from furniture_amd.mjcf.model import load_compiled
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    from tests.scenarios import spread_layout
    m = load_compiled("Sawyer", "bookcase_grevback_0484")
    lay = spread_layout(m)
    q = np.array(m.qpos0, dtype=float)
    q[m.arm_qposadr], q[m.grip_qposadr] = m.arm_initqpos, m.grip_initqpos
    for p in range(m.nparts):
        q[m.part_qposadr[p]:m.part_qposadr[p] + 7] = lay[p]
    init = {"qpos": q, "qvel": np.zeros(m.nv)}
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="bookcase_grevback_0484", max_episode_steps=50, seed=3)
    env = FurnitureSawyerEnv(make_config(**kw))
    orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=3, solver_tolerance=1e-10))
    env.set_init_qpos(init), orc.set_init_qpos(init)
    o = orc.flat_obs(orc.reset())
    d = env.reset()
    assert len(orc.sim.contacts()) == 56
    assert np.abs(np.concatenate([d["object_ob"], d["robot_ob"]]) - o).max() < 5e-4
    rng = np.random.RandomState(2)
    for t in range(4):
        a = rng.uniform(-1, 1, 9)
        ob, r, done, info = env.step(a)
        ob_o, r_o, done_o, _ = orc.step(a)
        assert np.abs(np.concatenate([ob["object_ob"], ob["robot_ob"]]) - orc.flat_obs(ob_o)).max() < 1e-3, t
        assert abs(r - r_o) < 1e-4 and done == done_o
        assert int(info["contact_overflow"]) == 0
    env.close()
