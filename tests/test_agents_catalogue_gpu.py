"""The reference's one integration test resets ALL 64 furniture on BAXTER (furniture/tests/test_furniture_init.py:14-55: instantiate the
scene, reset, no RandomizationError).  Here: every furniture compiled for the Baxter and the Cursor agents (scripts/compile_assets.py
--all --agents Sawyer,Baxter,Cursor; the Sawyer sweep is tests/test_all_furniture_gpu.py) runs reset + random-action steps across an
in-kernel auto-reset on the device -- finite observations, parts on the floor, no dropped contacts -- with an explicit exception list per
agent, each entry with its reason; plus device-vs-oracle-env resets on three Baxter models that are not desk_mikael."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# Baxter: one furniture does not compile -- 19 moving robot bodies + 14 parts exceed the 31 moving bodies (+ the world body) that a 32-bit subtree
# mask holds (furniture_amd/mjcf/reduce.py), and its 3 + 14 kinematic trees the 16 of the island bookkeeping (tried at the end of round 6 with 64-bit masks
# in the kernels with several slot sets: the masks compile and run, fsim_create then stops at 17 trees; widening the per-tree tables of every kernel's LDS
# image for one model of 192 was not done).  table_liden_0921 (12 parts: 32 bodies with the world, bit 31 in use) compiles since round 6.
NOT_COMPILED = {"Baxter": {"bookcase_grevback_0484"}, "Cursor": set()}
# the reference's own UniformRandomSampler raises RandomizationError for these with the default jitter (checked by running it in round 4)
UNPLACEABLE = {"bookcase_grevback_0484", "cabinet_akurum_0021", "table_hemnes_0539"}
# resets that drop contacts: none since round 6 (bookcase_billy_0191 and table_liden_0921 pass through 240-270 simultaneous contacts while the planks
# are thrown apart -- beyond 128 slots; the re-step ladder's last rung, 512 slots, takes them: tests/test_overflow_restep_gpu.py)
OVERFLOWS = {"Baxter": set(), "Cursor": set()}


def sweep(agent):
    import torch
    from furniture_amd.envs import ContactOverflowError, make_vec_env
    from furniture_amd.mjcf.model import _COMPILED_DIR, load_compiled
    from furniture_amd.sim import FsimError

    names = sorted(os.path.basename(p)[len(agent) + 2:-len("__vel.npz")] for p in glob.glob(os.path.join(_COMPILED_DIR, agent + "__*__vel.npz")))
    ran, refused, unplaceable, troubled = [], [], [], []
    for name in names:
        m = load_compiled(agent, name)
        try:
            env = make_vec_env(agent, 4, furniture_name=name, max_episode_steps=3, seed=11, record_vid=False, unity=False, control_type="impedance")
        except FsimError as e:
            refused.append((name, str(e)))
            continue
        try:
            ob = env.reset()
        except ContactOverflowError:
            troubled.append((name, m.nparts, 2))
            env.close()
            continue
        except RuntimeError as e:
            assert "Cannot place all objects" in str(e), (name, str(e))
            unplaceable.append(name)
            env.close()
            continue
        assert ob["object_ob"].shape == (4, 7 * m.nparts)
        g = torch.Generator(device=env.sim.device)
        g.manual_seed(1)
        trouble = 0
        for t in range(4):  # crosses an in-kernel auto-reset (max_episode_steps = 3)
            a = torch.empty((4, env.dof), device=env.sim.device).uniform_(-1, 1, generator=g)
            if agent == "Cursor":
                a[:, [6, 13, 14]] = -1.0  # ((move, rotate, select) x 2 + connect, furniture_cursor.py:56: no select / connect requests -- this sweep is about reset + physics; the Cursor logic has its own tests)
            try:
                ob, rew, done, info = env.step(a)
            except ContactOverflowError:
                trouble = 2
                break
            fin = all(bool(torch.isfinite(v).all()) for v in ob.values()) and bool(torch.isfinite(rew).all())
            assert fin, (name, t)
            trouble |= int(info["fail"].max()) | (int(info["contact_overflow"].max()) << 1)
            if not trouble:
                assert bool(done.all()) == (t == 2), (name, t)
        if trouble:
            troubled.append((name, m.nparts, trouble))
            env.close()
            continue
        z = ob["object_ob"].reshape(4, m.nparts, 7)[:, :, 2]
        assert float(z.min()) > -0.01 and float(ob["object_ob"].reshape(4, m.nparts, 7)[:, :, :3].abs().max()) < 10.0, name
        env.close()
        ran.append(name)
    return names, ran, refused, unplaceable, troubled


@pytest.mark.parametrize("agent", ["Baxter", "Cursor"])
def test_every_compiled_furniture_resets_and_steps(agent):
    names, ran, refused, unplaceable, troubled = sweep(agent)
    print("%s: ran %d of %d; refused %s; placement sampler gives up on %s; overflowed or failed (name, parts, fail | overflow << 1): %s" % (
        agent, len(ran), len(names), refused, unplaceable, troubled))
    assert len(names) == 64 - len(NOT_COMPILED[agent])
    assert refused == []
    assert set(unplaceable) <= UNPLACEABLE, unplaceable
    assert {x[0] for x in troubled} <= OVERFLOWS[agent], troubled
    assert len(ran) >= len(names) - len(UNPLACEABLE) - len(OVERFLOWS[agent])


@pytest.mark.parametrize("furniture", ["table_lack_0825", "bench_bjoderna_0208", "chair_agne_0007"])
def test_baxter_reset_matches_the_oracle_env(furniture):
    """device vs fp64 oracle env, Baxter + three furniture that are not desk_mikael: the reset (401 substeps, two arms, the pedestal's
    capsule) and two random-action steps"""
    from furniture_amd.envs import FurnitureBaxterEnv, make_config
    from furniture_amd.mjcf.model import load_compiled
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Baxter", furniture)
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name=furniture, max_episode_steps=50, seed=3)
    env = FurnitureBaxterEnv(make_config(**kw))
    orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=3, solver_tolerance=1e-10))
    o = orc.flat_obs(orc.reset())
    d = env.reset()
    assert np.abs(np.concatenate([d["object_ob"], d["robot_ob"]]) - o).max() < 5e-4
    rng = np.random.RandomState(2)
    for t in range(2):
        a = rng.uniform(-1, 1, env.dof)
        ob, r, done, info = env.step(a)
        ob_o, r_o, done_o, _ = orc.step(a)
        assert np.abs(np.concatenate([ob["object_ob"], ob["robot_ob"]]) - orc.flat_obs(ob_o)).max() < 2e-3, t
        assert abs(r - r_o) < 1e-4 and done == done_o
    env.close()


def test_baxter_with_the_last_body_on_bit_31_resets_like_the_oracle_env():
    """Baxter + table_liden_0921: 19 robot bodies + 12 parts + the world = 32 reduced bodies, the most a 32-bit subtree mask holds (the last
    part sits on bit 31: every mask loop of the kernels is sign-agnostic -- `mm &= mm - 1`, `__ffs`).  91 dofs, 128 contact slots; the reset
    starts with the parts inside each other (268 contacts, counted with the oracle) and takes the re-step ladder's last rung; against the fp64
    oracle env: the robot within 5e-4, the observation in the median (parts still moving when the reset ends: tests/test_overflow_restep_gpu.py)."""
    from furniture_amd.envs import FurnitureBaxterEnv, make_config
    from furniture_amd.mjcf.model import load_compiled
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Baxter", "table_liden_0921")
    assert len(m.r_ancmask) == 32 and int(m.r_ancmask[-1]) < 0 and m.nv == 91
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="table_liden_0921", max_episode_steps=50, seed=3)
    env = FurnitureBaxterEnv(make_config(**kw))
    orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=3, solver_tolerance=1e-10))
    o = orc.flat_obs(orc.reset())
    d = env.reset()
    assert env._b.sim.overflow_resteps() >= 1
    got = np.concatenate([d["object_ob"], d["robot_ob"]])
    dd = np.abs(got - o)
    assert np.isfinite(got).all() and np.median(dd) < 5e-4 and dd[7 * m.nparts:].max() < 5e-4, (float(np.median(dd)), float(dd.max()))
    print("Baxter + table_liden_0921 reset vs oracle env: max %.2e median %.1e" % (dd.max(), np.median(dd)))
    rng = np.random.RandomState(2)
    for t in range(2):
        a = rng.uniform(-1, 1, env.dof)
        ob, r, done, info = env.step(a)
        ob_o, r_o, done_o, _ = orc.step(a)
        g = np.concatenate([ob["object_ob"], ob["robot_ob"]])
        assert np.isfinite(g).all() and int(info["contact_overflow"]) == 0 and done == done_o
        assert np.median(np.abs(g - orc.flat_obs(ob_o))) < 1e-3, t
    env.close()
