"""Look-ahead reset (include/fsim.h fsim_config_t::lookahead_reset, csrc/fsim.hip env_shadow_job): the reset of every env's next
episode is computed ahead of time from the reset table the host has already uploaded -- a chunk of reset substeps per step launch, by
waves of that launch that have no env left to step -- into a shadow record that the terminal step copies in.  The claim under test: the swapped-in record, observation and info words are BIT-IDENTICAL to what the reset inside the
launch leaves (reference flow: furniture/env/furniture.py:1406-1663 run by the vec-env worker on `done`, util/subproc_vec_env.py:15-20),
whether a shadow was ready or not -- i.e. the option changes when the 301 / 401 reset substeps run, never what they compute."""
import numpy as np
import pytest
import torch

from furniture_amd.envs import make_vec_env

pytestmark = pytest.mark.gpu

SNAP = ["qpos", "qvel", "qacc_warmstart", "qfrc_bias", "ctrl", "qfrc_applied", "xfrc_applied", "eq_active", "eq_data", "geom_contype", "geom_conaffinity", "group", "env_block"]


def _make(agent, lookahead, mw, monkeypatch, n=24, steps=3, dense=False, chunk=1000, jobs=64, **kw):
    # (development knobs of the job policy: start the shadows at once, `jobs` envs per launch, `chunk` reset substeps per job -- 1000: the
    #  whole reset in one job, so that the short episodes of this test find the shadows complete)
    monkeypatch.setenv("FSIM_LA_DEFER", "0")
    monkeypatch.setenv("FSIM_LA_JOBS", str(jobs))
    monkeypatch.setenv("FSIM_LA_CHUNK", str(chunk))
    env_id = "IKEASawyerDense-v0" if dense else {"Sawyer": "IKEASawyer-v0", "Cursor": "IKEACursor-v0", "Baxter": "IKEABaxter-v0"}[agent]
    from furniture_amd.envs import FurnitureBatchEnv, make_config, DENSE_OVERRIDES
    over = dict(DENSE_OVERRIDES) if dense else {}
    over.update(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=steps, seed=11,
                lookahead_reset=lookahead, multi_wave=mw)
    over.update(kw)
    return FurnitureBatchEnv(agent, n, config=make_config(**over), dense=dense)


def _run(env, nsteps, sync_shadows, seed=4):
    g = torch.Generator(device=env.sim.device)
    g.manual_seed(seed)
    out = [{k: v.clone() for k, v in env.reset().items()}]
    trace = []
    for t in range(nsteps):
        a = torch.empty((env.num_envs, env.dof), device=env.sim.device).uniform_(-1, 1, generator=g)
        ob, rew, done, info = env.step(a)
        out.append({k: v.clone() for k, v in ob.items()})
        trace.append((rew.clone(), done.clone(), env._info.clone()))
    snap = {k: v.clone() for k, v in env.sim.get_state(*SNAP).items()}
    stats = env.sim.lookahead_stats()
    return out, trace, snap, stats


def _same(a, b):
    oa, ta, sa, _ = a
    ob, tb, sb, _ = b
    for t, (x, y) in enumerate(zip(oa, ob)):
        for k in x:
            assert torch.equal(x[k], y[k]), ("observation", t, k, float((x[k].float() - y[k].float()).abs().max()))
    for t, ((r1, d1, i1), (r2, d2, i2)) in enumerate(zip(ta, tb)):
        assert torch.equal(r1, r2) and torch.equal(d1, d2), ("reward / done", t)
        assert torch.equal(i1, i2), ("info", t, (i1 != i2).nonzero()[:4].tolist())
    for k in sa:
        assert torch.equal(sa[k], sb[k]), ("record", k, (sa[k] != sb[k]).nonzero()[:4].tolist())


@pytest.mark.parametrize("mw", ["off", "rule"])
def test_swapped_in_reset_is_bit_identical_to_the_reset_inside_the_step(monkeypatch, mw):
    n, T, nsteps = 24, 3, 10  # three batch-wide episode ends
    ref = _run(_make("Sawyer", False, mw, monkeypatch, n, T), nsteps, False)
    assert ref[3]["enabled"] == 0 and ref[3]["inline"] >= 3 * n  # (+ the n resets of reset())
    la = _run(_make("Sawyer", True, mw, monkeypatch, n, T), nsteps, True)
    st = la[3]
    assert st["enabled"] == 1 and st["swapped"] == 3 * n, st  # every auto-reset took its shadow record
    assert st["units"] >= 3 * n * 401
    _same(ref, la)
    # the reset in PIECES (the shipped policy: 51 substeps per job, eight jobs per reset) on longer episodes ...
    T, nsteps = 12, 26
    ref = _run(_make("Sawyer", False, mw, monkeypatch, n, T), nsteps, False)
    la = _run(_make("Sawyer", True, mw, monkeypatch, n, T, chunk=51), nsteps, True)
    assert la[3]["swapped"] == 2 * n and la[3]["units_per_job"] == 51, la[3]
    _same(ref, la)
    # ... and with too few jobs per launch for every shadow to be complete in time: some resets swap, the others run inside the step
    # (their half-done shadows are dropped) -- same bits either way
    mixed = _run(_make("Sawyer", True, mw, monkeypatch, n, T, chunk=51, jobs=10), nsteps, False)
    assert 0 < mixed[3]["swapped"] < 2 * n and mixed[3]["swapped"] + mixed[3]["inline"] == ref[3]["inline"], mixed[3]
    _same(ref, mixed)


def test_lookahead_for_the_dense_reward_env_and_the_cursor_agent(monkeypatch):
    for agent, dense in (("Sawyer", True), ("Cursor", False)):
        n, T, nsteps = 8, 2, 5
        ref = _run(_make(agent, False, "off", monkeypatch, n, T, dense=dense), nsteps, False)
        la = _run(_make(agent, True, "off", monkeypatch, n, T, dense=dense), nsteps, True)
        assert la[3]["swapped"] == 2 * n, (agent, la[3])
        _same(ref, la)


def test_reset_call_takes_the_shadow_record_and_new_tables_void_it(monkeypatch):
    """auto_reset off: reset() itself swaps (the table on the device is the env's next draw; its shadow was computed while the episode
    ran), and changing what a reset starts from (set_init_qpos) voids the shadows."""
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    monkeypatch.setenv("FSIM_LA_DEFER", "0")
    monkeypatch.setenv("FSIM_LA_JOBS", "64")
    monkeypatch.setenv("FSIM_LA_CHUNK", "1000")
    mk = lambda la: FurnitureBatchEnv("Sawyer", 6, auto_reset=False, config=make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825",
                                                                                       max_episode_steps=50, seed=3, lookahead_reset=la, multi_wave="off"))
    a, b = mk(True), mk(False)
    g = torch.Generator(device=a.sim.device)
    g.manual_seed(1)
    oa, ob = a.reset(), b.reset()
    for k in oa:
        assert torch.equal(oa[k], ob[k])
    for t in range(3):
        act = torch.empty((6, a.dof), device=a.sim.device).uniform_(-1, 1, generator=g)
        a.step(act), b.step(act)
    before = a.sim.lookahead_stats()
    oa, ob = a.reset(), b.reset()
    st = a.sim.lookahead_stats()
    assert st["swapped"] - before["swapped"] == 6 and st["inline"] == before["inline"], (before, st)
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k
    sa, sb = a.sim.get_state(*SNAP), b.sim.get_state(*SNAP)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    # set_init_qpos: the next reset starts from the given state, whatever shadow was there
    init = {k: v[0].cpu().numpy() for k, v in a.sim.get_state("qpos", "qvel").items()}
    for e in (a, b):
        e.step(act)
        e.set_init_qpos(init)
    oa, ob = a.reset(), b.reset()
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k
    assert torch.equal(a.sim.get_state("qpos")["qpos"], b.sim.get_state("qpos")["qpos"])
    a.close(), b.close()


def test_sticky_overflow_flag_survives_the_steps_between_two_host_reads(monkeypatch):
    """the contact-overflow report must not depend on the step it happens in.  bookcase_billy_0191 (eleven planks placed inside
    each other by the reference's own sampler) overflows the 128 contact slots during reset(): with the re-step ladder switched off (round 6:
    its last rung takes this reset, tests/test_overflow_restep_gpu.py) the reset raises, and with the error downgraded the sticky bits are
    still set many steps later."""
    import os
    monkeypatch.setenv("FSIM_NO_OVERFLOW_REDO", "1")
    from furniture_amd.envs import ContactOverflowError, FurnitureBatchEnv, make_config
    cfg = lambda: make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="bookcase_billy_0191", max_episode_steps=50, seed=1)
    env = FurnitureBatchEnv("Sawyer", 2, config=cfg())
    with pytest.raises(ContactOverflowError):
        env.reset()
    env.close()
    os.environ["FSIM_ALLOW_OVERFLOW"] = "1"
    try:
        env = FurnitureBatchEnv("Sawyer", 2, config=cfg())
        with pytest.warns(UserWarning):
            env.reset()
        for t in range(5):
            ob, rew, done, info = env.step(torch.zeros((2, env.dof), device=env.sim.device))
        assert bool(((env._info[:, 12] >> 8) != 0).all())  # sticky bits, steps after the launch that dropped the contacts
        env.close()
    finally:
        os.environ.pop("FSIM_ALLOW_OVERFLOW", None)
