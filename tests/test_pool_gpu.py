"""Shared work pool (include/fsim.h fsim_pool_*, csrc/fsim.hip k_pool): several handles of one model post their steps to ONE resident
kernel instead of launching a scheduler + step kernel each.  The claim under test: a pooled handle's observations, rewards, done flags,
info words and full env records are BIT-IDENTICAL to the same handle stepped on launches of its own -- multi-wave envs, deferred
resets of multi-wave envs, look-ahead resets and resets inside the step included -- and nothing depends on how the members' steps
interleave.  (Reference: the workers of furniture/env/base.py:55-80 step their envs independently; results do not depend on the
other workers.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SNAP = ["qpos", "qvel", "qacc_warmstart", "qfrc_bias", "ctrl", "qfrc_applied", "xfrc_applied", "eq_active", "eq_data", "geom_contype", "geom_conaffinity", "group", "env_block"]


def _make(n, first, T, la, **kw):
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    cfg = make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=T, seed=11,
                      lookahead_reset=la, multi_wave="rule", **kw)
    return FurnitureBatchEnv("Sawyer", n, config=cfg, first_env_index=first)


def _run(sizes, T, nsteps, la, pooled, order=None, seed=4):
    """len(sizes) batch envs over consecutive global env indices, stepped `nsteps` times with step_async on all, then step_wait on all
    (in `order`).  Returns per env: observations, (reward, done, info) per step, final records, look-ahead stats."""
    from furniture_amd.sim import FSimPool
    envs, first = [], 0
    for n in sizes:
        envs.append(_make(n, first, T, la))
        first += n
    pool = None
    if pooled:
        pool = FSimPool()
        for e in envs:
            pool.attach(e.sim)
    dev = envs[0].sim.device
    gens = []
    for i, e in enumerate(envs):
        g = torch.Generator(device=dev)
        g.manual_seed(seed + i)
        gens.append(g)
    obs = [[{k: v.clone() for k, v in e.reset().items()}] for e in envs]
    trace = [[] for _ in envs]
    for t in range(nsteps):
        acts = [torch.empty((e.num_envs, e.dof), device=dev).uniform_(-1, 1, generator=g) for e, g in zip(envs, gens)]
        torch.cuda.synchronize()
        for e, a in zip(envs, acts):
            e.step_async(a)
        for i in (order or range(len(envs))):
            e = envs[i]
            ob, rew, done, info = e.step_wait()
            obs[i].append({k: v.clone() for k, v in ob.items()})
            trace[i].append((rew.clone(), done.clone(), e._info.clone()))
    st = pool.stats() if pool else None
    snaps = [{k: v.clone() for k, v in e.sim.get_state(*SNAP).items()} for e in envs]
    stats = [e.sim.lookahead_stats() for e in envs]
    mw_steps = [int(e.sim.get_state("env_block")["env_block"][:, 36].sum()) for e in envs]  # E_MW_STEPS of the running episodes
    if pool:
        pool.close()
    for e in envs:
        e.close()
    return obs, trace, snaps, stats, st, mw_steps


def _same(a, b):
    for i in range(len(a[0])):
        for t, (x, y) in enumerate(zip(a[0][i], b[0][i])):
            for k in x:
                assert torch.equal(x[k], y[k]), ("observation", i, t, k, float((x[k].float() - y[k].float()).abs().max()))
        for t, ((r1, d1, i1), (r2, d2, i2)) in enumerate(zip(a[1][i], b[1][i])):
            assert torch.equal(r1, r2) and torch.equal(d1, d2), ("reward / done", i, t)
            # (the scheduler key is a cycle count: not part of the info block; every info word must agree)
            assert torch.equal(i1, i2), ("info", i, t, (i1 != i2).nonzero()[:4].tolist())
        for k in a[2][i]:
            assert torch.equal(a[2][i][k], b[2][i][k]), ("record", i, k, (a[2][i][k] != b[2][i][k]).nonzero()[:4].tolist())


def test_pooled_handles_are_bit_identical_to_handles_on_their_own_launches(monkeypatch):
    # multi-wave rule at 51 Newton iterations per step (one per substep is 50): a good part of the envs are stepped by four-wave teams, and with episodes of
    # four steps and no look-ahead every terminal step of such an env defers its reset to a one-wave worker
    monkeypatch.setenv("FSIM_MW_K", "51")
    sizes, T, nsteps = [40, 24, 64], 4, 11
    ref = _run(sizes, T, nsteps, False, pooled=False)
    got = _run(sizes, T, nsteps, False, pooled=True, order=[2, 0, 1])
    assert got[4]["members"] == 3 and got[4]["launches"] >= 1, got[4]
    assert sum(ref[5]) > 0 and ref[5] == got[5], (ref[5], got[5])  # four-wave steps happened, the same ones
    _same(ref, got)


def test_pooled_lookahead_resets_are_bit_identical(monkeypatch):
    monkeypatch.setenv("FSIM_LA_DEFER", "0")
    monkeypatch.setenv("FSIM_LA_JOBS", "64")
    monkeypatch.setenv("FSIM_LA_CHUNK", "51")
    monkeypatch.setenv("FSIM_POOL_JOB_AGE_MS", "1000")  # (tiny batches: the epochs are short, every listed job may start)
    sizes, T, nsteps = [24, 24], 12, 26
    ref = _run(sizes, T, nsteps, False, pooled=False)          # resets inside the step launch
    got = _run(sizes, T, nsteps, True, pooled=True)            # look-ahead jobs run by the pool's one-wave workers
    assert all(s["enabled"] == 1 and s["swapped"] > 0 for s in got[3]), got[3]
    _same(ref, got)


def test_pool_refuses_what_it_cannot_serve():
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    from furniture_amd.sim import FSimPool, FsimError
    a = _make(8, 0, 10, False)
    b = FurnitureBatchEnv("Sawyer", 8, config=make_config(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825",
                                                         max_episode_steps=11, seed=11, multi_wave="rule"))
    pool = FSimPool()
    pool.attach(a.sim)
    with pytest.raises(FsimError):  # another configuration
        pool.attach(b.sim)
    with pytest.raises(FsimError):  # twice
        pool.attach(a.sim)
    pool.close()
    a.close(), b.close()
