"""k_env_step_x's deferred resets: a multi-wave env whose step ends its episode with no look-ahead record ready has its reset run by wave 0
of the workgroup as ONE-wave work after the multi-wave phase.  The list is the workgroup's row of a device array and holds any number of
deferrals (before round 5: four env indices in registers, and a workgroup whose list was full left the queue for good -- with every
workgroup full the rest of the queue was never stepped, their rows stayed stale, silently).  Forced here: every env on the multi-wave queue (rule at 0 iterations), 64 envs on 4
workgroups, every env unstable in the same step (furniture.py:2889-2897), no look-ahead -- 64 deferrals against a capacity of 16."""
import numpy as np
import pytest
import torch

from furniture_amd.envs import ResetTableSampler, make_config
from furniture_amd.sim import FSim, INFO_DIM, INFO_FAIL, INFO_NEEDS_TABLE, default_config

pytestmark = pytest.mark.gpu


def _run(m, n, mw, monkeypatch):
    monkeypatch.setenv("FSIM_MW", mw)
    monkeypatch.setenv("FSIM_MW_K", "0")
    monkeypatch.setenv("FSIM_X_GRID", "4")
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset, cfg.lookahead_reset = 100, 1, 0
    sim = FSim(m, n, config=cfg)
    ecfg = make_config(unity=False, record_vid=False, furniture_name="table_lack_0825", max_episode_steps=100, seed=9)
    tabs = ResetTableSampler(m, ecfg, 9, 0, n).draw()
    sim.set_reset_tables(*tabs)
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    rew, done, info = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    sim.reset(None, obs)
    sim.sync()
    obs0 = obs.clone()
    sim.set_reset_tables(*tabs)
    act = torch.zeros((n, 9), device=dev)
    sim.step(act, obs, rew, done, info)  # (gives every env an iteration count: the rule then puts all of them on the multi-wave queue)
    sim.sync()
    assert not done.any()
    name = sim.step_kernel
    qv = sim.get_state("qvel")["qvel"].clone()
    qv[:] = float("nan")
    sim.set_state(qvel=qv)
    obs.fill_(-7.0)
    done.fill_(9)
    sim.step(act, obs, rew, done, info)
    sim.sync()
    out = (obs.clone(), rew.clone(), done.clone(), info.clone())
    mw_steps = int((sim.get_state("env_block")["env_block"][:, 36] > 0).sum())
    sim.step(act, obs, rew, done, info)
    sim.sync()
    assert not done.any() and torch.isfinite(obs).all()
    sim.close()
    return obs0, out, name, mw_steps


def test_more_deferred_resets_than_the_workgroups_hold(sawyer_lack, monkeypatch):
    n = 64
    obs0, (obs, rew, done, info), name, _ = _run(sawyer_lack, n, "1", monkeypatch)
    assert "k_env_step_x" in name
    assert (done == 1).all(), "envs left on the multi-wave queue: %s" % (done != 1).nonzero().flatten().tolist()
    assert (info[:, INFO_FAIL] == 1).all() and (info[:, INFO_NEEDS_TABLE] == 2).all()
    assert torch.allclose(rew, torch.full_like(rew, -100.0)) and torch.isfinite(obs).all()
    # the reset ran from the same table as the first one: the same observation up to the one integration step fsim.h documents for the
    # end-effector words of a failed step
    assert (obs[:, :35] - obs0[:, :35]).abs().max() < 2e-4
    # a reset is one-wave arithmetic wherever it runs: the one-wave kernel's rows are the same bits
    _, (obs1, rew1, done1, info1), name1, _ = _run(sawyer_lack, n, "0", monkeypatch)
    assert "k_env_step_x" not in name1
    assert torch.equal(obs, obs1) and torch.equal(done, done1) and torch.equal(rew, rew1)
