"""No result may depend on what the LDS held before the kernel started.  Work arrays live in LDS and do not survive a launch; a code path
that reads one before this launch wrote it (round 5: the Cursor agent's _try_connect read the part poses of "the last forward pass" in a step
in which no forward pass had run yet) returns whatever the CU's previous tenant left -- run-to-run nondeterminism that a fresh process on an
idle GPU never shows.  Here every CU's LDS is filled with NaNs, then with zeros, before each launch (tests/lds_poison.hip): the two runs
must be bit-identical."""
import ctypes
import os

import numpy as np
import pytest
import torch

from furniture_amd.envs import ResetTableSampler, make_config
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
from tests.scenarios import counter_actions

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    """tests/liblds_poison.so: built by __graft_entry__.build(); compiled on the spot (hipcc, a few seconds) in a tree that does not have it"""
    so = os.path.join(ROOT, "tests", "liblds_poison.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "lds_poison.hip")])
    return ctypes.CDLL(so)


def _run(agent, furniture, n, steps, pattern, tool, control="impedance", dense=False):
    from furniture_amd.envs import CONTROLLER_CODES
    m = load_compiled(agent, furniture, control)
    ecfg = make_config(unity=False, record_vid=False, furniture_name=furniture, max_episode_steps=4, seed=200)
    tabs = ResetTableSampler(m, ecfg, 200, 0, n)
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset = 4, 1  # in-kernel resets and look-ahead jobs inside the run
    cfg.control_type, cfg.dense_reward = CONTROLLER_CODES.get(control, 0), 1 if dense else 0

    def poison():
        for _ in range(3):  # (more workgroups than one pass places on every CU)
            assert tool.lds_poison(ctypes.c_uint(pattern)) == 0
    sim = FSim(m, n, config=cfg)
    if dense:
        from furniture_amd.dense import pack_dense
        sim.set_dense_reward(*pack_dense(m))
    p, nz = tabs.draw()
    sim.set_reset_tables(p, nz if agent != "Cursor" else None)
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    poison()
    sim.reset(None, obs)
    sim.sync()
    p, nz = tabs.draw()
    sim.set_reset_tables(p, nz if agent != "Cursor" else None)
    dof = sim.dof_action
    act, rew = torch.zeros((n, dof), device=dev), torch.zeros(n, device=dev)
    done, info = torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    out = [obs.cpu().numpy().copy()]
    for t in range(steps):
        a = np.stack([counter_actions(1, i, t, dof) for i in range(n)])  # (keyed by (env, t): the same rows whatever the batch size)
        if agent == "Cursor":  # select mostly on: parts get picked up, carried against the boundary, connected
            for k in (6, 13):
                a[:, k] = np.abs(a[:, k]) * np.where(np.abs(a[:, k - 1]) < 0.8, 1, -1)
        act.copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        poison()
        sim.step(act, obs, rew, done, info)
        sim.sync()
        out.append(np.concatenate([obs.cpu().numpy().view(np.uint32), rew.cpu().numpy().view(np.uint32)[:, None], info.cpu().numpy()[:, :12].astype(np.uint32)], axis=1))
        need = info[:, 7].cpu().numpy() > 0
        if need.any():
            p, nz = tabs.draw(need)
            sim.set_reset_tables(p, nz if agent != "Cursor" else None, mask=need)
    kernel = sim.step_kernel
    sim.close()
    return out, kernel


@pytest.mark.parametrize("agent,furniture,mw", [("Cursor", "toy_table", None), ("Cursor", "toy_table", "0"), ("Cursor", "table_lack_0825", None), ("Sawyer", "table_lack_0825", None),
                                                ("Sawyer", "table_lack_0825", "all"), ("Baxter", "desk_mikael_1064", None), ("Sawyer", "toy_table", None)])
def test_results_do_not_depend_on_what_the_lds_held(agent, furniture, mw, monkeypatch):
    tool = _tool()
    if mw is not None:
        monkeypatch.setenv("FSIM_MW", mw)
    a, kernel = _run(agent, furniture, 32, 9, 0x7FC00000, tool)
    b, _ = _run(agent, furniture, 32, 9, 0, tool)
    for t, (x, y) in enumerate(zip(a, b)):
        bad = np.nonzero((x.view(np.uint32) != y.view(np.uint32)).any(axis=1))[0]
        assert len(bad) == 0, "%s: step %d, envs %s differ between an LDS full of NaNs and an LDS full of zeros" % (kernel, t - 1, bad[:8].tolist())


@pytest.mark.parametrize("control,dense,mw_k", [("impedance", False, "0"), ("impedance", True, None), ("ik", False, None), ("ik_quaternion", True, None), ("position_orientation", False, None),
                                                ("joint_torque", False, None)])
def test_other_control_paths_do_not_depend_on_what_the_lds_held(control, dense, mw_k, monkeypatch):
    """(Sawyer + table_lack_0825: every env on a four-wave team -- the rule at 0 iterations --, the dense-reward env, IK control, two of the
    torque-level controllers)"""
    tool = _tool()
    if mw_k is not None:
        monkeypatch.setenv("FSIM_MW", "1")
        monkeypatch.setenv("FSIM_MW_K", mw_k)
    a, kernel = _run("Sawyer", "table_lack_0825", 32, 9, 0x7FC00000, tool, control, dense)
    b, _ = _run("Sawyer", "table_lack_0825", 32, 9, 0, tool, control, dense)
    for t, (x, y) in enumerate(zip(a, b)):
        bad = np.nonzero((x.view(np.uint32) != y.view(np.uint32)).any(axis=1))[0]
        assert len(bad) == 0, "%s / %s: step %d, envs %s differ between an LDS full of NaNs and an LDS full of zeros" % (kernel, control, t - 1, bad[:8].tolist())


@pytest.mark.parametrize("agent,furniture", [("Cursor", "toy_table"), ("Baxter", "desk_mikael_1064"), ("Sawyer", "chair_bertil_0148")])
def test_an_envs_bits_do_not_depend_on_the_batch_it_is_stepped_in(agent, furniture):
    """(tests/test_determinism_gpu.py holds this for the benchmark model; here the Cursor agent, Baxter and a mesh furniture on the generic kernels, with
    auto-resets inside the run and a differently filled LDS on the two sides)  Env i is seeded seed + i and its actions are keyed by (i, t): the first
    eight envs of a batch of 40 and a batch of 8 are bit-identical."""
    tool = _tool()
    a, _ = _run(agent, furniture, 40, 9, 0, tool)
    b, _ = _run(agent, furniture, 8, 9, 0x7FC00000, tool)
    for t, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x[:8].view(np.uint32), y.view(np.uint32)), (t - 1, np.nonzero((x[:8].view(np.uint32) != y.view(np.uint32)).any(axis=1))[0].tolist())
