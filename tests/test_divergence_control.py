"""The two CONTROL trajectories of scripts/divergence_control.py, on the CPU: the checker's own sources compiled in
fp32 (oracle/libfsim_cpu32.so) and the fp64 checker started from a state moved by 1e-7 (FSIM_CPU_PERTURB).  Neither is a checker; they say
what fp32 arithmetic alone, and the system's own sensitivity, do to a trajectory -- the yardstick the device's distance from the fp64
checker is read against (profiles/r06_divergence_control.txt)."""
import os
import subprocess

import numpy as np
import pytest

from furniture_amd.envs import ResetTableSampler, make_config
from tests.abi_session import Abi, Session, CPU_LIB, ROOT
from tests.scenarios import counter_actions

CPU32_LIB = os.path.join(ROOT, "oracle", "libfsim_cpu32.so")


@pytest.fixture(scope="module")
def libs():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libfsim_cpu.so", "libfsim_cpu32.so"])
    return Abi(CPU_LIB), Abi(CPU32_LIB)


def _run(abi, m, n, T, steps, tabs0, tabs1, perturb=None):
    if perturb is not None:
        os.environ["FSIM_CPU_PERTURB"] = perturb
    try:
        ses = Session(abi, m.to_blob(), n, max_episode_steps=T, auto_reset=1)
    finally:
        os.environ.pop("FSIM_CPU_PERTURB", None)
    ses.set_reset_tables(*tabs0)
    out = [(ses.reset(), None, None, None)]
    ses.set_reset_tables(*tabs1)
    for t in range(steps):
        a = np.stack([counter_actions(5, i, t, ses.dof) for i in range(n)])
        out.append(ses.step(a))
    v = ses.variant()
    ses.close()
    return v, out


def test_fp32_build_and_perturbed_twin_follow_the_fp64_checker(libs, sawyer_lack):
    m, n, T, steps = sawyer_lack, 8, 6, 9
    ecfg = make_config(unity=False, record_vid=False, furniture_name="table_lack_0825", max_episode_steps=T, seed=77)
    tabs = ResetTableSampler(m, ecfg, 77, 0, n)
    t0, t1 = tabs.draw(), tabs.draw()
    v64, o64 = _run(libs[0], m, n, T, steps, t0, t1)
    v32, o32 = _run(libs[1], m, n, T, steps, t0, t1)
    v64p, o64p = _run(libs[0], m, n, T, steps, t0, t1, perturb="1e-7")
    assert (v64, v32, v64p) == ("cpu-fp64", "cpu-fp32", "cpu-fp64")
    # the reset: 401 settling substeps in fp32 end within 1e-4 of the fp64 ones; the twin's reset observation differs by the 1e-7 its joint angles
    # were moved by (the poses are those of the reset's last forward pass)
    assert np.abs(o32[0][0] - o64[0][0]).max() < 1e-4
    assert np.abs(o64p[0][0] - o64[0][0]).max() < 5e-7
    moved = False
    for t in range(1, steps + 1):
        for other in (o32, o64p):
            assert np.array_equal(other[t][2], o64[t][2])                                  # done
            assert np.array_equal(other[t][3][:, [0, 1, 2, 5, 6, 7]], o64[t][3][:, [0, 1, 2, 5, 6, 7]])  # connect / success / fail / length / needs-table
        moved = moved or np.abs(o64p[t][0] - o64[t][0]).max() > 0
        # most envs stay together over a handful of steps (the median; single envs part at the first finger-pad contact: DESIGN.md section 5)
        assert np.median(np.abs(o32[t][0] - o64[t][0]).max(axis=1)) < 1e-4
        assert np.median(np.abs(o64p[t][0] - o64[t][0]).max(axis=1)) < 1e-5
    assert moved  # the perturbation is really applied
    # the auto-reset at the time limit (step T) re-synchronises the fp32 build with the fp64 checker: same table, same reset
    assert o64[T][2].all() and np.abs(o32[T][0] - o64[T][0]).max() < 1e-4
