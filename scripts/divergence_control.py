"""Divergence control: is the device's distance from the fp64 checker over a benchmark episode what fp32 arithmetic
and event timing of a hybrid system produce by themselves, or is there something else in it?

Four trajectories of the SAME workload (same reset tables, same actions, whole episodes with auto-resets), all through the one C-ABI
session of tests/abi_session.py:

    o64   oracle/libfsim_cpu.so            the fp64 checker                                           -- the reference line
    dev   furniture_amd/csrc/libfsim.so    the device (fp32, HIP)                       (needs a GPU; left out with --no-device)
    o32   oracle/libfsim_cpu32.so          the checker's own sources compiled with real = float       -- control (a): fp32 alone
    o64'  oracle/libfsim_cpu.so            the fp64 checker from a state moved by 1e-7 after every reset (FSIM_CPU_PERTURB) -- control
                                           (b): how fast the system itself amplifies a difference of the size of one fp32 rounding

and, per step, the SURVIVAL CURVES against o64: how many envs are still within 1e-4 / 1e-3 of it (whole observation, parts only, robot
only).  Verdict line at the end: at how many steps the device's curve lies BELOW the lower of the two controls (by more than the
binomial noise of the env count), and the first such step.

usage: python scripts/divergence_control.py N_ENVS EPISODE_STEPS STEPS [agent furniture] [--no-device] > profiles/r06_divergence_control.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from furniture_amd.envs import ResetTableSampler, make_config  # noqa: E402
from furniture_amd.mjcf.model import load_compiled  # noqa: E402
from tests.abi_session import Abi, Session, CPU_LIB, GPU_LIB  # noqa: E402
from tests.scenarios import counter_actions  # noqa: E402

CPU32_LIB = os.path.join(ROOT, "oracle", "libfsim_cpu32.so")


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    n, T, steps = int(args[0]), int(args[1]), int(args[2])
    agent, furn = (args[3], args[4]) if len(args) > 4 else ("Sawyer", "table_lack_0825")
    use_dev = "--no-device" not in argv
    perturb = "1e-7"
    m = load_compiled(agent, furn)
    ecfg = make_config(unity=False, record_vid=False, furniture_name=furn, max_episode_steps=T, seed=77)
    tabs = ResetTableSampler(m, ecfg, 77, 0, n)
    blob = m.to_blob()
    names, sess = [], []

    def add(name, abi):
        names.append(name)
        sess.append(Session(abi, blob, n, max_episode_steps=T, auto_reset=1))

    add("o64", Abi(CPU_LIB))
    if use_dev:
        import torch
        add("dev", Abi(GPU_LIB, torch.device("cuda:0")))
    add("o32", Abi(CPU32_LIB))
    os.environ["FSIM_CPU_PERTURB"] = perturb  # (read by fsim_create of the CPU library)
    add("o64'", Abi(CPU_LIB))
    del os.environ["FSIM_CPU_PERTURB"]
    others = names[1:]
    print("# divergence control: %s + %s, %d envs, episodes of %d steps, %d steps; actions tests/scenarios.counter_actions(5, env, t)" % (agent, furn, n, T, steps))
    print("# o64 = fp64 checker (reference line); dev = device; o32 = the checker's sources in fp32; o64' = fp64 checker, state + %s after every reset" % perturb)
    print("# columns per trajectory X: envs whose observation is within 1e-4 | 1e-3 of o64 (all / parts / robot at 1e-3), worst env")
    t0 = tabs.draw()
    for s in sess:
        s.set_reset_tables(*t0)
    obs = [s.reset() for s in sess]
    print("reset: " + "  ".join("%s max|d obs| %.2e" % (nm, np.abs(o - obs[0]).max()) for nm, o in zip(others, obs[1:])))
    t1 = tabs.draw()
    for s in sess:
        s.set_reset_tables(*t1)
    npart = 7 * m.nparts
    curves = {nm: [] for nm in others}
    integer_mismatch = {nm: 0 for nm in others}
    for t in range(steps):
        a = np.stack([counter_actions(5, i, t, sess[0].dof) for i in range(n)])
        out = [s.step(a) for s in sess]
        o0, r0, d0, i0 = out[0]
        line = "t %3d" % t
        for nm, (o, r, d, info) in zip(others, out[1:]):
            D = np.abs(o - o0)
            dall, dp, dr = D.max(axis=1), D[:, :npart].max(axis=1), D[:, npart:].max(axis=1)
            c = ((dall < 1e-4).sum(), (dall < 1e-3).sum(), (dp < 1e-3).sum(), (dr < 1e-3).sum())
            curves[nm].append(c)
            eq = np.array_equal(d, d0) and np.array_equal(info[:, [1, 2, 7]], i0[:, [1, 2, 7]])
            integer_mismatch[nm] += 0 if eq else 1
            line += "  | %s <1e-4 %4d <1e-3 %4d (parts %4d robot %4d) max %.1e%s" % (nm, c[0], c[1], c[2], c[3], dall.max(), "" if eq else " INT!")
        if d0.any():
            line += "  | %d resets" % int(d0.sum())
        print(line, flush=True)
        need = np.zeros(n, dtype=bool)
        for (_, _, _, info) in out:
            need |= info[:, 7] > 0
        if need.any():
            p, nz = tabs.draw(need)
            for s in sess:
                s.set_reset_tables(p, nz, mask=need)
    for s in sess:
        s.close()
    # ---- the verdict: the device against the lower of the two controls, per step, at both thresholds
    print("# steps at which done / success / fail / needs-table differ from o64: " + ", ".join("%s %d" % kv for kv in integer_mismatch.items()))
    if use_dev:
        for k, label in ((0, "1e-4"), (1, "1e-3")):
            dev = np.array([c[k] for c in curves["dev"]], dtype=float)
            ctl = np.minimum(np.array([c[k] for c in curves["o32"]], dtype=float), np.array([c[k] for c in curves["o64'"]], dtype=float))
            slack = 2.0 * np.sqrt(np.maximum(ctl * (1 - ctl / n), 1.0))  # two binomial standard deviations of a count out of n
            below = np.nonzero(dev < ctl - slack)[0]
            print("# within %s: device curve below min(o32, o64') - 2 sigma at %d of %d steps%s; mean survivors dev %.1f  o32 %.1f  o64' %.1f" % (
                label, len(below), steps, (" (first: t = %d)" % below[0]) if len(below) else "", dev.mean(),
                np.mean([c[k] for c in curves["o32"]]), np.mean([c[k] for c in curves["o64'"]])))


if __name__ == "__main__":
    main(sys.argv[1:])
