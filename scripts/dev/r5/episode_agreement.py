"""development: device (libfsim.so) against the native fp64 checker (oracle/libfsim_cpu.so) over whole episodes with auto-resets, through the
one C-ABI session of tests/abi_session.py.  Prints, per step, how many envs agree to 1e-3 / 1e-2 on the observation, the reward and done
agreement, and the agreement right after every auto-reset (both sides restart from the same table: the envs re-synchronise)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from furniture_amd.envs import ResetTableSampler, make_config
from furniture_amd.mjcf.model import load_compiled
from tests.abi_session import Abi, Session, CPU_LIB, GPU_LIB
from tests.scenarios import counter_actions

n, T, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
agent, furn = (sys.argv[4], sys.argv[5]) if len(sys.argv) > 5 else ("Sawyer", "table_lack_0825")
m = load_compiled(agent, furn)
ecfg = make_config(unity=False, record_vid=False, furniture_name=furn, max_episode_steps=T, seed=77)
tabs = ResetTableSampler(m, ecfg, 77, 0, n)
pair = [Session(Abi(GPU_LIB, torch.device("cuda:0")), m.to_blob(), n, max_episode_steps=T, auto_reset=1),
        Session(Abi(CPU_LIB), m.to_blob(), n, max_episode_steps=T, auto_reset=1)]
t0 = tabs.draw()
for s in pair:
    s.set_reset_tables(*t0)
og, oc = [s.reset() for s in pair]
print("reset: max |d obs| %.2e" % np.abs(og - oc).max())
t1 = tabs.draw()
for s in pair:
    s.set_reset_tables(*t1)
age = np.zeros(n, dtype=int)
for t in range(steps):
    a = np.stack([counter_actions(5, i, t, pair[0].dof) for i in range(n)])
    (og, rg, dg, ig), (oc, rc, dc, ic) = [s.step(a) for s in pair]
    d = np.abs(og - oc).max(axis=1)
    npart = 7 * m.nparts
    dp, dr = np.abs(og - oc)[:, :npart].max(axis=1), np.abs(og - oc)[:, npart:].max(axis=1)
    age += 1
    fresh = dg.astype(bool)
    line = "t %3d  obs<1e-3 %4d  <1e-2 %4d  max %.2e  rew eq %4d  done eq %s  ndone %d" % (t, (d < 1e-3).sum(), (d < 1e-2).sum(), d.max(), (np.abs(rg - rc) < 1e-4).sum(), np.array_equal(dg, dc), dg.sum())
    line += "  parts<1e-3 %4d robot<1e-3 %4d robot max %.1e" % ((dp < 1e-3).sum(), (dr < 1e-3).sum(), dr.max())
    if fresh.any():
        line += "  after reset: max %.2e" % d[fresh].max()
    if not np.array_equal(ig[:, [1, 2, 7]], ic[:, [1, 2, 7]]):
        line += "  info(success, fail, needs_table) differ in %d envs" % (ig[:, [1, 2, 7]] != ic[:, [1, 2, 7]]).any(axis=1).sum()
    print(line)
    if os.environ.get("DETAIL") and t % 30 in (10, 20, 28):
        D = np.abs(og - oc)[:, npart:]
        names = ["jpos"] * 7 + ["jvel"] * 7 + ["grip"] * 2 + ["eefp"] * 3 + ["quat"] * 4 + ["velp"] * 3 + ["velr"] * 3
        bad = np.nonzero(dr >= 1e-3)[0]
        for e in bad[:12]:
            grp = {}
            for k, nm in enumerate(names):
                grp[nm] = max(grp.get(nm, 0), D[e, k])
            st = pair[0].get_state(m, "ncon") if False else None
            print("    env %3d " % e + " ".join("%s %.1e" % kv for kv in grp.items()) + "  jpos argmax %d  parts %.1e" % (int(D[e, :7].argmax()), dp[e]))
    need = ig[:, 7] > 0
    if need.any() or (ic[:, 7] > 0).any():
        need = need | (ic[:, 7] > 0)
        p, nz = tabs.draw(need)
        for s in pair:
            s.set_reset_tables(p, nz, mask=need)
for s in pair:
    s.close()
