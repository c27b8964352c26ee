"""development: why does the ROBOT's observation drift between the device and the fp64 checker in envs whose parts nobody touches?
Follows a few envs step by step: joint positions against their ranges, contact counts on both sides."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from furniture_amd.envs import ResetTableSampler, make_config
from furniture_amd.mjcf.model import load_compiled
from tests.abi_session import Abi, Session, CPU_LIB, GPU_LIB
from tests.scenarios import counter_actions

n, steps = 256, int(sys.argv[1])
watch = [int(x) for x in sys.argv[2].split(",")]
m = load_compiled("Sawyer", "table_lack_0825")
ecfg = make_config(unity=False, record_vid=False, furniture_name="table_lack_0825", max_episode_steps=30, seed=77)
tabs = ResetTableSampler(m, ecfg, 77, 0, n)
pair = [Session(Abi(GPU_LIB, torch.device("cuda:0")), m.to_blob(), n, max_episode_steps=30, auto_reset=1),
        Session(Abi(CPU_LIB), m.to_blob(), n, max_episode_steps=30, auto_reset=1)]
t0 = tabs.draw()
for s in pair:
    s.set_reset_tables(*t0)
    s.reset()
rng = np.asarray(m.jnt_range).reshape(-1, 2)
lim = np.asarray(m.jnt_limited).reshape(-1)
aq = np.asarray(m.arm_qposadr)
jid = [int(np.nonzero(np.asarray(m.jnt_qposadr) == a)[0][0]) for a in aq]
print("arm joint ranges:", [(round(float(rng[j, 0]), 3), round(float(rng[j, 1]), 3), int(lim[j])) for j in jid])
for t in range(steps):
    a = np.stack([counter_actions(5, i, t, 9) for i in range(n)])
    for s in pair:
        s.step(a)
        s.forward()
    sg = pair[0].get_state(m, "qpos", "qvel", "ncon")
    sc = pair[1].get_state(m, "qpos", "qvel", "ncon")
    if t == 0:
        cg, cc = pair[0].get_state(m, "contact_geoms")["contact_geoms"], pair[1].get_state(m, "contact_geoms")["contact_geoms"]
        gn = m.meta["geom_names"]
        for e in watch[:1]:
            A = sorted((min(a, b), max(a, b)) for a, b in cg[e].reshape(-1, 2) if a >= 0)
            B = sorted((min(a, b), max(a, b)) for a, b in cc[e].reshape(-1, 2) if a >= 0)
            from collections import Counter
            ca, cb = Counter(A), Counter(B)
            print("contacts per geom pair that differ (device, checker):", [(gn[k[0]], gn[k[1]], ca[k], cb[k]) for k in sorted(set(ca) | set(cb)) if ca[k] != cb[k]])
    for e in watch:
        dq = np.abs(sg["qpos"][e][aq] - sc["qpos"][e][aq])
        k = int(dq.argmax())
        q = sc["qpos"][e][aq]
        at = [("L" if q[i] <= rng[jid[i], 0] + 1e-3 else ("U" if q[i] >= rng[jid[i], 1] - 1e-3 else ".")) for i in range(7)]
        print("t %2d env %3d  max dq %.1e (joint %d)  dq %s  limits %s  ncon dev %d cpu %d  act %s" % (
            t, e, dq.max(), k, " ".join("%.0e" % x for x in dq), "".join(at), sg["ncon"][e], sc["ncon"][e], " ".join("%+.2f" % x for x in a[e][:7])))
