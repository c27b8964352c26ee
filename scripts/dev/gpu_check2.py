import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config
from oracle.oracle_sim import OracleSim
m = load_compiled("Sawyer", "table_lack_0825")
N = 512
rng = np.random.RandomState(1)
def initial(n, arm_noise):
    q = np.tile(m.qpos0, (n, 1))
    q[:, m.arm_qposadr] = m.arm_initqpos + rng.uniform(-arm_noise, arm_noise, (n, len(m.arm_qposadr)))
    q[:, m.grip_qposadr] = m.grip_initqpos
    for i in range(m.nparts):
        a = m.part_qposadr[i]
        q[:, a:a + 7] = m.part_initqpos[i]
        q[:, a:a + 2] += rng.uniform(-0.02, 0.02, (n, 2))
        q[:, a + 2] += 0.01
    return q
cfg = default_config()
sim = FSim(m, N, config=cfg)
q0 = initial(N, 0.3)
sim.set_state(qpos=q0, qvel=np.zeros((N, m.nv)))
sim.physics_forward()
st = sim.get_state("qacc", "ncon", "solver_iters", "contact_geoms")
qacc = st["qacc"].cpu().numpy(); ncon = st["ncon"].cpu().numpy()[:, 0]; it = st["solver_iters"].cpu().numpy()[:, 0]
print("forward: nan envs", np.isnan(qacc).any(axis=1).sum(), "ncon hist", np.bincount(ncon), "iters hist", np.bincount(it))
orc = OracleSim(m); orc.set_solver(100, 1e-10, "newton")
worst = []
gn = m.meta["geom_names"]
for e in range(N):
    if ncon[e] <= 1 and not np.isnan(qacc[e]).any(): continue
    orc.reset(); orc.data.qpos[:] = q0[e]; orc.forward()
    err = np.abs(qacc[e] - orc.data.qacc).max() / (1 + np.abs(orc.data.qacc).max())
    cg = st["contact_geoms"][e].cpu().numpy().reshape(-1, 2); cg = [tuple(int(v) for v in x) for x in cg[cg[:, 0] >= 0]]
    worst.append((err if not np.isnan(err) else 1e9, e, ncon[e], orc.ncon, it[e], orc.last_solver_iters, cg, orc.contacts()))
worst.sort(key=lambda x: -x[0])
for w in worst[:8]:
    print("env %d relerr %.2e ncon %d/%d iters %d/%d" % (w[1], w[0], w[2], w[3], w[4], w[5]))
    print("   gpu:", [(gn[a], gn[b]) for a, b in w[6]]); print("   orc:", [(gn[a], gn[b]) for a, b in w[7]])
print("n compared", len(worst), "median relerr", np.median([w[0] for w in worst]) if worst else None)
# dynamics: 100 substeps, track which envs go NaN first
sim.physics_forward()
bias = sim.get_state("qfrc_bias")["qfrc_bias"].cpu().numpy()
app = np.zeros((N, m.nv)); rd = np.concatenate([m.arm_dofadr, m.grip_dofadr]); app[:, rd] = bias[:, rd]
sim.set_state(qfrc_applied=app)
bad_first = {}
for k in range(20):
    sim.physics_step(5)
    s = sim.get_state("qpos", "qvel", "solver_iters", "ncon")
    qv = s["qvel"].cpu().numpy(); bad = np.where(np.isnan(qv).any(axis=1) | (np.abs(qv).max(axis=1) > 50))[0]
    for e in bad:
        if e not in bad_first: bad_first[int(e)] = (k + 1) * 5
print("bad envs (first substep):", dict(list(bad_first.items())[:10]), "count", len(bad_first))
for e in list(bad_first)[:3]:
    o = OracleSim(m); o.set_solver(100, 1e-10, "newton"); o.reset(); o.data.qpos[:] = q0[e]; o.forward(); o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]
    try:
        for k in range(100):
            o.step()
        print("oracle env", e, "ok: max|qvel|", np.abs(o.data.qvel).max(), "ncon", o.ncon, [(gn[a], gn[b]) for a, b in o.contacts() if a != 0][:6])
    except Exception as ex:
        print("oracle env", e, "unstable too:", ex, "after", k)
# clean timing
big = FSim(m, 4096, config=cfg)
qb = initial(4096, 0.0)
big.set_state(qpos=qb, qvel=np.zeros((4096, m.nv)))
big.physics_forward(); bias = big.get_state("qfrc_bias")["qfrc_bias"]; app = torch.zeros_like(bias); idx = torch.as_tensor(rd, device=bias.device).long(); app[:, idx] = bias[:, idx]
big.set_state(qfrc_applied=app)
for rep in range(4):
    torch.cuda.synchronize(); t = time.time(); big.physics_step(50); big.sync(); dt = time.time() - t
    s = big.get_state("ncon", "solver_iters", "qvel")
    print("rep", rep, "%.2f ms per 50 substeps -> %.0f env-steps/s" % (dt * 1e3, 4096 / dt), "mean ncon %.1f mean iters(last) %.2f nan %s" % (float(s["ncon"].float().mean()), float(s["solver_iters"].float().mean()), bool(torch.isnan(s["qvel"]).any())))
