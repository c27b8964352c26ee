"""Quick GPU-vs-oracle physics diagnostics (development aid; the assertions live in tests/)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config
from oracle.oracle_sim import OracleSim

agent, furn = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("Sawyer", "table_lack_0825")
m = load_compiled(agent, furn)
N = 64
rng = np.random.RandomState(0)

def initial(n):
    q = np.tile(m.qpos0, (n, 1))
    q[:, m.arm_qposadr] = m.arm_initqpos + rng.uniform(-0.3, 0.3, (n, len(m.arm_qposadr)))
    q[:, m.grip_qposadr] = m.grip_initqpos
    for i in range(m.nparts):
        a = m.part_qposadr[i]
        q[:, a:a + 7] = m.part_initqpos[i]
        q[:, a:a + 2] += rng.uniform(-0.02, 0.02, (n, 2))
        q[:, a + 2] += 0.01
    v = rng.uniform(-0.2, 0.2, (n, m.nv))
    return q, v

cfg = default_config()
cfg.solver_tolerance = float(os.environ.get("FSIM_TOL", "1e-6"))
sim = FSim(m, N, config=cfg)
q0, v0 = initial(N)
sim.set_state(qpos=q0, qvel=v0)
sim.physics_forward()
st = sim.get_state("qacc", "xpos", "xquat", "qfrc_bias", "ncon", "solver_iters")
orc = OracleSim(m)
orc.set_solver(100, 1e-10, "newton")
print("== forward (free space + velocities) ==")
for e in range(3):
    orc.reset(); orc.data.qpos[:] = q0[e]; orc.data.qvel[:] = v0[e]; orc.forward()
    print("env", e, "ncon", int(st["ncon"][e]), orc.ncon, "qacc err", np.abs(st["qacc"][e].cpu().numpy() - orc.data.qacc).max(), "|qacc|", np.abs(orc.data.qacc).max(),
          "bias err", np.abs(st["qfrc_bias"][e].cpu().numpy() - orc.data.qfrc_bias).max(), "xpos err", np.abs(st["xpos"][e].cpu().numpy().reshape(-1, 3) - orc.data.xpos).max(),
          "xquat err", np.abs(st["xquat"][e].cpu().numpy().reshape(-1, 4) - orc.data.xquat).max())

print("== trajectory with contacts: 400 substeps, gravity-compensated arm ==")
q0, _ = initial(N)
v0 = np.zeros((N, m.nv))
sim.set_state(qpos=q0, qvel=v0, qacc_warmstart=np.zeros((N, m.nv)))
sim.physics_forward()
bias = sim.get_state("qfrc_bias")["qfrc_bias"].cpu().numpy()
app = np.zeros((N, m.nv)); rd = np.concatenate([m.arm_dofadr, m.grip_dofadr]); app[:, rd] = bias[:, rd]
sim.set_state(qfrc_applied=app)
orcs = []
for e in range(2):
    o = OracleSim(m); o.set_solver(100, 1e-10, "newton"); o.reset(); o.data.qpos[:] = q0[e]; o.forward()
    o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]; orcs.append(o)
for chunk in range(8):
    sim.physics_step(50)
    s = sim.get_state("qpos", "qvel", "ncon", "solver_iters", "contact_geoms")
    for e, o in enumerate(orcs):
        for _ in range(50): o.step()
        dq = np.abs(s["qpos"][e].cpu().numpy() - o.data.qpos); dv = np.abs(s["qvel"][e].cpu().numpy() - o.data.qvel)
        print("t=%3d env %d ncon gpu/orc %d/%d iters %d  dqpos %.2e (at %d) dqvel %.2e" % ((chunk + 1) * 50, e, int(s["ncon"][e]), o.ncon, int(s["solver_iters"][e]), dq.max(), dq.argmax(), dv.max()))
cg = s["contact_geoms"][0].cpu().numpy().reshape(-1, 2)
print("contacts gpu env0:", [tuple(x) for x in cg[cg[:, 0] >= 0]][:12]); print("contacts orc env0:", orcs[0].contacts()[:12])
print("nan check:", bool(torch.isnan(s["qpos"]).any()), "max |qvel| all envs", float(s["qvel"].abs().max()))

print("== timing: 4096 envs x 50 substeps ==")
big = FSim(m, 4096, config=cfg)
qb, _ = initial(4096)
big.set_state(qpos=qb, qvel=np.zeros((4096, m.nv)))
for rep in range(3):
    torch.cuda.synchronize(); t = time.time(); big.physics_step(50); big.sync(); dt = time.time() - t
    print("rep", rep, "%.2f ms per 50 substeps -> %.0f env-steps/s, %.2fM substeps/s" % (dt * 1e3, 4096 / dt, 4096 * 50 / dt / 1e6))
s = big.get_state("ncon", "solver_iters", "qvel")
print("mean ncon", float(s["ncon"].float().mean()), "mean iters", float(s["solver_iters"].float().mean()), "nan", bool(torch.isnan(s["qvel"]).any()))
