"""development: single physics substeps from the reset's start state (planks inside each other: one island of every part), device vs OracleSim"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from furniture_amd.envs import ResetTableSampler, make_config
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim
from oracle.oracle_sim import OracleSim
name = sys.argv[1]
m = load_compiled("Sawyer", name)
parts, noise = ResetTableSampler(m, make_config(furniture_name=name), 3, 0, 1).draw()
q = np.array(m.qpos0, dtype=np.float64)
q[m.arm_qposadr] = m.arm_initqpos; q[m.grip_qposadr] = m.grip_initqpos
pq = np.asarray(parts).reshape(-1, 7)
for i in range(m.nparts):
    a = m.part_qposadr[i]; q[a:a + 7] = pq[i]
sim = FSim(m, 1)
print(name, "nv", m.nv, "kernel", sim.kernel_variant, "slots", sim.max_contacts)
sim.set_state(qpos=q[None], qvel=np.zeros((1, m.nv)), qacc_warmstart=np.zeros((1, m.nv)))
o = OracleSim(m); o.set_solver(100, 1e-10, "newton"); o.reset(); o.data.qpos[:] = q; o.forward()
done = 0
for k in (1, 1, 3, 5, 10, 30, 50):
    sim.physics_step(k)
    for _ in range(k): o.step()
    done += k
    st = sim.get_state("qpos", "qvel", "ncon", "solver_iters")
    dq = np.abs(st["qpos"][0].cpu().numpy() - o.data.qpos).max(); dv = np.abs(st["qvel"][0].cpu().numpy() - o.data.qvel).max()
    print("  substep %3d: ncon dev %s oracle %d  iters dev %s oracle %d  |dq| %.2e |dv| %.2e  |v| %.2e" % (done, st["ncon"][0].cpu().numpy().tolist(), o.ncon, st["solver_iters"][0].cpu().numpy().tolist(), o.last_solver_iters, dq, dv, np.abs(o.data.qvel).max()), flush=True)
