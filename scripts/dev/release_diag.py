"""development (round-3 verdict weak 2): Sawyer_7 frames 3-5 -- the gripper opens and the column leaves the fingers.  Device and fp64 oracle
replayed substep by substep from the same frame start: column height / velocity, the finger-column contacts either side lists, Newton
iterations.  usage: release_diag.py [first frame] [last frame] [pgs | newton: solver of the oracle that replays the whole path]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_demo_sawyer_replay import D, N_SUB, H, kinematic_robot_model, robot, start_state
from furniture_amd.sim import FSim, default_config
from oracle.oracle_sim import OracleSim

m = kinematic_robot_model()
COL = 1
SAVE = {}
f0, f1 = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 6)
sim = FSim(m, 1, config=default_config())
o = OracleSim(m); o.set_solver(100, 1e-10, sys.argv[3] if len(sys.argv) > 3 else "pgs"); o.reset()  # (pgs: what rounds 2-3 replayed with)
o2 = OracleSim(m); o2.set_solver(100, 1e-10, "newton"); o2.reset()  # one substep from the DEVICE's state: is a difference the substep's or the path's?
q = start_state(m, 0)
zero = lambda k: np.zeros((1, k))
sim.set_state(qpos=q[None], qvel=zero(m.nv), qacc_warmstart=zero(m.nv), ctrl=zero(m.nu), qfrc_applied=zero(m.nv), xfrc_applied=zero(6 * m.nparts))
o.data.qpos[:] = q; o.data.qvel[:] = 0; o.forward()
a0, d0 = int(m.part_qposadr[COL]), int(m.part_dofadr[COL])
colg = set(int(g) for g in np.where(np.asarray(m.cg_partid) == COL)[0]) if hasattr(m, "cg_partid") else set()
fing = np.asarray(m.cg_fingerrole)
print("column collision geoms", sorted(colg), "finger geoms", [int(g) for g in np.where(fing > 0)[0]], "margin", np.asarray(m.cg_margin)[sorted(colg)], "solimp", np.asarray(m.cg_solimp)[sorted(colg)][:1], "solref", np.asarray(m.cg_solref)[sorted(colg)][:1], "friction", np.asarray(m.cg_friction)[sorted(colg)][:1])
orig = np.asarray(m.cg_orig)
for t in range(0, f1):
    v = (robot(t + 1) - robot(t)) / (N_SUB * H)
    st = sim.get_state("qpos", "qvel")
    qp, qv = st["qpos"].clone(), st["qvel"].clone()
    qp[:, :9] = torch.as_tensor(robot(t), dtype=torch.float32, device=qp.device); qv[:, :9] = torch.as_tensor(v, dtype=torch.float32, device=qp.device)
    ctrl = np.concatenate([v[:7], robot(t + 1)[7:9]])
    sim.set_state(qpos=qp, qvel=qv, ctrl=ctrl[None])
    o.data.qpos[:9], o.data.qvel[:9] = robot(t), v
    o.data.ctrl[:7], o.data.ctrl[7:9] = v[:7], robot(t + 1)[7:9]
    if t < f0:
        sim.physics_step(N_SUB); sim.sync()
        for _ in range(N_SUB): o.step()
        continue
    print("frame %d: finger joints %s -> %s" % (t, np.round(robot(t)[7:9], 5), np.round(robot(t + 1)[7:9], 5)))
    for k in range(N_SUB):
        pre = {a: b[0].cpu().numpy().astype(np.float64) for a, b in sim.get_state("qpos", "qvel", "qacc_warmstart", "ctrl").items()}
        o2.data.qpos[:], o2.data.qvel[:], o2.data.qacc_warmstart[:], o2.data.ctrl[:] = pre["qpos"], pre["qvel"], pre["qacc_warmstart"], pre["ctrl"]
        o2.step()
        az2, it2, nc2 = o2.data.qacc[d0 + 2], o2.last_solver_iters, len(o2.contacts())
        # hypothesis test: the same substep with the FINGER velocities zeroed (does the device's force match a contact that sees no opening speed?)
        o2.data.qpos[:], o2.data.qvel[:], o2.data.qacc_warmstart[:], o2.data.ctrl[:] = pre["qpos"], pre["qvel"], pre["qacc_warmstart"], pre["ctrl"]
        o2.data.qvel[7:9] = 0
        o2.step()
        az3 = o2.data.qacc[d0 + 2]
        sim.physics_step(1); o.step()
        if t == 4 and 30 <= k <= 48:
            SAVE.setdefault("pre_qpos", []).append(pre["qpos"]); SAVE.setdefault("pre_qvel", []).append(pre["qvel"]); SAVE.setdefault("pre_ws", []).append(pre["qacc_warmstart"]); SAVE.setdefault("ctrl", []).append(pre["ctrl"])
            SAVE.setdefault("dev_qacc", []).append(sim.get_state("qacc")["qacc"][0].cpu().numpy().astype(np.float64)); SAVE.setdefault("k", []).append(k)
        s = {a: b[0].cpu().numpy() for a, b in sim.get_state("qpos", "qvel", "qacc", "contact_geoms", "ncon", "solver_iters").items()}
        cd = s["contact_geoms"].reshape(-1, 2)[:int(s["ncon"][0])]
        # (both sides list MODEL geom ids: map through cg_orig to collision-geom indices)
        inv = {int(g): i for i, g in enumerate(orig)}
        cdm = [(inv.get(int(a), -1), inv.get(int(b), -1)) for a, b in cd]
        dev_fc = sorted((a, b) for a, b in cdm if a >= 0 and b >= 0 and ((a in colg and fing[b]) or (b in colg and fing[a])))
        oc = o.contacts(); od = o.contact_dists()
        ora_fc = sorted(((inv.get(int(a), -1), inv.get(int(b), -1)), round(dd * 1e3, 4)) for (a, b), dd in zip(oc, od) if (inv.get(int(a), -1) in colg and inv.get(int(b), -1) >= 0 and fing[inv[int(b)]]) or (inv.get(int(b), -1) in colg and inv.get(int(a), -1) >= 0 and fing[inv[int(a)]]))
        zd, zo = s["qpos"][a0 + 2], o.data.qpos[a0 + 2]
        if len(dev_fc) or len(ora_fc) or abs(zd - zo) > 1e-5 or k % 25 == 0:
            print("  sub %3d | col z dev %.5f orc %.5f vz %.4f %.4f az %.2f %.2f | finger-column contacts dev %s | orc (pair, dist mm) %s | iters %d %d | fingers dev %s | oracle substep from the device's state: az %.2f iters %d contacts %d" % (
                k, zd, zo, s["qvel"][d0 + 2], o.data.qvel[d0 + 2], s["qacc"][d0 + 2], o.data.qacc[d0 + 2], dev_fc, ora_fc, int(s["solver_iters"][0]), o.last_solver_iters, np.round(s["qpos"][7:9], 5), az2, it2, nc2) + " | with finger qvel = 0: az %.2f" % az3, flush=True)
sim.close()
if SAVE:
    np.savez(os.environ.get('FSIM_RELEASE_DUMP', 'gpurun_out/r4n/release_states.npz'), **{a: np.array(b) for a, b in SAVE.items()})
