"""development (round 4): GPU diagnostics run by scripts/gpu_r4c.sh -- (a) which multi-wave situations abort, (b) grevback laid-out start: where device and oracle part company"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def ik_case(n, control, steps=3):
    import torch
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    env = FurnitureBatchEnv("Sawyer", n, config=make_config(unity=False, record_vid=False, control_type=control, furniture_name="table_lack_0825", max_episode_steps=50, seed=9), auto_reset=False)
    env.reset()
    rng = np.random.RandomState(2)
    for t in range(steps):
        env.step(rng.uniform(-1, 1, (n, env.dof)).astype(np.float32))
        eb = env.sim.get_state("env_block")["env_block"].cpu().numpy()
        print("  step", t, "kernel", env.sim.step_kernel, "E_NITER", eb[:, 35].tolist(), "mw_steps", eb[:, 36].tolist(), flush=True)
    env.close()
    print("  ok", flush=True)


def grev():
    from furniture_amd.envs import FurnitureSawyerEnv, make_config
    from furniture_amd.mjcf.model import load_compiled
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    from tests.scenarios import spread_layout
    for name in ("bookcase_grevback_0484", "bed_dalselv_0270"):
        m = load_compiled("Sawyer", name)
        lay = spread_layout(m)
        q = np.array(m.qpos0, dtype=float)
        q[m.arm_qposadr], q[m.grip_qposadr] = m.arm_initqpos, m.grip_initqpos
        for p in range(m.nparts):
            q[m.part_qposadr[p]:m.part_qposadr[p] + 7] = lay[p]
        init = {"qpos": q, "qvel": np.zeros(m.nv)}
        kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name=name, max_episode_steps=50, seed=3)
        env = FurnitureSawyerEnv(make_config(**kw))
        orc = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=3, solver_tolerance=1e-10))
        env.set_init_qpos(init), orc.set_init_qpos(init)
        orc.reset(); env.reset()
        rng = np.random.RandomState(2)
        rd = np.concatenate([m.arm_dofadr, m.grip_dofadr])
        for t in range(3):
            st = {k: v[0].cpu().numpy() for k, v in env._b.sim.get_state("qpos", "qvel", "qfrc_bias", "qfrc_applied", "ctrl").items()}
            print(name, "before step", t, "|dq robot| %.2e |dv robot| %.2e |dbias| %.2e |dapplied| %.2e |dq all| %.2e |dv all| %.2e" % (
                np.abs(st["qpos"][m.arm_qposadr] - orc.sim.data.qpos[m.arm_qposadr]).max(), np.abs(st["qvel"][rd] - orc.sim.data.qvel[rd]).max(),
                np.abs(st["qfrc_bias"][rd] - orc.sim.data.qfrc_bias[rd]).max(), np.abs(st["qfrc_applied"][rd] - orc.sim.data.qfrc_applied[rd]).max(),
                np.abs(st["qpos"] - orc.sim.data.qpos).max(), np.abs(st["qvel"] - orc.sim.data.qvel).max()), flush=True)
            a = rng.uniform(-1, 1, 9)
            ob, r, d, info = env.step(a)
            ob_o, r_o, d_o, _ = orc.step(a)
            e = np.abs(np.concatenate([ob["object_ob"], ob["robot_ob"]]) - orc.flat_obs(ob_o))
            print("   obs err max %.2e at %d of %d; robot_ob err %s" % (e.max(), int(e.argmax()), len(e), np.array2string(e[-29:], precision=1)), "ncon oracle", len(orc.sim.contacts()), "overflow", info["contact_overflow"], flush=True)
        env.close()


def grev_substeps(name="bookcase_grevback_0484", nsub=60):
    """device fsim_physics_step(1) vs OracleSim.step() from the laid-out start, substep by substep: which dofs part company first"""
    import torch
    from furniture_amd.mjcf.model import load_compiled
    from furniture_amd.sim import FSim
    from oracle.oracle_sim import OracleSim
    from tests.scenarios import spread_layout
    m = load_compiled("Sawyer", name)
    lay = spread_layout(m)
    q = np.array(m.qpos0, dtype=float)
    q[m.arm_qposadr], q[m.grip_qposadr] = m.arm_initqpos, m.grip_initqpos
    for p in range(m.nparts):
        q[m.part_qposadr[p]:m.part_qposadr[p] + 7] = lay[p]
    sim = FSim(m, 2)
    sim.set_state(qpos=np.tile(q, (2, 1)), qvel=np.zeros((2, m.nv)), qacc_warmstart=np.zeros((2, m.nv)))
    sim.physics_forward()
    bias = sim.get_state("qfrc_bias")["qfrc_bias"].cpu().numpy()
    rd = np.concatenate([m.arm_dofadr, m.grip_dofadr])
    app = np.zeros((2, m.nv)); app[:, rd] = bias[:, rd]
    sim.set_state(qfrc_applied=app)
    o = OracleSim(m); o.set_solver(100, 1e-10, "newton"); o.reset()
    o.data.qpos[:] = q; o.forward(); o.data.qfrc_applied[rd] = o.data.qfrc_bias[rd]
    print(name, "kernel", sim.kernel_variant, "max_contacts", sim.max_contacts, "|bias dev - orc| robot %.2e" % np.abs(bias[0, rd] - o.data.qfrc_bias[rd]).max())
    for k in range(nsub):
        sim.physics_step(1); o.step()
        st = {a: b[0].cpu().numpy() for a, b in sim.get_state("qpos", "qvel", "qacc", "ncon", "solver_iters").items()}
        dq, dv, da = np.abs(st["qpos"] - o.data.qpos), np.abs(st["qvel"] - o.data.qvel), np.abs(st["qacc"] - o.data.qacc)
        if k in (0, 10, 11):
            cg = sim.get_state("contact_geoms")["contact_geoms"][0].cpu().numpy().reshape(-1, 2)[:int(st["ncon"][0])]
            dc, oc = sorted(map(tuple, np.sort(cg, axis=1).tolist())), sorted(tuple(sorted(x)) for x in o.contacts())
            from collections import Counter
            cd, co = Counter(dc), Counter(oc)
            print("   contacts only on the device:", dict(cd - co), "only in the oracle:", dict(co - cd), "| oracle iterations", o.last_solver_iters, flush=True)
            pt = [float(da[m.part_dofadr[p_]:m.part_dofadr[p_] + 6].max()) for p_ in range(m.nparts)]
            print("   |da| per part:", " ".join("%.1e" % v for v in pt), flush=True)
        if k < 14 or k % 10 == 9:
            print(" substep %2d: ncon dev %d orc %d iters %d | robot |dq| %.2e |dv| %.2e |da| %.2e | parts |dq| %.2e |dv| %.2e |da| %.2e (worst dof %d) oracle iters %d" % (
                k, int(st["ncon"][0]), len(o.contacts()), int(st["solver_iters"][0]), dq[m.arm_qposadr].max(), dv[rd].max(), da[rd].max(),
                np.delete(dq, m.arm_qposadr).max(), np.delete(dv, rd).max(), np.delete(da, rd).max(), int(da.argmax()), o.last_solver_iters), flush=True)
    sim.close()


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "ik":
        ik_case(int(sys.argv[2]), sys.argv[3])
    elif what == "grev":
        grev()
    elif what == "grevsub":
        grev_substeps(*(sys.argv[2:3]))
