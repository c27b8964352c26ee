"""development: is the 5-iterations-per-substep cost of a resting pile of parts an fp32 effect?  Takes the slowest env of a
device batch after 62 steps, replays 30 substeps from its state on the device (one env) and on the fp64 oracle, prints the
Newton iteration counts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FSIM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "furniture_amd", "csrc", "libfsim_prof.so"))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
from oracle.oracle_sim import OracleSim
m = load_compiled("Sawyer", "table_lack_0825")
N = 2048
cfg = default_config(); cfg.max_episode_steps = 150
sim = FSim(m, N, config=cfg)
sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, 0, N).draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 64):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
p = sim.get_state("qacc")["qacc"].view(torch.int32).cpu().numpy().astype(np.int64)
tot = (p[:, 1] + p[:, 3] + p[:, 4] + p[:, 16:22].sum(axis=1)) * 16
names = ("qpos", "qvel", "qacc_warmstart", "ctrl", "qfrc_applied", "xfrc_applied", "geom_contype", "geom_conaffinity", "eq_active", "eq_data")
for rank in (0, 1, 2):
    e = int(np.argsort(-tot)[rank])
    st = {k: v[e].cpu().numpy() for k, v in sim.get_state(*names).items()}
    one = FSim(m, 1, config=cfg)
    one.set_state(**{k: v[None].copy() for k, v in st.items()})
    o = OracleSim(m); o.set_solver(100, 1e-6, "newton"); o.reset()
    o.data.qpos[:] = st["qpos"]; o.data.qvel[:] = st["qvel"]; o.data.qacc_warmstart[:] = st["qacc_warmstart"]; o.data.ctrl[:] = st["ctrl"]
    o.data.qfrc_applied[:] = st["qfrc_applied"]
    xf = st["xfrc_applied"].reshape(-1, 6)
    for i, b in enumerate(m.part_bodyid): o.data.xfrc_applied[int(b)] = xf[i]
    o.model.geom_contype[:] = st["geom_contype"]; o.model.geom_conaffinity[:] = st["geom_conaffinity"]
    o.model.eq_active[:] = st["eq_active"]; o.model.eq_data[:] = st["eq_data"].reshape(-1, 7)
    itd, ito, ncs, vn, dists = [], [], [], [], []
    for s in range(30):
        one.physics_step(1)
        itd.append(int(one.get_state("solver_iters")["solver_iters"][0, 0]))
        o.step(); ito.append(o.last_solver_iters)
        ncs.append(len(o.contacts())); vn.append(float(np.abs(o.data.qvel[9:]).max())); dists.append(float(min(o.contact_dists())) if len(o.contacts()) else 0.0)
    cg = sorted(set((int(a), int(b)) for a, b in o.contacts()))
    rob = m.geom_is_robot.astype(bool)
    nrp = sum(1 for a, b in cg if rob[a] != rob[b] and not m.geom_is_floor[a] and not m.geom_is_floor[b]) if hasattr(m, "geom_is_floor") else -1
    print("   oracle per substep: contacts %s | max |qvel| of the parts %s | deepest contact %s" % (ncs[:16], [round(v, 3) for v in vn[:16]], [round(d * 1e3, 2) for d in dists[:16]]))
    print("env %d (%.1f Mcyc, it/substep %.2f): device iterations %s | oracle (fp64, tol 1e-6) %s | contacts %d" % (
        e, tot[e] / 1e6, p[e, 6] / max(1, p[e, 5]), itd[:24], ito[:24], len(o.contacts())))
    one.close(); o.close()
