"""GPU env logic vs oracle env diagnostics (development aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM, N_NOISE
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
from tests.scenarios import pinch_attach_state, counter_actions

m = load_compiled("Sawyer", "table_lack_0825")
N = 8
cfg = default_config(); cfg.max_episode_steps = 150; cfg.auto_reset = 0
sim = FSim(m, N, config=cfg)
envs = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=150, seed=123 + i, solver_tolerance=1e-10)) for i in range(N)]
obs_o = [e.reset() for e in envs]
parts = np.stack([e.reset_draws["part_qpos"].reshape(-1) for e in envs])
noise = np.stack([np.stack(e.reset_draws["noise"]).reshape(-1) for e in envs])
print("noise draws per env", noise.shape[1] // 7)
sim.set_reset_tables(parts, noise)
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev)
t = time.time(); sim.reset(None, obs); sim.sync(); print("device reset %.1f ms" % ((time.time() - t) * 1e3))
st = sim.get_state("qpos", "qvel")
for e in range(3):
    print("reset env", e, "dqpos %.2e dqvel %.2e dobs %.2e" % (np.abs(st["qpos"][e].cpu().numpy() - envs[e].sim.data.qpos).max(), np.abs(st["qvel"][e].cpu().numpy() - envs[e].sim.data.qvel).max(),
          np.abs(obs[e].cpu().numpy() - envs[e].flat_obs(obs_o[e])).max()))
act = torch.zeros((N, 9), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
for t_ in range(10):
    a = np.stack([counter_actions(123, i, t_, 9) for i in range(N)])
    act.copy_(torch.as_tensor(a)); sim.step(act, obs, rew, done, info); sim.sync()
    res = [envs[i].step(a[i]) for i in range(3)]
    print("step", t_, " ".join("e%d dobs %.1e drew %.1e" % (i, np.abs(obs[i].cpu().numpy() - envs[i].flat_obs(res[i][0])).max(), abs(float(rew[i]) - res[i][1])) for i in range(3)))
# scripted attach on env 0..N-1 (same state everywhere) vs oracle env 0
st = sim.get_state("qpos", "xpos", "xquat")
q0 = st["qpos"][0].cpu().numpy().astype(np.float64)
q, xfrc, masks = pinch_attach_state(m, envs[0].sim.data.qpos.copy(), envs[0].sim.data.xpos.copy(), envs[0].sim.data.xquat.copy())
o = envs[0]
o.sim.data.qpos[:] = q; o.sim.data.qvel[:] = 0; o.sim.data.qacc_warmstart[:] = 0
for i in range(m.nparts): o.sim.data.xfrc_applied[m.part_bodyid[i]] = xfrc.reshape(-1, 6)[i]
gm = sim.get_state("geom_contype", "geom_conaffinity")
for g, (ct, ca) in masks.items():
    o.sim.model.geom_contype[g], o.sim.model.geom_conaffinity[g] = ct, ca
    gm["geom_contype"][:, g] = ct; gm["geom_conaffinity"][:, g] = ca
sim.set_state(qpos=q[None], qvel=np.zeros((1, m.nv)), qacc_warmstart=np.zeros((1, m.nv)), xfrc_applied=xfrc[None], geom_contype=gm["geom_contype"], geom_conaffinity=gm["geom_conaffinity"])
a = np.zeros(9, dtype=np.float32); a[7] = 1.0; a[8] = 1.0
for t_ in range(3):
    act.copy_(torch.as_tensor(np.tile(a, (N, 1)))); sim.step(act, obs, rew, done, info); sim.sync()
    ro = o.step(a)
    inf = info[0].cpu().numpy()
    print("attach step", t_, "gpu: nconn %d site %d,%d connected %d rew %.3f | oracle: nconn %d site %d,%d connected %d rew %.3f | dobs %.2e" % (
        inf[0], inf[3], inf[4], inf[6], float(rew[0]), ro[3]["num_connected"], ro[3]["site1"], ro[3]["site2"], ro[3]["connected_this_step"], ro[1], np.abs(obs[0].cpu().numpy() - o.flat_obs(ro[0])).max()))
s2 = sim.get_state("eq_active", "eq_data", "geom_contype", "geom_conaffinity", "group", "qpos")
print("gpu eq_active", s2["eq_active"][0].cpu().numpy(), "oracle", o.sim.model.eq_active)
print("eq_data diff", np.abs(s2["eq_data"][0].cpu().numpy().reshape(-1, 7) - o.sim.model.eq_data).max())
pc = m.geom_is_partcol.astype(bool)
print("masks gpu", s2["geom_contype"][0].cpu().numpy()[pc], s2["geom_conaffinity"][0].cpu().numpy()[pc]); print("masks orc", o.sim.model.geom_contype[pc], o.sim.model.geom_conaffinity[pc])
print("group gpu", s2["group"][0].cpu().numpy(), "orc", o._group, "dqpos", np.abs(s2["qpos"][0].cpu().numpy() - o.sim.data.qpos).max())
# throughput of the full env step
cfg2 = default_config(); cfg2.max_episode_steps = 150; cfg2.auto_reset = 1
NB = 4096
big = FSim(m, NB, config=cfg2)
pb = np.tile(parts[0], (NB, 1)); nb = np.tile(noise[0], (NB, 1))
big.set_reset_tables(pb, nb)
obsb = torch.zeros((NB, big.obs_dim), device=dev); t = time.time(); big.reset(None, obsb); big.sync(); print("reset 4096 envs: %.1f ms" % ((time.time() - t) * 1e3))
actb = torch.empty((NB, 9), device=dev); rewb = torch.zeros(NB, device=dev); doneb = torch.zeros(NB, dtype=torch.uint8, device=dev); infob = torch.zeros((NB, INFO_DIM), dtype=torch.int32, device=dev)
g = torch.Generator(device=dev); g.manual_seed(0)
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    for k in range(10):
        actb.uniform_(-1, 1, generator=g); big.step(actb, obsb, rewb, doneb, infob)
    big.sync(); dt = time.time() - t
    ms, n = big.kernel_time_ms()
    print("rep %d: %.1f ms/step -> %.0f env-steps/s ; kernel avg %.2f ms (%d launches); nan obs %s" % (rep, dt / 10 * 1e3, NB * 10 / dt, ms, n, bool(torch.isnan(obsb).any())))
