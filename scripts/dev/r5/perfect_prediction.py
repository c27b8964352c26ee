"""development (round 5): what would a PERFECT multi-wave prediction buy?  Every step is run twice from the same snapshot: once to learn
each env's Newton iterations of THIS step, then -- timed -- with E_NITER of every record set to that number (ORACLE=1), so that the
K = 150 rule selects exactly the envs that will be slow, from their first substep on.  An upper bound for mid-step escalation (which
hands an env to a team only after it has shown itself slow).  ORACLE=0: the same procedure without the overwrite (the baseline under
this harness).  G slabs, launch all, then sync all, per step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM, E_NITER
from furniture_amd.envs import ResetTableSampler, make_config

m = load_compiled("Sawyer", "table_lack_0825")
G = int(os.environ.get("G", "4")); N = 4096; ng = N // G
ORACLE = os.environ.get("ORACLE", "1") == "1"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = default_config(); cfg.max_episode_steps = 150
SNAP = ["qpos", "qvel", "qacc_warmstart", "qfrc_bias", "ctrl", "qfrc_applied", "xfrc_applied", "eq_active", "eq_data", "geom_contype", "geom_conaffinity", "group", "env_block"]
sims, bufs = [], []
for g in range(G):
    s = FSim(m, ng, config=cfg)
    s.set_reset_tables(*ResetTableSampler(m, make_config(), 123, g * ng, ng).draw())
    dev = s.device
    b = dict(obs=torch.zeros((ng, s.obs_dim), device=dev), rew=torch.zeros(ng, device=dev), done=torch.zeros(ng, dtype=torch.uint8, device=dev), info=torch.zeros((ng, INFO_DIM), dtype=torch.int32, device=dev))
    gen = torch.Generator(device=dev); gen.manual_seed(123 + g)
    b["act"] = torch.empty((T, ng, 9), device=dev).uniform_(-1, 1, generator=gen)
    s.reset(None, b["obs"]); s.sync()
    sims.append(s); bufs.append(b)
torch.cuda.synchronize()
times, nsel = [], []
for t in range(T):
    snaps = [s.get_state(*SNAP) for s in sims]
    for s, b in zip(sims, bufs):
        s.step(b["act"][t], b["obs"], b["rew"], b["done"], b["info"])
    for s in sims:
        s.sync()
    nit = [s.get_state("env_block")["env_block"][:, E_NITER].clone() for s in sims]
    for s, sn, ni in zip(sims, snaps, nit):
        if ORACLE:
            sn["env_block"][:, E_NITER] = ni
        s.set_state(**sn)
    nsel.append(sum(int((sn["env_block"][:, E_NITER] >= 150).sum()) for sn in snaps))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, b in zip(sims, bufs):
        s.step(b["act"][t], b["obs"], b["rew"], b["done"], b["info"])
    for s in sims:
        s.sync()
    times.append(time.perf_counter() - t0)
print("G %d oracle %d: %.3f ms per batched step over steps 5.. (%.0f env-steps/s), multi-wave envs per step %.0f" % (G, ORACLE, np.mean(times[5:]) * 1e3, N / np.mean(times[5:]), np.mean(nsel[5:])))
