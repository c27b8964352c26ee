"""development: how often does the benchmark workload drop contacts?  Sawyer + table_lack_0825, N envs, random actions, T-step episodes with
auto-reset: the sticky overflow word of every env record (E_OVERFLOW: bit 0 broadphase lists, bit 1 contact slots) after K steps.
usage: overflow_census.py [n_envs] [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM, E_OVERFLOW
from furniture_amd.envs import ResetTableQueue, ResetTableSampler, make_config, INFO_NEEDS_TABLE
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 450
m = load_compiled("Sawyer", "table_lack_0825")
cfg = default_config(); cfg.max_episode_steps = 150
sim = FSim(m, N, config=cfg)
tables = ResetTableQueue(ResetTableSampler(m, make_config(), 123, 0, N))
sim.set_reset_tables(*tables.take())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync(); sim.set_reset_tables(*tables.take())
seen = np.zeros(N, dtype=np.int64); first = {}
for t in range(K):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    if sim.tables_needed():
        need = info.cpu().numpy()[:, INFO_NEEDS_TABLE]; mask = need > 0
        if (need > 1).any(): tables.take(need > 1)
        p, nz = tables.take(mask); sim.set_reset_tables(p, nz, mask=mask)
    w = sim.get_state("env_block")["env_block"].view(torch.int32)[:, E_OVERFLOW].cpu().numpy()
    new = (w != 0) & (seen == 0)
    for e in np.nonzero(new)[0]: first[int(e)] = (t, int(w[e]))
    seen |= w
print("%d envs x %d steps: %d envs dropped contacts at least once (bit 0 = broadphase lists: %d, bit 1 = contact slots: %d)" % (N, K, int((seen != 0).sum()), int((seen & 1 != 0).sum()), int((seen & 2 != 0).sum())))
print("first events (env: step, bits):", dict(list(first.items())[:12]))
