import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FSIM_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "furniture_amd", "csrc", "libfsim_prof.so")
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
m = load_compiled("Sawyer", "table_lack_0825")
N = 4096
cfg = default_config(); cfg.max_episode_steps = 150; cfg.auto_reset = 0
sim = FSim(m, N, config=cfg)
sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, 0, N).draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
for t in range(8):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    p = sim.get_state("qacc")["qacc"].view(torch.int32)[:, :16].cpu().numpy()
    fails = np.where(info[:, 2].cpu().numpy() != 0)[0]
    codes = p[:, 11]; first = p[:, 12]
    print("step", t, "fails", len(fails), "nan codes among fails:", sorted(zip(codes[fails].tolist(), first[fails].tolist()))[:12], "| envs with code but no fail", int(((codes != 0) & (info[:, 2].cpu().numpy() == 0)).sum()))
