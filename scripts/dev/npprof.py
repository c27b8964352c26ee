import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["FSIM_LIB"] = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "furniture_amd", "csrc", "libfsim_npprof.so")
import torch
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, default_config, INFO_DIM
from furniture_amd.envs import ResetTableSampler, make_config
m = load_compiled("Sawyer", "table_lack_0825")
N = 4096
cfg = default_config(); cfg.max_episode_steps = 150
sim = FSim(m, N, config=cfg)
sim.set_reset_tables(*ResetTableSampler(m, make_config(), 123, 0, N).draw())
dev = sim.device
obs = torch.zeros((N, sim.obs_dim), device=dev); rew = torch.zeros(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev); info = torch.zeros((N, INFO_DIM), dtype=torch.int32, device=dev)
act = torch.empty((N, 9), device=dev); g = torch.Generator(device=dev); g.manual_seed(123)
sim.reset(None, obs); sim.sync()
for t in range(6):
    act.uniform_(-1, 1, generator=g); torch.cuda.synchronize()
    sim.step(act, obs, rew, done, info); sim.sync()
    pall = sim.get_state("qacc")["qacc"].view(torch.int32)[:, :39].cpu().numpy().astype(np.int64)
    med = np.median(pall[:, 32:39] * 16, axis=0) / 50e3
    print("step %d narrowphase path kcyc/substep (median env): MPR(cyl-*) %.1f  sphere/plane-other %.1f  plane_box %.1f  box_box %.1f | total narrow %.1f" % (t, med[0], med[1], med[5], med[6], np.median(pall[:, 31] * 16) / 50e3))
