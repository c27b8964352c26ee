"""development: how far does the farthest part of chair_agam_0005 travel in the 4 steps of tests/test_all_furniture_gpu.py?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from furniture_amd.envs import make_vec_env
from furniture_amd.mjcf.model import load_compiled
name = sys.argv[1] if len(sys.argv) > 1 else "chair_agam_0005"
m = load_compiled("Sawyer", name)
env = make_vec_env("Sawyer", 4, furniture_name=name, max_episode_steps=3, seed=11, record_vid=False, unity=False, control_type="impedance")
ob = env.reset()
g = torch.Generator(device=env.sim.device); g.manual_seed(1)
for t in range(4):
    a = torch.empty((4, 9), device=env.sim.device).uniform_(-1, 1, generator=g)
    ob, rew, done, info = env.step(a)
    p = ob["object_ob"].reshape(4, m.nparts, 7)[:, :, :3]
    print("step", t, "max |pos| per env", [round(float(x), 3) for x in p.abs().amax(dim=(1, 2))], "fail", info["fail"].tolist())
