import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from furniture_amd.envs import FurnitureSawyerEnv, make_config, make_vec_env
from furniture_amd.mjcf.model import load_compiled
name = sys.argv[1] if len(sys.argv) > 1 else "bookcase_grevback_0484"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = load_compiled("Sawyer", name)
kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name=name, max_episode_steps=50, seed=seed)
env = FurnitureSawyerEnv(make_config(**kw))
d = env.reset()
print("single env reset: object_ob z", d["object_ob"].reshape(-1, 7)[:, 2].round(3), "robot_ob", d["robot_ob"][:6].round(3))
ob, r, done, info = env.step(np.zeros(9))
print("after a step: z", ob["object_ob"].reshape(-1, 7)[:, 2].round(3), "robot", ob["robot_ob"][:6].round(3), info)
env.close()
v = make_vec_env("Sawyer", 4, furniture_name=name, max_episode_steps=50, seed=seed, record_vid=False, unity=False, control_type="impedance")
o = v.reset()
print("vec env reset: z", o["object_ob"][0].reshape(-1, 7)[:, 2].cpu().numpy().round(3), "robot", o["robot_ob"][0][:6].cpu().numpy().round(3))
