import sys, numpy as np, torch
sys.path.insert(0,'.')
from furniture_amd.envs import make_vec_env
from furniture_amd.mjcf.model import load_compiled
for name in sys.argv[1:]:
    m = load_compiled("Sawyer", name)
    env = make_vec_env("Sawyer", 2, furniture_name=name, max_episode_steps=30, seed=11, record_vid=False, unity=False, control_type="impedance")
    ob = env.reset()
    st = env.sim.get_state("ncon", "solver_iters", "qpos")
    print(name, "nv", m.nv, "nparts", m.nparts, "after reset ncon", st["ncon"][:,0].tolist(), "iters", st["solver_iters"][:,0].tolist(), "z", ob["object_ob"].reshape(2,m.nparts,7)[0,:,2].cpu().numpy().round(3))
    for t in range(2):
        ob, rew, done, info = env.step(torch.zeros((2,9), device=env.sim.device))
        print("  step", t, "done", done.tolist(), "fail", info["fail"].tolist(), "overflow", info["contact_overflow"].tolist(), "rew", rew.tolist())
    env.close()
