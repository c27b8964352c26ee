import os, sys, numpy as np, torch, time
sys.path.insert(0,'.')
os.environ["FSIM_LIB"]=os.path.abspath("furniture_amd/csrc/libfsim_prof.so")
from furniture_amd.envs import make_vec_env
for ct in ("ik","impedance"):
    env = make_vec_env("Sawyer", 2048, furniture_name="table_lack_0825", max_episode_steps=150, seed=123, record_vid=False, unity=False, control_type=ct)
    env.reset()
    g = torch.Generator(device=env.sim.device); g.manual_seed(1)
    for t in range(12):
        a = torch.empty((2048, env.sim.dof_action), device=env.sim.device).uniform_(-1, 1, generator=g)
        t0=time.time(); env.step(a); dt=time.time()-t0
    p = env.sim.get_state("qacc")["qacc"].view(torch.int32).cpu().numpy().astype(np.int64)
    tot = (p[:, 1] + p[:, 3] + p[:, 4] + p[:, 16:22].sum(axis=1)) * 16
    nsub, nit, ncoup = p[:,5], p[:,6], p[:,7]
    print(ct, "step %.1f ms | substeps/env %.0f | mean Mcyc/env %.2f p50 %.2f p99 %.2f max %.2f | it/substep %.2f | coupled frac %.3f | kcyc per substep (mean) %.1f" % (dt*1e3, nsub.mean(), tot.mean()/1e6, np.percentile(tot,50)/1e6, np.percentile(tot,99)/1e6, tot.max()/1e6, nit.sum()/nsub.sum(), ncoup.sum()/nsub.sum(), tot.sum()/nsub.sum()/1e3))
    env.close()
