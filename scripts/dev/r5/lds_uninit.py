"""development: which LDS words of an env image are read before they are written?  Needs furniture_amd/csrc/libfsim_dbg.so (-DFSIM_DBG_FILL) as
FSIM_LIB and tests/liblds_poison.so.  Every CU's LDS is zeroed by another kernel before each step (reference run); then, range by
range, the words [lo, hi) of every env image are set to NaN before the env runs: a result that changes names a range with such a word.
usage: FSIM_LIB=.../libfsim_dbg.so [FSIM_MW=0] lds_uninit.py <agent> <furniture> <n> <steps>"""
import sys, os, re, ctypes, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
agent, furn, n, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
if len(sys.argv) > 5 and sys.argv[5] == "layout":
    from furniture_amd.mjcf.model import load_compiled
    from furniture_amd.sim import FSim, default_config
    from furniture_amd.envs import CONTROLLER_CODES
    cfg0 = default_config()
    cfg0.control_type = CONTROLLER_CODES.get(os.environ.get("CONTROL", "impedance"), 0)
    FSim(load_compiled(agent, furn, os.environ.get("CONTROL", "impedance")), 1, config=cfg0).close()
    sys.exit(0)
lay = subprocess.run([sys.executable, __file__, agent, furn, "1", "1", "layout"], capture_output=True, text=True).stderr
line = [l for l in lay.splitlines() if l.startswith("[fsim dbg]")][0]
names = re.findall(r"(\w+(?:\(H\))?) (\d+)", line[len("[fsim dbg] "):])
off = {k: int(v) for k, v in names}
print(line)
import torch
from furniture_amd.envs import ResetTableSampler, make_config
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.sim import FSim, INFO_DIM, default_config
PL = ctypes.CDLL(os.path.join(ROOT, "tests", "liblds_poison.so"))
CT = os.environ.get("CONTROL", "impedance")
m = load_compiled(agent, furn, CT)
ecfg = make_config(unity=False, record_vid=False, furniture_name=furn, max_episode_steps=1000, seed=200)
parts, noise = ResetTableSampler(m, ecfg, 200, 0, n).draw()


def run(fill):
    os.environ["FSIM_DBG_FILL"] = "%d,%d,7fc00000" % fill if fill else "0,0,0"
    cfg = default_config()
    cfg.max_episode_steps, cfg.auto_reset, cfg.lookahead_reset, cfg.overflow_restep = 1000, 0, 0, 0
    from furniture_amd.envs import CONTROLLER_CODES
    cfg.control_type = CONTROLLER_CODES.get(CT, 0)
    sim = FSim(m, n, config=cfg)
    sim.set_reset_tables(parts, noise if agent != "Cursor" else None)
    dev = sim.device
    obs = torch.zeros((n, sim.obs_dim), device=dev)
    for _ in range(3):
        PL.lds_poison(ctypes.c_uint(0))
    sim.reset(None, obs)
    sim.sync()
    dof = sim.dof_action
    act, rew = torch.zeros((n, dof), device=dev), torch.zeros(n, device=dev)
    done, info = torch.zeros(n, dtype=torch.uint8, device=dev), torch.zeros((n, INFO_DIM), dtype=torch.int32, device=dev)
    rng = np.random.RandomState(1)
    out = [obs.cpu().numpy().copy()]
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, dof)).astype(np.float32)
        if agent == "Cursor":
            a[:, 6] = np.abs(a[:, 6]) * np.where(rng.rand(n) < 0.8, 1, -1)
            a[:, 13] = np.abs(a[:, 13]) * np.where(rng.rand(n) < 0.8, 1, -1)
        act.copy_(torch.as_tensor(a))
        torch.cuda.synchronize()
        for _ in range(3):
            PL.lds_poison(ctypes.c_uint(0))
        sim.step(act, obs, rew, done, info)
        sim.sync()
        out.append(np.concatenate([obs.cpu().numpy().view(np.uint32), info.cpu().numpy()[:, [0, 2, 12]].astype(np.uint32)], axis=1))
    sim.close()
    return out


def same(a, b):
    return all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b))


ref = run(None)
assert same(ref, run(None)), "the zero-poisoned reference itself is not reproducible"
bounds = sorted(set(v for k, v in off.items() if k not in ("k_begin", "k_end")) | {off["lds_words"]})
bounds = [b for b in bounds if b >= off["stride"]]
inv = {}
for k, v in off.items():
    inv.setdefault(v, []).append(k)
bad = []
for lo, hi in zip(bounds[:-1], bounds[1:]):
    ok = same(ref, run((lo, hi)))
    print("[%5d, %5d) %-24s %s" % (lo, hi, "+".join(inv.get(lo, ["?"])), "ok" if ok else "READ BEFORE WRITTEN"))
    if not ok:
        bad.append((lo, hi))
for lo, hi in bad:  # refine to single words
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if not same(ref, run((lo, mid))):
            hi = mid
        else:
            lo = mid
    print("first offending word: %d (array %s + %d)" % (lo, "+".join(inv.get(max(b for b in bounds if b <= lo), ["?"])), lo - max(b for b in bounds if b <= lo)))
