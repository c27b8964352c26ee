"""Generate tests/golden/step_scan.npz: FurnitureEnv._step_continuous's finger-touch scan (furniture.py:1262-1330) run by the
REFERENCE on a fake self.  Build container only (needs /root/reference).  The fake env carries the real id tables of
Sawyer + table_lack_0825 and Baxter + desk_mikael_1064 (geom -> body, part bodies in order, finger geoms per arm, all from the
compiled models) and random contact lists; `_setup_action`, `_do_simulation` are inert and `_try_connect` answers from a script
and records which part it was asked about."""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from make_golden_env_logic import import_reference  # noqa: E402
from furniture_amd.mjcf.model import load_compiled  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "step_scan.npz")


def main():
    F = import_reference()
    rng = np.random.RandomState(77)
    out = {}
    for agent, furn, arms in (("Sawyer", "table_lack_0825", ["right"]), ("Baxter", "desk_mikael_1064", ["right", "left"])):
        m = load_compiled(agent, furn)
        part_bodies = [int(b) for b in m.part_bodyid]
        names = {b: "part%d" % i for i, b in enumerate(part_bodies)}
        lf = {a: [g for g in range(m.ngeom) if m.geom_fingerrole[g] & (1 << (2 * k))] for k, a in enumerate(arms)}
        rf = {a: [g for g in range(m.ngeom) if m.geom_fingerrole[g] & (1 << (2 * k + 1))] for k, a in enumerate(arms)}
        part_geoms = [g for g in range(m.ngeom) if m.geom_bodyid[g] in part_bodies and m.geom_is_partcol[g]]
        other = [g for g in range(m.ngeom) if g not in part_geoms and all(g not in lf[a] + rf[a] for a in arms)]
        cons, scripts, tried, nsim = [], [], [], []
        for t in range(200):
            ncon = rng.randint(0, 10)
            pairs = []
            for _ in range(ncon):
                kind = rng.randint(4)
                a = arms[rng.randint(len(arms))]
                if kind == 0:
                    g1, g2 = rng.choice(lf[a]), rng.choice(part_geoms)
                elif kind == 1:
                    g1, g2 = rng.choice(rf[a]), rng.choice(part_geoms)
                elif kind == 2:
                    g1, g2 = rng.choice(other), rng.choice(part_geoms)
                else:
                    g1, g2 = rng.choice(lf[a] + rf[a]), rng.choice(other)
                if rng.rand() < 0.5:
                    g1, g2 = g2, g1
                pairs.append((int(g1), int(g2)))
            # focus: make double touches likely
            if rng.rand() < 0.7 and ncon:
                a = arms[rng.randint(len(arms))]
                pg = int(rng.choice(part_geoms))
                pairs += [(int(rng.choice(lf[a])), pg), (pg, int(rng.choice(rf[a])))]
            script = rng.rand(4) < 0.4
            env = types.SimpleNamespace()
            env._control_type, env._arms, env._record_demo = "impedance", arms, False
            env._object_body_ids = part_bodies
            env.l_finger_geom_ids, env.r_finger_geom_ids = lf, rf
            env.sim = types.SimpleNamespace(
                data=types.SimpleNamespace(ncon=len(pairs), contact=[types.SimpleNamespace(geom1=p[0], geom2=p[1]) for p in pairs]),
                model=types.SimpleNamespace(geom_bodyid=m.geom_bodyid, body_id2name=lambda b: names[b]))
            log = dict(tried=[], nsim=0)
            env._setup_action = lambda a: a
            env._do_simulation = lambda a, log=log: log.__setitem__("nsim", log["nsim"] + 1)

            def try_connect(name, log=log, script=script):
                log["tried"].append(int(name[4:]))
                return bool(script[len(log["tried"]) - 1])

            env._try_connect = try_connect
            dof = 9 if agent == "Sawyer" else 17
            action = rng.uniform(-1, 1, dof)
            action[-1] = 1.0 if t % 5 else -1.0  # connect > 0 most of the time
            F.FurnitureEnv._step_continuous(env, action)
            pad = np.full((12, 2), -1)
            pad[:len(pairs)] = np.array(pairs).reshape(-1, 2) if pairs else np.zeros((0, 2))
            cons.append(pad); scripts.append(script); nsim.append(log["nsim"])
            tr = np.full(4, -1); tr[:len(log["tried"])] = log["tried"]
            tried.append(tr)
            out.setdefault(agent + "_connect", []).append(action[-1])
        out[agent + "_contacts"], out[agent + "_script"] = np.array(cons), np.array(scripts)
        out[agent + "_tried"], out[agent + "_nsim"] = np.array(tried), np.array(nsim)
        out[agent + "_connect"] = np.array(out[agent + "_connect"])
    # ---- FurnitureEnv._after_step (furniture.py:451-480): counters, equality time limit, failure penalty, terminal step_log ------
    env = types.SimpleNamespace(_episode_reward=0.0, _episode_length=0, _max_episode_steps=7, _fail=False, _success=False, _num_connected=0,
                                _episode_time=0.0, _config=types.SimpleNamespace(unstable_penalty_coef=100), get_env_state=lambda: {})
    rows = []
    for t in range(60):
        reward = float(rng.uniform(-1, 1))
        terminal_in = bool(rng.rand() < 0.1)
        env._fail = bool(rng.rand() < 0.1)
        env._success = bool(rng.rand() < 0.1)
        env._num_connected = int(rng.randint(0, 4))
        fail_in = env._fail
        term, log, pen = F.FurnitureEnv._after_step(env, reward, terminal_in, {})
        rows.append([reward, terminal_in, fail_in, term, pen, env._episode_length, env._episode_reward, log.get("episode_reward", np.nan),
                     log.get("episode_length", -1), log.get("episode_unstable", np.nan), env._fail])
        if term:  # _after_reset (furniture.py:336-346)
            env._episode_reward, env._episode_length = 0.0, 0
    out["after_step"] = np.array(rows, dtype=float)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()
