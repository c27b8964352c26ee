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
    # ---- call traces of _do_simulation (furniture.py:2857-2897) and of _step's post-connect block (:405-449) -------------------------
    T_TOK = ["setctrl", "forward", "step", "stop1", "stop0", "reset", "fail", "move_target", "get_obs", "scan"]
    out["trace_tokens"] = np.array(T_TOK)

    def fake(agent, log, selected=(None, None), groups=None, raise_at=None):
        env = types.SimpleNamespace()
        env._agent_type, env._control_type = agent, "impedance"
        env._control_timestep, env._model_timestep, env._cur_time = 0.1, 0.002, 0.0
        env._object_names = ["p0", "p1", "p2", "p3"]
        env._cursor_selected = list(selected)
        grp = groups or {n: i for i, n in enumerate(env._object_names)}
        env._find_group = lambda n: grp[n]
        env._stop_object = lambda n, gravity=1: log.append("stop%d" % gravity)

        class Ctrl:
            def __setitem__(self, k, v): log.append("setctrl")
        count = dict(n=0)

        def step():
            count["n"] += 1
            if raise_at is not None and count["n"] == raise_at:
                raise RuntimeError("unstable")
            log.append("step")
        env.sim = types.SimpleNamespace(data=types.SimpleNamespace(ctrl=Ctrl()), forward=lambda: log.append("forward"), step=step)
        env.set_init_qpos = lambda q: None
        env.reset = lambda: log.append("reset")
        env._fail = False
        return env

    traces = {}
    log = []; e = fake("Sawyer", log); F.FurnitureEnv._do_simulation(e, np.zeros(9)); traces["dosim_sawyer"] = log
    log = []; e = fake("Cursor", log, selected=("p1", None), groups={"p0": 0, "p1": 1, "p2": 1, "p3": 3}); F.FurnitureEnv._do_simulation(e, None); traces["dosim_cursor"] = log
    log = []; e = fake("Sawyer", log, raise_at=7); F.FurnitureEnv._do_simulation(e, np.zeros(9)); traces["dosim_unstable"] = log + (["fail"] if e._fail else [])
    # _step: Sawyer, a connect happened inside _step_continuous
    log = []
    e = fake("Sawyer", log)
    e._connected_body1, e._connected_body1_pos, e._connected_body1_quat, e._gravity_compensation = "p0", np.zeros(3), np.array([1.0, 0, 0, 0]), 0
    e._step_continuous = lambda a: log.append("scan")
    e._move_objects_target = lambda *a: log.append("move_target")
    e._get_obs = lambda: (log.append("get_obs"), {})[1]
    e._num_connected, e._success_num_conn, e._success = 4, 4, False
    ob, rew, done, info = F.FurnitureEnv._step(e, np.zeros(9))
    traces["step_postconnect"] = log
    out["step_postconnect_result"] = np.array([rew, float(done), float(e._success), float(e._connected_body1 is None)])
    for k, v in traces.items():
        out["trace_" + k] = np.array([T_TOK.index(t) for t in v], dtype=np.int16)
        print(k, len(v), v[:6], "...", v[-3:])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()
