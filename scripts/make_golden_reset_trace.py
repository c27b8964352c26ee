"""Generate tests/golden/reset_trace.npz: the call trace of FurnitureEnv._reset (furniture.py:1406-1663) run by the REFERENCE on a
fake self -- the order of sim.reset / forward / step, the stabilisation loops, gravity-compensation writes, robot-collision mask
off / on, robot re-posing and the state zeroing that make a reset cost 301 or 401 physics substeps.  Build container only."""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_env_logic import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reset_trace.npz")
TOKENS = ["sim.reset", "forward", "step", "stop0", "slow", "initrobot", "gravcomp", "robotcol_off", "robotcol_on", "partcol_on", "place",
          "setqpos", "zero_ctrl", "zero_applied", "zero_xfrc", "zero_qacc", "zero_warm", "next_subtask", "ik_sync", "weld_off"]


class Rec(np.ndarray):
    """numpy array whose writes are logged as tokens (the reference manipulates sim.data / sim.model arrays in place)."""
    def __new__(cls, arr, log, on_write):
        o = np.asarray(arr).view(cls)
        o._log, o._on = log, on_write
        return o

    def __array_finalize__(self, obj):
        self._log, self._on = getattr(obj, "_log", None), getattr(obj, "_on", None)

    def __setitem__(self, k, v):
        if self._log is not None and self._on is not None:
            tok = self._on(k, v)
            if tok:
                self._log.append(tok)
        np.ndarray.__setitem__(self, k, v)


def run(F, agent, recipe, control_type):
    log = []
    ngeom, nparts = 10, 3
    names = ["robot0", "robot1", "p0", "p1", "p2"]
    geom_body = np.array([0, 0, 1, 1, 2, 2, 3, 3, 4, 4])
    geom_names = ["r_a", "r_b", "r_c", "r_d", "p0_collision", "p0_vis", "p1_collision", "p1_vis", "p2_collision", "p2_vis"]
    env = types.SimpleNamespace()
    env._config = types.SimpleNamespace(furniture_name="x", furniture_id=0, furn_size_rand=0, fix_init=False, assembled=False, background=None)
    env._furniture_id, env._manual_resize, env._load_init_states = 0, None, None
    env._agent_type, env._control_type = agent, control_type
    env._object_names = ["p0", "p1", "p2"]
    env._object_body_ids = [2, 3, 4]
    env._object_body_id2name = {2: "p0", 3: "p1", 4: "p2"}
    env._num_connects, env._preassembled, env._recipe = None, [], ({"site_recipe": []} if recipe else None)
    env._init_qpos, env.init_pos, env.init_quat = None, None, None
    env._record_demo, env._unity, env._background = False, None, None
    env.mujoco_robot = types.SimpleNamespace(is_robot_part=lambda g: g.startswith("r_"))
    env._ref_joint_vel_indexes_all, env._ref_gripper_joint_vel_indexes_all = [0, 1], [2]
    env._right_hand_quat = np.array([0.0, 0, 0, 1])
    env._controller = types.SimpleNamespace(sync_state=lambda: log.append("ik_sync"))
    ct = Rec(np.array([1, 1, 1, 1, 0, 0, 0, 0, 0, 0]), log, None)
    ca = Rec(np.ones(ngeom, dtype=int), log, None)
    state = dict(phase="off")

    def on_ct(k, v):
        if isinstance(k, (int, np.integer)) and k < 4:
            return "robotcol_off" if v == 0 else "robotcol_on"
        if isinstance(k, (int, np.integer)) and k >= 4:
            return "partcol_on"
        return None
    ct._on = on_ct
    data = types.SimpleNamespace(
        ctrl=Rec(np.zeros(3), log, lambda k, v: "zero_ctrl"), qfrc_bias=np.arange(3.0),
        qfrc_applied=Rec(np.zeros(3), log, lambda k, v: "zero_applied" if isinstance(k, slice) else "gravcomp"),
        xfrc_applied=Rec(np.zeros((5, 6)), log, lambda k, v: "zero_xfrc"), qacc=Rec(np.zeros(3), log, lambda k, v: "zero_qacc"),
        qacc_warmstart=Rec(np.zeros(3), log, lambda k, v: "zero_warm"), time=0.0)
    model = types.SimpleNamespace(geom_bodyid=geom_body, body_names=names, geom_id2name=lambda g: geom_names[g], geom_contype=ct, geom_conaffinity=ca,
                                  eq_obj1id=np.array([2, 3]), eq_obj2id=np.array([4, 4]),
                                  eq_active=Rec(np.ones(2, dtype=int), log, lambda k, v: "weld_off" if v == 0 else "weld_on"))
    env.sim = types.SimpleNamespace(reset=lambda: log.append("sim.reset"), forward=lambda: log.append("forward"), step=lambda: log.append("step"),
                                    data=data, model=model)
    env._place_objects = lambda: (log.append("place"), ({n: np.zeros(3) for n in env._object_names}, {n: np.array([1.0, 0, 0, 0]) for n in env._object_names}))[1]
    env._set_qpos = lambda name, pos, rot=None: log.append("setqpos")
    env._stop_objects = lambda gravity=1: log.append("stop%d" % gravity)
    env._slow_objects = lambda: log.append("slow")
    env._initialize_robot_pos = lambda: log.append("initrobot")
    env._get_next_subtask = lambda: log.append("next_subtask")
    env._merge_groups = lambda a, b: None
    F.FurnitureEnv._reset(env)
    return log, env


def main():
    F = import_reference()
    out = {"tokens": np.array(TOKENS)}
    for agent, recipe, ctype in (("Sawyer", True, "impedance"), ("Sawyer", False, "impedance"), ("Cursor", True, "impedance"), ("Sawyer", True, "ik")):
        log, env = run(F, agent, recipe, ctype)
        unknown = sorted(set(log) - set(TOKENS))
        assert not unknown, unknown
        key = "%s_%s_%s" % (agent, "recipe" if recipe else "norecipe", ctype)
        out[key] = np.array([TOKENS.index(t) for t in log], dtype=np.int16)
        print(key, len(log), "events;", log.count("step"), "sim.step calls,", log.count("forward"), "forward calls")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
