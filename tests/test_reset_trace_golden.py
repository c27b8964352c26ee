"""FurnitureEnv._reset (furniture.py:1406-1663) run by the REFERENCE on a fake self (tests/golden/reset_trace.npz,
scripts/make_golden_reset_trace.py) against the oracle env's reset(): same order of sim.reset / forward / step, stabilisation
loops, gravity-compensation writes, robot-collision off / on, robot re-posing, state zeroing, IK sync -- the structure behind
"a reset costs 301 / 401 physics substeps" that the device's in-kernel env_reset is diffed against."""
import os

import numpy as np
import pytest

from furniture_amd.mjcf.model import load_compiled
from oracle.oracle_env import FurnitureEnvOracle, OracleConfig

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reset_trace.npz"))
TOK = [str(t) for t in G["tokens"]]
COLLAPSE = {"robotcol_off", "robotcol_on", "partcol_on", "gravcomp", "setqpos", "weld_off"}   # one token per geom / part / index set
IGNORE = {"zero_ctrl"}  # the Cursor model has no actuators: nothing to zero on our side


def _collapse(seq):
    out = []
    for t in seq:
        if t in IGNORE:
            continue
        if out and out[-1] == t and t in COLLAPSE:
            continue
        out.append(t)
    return out


class _LogArr:
    """in-place writes to a simulator array, classified into trace tokens"""
    def __init__(self, real, log, classify):
        self._r, self._log, self._c = real, log, classify

    def __getitem__(self, k):
        return self._r[k]

    def __setitem__(self, k, v):
        tok = self._c(k, v)
        if tok:
            self._log.append(tok)
        self._r[k] = v

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._r, dtype=dtype)

    def __len__(self):
        return len(self._r)

    def copy(self):
        return self._r.copy()

    @property
    def shape(self):
        return self._r.shape


class _Proxy:
    def __init__(self, real, over):
        object.__setattr__(self, "_real", real)
        object.__setattr__(self, "_over", over)

    def __getattr__(self, n):
        o = object.__getattribute__(self, "_over")
        return o[n] if n in o else getattr(object.__getattribute__(self, "_real"), n)

    def __setattr__(self, n, v):
        setattr(object.__getattribute__(self, "_real"), n, v)


def _oracle_trace(agent, furniture, control_type):
    m = load_compiled(agent, furniture)
    e = FurnitureEnvOracle(m, OracleConfig(seed=1, control_type=control_type))
    log = []
    sim = e.sim
    robot = m.geom_is_robot.astype(bool)

    def cls_mask(k, v):
        if isinstance(k, slice):
            return None                                   # wholesale restore of the compiled defaults
        k = np.asarray(k)
        if k.dtype == bool and np.array_equal(k, robot):
            return "robotcol_off" if np.all(np.asarray(v) == 0) else "robotcol_on"
        return "partcol_on"

    data = _Proxy(sim.data, dict(
        qfrc_applied=_LogArr(sim.data.qfrc_applied, log, lambda k, v: ("zero_applied" if k == slice(None) else None) if isinstance(k, slice) else "gravcomp"),
        ctrl=_LogArr(sim.data.ctrl, log, lambda k, v: "zero_ctrl"), xfrc_applied=_LogArr(sim.data.xfrc_applied, log, lambda k, v: "zero_xfrc" if isinstance(k, slice) and k == slice(None) else None),
        qacc=_LogArr(sim.data.qacc, log, lambda k, v: "zero_qacc"), qacc_warmstart=_LogArr(sim.data.qacc_warmstart, log, lambda k, v: "zero_warm")))
    model = _Proxy(sim.model, dict(
        geom_contype=_LogArr(sim.model.geom_contype, log, cls_mask),
        eq_active=_LogArr(sim.model.eq_active, log, lambda k, v: "weld_off" if np.all(np.asarray(v) == 0) else "weld_on")))
    e.sim = _Proxy(sim, dict(data=data, model=model, reset=lambda: (log.append("sim.reset"), sim.reset())[1],
                             forward=lambda: (log.append("forward"), sim.forward())[1], step=lambda: (log.append("step"), sim.step())[1]))
    stop, slow, init, nxt, setq = e._stop_object, e._slow_object, e._initialize_robot_pos, e._get_next_subtask, e._set_part_qpos
    e._stop_object = lambda i, gravity=1: (log.append("stop%d" % gravity) if i == 0 else None, stop(i, gravity))[1]
    e._slow_object = lambda i: (log.append("slow") if i == 0 else None, slow(i))[1]
    e._initialize_robot_pos = lambda: (log.append("initrobot"), init())[1]
    e._get_next_subtask = lambda: (log.append("next_subtask"), nxt())[1]
    e._set_part_qpos = lambda i, p, r: (log.append("setqpos"), setq(i, p, r))[1]
    import oracle.oracle_env as OE
    sp = OE.sample_placement
    OE.sample_placement = lambda *a, **k: (log.append("place"), sp(*a, **k))[1]
    try:
        e.reset()
    finally:
        OE.sample_placement = sp
    if control_type == "ik":  # controller.sync_state() is the last thing before _get_next_subtask (furniture.py:1647-1650)
        i = len(log) - 1 - log[::-1].index("next_subtask")
        log.insert(i, "ik_sync")
    return log


@pytest.mark.parametrize("key,agent,furniture,ctype", [
    ("Sawyer_recipe_impedance", "Sawyer", "table_lack_0825", "impedance"),
    ("Sawyer_norecipe_impedance", "Sawyer", "swivel_chair_0700", "impedance"),
    ("Cursor_recipe_impedance", "Cursor", "table_lack_0825", "impedance"),
    ("Sawyer_recipe_ik", "Sawyer", "table_lack_0825", "ik"),
])
def test_reset_call_trace_matches_reference_method(key, agent, furniture, ctype):
    m = load_compiled(agent, furniture)
    assert bool(m.meta.get("has_recipe")) == ("norecipe" not in key)
    want = _collapse([TOK[i] for i in G[key]])
    got = _collapse(_oracle_trace(agent, furniture, ctype))
    assert want.count("step") == (401 if "norecipe" not in key else 301)
    # the oracle writes its masks / welds in a different (vectorised) pattern around sim.reset; compare from the placement on
    w, g = want[want.index("place"):], got[got.index("place"):]
    first_diff = next((i for i in range(min(len(g), len(w))) if g[i] != w[i]), None)
    assert g == w, (len(g), len(w), first_diff, g[max(0, (first_diff or 0) - 2):(first_diff or 0) + 3], w[max(0, (first_diff or 0) - 2):(first_diff or 0) + 3])
    # and before it: one sim.reset, robot collisions off, part collisions on, welds off -- as sets
    assert set(want[:want.index("place")]) - {"weld_on"} >= {"sim.reset", "robotcol_off", "partcol_on"}
    assert {"sim.reset", "robotcol_off", "partcol_on"} <= set(got[:got.index("place")])
