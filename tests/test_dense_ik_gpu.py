"""The dense-reward env under IK control, driven through its whole recipe by the scripted policy (furniture_amd/scripted.py) -- the
reference's training env (furniture_sawyer_dense.py) finishing its task: four subtasks, 8 phases each, success.

The recipe (dense_subtasks) prescribes the leg order and the table connector of each leg; any other connection ends the episode."""
import numpy as np
import pytest

from furniture_amd.dense import dense_subtasks
from furniture_amd.mjcf.model import load_compiled
from furniture_amd.scripted import PickAndAttach


# config/furniture_sawyer_dense.py:5-14: no auto-align, strict alignment thresholds
STRICT = dict(auto_align=False, alignment_pos_dist=0.02, alignment_rot_dist_up=0.99, alignment_rot_dist_forward=0.99, alignment_project_dist=0.0)


def _recipe(m):
    sub = dense_subtasks(m)[0]
    return [int(d["leg_part"]) for d in sub], [int(d["k_table"]) for d in sub]


def test_scripted_recipe_on_the_dense_oracle_env():
    from oracle.dense_reward import DenseConfig
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled("Sawyer", "table_lack_0825")
    legs, conns = _recipe(m)
    e = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=4000, seed=123, control_type="ik_quaternion", dense=DenseConfig(), **STRICT))
    ob = e.reset()
    log = dict(done_at=None, n=0, info=None)

    def step(a):
        ob, r, d, info = e.step(a[0].astype(np.float64))
        log["n"] += 1
        if d and log["done_at"] is None:
            log["done_at"], log["info"] = log["n"], info
        return {"object_ob": ob["object_ob"][None], "robot_ob": ob["robot_ob"][None]}, np.array([r]), d, {"num_connected": np.array([info["num_connected"]])}
    total, ncon, _ = PickAndAttach(m, 1, hover=0.008).run(step, {"object_ob": ob["object_ob"][None], "robot_ob": ob["robot_ob"][None]}, legs=legs, table_connectors=conns)
    # done only once, on the last step, by success; every subtask's bonuses were paid (success alone pays ~1e4)
    assert ncon[0] == 4 and log["done_at"] == log["n"] and log["info"]["success"] == 1 and log["info"]["subtask"] == 4
    assert total[0] > 5e4


@pytest.mark.gpu
def test_scripted_recipe_on_the_dense_device_env():
    """8 placements on the HIP path: at least half walk the whole recipe (done by success exactly when num_connected reaches 4, phase
    counter at subtask 4), env 0 among them with a summed reward within 25 % of the fp64 oracle env's for the same seed (600-step
    closed-loop trajectories diverge in fp32; the phase bonuses dominate the sum)."""
    from furniture_amd.envs import FurnitureBatchEnv, make_config, DENSE_OVERRIDES
    m = load_compiled("Sawyer", "table_lack_0825")
    legs, conns = _recipe(m)
    n = 8
    kw = dict(DENSE_OVERRIDES)
    kw.update(unity=False, record_vid=False, control_type="ik_quaternion", furniture_name="table_lack_0825", max_episode_steps=6000, seed=123)
    env = FurnitureBatchEnv("Sawyer", n, config=make_config(**kw), auto_reset=False, dense=True)
    ob = env.reset()
    done_any, succ, phase = np.zeros(n, bool), np.zeros(n, bool), np.zeros(n, int)

    def step(a):
        nonlocal done_any, succ, phase
        ob, r, d, info = env.step(a)
        live = ~done_any
        dn = d.cpu().numpy().astype(bool)
        succ |= live & dn & (info["episode_success"].cpu().numpy() != 0)
        phase = np.where(live, info["phase_i"].cpu().numpy(), phase)
        done_any |= dn
        return ob, np.where(live, r.cpu().numpy(), 0.0), d, {"num_connected": info["num_connected"]}
    total, ncon, _ = PickAndAttach(m, n, hover=0.008).run(step, ob, legs=legs, table_connectors=conns)
    print("dense env under the scripted recipe: num_connected", ncon.tolist(), "success", succ.tolist(), "phase_i", phase.tolist(), "reward", total.round(0).tolist())
    assert (succ == (ncon == 4)).all() and succ.sum() >= n // 2 and succ[0]
    assert (phase[succ] // 8 >= 3).all()
    env.close()
    from oracle.dense_reward import DenseConfig
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    e = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=6000, seed=123, control_type="ik_quaternion", dense=DenseConfig(), **STRICT))
    ob = e.reset()
    fin = [False]

    def ostep(a):
        ob, r, d, info = e.step(a[0].astype(np.float64))
        r = 0.0 if fin[0] else r
        fin[0] |= d
        return {"object_ob": ob["object_ob"][None], "robot_ob": ob["robot_ob"][None]}, np.array([r]), d, {"num_connected": np.array([info["num_connected"]])}
    t_o, n_o, _ = PickAndAttach(m, 1, hover=0.008).run(ostep, {"object_ob": ob["object_ob"][None], "robot_ob": ob["robot_ob"][None]}, legs=legs, table_connectors=conns)
    print("oracle env, seed 123: reward %.0f; device env 0: %.0f" % (t_o[0], total[0]))
    assert n_o[0] == 4 and abs(total[0] - t_o[0]) < 0.25 * t_o[0]
