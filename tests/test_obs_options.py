"""Observation options of FurnitureEnv._get_obs (furniture.py:1344-1387): object_ob_all=False keeps only the two parts of the
current subtask (in part order; a 14-zero dummy when no subtask is left), subtask_ob=True adds their 1-based ids."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from furniture_amd.envs import FurnitureBatchEnv, make_config


def _fake(n_obj, **cfg):
    m = SimpleNamespace(neq=4, eq_part1=np.array([0, 2, 3, 1]), eq_part2=np.array([4, 4, 4, 4]))
    return SimpleNamespace(config=make_config(**cfg), sim=SimpleNamespace(torch=torch), n_obj=n_obj, model=m)


def test_split_selects_subtask_parts_in_part_order():
    n_obj, n = 5, 3
    flat = torch.arange(n * (7 * n_obj + 4), dtype=torch.float32).reshape(n, -1)
    sub = torch.tensor([[4, 1], [0, 4], [-1, -1]])
    ob = FurnitureBatchEnv._split(_fake(n_obj, object_ob_all=False, subtask_ob=True), flat, sub)
    assert list(ob) == ["object_ob", "subtask_ob", "robot_ob"]
    parts = flat[:, :35].reshape(n, 5, 7)
    assert torch.equal(ob["object_ob"][0], torch.cat([parts[0, 1], parts[0, 4]]))   # ascending part index, not (part1, part2)
    assert torch.equal(ob["object_ob"][1], torch.cat([parts[1, 0], parts[1, 4]]))
    assert torch.equal(ob["object_ob"][2], torch.zeros(14))                          # dummy
    assert ob["subtask_ob"].tolist() == [[5, 2], [1, 5], [0, 0]]
    assert torch.equal(ob["robot_ob"], flat[:, 35:])
    # right after a reset the subtask is weld 0's parts
    ob0 = FurnitureBatchEnv._split(_fake(n_obj, object_ob_all=False, subtask_ob=True), flat, None)
    assert ob0["subtask_ob"].tolist() == [[1, 5]] * n
    # defaults: everything, no subtask_ob
    obd = FurnitureBatchEnv._split(_fake(n_obj), flat, sub)
    assert list(obd) == ["object_ob", "robot_ob"] and obd["object_ob"].shape == (n, 35)


@pytest.mark.gpu
def test_obs_options_match_oracle_env():
    from furniture_amd.mjcf.model import load_compiled
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    from tests.scenarios import counter_actions
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=50,
              object_ob_all=False, subtask_ob=True, seed=123)
    env = FurnitureBatchEnv("Sawyer", 2, config=make_config(**kw), auto_reset=False)
    assert env.observation_space.spaces["object_ob"].shape == (14,) and env.observation_space.spaces["subtask_ob"].shape == (2,)
    m = load_compiled("Sawyer", "table_lack_0825")
    oracles = [FurnitureEnvOracle(m, OracleConfig(max_episode_steps=50, seed=123 + i, solver_tolerance=1e-10, object_ob_all=False, subtask_ob=True))
               for i in range(2)]
    ob = env.reset()
    obs_o = [e.reset() for e in oracles]
    for i, e in enumerate(oracles):
        for k in ("object_ob", "subtask_ob", "robot_ob"):
            assert np.abs(ob[k][i].cpu().numpy() - obs_o[i][k]).max() < 1e-4, k
    for t in range(2):
        a = np.stack([counter_actions(5, i, t, env.dof) for i in range(2)])
        ob, rew, done, info = env.step(a)
        for i, e in enumerate(oracles):
            o, r, d, _ = e.step(a[i])
            for k in ("object_ob", "subtask_ob", "robot_ob"):
                assert np.abs(ob[k][i].cpu().numpy() - o[k]).max() < 5e-4, k
    env.close()


@pytest.mark.parametrize("flag", ["load_demo", "record_demo"])
def test_unsupported_reference_options_fail_loudly(flag):
    """Options of furniture/config/furniture.py that change the reset / connect flow and are not built raise before any device work
    (no silent ignore); checked on CPU: the guard sits ahead of the FSim construction."""
    with pytest.raises(NotImplementedError, match=flag):
        FurnitureBatchEnv("Sawyer", 1, config=make_config(unity=False, record_vid=False, control_type="impedance",
                                                          furniture_name="table_lack_0825", **{flag: True}))


@pytest.mark.gpu
def test_bf16_observation_slab_is_the_rounded_f32_one():
    """fsim_config_t.obs_bf16 (BASELINE config 2's narrow observation slab; SURVEY 8b `void* obs f32|bf16`): same state, same
    arithmetic, the finished observation stored as bfloat16 with round-to-nearest-even -- i.e. exactly torch's f32 -> bf16 cast
    of the float32 slab, after a reset and after steps; reward / done are unaffected."""
    import torch
    from furniture_amd.envs import FurnitureBatchEnv, make_config
    kw = dict(unity=False, record_vid=False, control_type="impedance", furniture_name="table_lack_0825", max_episode_steps=150, seed=5)
    a = FurnitureBatchEnv("Sawyer", 16, config=make_config(**kw))
    b = FurnitureBatchEnv("Sawyer", 16, config=make_config(**kw), obs_bf16=True)
    oa, ob = a.reset(), b.reset()
    assert ob["object_ob"].dtype == torch.bfloat16 and ob["robot_ob"].shape == oa["robot_ob"].shape
    rng = np.random.RandomState(0)
    for t in range(4):
        for k in oa:
            assert torch.equal(oa[k].to(torch.bfloat16), ob[k]), (t, k)
        act = rng.uniform(-1, 1, (16, 9)).astype(np.float32)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(ra, rb) and torch.equal(da, db)
    a.close()
    b.close()


def test_argparse_intake_has_the_reference_option_names():
    """furniture/config/__init__.py:7-35: create_parser(env) -> namespace with the reference's option names and defaults"""
    import sys
    from furniture_amd.config import create_parser
    argv, sys.argv = sys.argv, ["prog"]
    try:
        cfg, _ = create_parser("IKEASawyer-v0").parse_known_args(["--max_episode_steps", "50", "--preassembled", "0,1", "--unity", "False"])
        assert cfg.max_episode_steps == 50 and cfg.preassembled == [0, 1] and cfg.unity is False and cfg.control_type == "ik"
        assert cfg.alignment_pos_dist == 0.1 and cfg.seed == 123 and cfg.num_connects is None
        d, _ = create_parser("IKEASawyerDense-v0").parse_known_args([])
        assert d.max_episode_steps == 150 and d.control_type == "impedance" and d.phase_bonus == 5000.0 and d.phase_ob is False
    finally:
        sys.argv = argv
