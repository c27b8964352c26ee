"""Command-line intake of the reference (furniture/config/__init__.py:7-35 create_parser, furniture_gym.py:16-46 FurnitureGym):
the same option names, parsed into the namespace the accelerated envs take.  Only the options that reach the hot path exist
(furniture_amd.envs.DEFAULTS, the dense-reward coefficients for the dense ids); an option that is given but not built fails in the
env constructor, not here."""
import argparse

from .dense import DENSE_COEF_DEFAULTS
from .envs import DEFAULTS, DENSE_OVERRIDES, GYM_IDS, make_env


def str2bool(v):  # furniture/util/__init__.py:17-18
    return v.lower() == "true"


def str2intlist(value):  # furniture/util/__init__.py:21-25
    return value if not value else [int(num) for num in value.split(",")]


def create_parser(env=None):
    """furniture/config/__init__.py:7-35"""
    parser = argparse.ArgumentParser("IKEA Furniture Assembly Environment (MI355X batched path)", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--env", type=str, default=env if env is not None else "IKEASawyer-v0", help="Environment name")
    args, _ = parser.parse_known_args()
    defaults = dict(DEFAULTS)
    dense = args.env in ("IKEASawyerDense-v0", "furniture-sawyer-densereward-v0")
    if dense:  # furniture/config/furniture_sawyer_dense.py
        defaults.update({k: v for k, v in DENSE_COEF_DEFAULTS if k not in ("z_finedist", "griptip_site", "grip_site", "phase_ob")})
        defaults.update(DENSE_OVERRIDES)
        defaults.update(phase_ob=False, diff_rew=True)
    defaults.update(preassembled=[], num_connects=None, debug=False)
    for k, v in sorted(defaults.items()):
        if isinstance(v, bool):
            parser.add_argument("--" + k, type=str2bool, default=v)
        elif k == "preassembled":
            parser.add_argument("--" + k, type=str2intlist, default=v, help="list of weld equality ids to activate at start")
        elif k in ("furniture_name", "num_connects"):
            parser.add_argument("--" + k, type=str if k == "furniture_name" else int, default=v)
        else:
            parser.add_argument("--" + k, type=type(v), default=v)
    return parser


class FurnitureGym:
    """furniture/env/furniture_gym.py:11-80: FurnitureGym(id=..., name=..., **config overrides)"""

    def __init__(self, **kwarg):
        parser = create_parser(env=kwarg["id"])
        config, _ = parser.parse_known_args()
        for key, value in GYM_IDS.get(kwarg["id"], (None, {}))[1].items():
            setattr(config, key, value)
        for key, value in kwarg.items():
            setattr(config, key, value)
        self.env = make_env(kwarg["name"], config)
        self.observation_space, self.action_space = self.env.observation_space, self.env.action_space
        self.num_subtask, self.set_subtask = self.env.num_subtask, self.env.set_subtask
        self.set_init_qpos, self.get_env_state = self.env.set_init_qpos, self.env.get_env_state
        self._max_episode_steps = config.max_episode_steps

    def set_max_episode_steps(self, max_episode_steps):
        self._max_episode_steps = max_episode_steps
        self.env.set_max_episode_steps(max_episode_steps)

    def reset(self):
        return self.env.reset()

    def step(self, action):
        return self.env.step(action)

    def render(self, mode="human"):
        return self.env.render(mode)

    def close(self):
        self.env.close()
