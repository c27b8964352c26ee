"""SubprocVecEnv-shaped view of the batched GPU env (furniture/util/vec_env.py:53-163, subproc_vec_env.py:15-121,
furniture/env/base.py:55-80 make_vec_env): numpy in / numpy out, one info dict per env, auto-reset on done with the reset
observation returned in place of the terminal one (subproc_vec_env.py:16-20), env i seeded seed + i (base.py:77).

The reference spawns num_env worker processes, each stepping MuJoCo; here the whole batch is one kernel launch, so
step_async() really is asynchronous (the launch is enqueued on the handle's HIP stream) and step_wait() is the sync."""
from collections import OrderedDict

import numpy as np

from .envs import GYM_IDS, REGISTRY, FurnitureBatchEnv, make_config
from .sim import INFO_EPISODE_REWARD_F


class AlreadySteppingError(Exception):
    """vec_env.py:31-40"""

    def __init__(self):
        Exception.__init__(self, "already running an async step")


class NotSteppingError(Exception):
    """vec_env.py:42-50"""

    def __init__(self):
        Exception.__init__(self, "not running an async step")


class FurnitureVecEnv:
    closed = False
    viewer = None
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, env_id, num_env, config=None, env_kwargs=None, device=0, first_env_index=0):
        """env_id: a registered gym id (furniture/env/__init__.py) or an env class name; config/env_kwargs as in
        make_vec_env (config attributes override env_kwargs, base.py:69-73)."""
        if env_id in GYM_IDS:
            name, defaults = GYM_IDS[env_id]
        elif env_id in REGISTRY:
            name, defaults = env_id, {}
        else:
            raise Exception("unknown env id %s" % env_id)
        kw = dict(defaults)
        kw.update(env_kwargs or {})
        if config is not None:
            kw.update(config.__dict__)
        cls = REGISTRY[name]
        self._b = FurnitureBatchEnv(cls._agent, num_env, config=make_config(**kw), device=device, first_env_index=first_env_index,
                                    auto_reset=True, dense=cls._dense)
        self.num_envs = num_env
        self.observation_space = self._b.observation_space
        self.action_space = self._b.action_space
        self.waiting = False

    # -- VecEnv surface ---------------------------------------------------------------------------------
    def _np_obs(self, ob):
        return OrderedDict((k, v.double().cpu().numpy()) for k, v in ob.items())

    def reset(self):
        self._assert_not_closed()
        if self.waiting:  # "that work will be cancelled" (vec_env.py:77-79): drain it, the reset overwrites the result
            self._b.step_wait()
            self.waiting = False
        return self._np_obs(self._b.reset())

    def step_async(self, actions):
        self._assert_not_closed()
        if self.waiting:
            raise AlreadySteppingError()
        if isinstance(actions, (list, tuple)) and len(actions) and isinstance(actions[0], dict):
            keys = list(self.action_space.spaces.keys())
            actions = np.stack([np.concatenate([a[k] for k in keys]) for a in actions])
        self._b.step_async(np.asarray(actions, dtype=np.float32))
        self.waiting = True

    def step_wait(self):
        self._assert_not_closed()
        if not self.waiting:
            raise NotSteppingError()
        ob, rew, done, info = self._b.step_wait()
        self.waiting = False
        done_np = done.cpu().numpy()
        cols = {k: v.cpu().numpy() for k, v in info.items()}
        ep_rew = self._b._info[:, INFO_EPISODE_REWARD_F].view(self._b.sim.torch.float32).cpu().numpy()
        infos = []
        for i in range(self.num_envs):
            d = {k: int(v[i]) for k, v in cols.items()}
            if done_np[i]:  # the terminal step_log of _after_step (furniture.py:466-476)
                d.update(episode_success=int(cols["episode_success"][i]), episode_reward=float(ep_rew[i]),
                         episode_length=int(cols["episode_length"][i]), episode_num_connected=int(cols["num_connected"][i]),
                         episode_unstable=-float(self._b.config.unstable_penalty_coef) if cols["fail"][i] else 0)
            infos.append(d)
        return self._np_obs(ob), rew.double().cpu().numpy(), done_np, tuple(infos)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self):
        raise NotImplementedError("rendering (Unity / MuJoCo viewer) is outside the accelerated hot path")

    def render(self, mode="human"):
        return self.get_images()

    @property
    def unwrapped(self):
        return self

    # -- full-state snapshot (the reference's get_env_state is {qpos, qvel} only and loses welds / masks / groups)
    def get_env_state(self):
        return {k: v.cpu().numpy() for k, v in self._b.get_env_state().items()}

    def set_env_state(self, state):
        self._b.set_env_state(state)

    def close_extras(self):
        if self.waiting:
            self._b.step_wait()
            self.waiting = False
        self._b.close()

    def close(self):
        if self.closed:
            return
        self.close_extras()
        self.closed = True

    def _assert_not_closed(self):
        assert not self.closed, "Trying to operate on a FurnitureVecEnv after calling close()"

    def __del__(self):
        if not self.closed:
            try:
                self.close()
            except Exception:
                pass


def make_vec_env(env_id, num_env, config=None, env_kwargs=None, device=0):
    """furniture/env/base.py:55-80."""
    return FurnitureVecEnv(env_id, num_env, config=config, env_kwargs=env_kwargs, device=device)
