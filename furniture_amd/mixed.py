"""Mixed-furniture batch (BASELINE config 5, SURVEY §8 d/e): the lanes of one batch cycle through several furniture
models with different nq / nv / contact counts.

The reference runs heterogeneous batches as one subprocess per env (``furniture/env/base.py:55-80``), each with its own
model.  Here a lane keeps its GLOBAL env index i (so it is still seeded ``seed + i``, ``base.py:77``) and is served by the
FSim handle of its furniture: lane i -> ``furniture_names[i % k]``.  Every furniture's lanes form one sub-batch with its own
handle and HIP stream; the sub-batches' step kernels are enqueued back to back and run concurrently on the GPU, so the
wave-per-env kernel never sees divergent model sizes inside a launch (no per-lane padding of the physics state).  Only the
observation slab is padded: rows are ``max_obs_dim`` wide, ``object_ob`` padded to the largest part count, which is the shape
the RCCL all-gather to the learner moves (``furniture_amd/dist.py``).

With four or more furniture models set ``GPU_MAX_HW_QUEUES`` >= 2 x models in the environment BEFORE the first HIP call: with the
runtime's default of 4 hardware queues four handles' step kernels ran two at a time (measured, DESIGN_HISTORY.md section 6); three models
overlap fine with the default.
"""

from collections import OrderedDict

import numpy as np

from .envs import FurnitureBatchEnv, make_config


def lane_assignment(num_envs, k, first_env_index=0):
    """Global lane i -> model i % k.  Returns, per model, the local rows it owns inside this rank's slab."""
    g = first_env_index + np.arange(num_envs)
    return [np.nonzero(g % k == j)[0] for j in range(k)]


def padded_layout(n_parts_each, robot_dim):
    """Column layout of the padded observation row: [object_ob (7 * max parts) | robot_ob]."""
    pmax = max(n_parts_each)
    return 7 * pmax, 7 * pmax + robot_dim


class FurnitureMixedBatchEnv:
    """VecEnv-shaped like FurnitureBatchEnv, over several furniture models of one agent."""

    def __init__(self, agent, furniture_names, num_envs, config=None, device=0, first_env_index=0, auto_reset=True, **kw):
        cfg = config if config is not None else make_config()
        for key, v in kw.items():
            setattr(cfg, key, v)
        self.agent, self.names, self.num_envs = agent, list(furniture_names), int(num_envs)
        k = len(self.names)
        self.rows = lane_assignment(num_envs, k, first_env_index)
        self.subs = []
        for j, name in enumerate(self.names):
            c = make_config(**vars(cfg))
            c.furniture_name = name
            idx = (first_env_index + self.rows[j]).tolist()
            self.subs.append(FurnitureBatchEnv(agent, len(idx), config=c, device=device, auto_reset=auto_reset, env_indices=idx)
                             if idx else None)
        live = [s for s in self.subs if s is not None]
        torch = live[0].sim.torch
        self.torch, self.device = torch, live[0].sim.device
        self.dof = live[0].dof
        assert all(s.dof == self.dof for s in live), "one agent, one control type: the action width is common"
        robot_dims = {s.sim.obs_dim - 7 * s.n_obj for s in live}
        assert len(robot_dims) == 1
        self.robot_dim = robot_dims.pop()
        self.obj_cols, self.obs_dim = padded_layout([s.n_obj for s in live], self.robot_dim)
        dev = self.device
        self._rows_t = [torch.as_tensor(r, device=dev, dtype=torch.long) for r in self.rows]
        self.obs = torch.zeros((num_envs, self.obs_dim), device=dev)
        self.reward = torch.zeros(num_envs, device=dev)
        self.done = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self.n_parts = torch.zeros(num_envs, dtype=torch.int32, device=dev)
        self.model_id = torch.zeros(num_envs, dtype=torch.int32, device=dev)
        for j, s in enumerate(self.subs):
            if s is not None:
                self.n_parts[self._rows_t[j]] = s.n_obj
                self.model_id[self._rows_t[j]] = j

    # -- scatter of one sub-batch's rows into the padded slab ---------------------------------------------------------
    def _scatter_obs(self, j, flat):
        s, r = self.subs[j], self._rows_t[j]
        ko = 7 * s.n_obj
        self.obs[r, :ko] = flat[:, :ko]
        if ko < self.obj_cols:
            self.obs[r, ko:self.obj_cols] = 0
        self.obs[r, self.obj_cols:] = flat[:, ko:]

    def _split(self):
        return OrderedDict([("object_ob", self.obs[:, :self.obj_cols]), ("robot_ob", self.obs[:, self.obj_cols:])])

    def reset(self):
        for j, s in enumerate(self.subs):
            if s is not None:
                s.reset()
                self._scatter_obs(j, s._obs)
        return self._split()

    def step_async(self, actions):
        torch = self.torch
        a = actions["default"] if isinstance(actions, dict) else actions
        if not torch.is_tensor(a):
            a = torch.as_tensor(np.asarray(a, dtype=np.float32))
        a = a.to(self.device).reshape(self.num_envs, -1)
        for j, s in enumerate(self.subs):  # all launches are enqueued before the first wait: the kernels overlap
            if s is not None:
                s.step_async(a.index_select(0, self._rows_t[j]))

    def step_wait(self):
        infos = []
        for j, s in enumerate(self.subs):
            if s is None:
                infos.append(None)
                continue
            _, rew, done, info = s.step_wait()
            r = self._rows_t[j]
            self._scatter_obs(j, s._obs)
            self.reward[r] = rew
            self.done[r] = done.to(self.torch.uint8)
            infos.append(info)
        keys = [k for k in next(i for i in infos if i is not None)]
        out = {}
        for key in keys:
            t = self.torch.zeros(self.num_envs, dtype=self.torch.int32, device=self.device)
            for j, info in enumerate(infos):
                if info is not None:
                    t[self._rows_t[j]] = info[key].to(self.torch.int32)
            out[key] = t
        out["n_parts"], out["model_id"] = self.n_parts, self.model_id
        return self._split(), self.reward, self.done.bool(), out

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def gather(self):
        """RCCL all-gather of the padded slab (+reward, done) to the learner: [world * num_envs, obs_dim] in global lane order."""
        from .dist import gather_observations
        return gather_observations(self.obs, self.reward, self.done)

    def close(self):
        for s in self.subs:
            if s is not None:
                s.close()
