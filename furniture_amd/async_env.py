"""Asynchronous batched env (EnvPool-style ``send`` / ``recv``) over ONE FSim handle -- for learners that do not need all
envs to advance in lockstep.

Why it exists: a synchronous ``step()`` of 4096 envs ends with its slowest wave, and ~3 % of the envs (a
gripper touching a part: 5-8 Newton iterations per substep on a coupled island) take 3-6x longer than the rest, so the chip idles
for a third of every step.  Envs are independent, so nothing forces the cheap 97 % to wait: here the envs that are *predicted*
expensive (the step kernel writes a per-env key: shader time of the last step, "hand within 10 cm of a part", "will reset next
step") are batched apart from the cheap ones and the two kinds of batches run concurrently on separate HIP streams of the same
handle (``fsim_step_subset``).  An env simply advances more slowly while it is expensive.  Per-env trajectories are
bit-identical to synchronous stepping (tests/test_async_gpu.py).

MEASURED (scripts/bench_async.py, 4096 envs Sawyer + table_lack_0825): no throughput gain -- 395 k aggregate env-steps/s against
445 k for the synchronous two-slab bench.  Envs that newly couple inside a cheap batch still give it a tail, expensive batches
share their SIMDs with the cheap batch and run slower than they do alone, and with a fixed number of steps per env the run ends with
the env that stayed coupled longest.  The API is kept for learners that want per-env asynchrony (the semantics), not for speed.

The reference's analogue is the worker loop of ``SubprocVecEnv`` (furniture/util/subproc_vec_env.py:15-48), which is already
asynchronous underneath: ``step_async`` sends to every worker, ``step_wait`` receives from every worker; ``send`` / ``recv`` expose
that per env instead of per batch.
"""

import time

import numpy as np

from .envs import FurnitureBatchEnv
from .sim import INFO_NEEDS_TABLE

NEAR_BIT = 1 << 30


class FurnitureAsyncBatchEnv:
    """send(actions, env_ids) / recv() -> (env_ids, obs rows, reward, done, info rows).  Device tensors throughout."""

    def __init__(self, agent, num_envs, config=None, device=0, first_env_index=0, cheap_min_fraction=0.5, cost_ratio=1.6,
                 n_expensive_queues=2, use_near_hint=True, **kw):
        self.b = FurnitureBatchEnv(agent, num_envs, config=config, device=device, first_env_index=first_env_index, auto_reset=True, **kw)
        self.b.refill_tables_every_step = False
        self.sim, self.num_envs, self.dof = self.b.sim, num_envs, self.b.dof
        torch = self.sim.torch
        self.torch, self.device = torch, self.sim.device
        self._keys = torch.zeros(num_envs, dtype=torch.int32, device=self.device)
        self._expensive = np.zeros(num_envs, dtype=bool)       # class of each env: predicted cost of its NEXT step
        self._pending = np.zeros(num_envs, dtype=bool)         # action received, not launched yet
        self._inflight = {}                                    # queue -> (ids numpy, ids tensor)
        self._c_queue, self._e_queues = 0, list(range(1, 1 + n_expensive_queues))
        self.cheap_min = max(1, int(cheap_min_fraction * num_envs))
        self.cost_ratio = cost_ratio
        self.use_near_hint = use_near_hint
        self._cheap_cost = None                                # running median of the cheap class's step cost (key units)
        self.stats = dict(cheap_batches=0, expensive_batches=0, cheap_envs=0, expensive_envs=0)

    # -- reference-shaped pieces ---------------------------------------------------------------------------------------------
    @property
    def observation_space(self):
        return self.b.observation_space

    @property
    def action_space(self):
        return self.b.action_space

    def reset(self):
        """Synchronous reset of every env; all envs are then idle (waiting for ``send``)."""
        assert not self._inflight
        ob = self.b.reset()
        self._pending[:] = False
        self._expensive[:] = False
        return ob

    # -- send / recv -----------------------------------------------------------------------------------------------------
    def send(self, actions, env_ids):
        """actions [len(env_ids), dof] (device tensor or array) for idle envs ``env_ids`` (array of ints)."""
        torch = self.torch
        ids = np.asarray(env_ids, dtype=np.int64)
        if len(ids) == 0:
            return
        assert not self._pending[ids].any(), "send(): an env already has an action pending"
        a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions, dtype=np.float32))
        idt = torch.as_tensor(ids, device=self.device)
        self.b._act.index_copy_(0, idt, a.to(self.device).reshape(len(ids), -1))
        self._pending[ids] = True
        self._pump()

    def _launch(self, queue, ids):
        torch = self.torch
        # longest job first inside the batch (the grid is dispatched in blockIdx order)
        k = self._keys_host[ids]
        order = np.argsort(-np.where(k < 0, np.int64(1) << 40, (k & (NEAR_BIT - 1)).astype(np.int64) + ((k & NEAR_BIT) != 0) * (np.int64(1) << 31)), kind="stable")
        ids = ids[order]
        idt = torch.as_tensor(ids.astype(np.int32), device=self.device)
        torch.cuda.current_stream(self.device).synchronize()  # actions / id list resident before the queue's stream runs
        b = self.b
        self.sim.step_subset(queue, idt, len(ids), b._act, b._obs, b._rew, b._done, b._info, self._keys)
        self._inflight[queue] = (ids, idt)
        self._pending[ids] = False

    _keys_host = None

    def _pump(self, force=False):
        if self._keys_host is None:
            self._keys_host = np.zeros(self.num_envs, dtype=np.int64)
        pend = np.nonzero(self._pending)[0]
        if len(pend) == 0:
            return
        pe = pend[self._expensive[pend]]
        for q in self._e_queues:
            if len(pe) and q not in self._inflight:
                self._launch(q, pe)
                self.stats["expensive_batches"] += 1
                self.stats["expensive_envs"] += len(pe)
                pe = pe[:0]
        pc = pend[~self._expensive[pend]]
        if len(pc) and self._c_queue not in self._inflight and (len(pc) >= self.cheap_min or force or not self._inflight):
            self._launch(self._c_queue, pc)
            self.stats["cheap_batches"] += 1
            self.stats["cheap_envs"] += len(pc)

    def _complete(self, queue):
        torch = self.torch
        ids, idt = self._inflight.pop(queue)
        b = self.b
        idl = idt.long()
        keys = self._keys.index_select(0, idl).cpu().numpy().astype(np.int64)
        self._keys_host[ids] = keys
        cost = keys & (NEAR_BIT - 1)
        cheap_now = ~self._expensive[ids]
        if cheap_now.any():
            med = float(np.median(cost[cheap_now & (keys >= 0)])) if (cheap_now & (keys >= 0)).any() else None
            if med is not None:
                self._cheap_cost = med if self._cheap_cost is None else 0.8 * self._cheap_cost + 0.2 * med
        thr = self.cost_ratio * self._cheap_cost if self._cheap_cost else np.inf
        self._expensive[ids] = (keys < 0) | (((keys & NEAR_BIT) != 0) & self.use_near_hint) | (cost > thr)
        need = b._info.index_select(0, idl)[:, INFO_NEEDS_TABLE].cpu().numpy().astype(bool)
        if need.any():  # host-side reference RNG stream for the envs that consumed their reset table (not in flight: safe)
            mask = np.zeros(self.num_envs, dtype=bool)
            mask[ids[need]] = True
            b._refill(mask)
        return ids, idl

    def recv(self):
        """Blocks until one batch has finished; returns (env_ids numpy, obs [k, obs_dim], reward [k], done [k], info [k, INFO_DIM])
        for the envs of that batch, which are idle again."""
        if not self._inflight:
            self._pump(force=True)
            if not self._inflight:
                raise RuntimeError("recv(): nothing in flight and nothing pending")
        while True:
            for q in list(self._inflight):
                if not self.sim.queue_busy(q):
                    ids, idl = self._complete(q)
                    b = self.b
                    out = (ids, b._obs.index_select(0, idl), b._rew.index_select(0, idl), b._done.index_select(0, idl).bool(),
                           b._info.index_select(0, idl))
                    self._pump()
                    return out
            time.sleep(0)

    def drain(self):
        """Receive everything that is in flight or pending (used at the end of a run)."""
        out = []
        while self._inflight or self._pending.any():
            out.append(self.recv())
        return out

    def close(self):
        for q in list(self._inflight):
            self.sim.queue_sync(q)
        self._inflight.clear()
        self.b.close()
