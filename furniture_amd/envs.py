"""Host-side mirror of the reference's env surface for the accelerated path.

* ``FurnitureBatchEnv``  -- the batched replacement of ``SubprocVecEnv`` (furniture/util/subproc_vec_env.py):
  n envs of one (agent, furniture) pair on one GPU, VecEnv-shaped (reset / step_async / step_wait / step).
* ``FurnitureSawyerEnv`` / ``FurnitureBaxterEnv`` / ``FurnitureCursorEnv`` -- single-env classes with the reference's
  ``reset()/step()/observation_space/action_space/dof`` (furniture/env/furniture_{sawyer,baxter,cursor}.py).
* ``make_env`` / ``make`` -- name and gym-id registry (furniture/env/base.py:14-52, furniture/env/__init__.py:19-114).

Everything numeric happens in libfsim.so; this file is plumbing (config intake, reset-table sampling with the
reference's RNG stream, tensor allocation).  No CPU fallback: without the HIP library / a GPU construction raises.
"""

from collections import OrderedDict
from types import SimpleNamespace

import math

import numpy as np

from . import spaces
from . import transform_utils as T
from .mjcf.model import load_compiled
from .dense import DENSE_COEF_DEFAULTS, pack_dense
from .sim import (FSim, INFO_OVERFLOW, INFO_CONNECTED_THIS_STEP, INFO_DENSE_PHASE, INFO_DIM, INFO_EPISODE_LENGTH, INFO_FAIL, INFO_LAST_SITE1, INFO_LAST_SITE2,
                  INFO_NEEDS_TABLE, INFO_NUM_CONNECTED, INFO_SUBTASK1, INFO_SUCCESS, INFO_SUCCESS_REWARD_F, INFO_TOUCH_REWARD_F,
                  INFO_PICK_REWARD_F, INFO_CTRL_PENALTY_F, N_NOISE, default_config)

# furniture/config/furniture.py defaults that matter on the hot path (file:line in the reference)
DEFAULTS = dict(
    unity=True,                 # :21-23  (must be disabled: rendering is out of scope)
    control_type="ik",          # :54-57
    control_freq=10,            # :71-73
    rescale_actions=True,       # :74-79
    discrete_grip=True,         # :89-94
    auto_align=True,            # :95-100
    record_vid=True,            # :146-148 (must be disabled)
    record_demo=False,
    max_episode_steps=2000,     # :163-168
    furn_xyz_rand=0.02, furn_rot_rand=3, agent_xyz_rand=0.001, furn_size_rand=0.0,   # :177-200
    alignment_pos_dist=0.1, alignment_rot_dist_up=0.9, alignment_rot_dist_forward=0.9, alignment_project_dist=0.3,  # :203-226
    robot_ob=True, object_ob=True, object_ob_all=True, visual_ob=False, subtask_ob=False,  # :229-252
    ctrl_penalty_coef=1e-3, unstable_penalty_coef=100, success_reward=100, touch_reward=10, pick_reward=100,  # :291-295
    reset_robot_after_attach=False, no_collision=False,
    furniture_name=None, furniture_id=0, seed=123, move_speed=0.1,
)
# furniture/config/furniture_sawyer_dense.py:5-14
DENSE_OVERRIDES = dict(max_episode_steps=150, control_type="impedance", furniture_name="table_lack_0825", unity=False,
                       auto_align=False, alignment_pos_dist=0.02, alignment_rot_dist_up=0.99, alignment_rot_dist_forward=0.99,
                       alignment_project_dist=0.0)

# furniture.py:41-47 (NEW_CONTROLLERS) -> fsim_config_t.control_type: torque-level arm controllers run per physics substep
CONTROLLER_CODES = {"position_orientation": 2, "position": 3, "joint_impedance": 4, "joint_velocity": 5, "joint_torque": 6,
                    "ik": 7, "ik_quaternion": 8}  # "ik" (the reference's default, furniture.py:2899-2991): batched DLS solver instead of pybullet

GYM_IDS = {  # furniture/env/__init__.py:19-114
    "IKEACursor-v0": ("FurnitureCursorEnv", dict(furniture_id=0)),
    "IKEASawyer-v0": ("FurnitureSawyerEnv", dict(furniture_name="swivel_chair_0700")),
    "IKEABaxter-v0": ("FurnitureBaxterEnv", dict(furniture_id=1)),
    "IKEASawyerDense-v0": ("FurnitureSawyerDenseRewardEnv", DENSE_OVERRIDES),
    "furniture-sawyer-densereward-v0": ("FurnitureSawyerDenseRewardEnv", DENSE_OVERRIDES),
}


def make_config(**kw):
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    return SimpleNamespace(**cfg)


def furniture_names():
    import os
    from .mjcf.model import _COMPILED_DIR
    with open(os.path.join(_COMPILED_DIR, "furniture_names.txt")) as f:
        return [x.strip() for x in f if x.strip()]


_HOSTLIB = None


def _host_lib():
    """libfsim_host.so (csrc/fsim_host.c, plain C): the reset-time RNG stream for many envs per call.  None if it is not built --
    the sampler then runs its per-env Python loop (same stream, 100x slower)."""
    global _HOSTLIB
    if _HOSTLIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libfsim_host.so")
        try:
            lib = ctypes.CDLL(path)
            lib.fsim_host_reset_draw.restype = ctypes.c_int
            lib.fsim_host_seed.restype = None
            lib.fsim_host_seed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            lib.fsim_host_reset_draw.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_int]
            _HOSTLIB = lib
        except OSError:
            _HOSTLIB = False
    return _HOSTLIB or None


def host_threads():
    """OpenMP threads one call of libfsim_host.so may use: this rank's share of the host cores.  The ranks of one node reach their
    batch-wide reset on the same step; uncapped, each of 8 ranks would start one thread per core for the same 14 ms."""
    import os
    if os.environ.get("FSIM_HOST_THREADS"):
        return max(1, int(os.environ["FSIM_HOST_THREADS"]))
    ranks = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1)
    return max(1, (os.cpu_count() or 1) // max(1, ranks))


class ResetTableSampler:
    """Per-env replay of the reference's reset-time RNG stream (``self._rng = RandomState(seed)``, furniture.py:72):
    UniformRandomSampler.sample() draws (tasks/placement_sampler.py:138-190) followed by the 101 joint-noise draws of
    _initialize_robot_pos() (furniture.py:1761-1779 called at :1580 and 100x at :1606-1611).  Env i of the global batch
    is seeded seed + i (furniture/env/base.py:77), independent of how the batch is split over GPUs.

    Two implementations of the same stream: ``draw()`` hands the whole batch to libfsim_host.so (MT19937 states kept as a
    [n, 625] array, ~1 ms for 4096 envs); the per-env Python loop over ``RandomState`` objects (``_draw_python``) is what the golden
    test pins against the reference's own sampler and what runs when the C helper is not built.  Touching ``.rngs`` / ``.hist``
    (the furniture switch of reset(furniture_id) hands the generators on) moves the sampler to the Python objects for good."""

    def __init__(self, model, cfg, seed, first_env_index, n_envs, env_indices=None):
        """env_indices: explicit global env index per row (mixed-furniture batches own non-contiguous lanes)."""
        self.m, self.cfg = model, cfg
        idx = [first_env_index + i for i in range(n_envs)] if env_indices is None else [int(i) for i in env_indices]
        assert len(idx) == n_envs
        self._native = _host_lib() is not None and all(0 <= seed + i < 2 ** 32 for i in idx)
        # (native: the generators exist as state rows; the Python objects are only made when somebody asks for them)
        self._rngs = [None] * n_envs if self._native else [np.random.RandomState(seed + i) for i in idx]
        self._seeds = [seed + i for i in idx]
        self._hist = [[] for _ in idx]  # RNG states before each of the last draws (rng_handover rolls unconsumed draws back)
        self._fixed = [None] * n_envs  # config.fix_init (furniture.py:1518-1525): the first placement of an env is kept for its later resets
        self.narm = len(model.arm_qposadr)
        # config.reset_robot_after_attach with config.preassembled on a furniture with a recipe: every _connect of the reset re-poses the
        # arm with one draw taken BETWEEN the placement and the 101 robot-initialisation draws (furniture.py:919-925 inside :1542-1557).
        # They travel behind the 101 rows of the noise table (rows 101 ..: read by the kernel's in-reset connects, fsim_env.hpp)
        self.n_attach_in_reset = (len(getattr(cfg, "preassembled", None) or []) if getattr(cfg, "reset_robot_after_attach", False)
                                  and model.meta.get("has_recipe", False) and self.narm else 0)
        if self._native:
            self._mt = np.empty((n_envs, 625), dtype=np.uint32)
            sd = np.ascontiguousarray(self._seeds, dtype=np.uint32)
            _host_lib().fsim_host_seed(self._mt.ctypes.data, n_envs, sd.ctypes.data)
            self._mt_hist = np.zeros((3, n_envs, 625), dtype=np.uint32)  # states before the last three draws of every env
            self._mt_nhist = np.zeros(n_envs, dtype=np.int64)
            self._placed = np.zeros(n_envs, dtype=bool)
            self._xy = np.zeros((n_envs, model.nparts, 2))

    # -- the generators as Python objects (leaves the native path) ---------------------------------------------------------
    def _to_python(self):
        if not self._native:
            return
        self._native = False
        for i in range(len(self._rngs)):
            r = self._rngs[i] = np.random.RandomState(0)
            r.set_state(("MT19937", self._mt[i, :624].copy(), int(self._mt[i, 624]), 0, 0.0))
            c = int(self._mt_nhist[i])  # draws so far; the ring holds the states before the last min(c, 3) of them, oldest first below
            self._hist[i] = [("MT19937", self._mt_hist[d % 3, i, :624].copy(), int(self._mt_hist[d % 3, i, 624]), 0, 0.0) for d in range(max(0, c - 3), c)]
            if self._placed[i]:
                self._fixed[i] = self._full_placement(self._xy[i]).reshape(-1)

    @property
    def rngs(self):
        self._to_python()
        return self._rngs

    @rngs.setter
    def rngs(self, v):
        self._to_python()
        self._rngs = list(v)

    @property
    def hist(self):
        self._to_python()
        return self._hist

    @hist.setter
    def hist(self, v):
        self._to_python()
        self._hist = v

    def _quats(self):
        if getattr(self, "_quat_cache", None) is None:
            # reference quirk: uniform(high=rot_hi, low=rot_hi) is the constant rot_hi (one draw is still consumed), so the
            # per-part orientation does not depend on the stream
            rot_hi = max(-self.cfg.furn_rot_rand, self.cfg.furn_rot_rand)
            self._quat_cache = [list(T.euler_to_quat([rot_hi, 0, 0], self.m.part_initqpos[i][3:7])) for i in range(self.m.nparts)]
        return self._quat_cache

    def _full_placement(self, xy):
        """[nparts, 7] part poses from the sampled x / y (z = the XML height + 1 cm, the constant orientation)"""
        m = self.m
        out = np.zeros((m.nparts, 7))
        out[:, :2] = xy
        out[:, 2] = np.asarray(m.part_initqpos)[:, 2] + 0.01
        out[:, 3:7] = np.asarray(self._quats())
        return out

    def _placement(self, rng):
        m, r = self.m, self.cfg.furn_xyz_rand
        lo, hi = min(-r, r), max(-r, r)
        rot_hi = max(-self.cfg.furn_rot_rand, self.cfg.furn_rot_rand)
        quats = self._quats()
        out = np.zeros((m.nparts, 7))
        placed = []
        uni = rng.uniform
        for i in range(m.nparts):
            base, rad = m.part_initqpos[i], float(m.part_hradius[i])
            bx, by = float(base[0]), float(base[1])
            for _ in range(10000):
                x = bx + uni(high=hi, low=lo)
                y = by + uni(high=hi, low=lo)
                if all(math.hypot(x - px, y - py) > pr + rad for px, py, pr in placed):
                    uni(high=rot_hi, low=rot_hi)
                    out[i, 0], out[i, 1], out[i, 2] = x, y, base[2] + 0.01
                    out[i, 3:7] = quats[i]
                    placed.append((x, y, rad))
                    break
            else:
                raise RuntimeError("Cannot place all objects on the desk")
        return out

    def draw(self, mask=None):
        """(part_qpos [n, nparts*7], robot_noise [n, 101*narm]) for the envs selected by mask (others zero)."""
        return self._draw_native(mask) if self._native and not self.n_attach_in_reset else self._draw_python(mask)

    def _draw_native(self, mask):
        m, cfg = self.m, self.cfg
        n = len(self._rngs)
        sel = np.ones(n, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        # the states before this draw (rng_handover rolls draws back that no reset consumed)
        idx = np.nonzero(sel)[0]
        self._mt_hist[self._mt_nhist[idx] % 3, idx] = self._mt[idx]  # ring of three per env: slot (draw count) mod 3
        self._mt_nhist[idx] += 1
        place = sel & ~(self._placed if getattr(cfg, "fix_init", False) else np.zeros(n, dtype=bool))
        r = cfg.furn_xyz_rand
        lo, hi = min(-r, r), max(-r, r)
        rot_hi = max(-cfg.furn_rot_rand, cfg.furn_rot_rand)
        base = np.ascontiguousarray(np.asarray(m.part_initqpos, dtype=np.float64)[:, :2])
        rad = np.ascontiguousarray(np.asarray(m.part_hradius, dtype=np.float64).reshape(-1))
        n_noise = N_NOISE * self.narm
        noise = np.zeros((n, N_NOISE * max(self.narm, 1)), dtype=np.float32)
        dm, pm = np.ascontiguousarray(sel.astype(np.uint8)), np.ascontiguousarray(place.astype(np.uint8))
        status = np.zeros(n, dtype=np.uint8)
        rc = _host_lib().fsim_host_reset_draw(self._mt.ctypes.data, n, dm.ctypes.data, pm.ctypes.data, m.nparts, base.ctypes.data, rad.ctypes.data, lo, hi, rot_hi,
                                              n_noise, float(cfg.agent_xyz_rand), self._xy.ctypes.data, noise.ctypes.data, status.ctypes.data, host_threads())
        if rc:
            # A failed draw consumes NOTHING: the tables of this call are not handed out (ResetTableQueue / _refill see the exception),
            # so every generator that took part goes back to where it was -- the env that could not be placed and the others alike.
            # (The reference has one env per process: its RandomizationError ends that env's reset and nobody else's stream moves.)
            self._mt[idx] = self._mt_hist[(self._mt_nhist[idx] - 1) % 3, idx]
            self._mt_nhist[idx] -= 1
            self.last_failed = np.nonzero(status)[0]
            raise RuntimeError("Cannot place all objects on the desk (env rows %s)" % self.last_failed[:8].tolist())
        self._placed |= place
        parts = np.zeros((n, m.nparts * 7), dtype=np.float32)
        if getattr(cfg, "assembled", False):  # furniture.py:1526-1530: the parts stay at the XML's assembled poses; the draw is still taken
            parts[sel] = np.asarray(m.part_initqpos, dtype=np.float64).reshape(-1)
        else:
            full = np.zeros((n, m.nparts, 7))
            full[:, :, :2] = self._xy
            full[:, :, 2] = np.asarray(m.part_initqpos)[:, 2] + 0.01
            full[:, :, 3:7] = np.asarray(self._quats())
            parts[sel] = full.reshape(n, -1)[sel]
        return parts, noise

    def _draw_python(self, mask=None):
        self._to_python()
        n = len(self._rngs)
        parts = np.zeros((n, self.m.nparts * 7), dtype=np.float32)
        k = self.n_attach_in_reset
        noise = np.zeros((n, (N_NOISE + k) * max(self.narm, 1)), dtype=np.float32)
        for i, rng in enumerate(self._rngs):
            if mask is not None and not mask[i]:
                continue
            self._hist[i] = self._hist[i][-2:] + [rng.get_state()]
            if getattr(self.cfg, "fix_init", False) and self._fixed[i] is not None:
                placement = self._fixed[i]  # (no placement draw: _place_objects is not called again)
            else:
                placement = self._placement(rng).reshape(-1)
                self._fixed[i] = placement
            # config.assembled (furniture.py:1526-1530): the parts stay where sim.reset() put them (the XML's assembled poses); the draw is still taken
            parts[i] = np.asarray(self.m.part_initqpos, dtype=np.float64).reshape(-1) if getattr(self.cfg, "assembled", False) else placement
            if self.narm:
                a = self.cfg.agent_xyz_rand
                extra = rng.uniform(low=-a, high=a, size=(k, self.narm)).reshape(-1) if k else np.zeros(0)
                # one (101, narm) draw consumes the Mersenne-Twister stream exactly like 101 successive size-narm draws
                noise[i] = np.concatenate([rng.uniform(low=-a, high=a, size=(N_NOISE, self.narm)).reshape(-1), extra])
        return parts, noise


def robot_without_collision(m):
    """config.no_collision (furniture.py:1961-1965): the robot's geoms collide with nothing -- their contype / conaffinity are 0 in
    the model, so the reset's "robot collision off / on" (furniture.py:1441-1461) restores zeros and the finger-touch scan never fires."""
    from .mjcf.model import CompiledModel
    arr = dict(m.arrays)
    for full, flag in (("geom_contype", "geom_is_robot"), ("geom_conaffinity", "geom_is_robot"), ("cg_contype0", "cg_isrobot"), ("cg_conaffinity0", "cg_isrobot")):
        a = np.array(arr[full]).copy()
        a[np.asarray(arr[flag]).astype(bool)] = 0
        arr[full] = a
    return CompiledModel(arr, m.meta)


class ContactOverflowError(RuntimeError):
    """a step dropped contacts (see FurnitureBatchEnv.step_wait)"""


class ResetTableQueue:
    """Keeps one reset table per env drawn AHEAD on the host, so that handing a table to the device after an in-kernel
    auto-reset is a copy, and the reference RNG stream of the consumed envs is advanced by a worker thread while the GPU
    runs the following steps (the main thread sits in hipStreamSynchronize with the GIL released).  Per-env draw order
    is unchanged, so results are identical to calling ResetTableSampler.draw() synchronously."""

    def __init__(self, sampler):
        from concurrent.futures import ThreadPoolExecutor
        self._s = sampler
        self._pool = ThreadPoolExecutor(1)
        self._parts, self._noise = sampler.draw()
        self._fut = None

    def _refill(self, mask):
        p, nz = self._s.draw(mask)
        self._parts[mask] = p[mask]
        self._noise[mask] = nz[mask]

    def take(self, mask=None):
        """Tables for the envs in mask (all if None); schedules their replacements."""
        if self._fut is not None:
            fut, self._fut = self._fut, None
            fut.result()  # (raises what the worker's draw raised, once)
        n = self._parts.shape[0]
        m = np.ones(n, dtype=bool) if mask is None else np.asarray(mask, dtype=bool).copy()
        parts, noise = self._parts.copy(), self._noise.copy()
        self._fut = self._pool.submit(self._refill, m)
        return parts, noise

    def close(self):
        if self._fut is not None:
            # (a draw AHEAD that could not place the parts belongs to a reset nobody asked for: the reference raises RandomizationError
            #  inside the reset that needs the placement, never when the env is closed)
            self._fut.exception()
            self._fut = None
        self._pool.shutdown()


_AGENT_OF = {"FurnitureSawyerEnv": "Sawyer", "FurnitureBaxterEnv": "Baxter", "FurnitureCursorEnv": "Cursor"}


class FurnitureBatchEnv:
    """n_envs copies of FurnitureEnv on one GPU.  Observations / rewards / dones are torch tensors on the device."""

    def __init__(self, agent, num_envs, config=None, device=0, first_env_index=0, auto_reset=True, dense=False, env_indices=None, obs_bf16=False, **kw):
        """dense=True: FurnitureSawyerDenseRewardEnv semantics (furniture_sawyer_dense.py) -- the config then carries the
        config/furniture_sawyer_dense.py overrides and, optionally, any of its reward coefficients.
        obs_bf16=True: the observation slab is stored (and returned) as bfloat16 -- state and arithmetic stay float32."""
        cfg = config if config is not None else make_config(**(DENSE_OVERRIDES if dense else {}))
        for k, v in kw.items():
            setattr(cfg, k, v)
        if dense:
            if agent != "Sawyer":
                raise ValueError("the dense-reward env exists for the Sawyer agent only")
            if not getattr(cfg, "diff_rew", True):
                raise NotImplementedError("diff_rew=False: the reference itself fails in grasp_leg (furniture_sawyer_dense.py:668)")
        if cfg.unity or cfg.record_vid:
            # the reference's defaults (config/furniture.py:21-23, 146-148) would launch the Unity binary / a video writer;
            # neither changes what reset()/step() return, so the accelerated env accepts the flags and switches them off
            import warnings
            warnings.warn("furniture_amd: unity / record_vid are rendering side channels outside the accelerated hot path -- disabled",
                          stacklevel=3)
            cfg.unity, cfg.record_vid = False, False
        if cfg.visual_ob:
            raise ValueError("visual_ob must be False: camera observations need the renderer, which is outside the accelerated hot path")
        # "torque" (furniture.py:1268): _do_simulation(action[:-1]) = the impedance flow -- _setup_action turns [7 arm, 1 grip] into the 9
        # actuator controls and rescales them to the ctrlrange -- on the motor-actuated robot (robot_torque.xml): same kernel, other model
        if agent != "Cursor" and cfg.control_type not in ("impedance", "torque") and cfg.control_type not in CONTROLLER_CODES:
            raise NotImplementedError("control_type %r: the accelerated path implements 'impedance', 'torque' and the torque-level arm "
                                      "controllers / ik %s" % (cfg.control_type, sorted(CONTROLLER_CODES)))
        if agent != "Cursor" and cfg.control_type in CONTROLLER_CODES and dense and cfg.control_type not in ("ik", "ik_quaternion"):
            raise NotImplementedError("the dense-reward env runs with control_type 'impedance' (config/furniture_sawyer_dense.py:7) or ik / ik_quaternion")
        if agent == "Baxter" and cfg.control_type in CONTROLLER_CODES and cfg.control_type not in ("ik", "ik_quaternion"):
            raise NotImplementedError("the torque-level arm controllers are built for Sawyer (the reference's Baxter path mis-indexes ctrl)")
        if cfg.furn_size_rand != 0:
            raise NotImplementedError("furn_size_rand != 0 (XML rescale) is out of scope")
        # reference options that change the reset / connect flow and are not built: fail loudly instead of ignoring them
        for flag, ref in (("load_demo", "furniture.py:121-124"), ("load_init_states", "furniture.py:126-129"), ("record_demo", "furniture.py:324-325")):
            if getattr(cfg, flag, None):
                raise NotImplementedError("config.%s (%s) is not part of the accelerated path" % (flag, ref))
        # config.reset_robot_after_attach (furniture.py:919-925): _connect re-poses the arm with ONE draw of the env's RandomState, taken
        # between the draws of two resets.  Built as a mode of its own (see _attach_*): the env's stream is kept on the host, the kernel is
        # handed the next attach draw and the next reset table ahead of time, resets of finished episodes are issued by the host (device
        # auto_reset off) once it knows whether the last step attached.
        self._attach_mode = bool(getattr(cfg, "reset_robot_after_attach", False)) and agent != "Cursor"
        # (config.preassembled / set_subtask: the reset's connects read their draws from rows 101.. of the noise table; config.assembled
        #  switches the welds on without calling _connect, config.fix_init only skips later placement draws, num_connects only moves the
        #  success count: the sampler replays all of them as they are)
        names = furniture_names()
        fname = cfg.furniture_name or names[cfg.furniture_id]
        self.agent, self.furniture_name, self.config = agent, fname, cfg
        self.model = load_compiled(agent, fname, cfg.control_type if agent != "Cursor" else "impedance")
        if getattr(cfg, "no_collision", False):  # furniture.py:1961-1965: every robot geom gets contype = conaffinity = 0 in the XML
            self.model = robot_without_collision(self.model)
        c = default_config()
        c.control_type = CONTROLLER_CODES.get(cfg.control_type, 0) if agent != "Cursor" else 0
        c.n_substeps = int((1.0 / cfg.control_freq) / float(self.model.opt[0]))
        c.max_episode_steps = int(cfg.max_episode_steps)
        c.discrete_grip, c.rescale_actions, c.auto_align = int(cfg.discrete_grip), int(cfg.rescale_actions), int(cfg.auto_align)
        c.auto_reset = 1 if (auto_reset and not self._attach_mode) else 0  # (attach mode: the host resets finished episodes, see _attach_after_step)
        c.reset_robot_after_attach = 1 if getattr(cfg, "reset_robot_after_attach", False) else 0
        for k in ("alignment_pos_dist", "alignment_rot_dist_up", "alignment_rot_dist_forward", "alignment_project_dist",
                  "ctrl_penalty_coef", "unstable_penalty_coef", "success_reward", "touch_reward", "pick_reward",
                  "furn_xyz_rand", "furn_rot_rand", "agent_xyz_rand"):
            setattr(c, k, float(getattr(cfg, k)))
        for k in ("move_speed", "rotate_speed", "cursor_boundary"):  # Cursor agent (config/furniture.py:84-90)
            if getattr(cfg, k, None) is not None:
                setattr(c, k, float(getattr(cfg, k)))
        if getattr(cfg, "solver_tolerance", None) is not None:
            c.solver_tolerance = float(cfg.solver_tolerance)
        c.dense_reward = 1 if dense else 0
        c.obs_bf16 = 1 if obs_bf16 else 0
        # accelerated-path options (not in the reference): which step kernel the handle runs ("auto" | "off" | "rule" | "all",
        # include/fsim.h fsim_config_t::multi_wave) and whether resets are computed ahead of time (lookahead_reset)
        from .sim import MULTI_WAVE
        c.multi_wave = MULTI_WAVE[getattr(cfg, "multi_wave", None) or "auto"]
        c.lookahead_reset = 1 if getattr(cfg, "lookahead_reset", True) else 0
        c.overflow_restep = 1 if getattr(cfg, "overflow_restep", True) else 0
        self.dense = bool(dense)
        self.sim = FSim(self.model, num_envs, device=device, config=c)
        if dense:
            coef = {k: float(getattr(cfg, k)) for k, _ in DENSE_COEF_DEFAULTS
                    if k not in ("z_finedist", "griptip_site", "grip_site") and getattr(cfg, k, None) is not None}
            coef["phase_ob"] = 1.0 if getattr(cfg, "phase_ob", False) else 0.0  # also switches the early-pick shortcuts off (furniture_sawyer_dense.py:306)
            self.sim.set_dense_reward(*pack_dense(self.model, coef))
        self._num_connects = getattr(cfg, "num_connects", None)
        if getattr(cfg, "assembled", False):  # furniture.py:1502-1503, 1526-1530: every weld on from the start, one group
            if getattr(cfg, "preassembled", None):
                raise ValueError("config.assembled and config.preassembled exclude each other (furniture.py:1493-1503)")
            self.sim.set_preassembled(list(range(self.model.neq)), self._num_connects, welds=True)
        elif getattr(cfg, "preassembled", None) or self._num_connects is not None:  # config.preassembled / num_connects (furniture.py:163, 1476-1503)
            self.sim.set_preassembled(list(getattr(cfg, "preassembled", None) or []), self._num_connects)
        self.num_envs = num_envs
        torch = self.sim.torch
        dev = self.sim.device
        self._obs = torch.zeros((num_envs, self.sim.obs_dim), dtype=torch.bfloat16 if obs_bf16 else torch.float32, device=dev)
        self._rew = torch.zeros(num_envs, dtype=torch.float32, device=dev)
        self._done = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self._info = torch.zeros((num_envs, INFO_DIM), dtype=torch.int32, device=dev)
        self._act = torch.zeros((num_envs, self.sim.dof_action), dtype=torch.float32, device=dev)
        self._sampler = ResetTableSampler(self.model, cfg, cfg.seed, first_env_index, num_envs, env_indices=env_indices)
        self._tables_fresh = np.zeros(num_envs, dtype=bool)  # env i's table on the device has not been consumed yet
        self._auto_reset = bool(auto_reset)
        self.n_obj = self.model.nparts
        self.refill_tables_every_step = True

    # -- spaces (furniture.py:215-310, furniture_sawyer.py:28-64) ---------------------------------------
    @property
    def dof(self):
        return self.sim.dof_action

    @property
    def observation_space(self):
        """furniture.py:215-252 (+ the agent's robot_ob): object_ob holds every part, or only the two parts of the current subtask
        when object_ob_all is False; subtask_ob adds their (1-based) part ids."""
        cfg, sp = self.config, []
        if getattr(cfg, "object_ob", True):
            sp.append(("object_ob", spaces.Box(-np.inf, np.inf, shape=(7 * self.n_obj if getattr(cfg, "object_ob_all", True) else 14,))))
        if getattr(cfg, "subtask_ob", False):
            sp.append(("subtask_ob", spaces.Box(0.0, float(self.n_obj), shape=(2,))))
        if getattr(cfg, "robot_ob", True):
            sp.append(("robot_ob", spaces.Box(-np.inf, np.inf, shape=(self.sim.obs_dim - 7 * self.n_obj,))))
        if self.dense and getattr(cfg, "phase_ob", False):  # furniture_sawyer_dense.py:98-108
            sp.append(("phase_ob", spaces.Box(0.0, 1.0, shape=(8,))))
        return spaces.Dict(sp)

    @property
    def action_space(self):
        return spaces.Dict([("default", spaces.Box(-1, 1, shape=(self.dof,), dtype=np.float32))])

    def _split(self, flat, subtask=None):
        """flat device slab -> the reference's observation dict (furniture.py:1344-1387).  subtask: [n, 2] int tensor of
        (_subtask_part1, _subtask_part2) for the observed state, None right after a reset (then it is the first weld of the model
        whose parts are not connected yet, i.e. weld 0: furniture.py:2723-2736)."""
        cfg, torch = self.config, self.sim.torch
        k, n = 7 * self.n_obj, flat.shape[0]
        all_parts, want_sub = getattr(cfg, "object_ob_all", True), getattr(cfg, "subtask_ob", False)
        if (not all_parts or want_sub) and subtask is None:
            first = (int(self.model.eq_part1[0]), int(self.model.eq_part2[0])) if self.model.neq else (-1, -1)
            subtask = torch.tensor([first], dtype=torch.int64, device=flat.device).expand(n, 2)
        out = OrderedDict()
        if getattr(cfg, "object_ob", True):
            if all_parts:
                out["object_ob"] = flat[:, :k]
            else:  # parts are visited in index order and kept if they are one of the two; no subtask left -> a 14-zero dummy
                idx, _ = torch.sort(subtask.long(), dim=1)
                parts = flat[:, :k].reshape(n, self.n_obj, 7)
                sel = parts.gather(1, idx.clamp(min=0)[:, :, None].expand(-1, -1, 7)).reshape(n, 14)
                out["object_ob"] = torch.where((subtask[:, :1] >= 0).expand(-1, 14), sel, torch.zeros_like(sel))
        if want_sub:
            out["subtask_ob"] = (subtask + 1).to(flat.dtype)
        if getattr(cfg, "robot_ob", True):
            out["robot_ob"] = flat[:, k:]
        if getattr(self, "dense", False) and getattr(cfg, "phase_ob", False):
            # one-hot of _phase_i of the state the observation describes (furniture_sawyer_dense.py:111-126): read from the
            # env's reward variables on the device -- the info word is the phase the step STARTED in
            ph = self.sim.get_state("dense")["dense"][:, 1].long().clamp(0, 7)
            out["phase_ob"] = torch.nn.functional.one_hot(ph, 8).to(flat.dtype)
        return out

    def _refill(self, mask=None, skip=None, lookahead=False):
        """Upload the next reset table of the envs in mask (all if None).  skip: envs whose stream first loses one draw (the
        reference draws twice when an unstable simulation resets inside step() and the vec-env worker resets again).
        lookahead: the tables are for a reset that has not been asked for yet -- if the placement sampler gives up on one of them
        ("Cannot place all objects": the reference's RandomizationError, common for furniture with many large parts), the error is
        kept for the reset() that would have drawn it, as in the reference."""
        if getattr(self, "_place_error", None) is not None and not lookahead:
            e, self._place_error = self._place_error, None
            raise e
        try:
            if getattr(self, "_table_queue", None) is None:
                self._table_queue = ResetTableQueue(self._sampler)
            if skip is not None and skip.any():
                self._table_queue.take(skip)  # drawn and dropped
            parts, noise = self._table_queue.take(mask)
        except RuntimeError as e:
            if lookahead and "Cannot place" in str(e):
                self._place_error = e
                return
            raise
        self.sim.set_reset_tables(parts, noise, mask=mask)
        if mask is None:
            self._tables_fresh[:] = True
        else:
            self._tables_fresh[np.asarray(mask, dtype=bool)] = True

    def set_subtask(self, subtask, num_connects=None):
        """furniture.py:204-207: the following resets start with recipe steps (weld ids, for a furniture without a recipe)
        0 .. subtask-1 already assembled; num_connects as in config.num_connects (success after that many further connects)."""
        self.config.preassembled = list(range(int(subtask)))
        self._num_connects = num_connects
        self.sim.set_preassembled(self.config.preassembled, num_connects)
        if self._attach_mode:
            # config.reset_robot_after_attach: the recipe connects of the following resets take one draw each (ResetTableSampler); the
            # speculative reset table on the device (read by a reset inside step()) was drawn for the old count
            self._sampler.n_attach_in_reset = len(self.config.preassembled) if self.model.meta.get("has_recipe", False) and self._sampler.narm else 0
            if getattr(self, "_attach_after_reset", None) is not None and self._attach_after_reset[0] is not None:
                self._attach_peek(np.ones(self.num_envs, dtype=bool))

    def num_subtask(self):
        """furniture.py:209-213"""
        return self._num_connects if self._num_connects is not None else self.n_obj - 1

    def set_init_qpos(self, init_qpos):
        """furniture.py:315-316: every following reset starts from this {qpos, qvel} state (get_env_state's format; [dim] or
        [n_envs, dim]) instead of a sampled placement; None switches back.  Such resets take no draw from the RNG stream."""
        self._init_qpos = init_qpos
        if init_qpos is None:
            self.sim.set_init_state(None)
        else:
            as_np = lambda v: v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            self.sim.set_init_state(as_np(init_qpos["qpos"]), as_np(init_qpos["qvel"]))

    # -- config.reset_robot_after_attach: the env's RNG stream with attach draws in it -----------------------------------------
    # sampler.rngs[i] is env i's RandomState at the position every CONSUMED draw has left it (committed).  Ahead of every step the
    # device holds two speculative draws taken from copies of it: the next reset table (read by a reset INSIDE step(): an unstable
    # simulation) and the next attach noise (read by _connect).  After the step the info block says which of the two the kernel
    # used (they exclude each other: the finger scan only runs after a stable simulation, furniture.py:1262-1330), the stream advances
    # past that one, finished episodes are reset from the stream as it then stands, and the speculative draws are renewed.
    def _attach_peek(self, mask):
        rngs, narm, a = self._sampler.rngs, self._sampler.narm, self.config.agent_xyz_rand
        idx = np.nonzero(mask)[0]
        keep = [rngs[i].get_state() for i in idx]
        parts, noise = self._sampler._draw_python(mask)
        no_draw = getattr(self, "_init_qpos", None) is not None  # (set_init_qpos: a reset takes nothing from the stream, furniture.py:1505-1519)
        for k, i in enumerate(idx):
            self._attach_after_reset[i] = keep[k] if no_draw else rngs[i].get_state()
            rngs[i].set_state(keep[k])
        att = np.zeros((self.num_envs, max(narm, 1)), dtype=np.float32)
        for k, i in enumerate(idx):
            att[i, :narm] = rngs[i].uniform(low=-a, high=a, size=narm)  # _init_random(init_qpos.shape, "agent"), furniture.py:336-349
            self._attach_after_attach[i] = rngs[i].get_state()
            rngs[i].set_state(keep[k])
        self.sim.set_reset_tables(parts, noise, mask=mask)
        self.sim.set_attach_noise(att, mask=mask)

    def _attach_reset(self, mask):
        """reset the envs in mask (numpy bool) from their committed streams; the observation rows of those envs are rewritten"""
        keep = [(i, self._sampler.rngs[i].get_state()) for i in np.nonzero(mask)[0]] if getattr(self, "_init_qpos", None) is not None else []
        parts, noise = self._sampler._draw_python(mask)  # advances the committed generators ...
        for i, st in keep:  # ... unless the reset starts from a given state: the kernel wants a table present but takes nothing from it
            self._sampler.rngs[i].set_state(st)
        self.sim.set_reset_tables(parts, noise, mask=mask)
        torch = self.sim.torch
        self.sim.reset(torch.as_tensor(mask.astype(np.uint8), device=self.sim.device), self._obs)
        self.sim.sync()
        self._attach_peek(mask)

    def _attach_after_step(self):
        info = self._info.cpu().numpy()
        attached, failed = info[:, INFO_CONNECTED_THIS_STEP] != 0, info[:, INFO_FAIL] != 0
        rngs = self._sampler.rngs
        for i in np.nonzero(attached & ~failed)[0]:
            rngs[i].set_state(self._attach_after_attach[i])
        for i in np.nonzero(failed)[0]:  # the reset inside step() took the table (furniture.py:2889-2897)
            rngs[i].set_state(self._attach_after_reset[i])
        done = self._done.cpu().numpy() != 0
        if self._auto_reset and done.any():
            self._attach_reset(done)  # SubprocVecEnv worker semantics: the observation of a finished env is that of its reset
        renew = (attached | failed) & ~(done if self._auto_reset else np.zeros_like(done))
        if renew.any():
            self._attach_peek(renew)

    def reset(self):
        if self._attach_mode:
            self._sampler.rngs  # (the Python generators: the attach draws are taken from copies of them)
            if not hasattr(self, "_attach_after_reset"):
                self._attach_after_reset, self._attach_after_attach = [None] * self.num_envs, [None] * self.num_envs
            self._attach_reset(np.ones(self.num_envs, dtype=bool))
            return self._split(self._obs)
        if getattr(self, "_init_qpos", None) is not None:
            if not self._tables_fresh.all():  # (the kernel insists on tables being present; these are not consumed)
                self._refill(None if not self._tables_fresh.any() else ~self._tables_fresh)
            self.sim.reset(None, self._obs)
            self.sim.sync()
            self._check_overflow_after_reset()
            return self._split(self._obs)
        # one table = one pass of the reference's reset-time RNG stream: a table that is on the device but was never
        # consumed (uploaded for an auto-reset that did not happen yet) IS the next draw of that env and is used as is
        stale = ~self._tables_fresh
        if stale.any():
            self._refill(None if stale.all() else stale)
        self.sim.reset(None, self._obs)
        self.sim.sync()  # the reset kernel reads the tables: the next ones may only be uploaded once it has finished
        self._tables_fresh[:] = False
        self._check_overflow_after_reset()
        # the next draw of every env's stream goes to the device now: the in-kernel resets read it (the auto-reset of a terminal
        # step; without auto_reset, the reset an unstable simulation triggers inside step(), furniture.py:2889-2897)
        self._refill(lookahead=True)
        return self._split(self._obs)

    def rng_handover(self):
        """The per-env RandomState objects positioned at the first draw no reset has consumed yet (tables drawn ahead -- one in
        the host queue, possibly one on the device -- are rolled back): what a rebuilt env (reset(furniture_id)) continues from,
        as the reference keeps ONE self._rng across furniture switches (furniture.py:72, 318-334)."""
        q = getattr(self, "_table_queue", None)
        if q is not None:
            q.close()
            self._table_queue = None
        for i, rng in enumerate(self._sampler.rngs):
            back = (1 if q is not None else 0) + int(self._tables_fresh[i])
            hist = self._sampler.hist[i]
            if back:
                rng.set_state(hist[-back])
        return self._sampler.rngs

    def step_async(self, actions):
        torch = self.sim.torch
        a = actions
        if isinstance(a, dict):
            a = a["default"]
        if not torch.is_tensor(a):
            a = torch.as_tensor(np.asarray(a, dtype=np.float32))
        self._act.copy_(a.reshape(self.num_envs, -1), non_blocking=True)
        torch.cuda.current_stream(self.sim.device).synchronize()  # action must be resident before the handle's stream runs
        self.sim.step(self._act, self._obs, self._rew, self._done, self._info)

    def step_wait(self):
        self.sim.sync()
        if self._attach_mode:
            self._attach_after_step()
        elif self.refill_tables_every_step:
            if self.sim.tables_needed():
                need = self._info.cpu().numpy()[:, INFO_NEEDS_TABLE]  # (whole block = a DMA copy; a column slice would launch a gather kernel)
                self._tables_fresh[need > 0] = False
                self._refill(need > 0, skip=need > 1, lookahead=True)
        if not self._auto_reset and not self._attach_mode and bool((self._info[:, INFO_FAIL] != 0).any()):
            # an unstable simulation reset the env inside step() and consumed the table on the device: upload the env's next draw
            failed = (self._info[:, INFO_FAIL] != 0).cpu().numpy()
            self._tables_fresh[failed] = False
            self._refill(failed, lookahead=True)
        info = self._info
        # a launch in which contacts did not fit the slots (48 / 64 / 128 by model size) or the broadphase list integrated WRONG physics
        # for that env: an error, not a warning (FSIM_ALLOW_OVERFLOW=1 downgrades it; the flags stay in info["contact_overflow"]).  The
        # device keeps the flags STICKY in the env's record (bits 8-9 of the info word: any step, any reset -- the one inside reset(),
        # the auto-reset of a terminal step, a look-ahead reset that was copied in), so reading the block every 16th step misses nothing.
        self._steps_done = getattr(self, "_steps_done", 0) + 1
        if self._steps_done % 16 == 1:
            self._check_overflow((info[:, INFO_OVERFLOW] >> 8) != 0)
        infos = dict(num_connected=info[:, INFO_NUM_CONNECTED], episode_success=info[:, INFO_SUCCESS], fail=info[:, INFO_FAIL],
                     site1=info[:, INFO_LAST_SITE1], site2=info[:, INFO_LAST_SITE2], episode_length=info[:, INFO_EPISODE_LENGTH],
                     connected=info[:, INFO_CONNECTED_THIS_STEP], contact_overflow=info[:, INFO_OVERFLOW] & 0xff)
        if self.dense:
            infos["phase_i"] = info[:, INFO_DENSE_PHASE]  # phase + 8 * subtask (furniture_sawyer_dense.py:347)
        else:  # the reward terms _compute_reward reports (furniture.py:535-540); float bits in the int32 info block
            fl = info[:, INFO_SUCCESS_REWARD_F:INFO_CTRL_PENALTY_F + 1].view(self.sim.torch.float32)
            infos.update(success_reward=fl[:, 0], touch_reward=fl[:, 1], pick_reward=fl[:, 2], ctrl_penalty=fl[:, 3])
        return self._split(self._obs, info[:, INFO_SUBTASK1:INFO_SUBTASK1 + 2]), self._rew, self._done.bool(), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _check_overflow(self, flags):
        """flags: bool tensor [n], env dropped contacts at some point (sticky on the device)"""
        if not bool(flags.any()):
            return
        import os
        msg = ("furniture_amd: %s overflowed the contact slots / broadphase list of the step kernel in %d env(s) (info['contact_overflow']): "
               "contacts were dropped -- this furniture has more simultaneous contacts than the accelerated path holds (%d slots)"
               % (self.furniture_name, int(flags.sum()), self.sim.max_contacts))
        if os.environ.get("FSIM_ALLOW_OVERFLOW") != "1":
            raise ContactOverflowError(msg)
        if not getattr(self, "_overflow_warned", False):
            import warnings
            warnings.warn(msg, stacklevel=3)
            self._overflow_warned = True

    def _check_overflow_after_reset(self):
        # reset launches write no info block: the sticky word of the env record (fsim_model.hpp E_OVERFLOW = word 6 of env_block)
        self._check_overflow(self.sim.get_state("env_block")["env_block"][:, 6] != 0)

    def set_max_episode_steps(self, max_episode_steps):
        """furniture.py:312-313 (forwarded by FurnitureGym, furniture_gym.py:46-48)."""
        self.config.max_episode_steps = int(max_episode_steps)
        self.sim.set_max_episode_steps(max_episode_steps)

    def get_env_state(self):
        """Full snapshot (the reference's {qpos, qvel} plus the weld/mask/group state it omits)."""
        # every field of the per-env record: restoring it reproduces the trajectory bit for bit (qfrc_bias of the last forward
        # pass is what the next step's gravity compensation reads, furniture.py:3346-3353)
        names = ["qpos", "qvel", "qacc_warmstart", "qfrc_bias", "ctrl", "qfrc_applied", "xfrc_applied", "eq_active", "eq_data",
                 "geom_contype", "geom_conaffinity", "group", "env_block"] + (["dense"] if self.dense else [])
        return self.sim.get_state(*names)

    def set_env_state(self, state):
        """Restore a snapshot taken by get_env_state (dict of [n_envs, dim] arrays / tensors).  The derived quantities (poses,
        contacts) are recomputed by the first forward pass of the next step, as after the reference's set_env_state +
        sim.forward() (furniture.py:1795-1803); the observation buffer is not refreshed until then."""
        self.sim.set_state(**{k: v for k, v in state.items()})

    def close(self):
        if getattr(self, "_table_queue", None) is not None:
            self._table_queue.close()
            self._table_queue = None
        self.sim.close()


class _SingleEnv:
    """n_envs = 1 view with numpy / python scalars, shaped like the reference's env classes."""

    _agent = None
    _dense = False

    def __init__(self, config=None, device=0, **kw):
        self._b = FurnitureBatchEnv(self._agent, 1, config=config, device=device, auto_reset=False, dense=self._dense, **kw)
        self._max_episode_steps = self._b.config.max_episode_steps

    # reference surface -------------------------------------------------------------------------------
    @property
    def dof(self):
        return self._b.dof

    @property
    def observation_space(self):
        return self._b.observation_space

    @property
    def action_space(self):
        return self._b.action_space

    @property
    def action_size(self):
        return spaces.flatdim(self.action_space)

    @property
    def max_episode_steps(self):
        return self._max_episode_steps

    def num_subtask(self):
        return self._b.num_subtask()

    def set_max_episode_steps(self, max_episode_steps):
        self._max_episode_steps = int(max_episode_steps)
        self._b.set_max_episode_steps(max_episode_steps)

    def set_subtask(self, subtask, num_connects=None):
        """furniture.py:204-207"""
        self._b.set_subtask(subtask, num_connects)

    def set_init_qpos(self, init_qpos):
        """furniture.py:315-316: {qpos, qvel} as returned by get_env_state(), or None"""
        self._b.set_init_qpos(init_qpos)

    def _np(self, ob):
        return OrderedDict((k, v[0].double().cpu().numpy()) for k, v in ob.items())

    def reset(self, furniture_id=None, background=None):
        if furniture_id is not None and self._b.furniture_name != furniture_names()[furniture_id]:
            # furniture.py:318-334: a new furniture id rebuilds the model (here: a new handle for the other compiled model); the
            # env's RNG stream carries on where the old furniture left it
            old = self._b
            cfg = old.config
            cfg.furniture_id, cfg.furniture_name = int(furniture_id), None
            rngs = old.rng_handover()
            dev = old.sim.device.index or 0
            old.close()
            self._b = FurnitureBatchEnv(self._agent, 1, config=cfg, device=dev, auto_reset=False, dense=self._dense)
            self._b._sampler.rngs = rngs
            self._b._sampler.hist = [[] for _ in rngs]
        return self._np(self._b.reset())

    def step(self, action):
        if isinstance(action, list):
            action = {k: v for a in action for k, v in a.items()}
        if isinstance(action, dict):
            action = np.concatenate([action[k] for k in self.action_space.spaces.keys()])
        # single-env semantics: no auto-reset inside step (the caller resets, as with the reference's gym.Env)
        ob, rew, done, info = self._b.step(np.asarray(action, dtype=np.float32)[None])
        return self._np(ob), float(rew[0]), bool(done[0]), {k: (float(v[0]) if v.dtype.is_floating_point else int(v[0])) for k, v in info.items()}

    def get_env_state(self):
        s = self._b.get_env_state()
        return {k: v[0].cpu().numpy() for k, v in s.items()}

    def render(self, mode="human"):
        raise NotImplementedError("rendering (Unity / MuJoCo viewer) is outside the accelerated hot path")

    def close(self):
        self._b.close()


class FurnitureSawyerEnv(_SingleEnv):
    _agent = "Sawyer"


class FurnitureSawyerDenseRewardEnv(_SingleEnv):
    """furniture_sawyer_dense.py: Sawyer + table_lack_0825 with the 8-phase dense reward."""
    _agent = "Sawyer"
    _dense = True


class FurnitureBaxterEnv(_SingleEnv):
    _agent = "Baxter"


class FurnitureCursorEnv(_SingleEnv):
    _agent = "Cursor"


REGISTRY = {"FurnitureSawyerEnv": FurnitureSawyerEnv, "FurnitureBaxterEnv": FurnitureBaxterEnv, "FurnitureCursorEnv": FurnitureCursorEnv,
            "FurnitureSawyerDenseRewardEnv": FurnitureSawyerDenseRewardEnv}


def make_env(name, config=None, **kw):
    """furniture/env/base.py:14-24: bad env name -> Exception."""
    if name not in REGISTRY:
        raise Exception("No such environment: %s (accelerated path: %s)" % (name, ", ".join(REGISTRY)))
    return REGISTRY[name](config=config, **kw)


def make(env_id, **kw):
    """gym.make(id, **kw) equivalent for the registered ids."""
    if env_id not in GYM_IDS:
        raise Exception("unknown env id %s" % env_id)
    name, defaults = GYM_IDS[env_id]
    merged = dict(defaults)
    merged.update(kw)
    return make_env(name, make_config(**merged))


def make_vec_env(agent, num_envs, config=None, device=0, first_env_index=0, **kw):
    """furniture/env/base.py:55-80 replacement: one batched env instead of num_envs subprocesses."""
    return FurnitureBatchEnv(agent, num_envs, config=config, device=device, first_env_index=first_env_index, **kw)
