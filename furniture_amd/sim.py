"""ctypes binding to libfsim.so (the C-ABI in include/fsim.h).

torch is used only as the device-memory allocator / stream plumbing: every pointer that
crosses the boundary is ``tensor.data_ptr()``.  There is NO CPU fallback: if the HIP
library is missing or no GPU is visible, construction raises.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIBPATH = os.environ.get("FSIM_LIB", os.path.join(_CSRC, "libfsim.so"))
_LIB = None

INFO_DIM = 17
INFO_SUBTASK1, INFO_SUBTASK2 = 15, 16
INFO_DENSE_PHASE = 13
INFO_EPISODE_REWARD_F = 14
DENSE_STATEW = 27  # FSIM_DENSE_STATEW; ED_* of csrc/fsim_dense.hpp (subtask, phase, flags, fine-aligned, 4 x vec3, 11 prev values)
INFO_OVERFLOW = 12  # bits 0-1: this launch, bits 8-9: sticky (include/fsim.h)
E_OVERFLOW, E_NITER, E_MW_STEPS, E_EPISODE_COUNT = 6, 35, 36, 37  # words of env_block (csrc/fsim_model.hpp E_*)
INFO_NUM_CONNECTED, INFO_SUCCESS, INFO_FAIL, INFO_LAST_SITE1, INFO_LAST_SITE2, INFO_EPISODE_LENGTH = range(6)
INFO_CONNECTED_THIS_STEP, INFO_NEEDS_TABLE, INFO_SUCCESS_REWARD_F, INFO_TOUCH_REWARD_F, INFO_PICK_REWARD_F, INFO_CTRL_PENALTY_F = range(6, 12)
N_NOISE = 101  # _initialize_robot_pos draws per reset (furniture.py:1580, 1606-1611)


class FsimConfig(ctypes.Structure):
    _fields_ = [
        ("control_type", ctypes.c_int32), ("n_substeps", ctypes.c_int32), ("max_episode_steps", ctypes.c_int32),
        ("discrete_grip", ctypes.c_int32), ("rescale_actions", ctypes.c_int32), ("auto_align", ctypes.c_int32),
        ("num_connect_steps", ctypes.c_int32), ("auto_reset", ctypes.c_int32), ("solver_iterations", ctypes.c_int32),
        ("reset_robot_after_attach", ctypes.c_int32),
        ("solver_tolerance", ctypes.c_float),
        ("alignment_pos_dist", ctypes.c_float), ("alignment_rot_dist_up", ctypes.c_float),
        ("alignment_rot_dist_forward", ctypes.c_float), ("alignment_project_dist", ctypes.c_float),
        ("ctrl_penalty_coef", ctypes.c_float), ("unstable_penalty_coef", ctypes.c_float), ("success_reward", ctypes.c_float),
        ("touch_reward", ctypes.c_float), ("pick_reward", ctypes.c_float),
        ("furn_xyz_rand", ctypes.c_float), ("furn_rot_rand", ctypes.c_float), ("agent_xyz_rand", ctypes.c_float),
        ("move_speed", ctypes.c_float), ("rotate_speed", ctypes.c_float), ("cursor_boundary", ctypes.c_float),
        ("dense_reward", ctypes.c_int32), ("obs_bf16", ctypes.c_int32),
        ("multi_wave", ctypes.c_int32), ("lookahead_reset", ctypes.c_int32), ("overflow_restep", ctypes.c_int32),
    ]


MULTI_WAVE = {"auto": 0, "off": 1, "rule": 2, "all": 3}  # fsim_config_t::multi_wave


class StatePtrs(ctypes.Structure):
    _names = ["qpos", "qvel", "qacc_warmstart", "qfrc_bias", "ctrl", "qfrc_applied", "xfrc_applied", "eq_data", "eq_active",
              "geom_contype", "geom_conaffinity", "group", "qacc", "xpos", "xquat", "ncon", "contact_geoms", "solver_iters", "cursor", "dense", "env_block"]
    _fields_ = [(n, ctypes.c_void_p) for n in _names]


def library_path():
    return _LIBPATH


def build(force=False, verbose=False):
    """Compile libfsim.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "fsim.h"))
    # the host helper is a library of its own with its own staleness: a checkout that has libfsim.so but no (or an old) libfsim_host.so
    # must not silently run the 100x slower Python sampler
    host_so, host_c = os.path.join(_CSRC, "libfsim_host.so"), os.path.join(_CSRC, "fsim_host.c")
    if force or not os.path.exists(host_so) or os.path.getmtime(host_c) > os.path.getmtime(host_so):
        try:
            build_host(verbose)
        except (OSError, subprocess.CalledProcessError) as e:  # optional: no gcc / no OpenMP -> the Python sampler (same stream, slower)
            import warnings
            warnings.warn("furniture_amd: libfsim_host.so could not be built (%s); reset tables will be drawn by the per-env Python loop" % e)
    if not force and os.path.exists(_LIBPATH) and all(os.path.getmtime(s) <= os.path.getmtime(_LIBPATH) for s in srcs):
        return _LIBPATH
    # (max-ilp: one wavefront per env has no other wave to hide latency behind, so the machine scheduler is asked to interleave
    #  independent chains rather than to minimise register pressure; +1.5 % on the benchmark, same register / scratch budget.
    #  -fno-optimize-sibling-calls: ONE tail call of an out-of-line device function is enough for the compiler to give up treating
    #  that function as "all callers known" -- it then saves its 113 callee-saved VGPRs to scratch on every call, 29 KB per wave.
    #  Without tail calls every local function drops its callee-saved area: 55 -> 33 KB of HBM traffic per env-step, DESIGN_HISTORY.md 6c)
    cmd = hipcc_command(_LIBPATH)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _LIBPATH


def hipcc_command(out, defines=()):
    """The one hipcc line that builds the library (``defines``: -D options of an opt-in variant, e.g. FSIM_MFMA_HESSIAN -- __graft_entry__.build()
    builds that one beside the default library so that the GPU suite can run it through the same C-ABI session)."""
    return ["hipcc", "--offload-arch=gfx950", "-O3", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
            "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fno-optimize-sibling-calls"] + ["-D" + d for d in defines] + ["-o", out, os.path.join(_CSRC, "fsim.hip")]


def build_variant(name, defines, force=False):
    """libfsim_<name>.so: the library with opt-in compile-time paths switched on (never loaded by the package itself)."""
    out = os.path.join(_CSRC, "libfsim_%s.so" % name)
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".hpp"))]
    if force or not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        subprocess.check_call(hipcc_command(out, defines))
    return out


def build_host(verbose=False):
    """libfsim_host.so: plain-C host helper of the env layer (the reference's reset-time RNG stream for a whole batch per call)."""
    # (-ffp-contract=off: low + (high - low) * u rounds twice, as NumPy's uniform does -- gcc's default may fuse it where the target has an FMA)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", os.path.join(_CSRC, "libfsim_host.so"), os.path.join(_CSRC, "fsim_host.c"), "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_LIBPATH):
            raise RuntimeError(
                "libfsim.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                "furniture_amd has no CPU fallback for the hot path")
        import torch  # noqa: F401  -- first: torch must bind ITS HIP runtime before libfsim.so pulls in /opt/rocm's (else
        #                             torch.cuda.is_available() turns False in this process)
        L = ctypes.CDLL(_LIBPATH)
        L.fsim_last_error.restype = ctypes.c_char_p
        L.fsim_default_config.argtypes = [ctypes.POINTER(FsimConfig)]
        L.fsim_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(FsimConfig),
                                  ctypes.POINTER(ctypes.c_void_p)]
        L.fsim_destroy.argtypes = [ctypes.c_void_p]
        L.fsim_destroy.restype = None
        L.fsim_dims.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int32)] * 7
        L.fsim_stream.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        L.fsim_sync.argtypes = [ctypes.c_void_p]
        L.fsim_physics_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.fsim_physics_forward.argtypes = [ctypes.c_void_p]
        L.fsim_get_state.argtypes = [ctypes.c_void_p, ctypes.POINTER(StatePtrs)]
        L.fsim_set_state.argtypes = [ctypes.c_void_p, ctypes.POINTER(StatePtrs)]
        L.fsim_max_contacts.argtypes = [ctypes.c_void_p]
        L.fsim_env_block_words.argtypes = [ctypes.c_void_p]
        L.fsim_kernel_variant.argtypes = [ctypes.c_void_p]
        L.fsim_tables_needed.argtypes = [ctypes.c_void_p]
        L.fsim_replay_is_aligned.argtypes = [ctypes.c_int] + [ctypes.c_float] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 8
        L.fsim_replay_try_connect.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6
        L.fsim_replay_touch_scan.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
        L.fsim_kernel_variant.restype = ctypes.c_char_p
        L.fsim_step_kernel.argtypes = [ctypes.c_void_p]
        L.fsim_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.fsim_step_kernel.restype = ctypes.c_char_p
        L.fsim_lookahead_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.fsim_overflow_resteps.argtypes = [ctypes.c_void_p]
        L.fsim_overflow_resteps.restype = ctypes.c_int64
        L.fsim_set_reset_tables.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.fsim_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.fsim_set_attach_noise.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.fsim_set_init_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.fsim_step.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
        L.fsim_set_max_episode_steps.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.fsim_kernel_time_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]
        L.fsim_set_dense_reward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.fsim_set_preassembled.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.fsim_dense_replay.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + \
            [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _LIB = L
    return _LIB


EXPORTED_SYMBOLS = [
    "fsim_last_error", "fsim_default_config", "fsim_create", "fsim_destroy", "fsim_dims", "fsim_stream", "fsim_sync",
    "fsim_physics_step", "fsim_physics_forward", "fsim_get_state", "fsim_set_state", "fsim_max_contacts",
    "fsim_set_reset_tables", "fsim_reset", "fsim_step", "fsim_kernel_time_ms", "fsim_set_dense_reward", "fsim_dense_replay",
    "fsim_env_block_words", "fsim_set_max_episode_steps", "fsim_kernel_variant",
    "fsim_step_kernel", "fsim_lookahead_stats", "fsim_overflow_resteps",
    "fsim_replay_is_aligned", "fsim_replay_try_connect", "fsim_replay_touch_scan", "fsim_set_init_state", "fsim_tables_needed", "fsim_set_preassembled", "fsim_set_attach_noise",
    "fsim_read",
]


def preassembled_rows(model, preassembled):
    """(ids, connector pairs, angles) of fsim_set_preassembled for the reference's `preassembled` list.  With a recipe, row i holds
    the connector-table indices of site_recipe[i]'s site2 and site1 -- the reset calls _connect(site2_id, site1_id)
    (furniture.py:1546-1554) -- and the recipe's angle (NaN: none)."""
    ids = np.ascontiguousarray([int(x) for x in preassembled], dtype=np.int32)
    if not model.meta.get("has_recipe"):
        return ids, None, None
    sites = list(model.meta["site_names"])
    conn = [int(x) for x in model.conn_siteid]
    pairs, angles = np.zeros((len(ids), 2), dtype=np.int32), np.full(len(ids), np.nan, dtype=np.float32)
    for r, i in enumerate(ids):
        row = model.meta["site_recipe"][int(i)]
        pairs[r] = [conn.index(sites.index(row[1])), conn.index(sites.index(row[0]))]
        if len(row) == 3:
            angles[r] = float(row[2])
    return ids, pairs, angles


def default_config():
    c = FsimConfig()
    lib().fsim_default_config(ctypes.byref(c))
    return c


class FsimError(RuntimeError):
    pass


def dense_replay(coef, subtasks, n_pre, obs0, obs, ac, connected, device=0):
    """fsim_dense_replay: the device reward state machine alone on recorded sensor values (parity hook).
    -> (reward float32 [T], flags int32 [T, 4] = done, success, phase, subtask)."""
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    coef, subtasks, obs0, obs, ac = f32(coef), f32(subtasks), f32(obs0), f32(obs), f32(ac)
    connected = np.ascontiguousarray(connected, dtype=np.uint8)
    T, nsub = len(ac), len(subtasks)
    assert obs.shape == (T, nsub, 40) and obs0.shape == (nsub, 40)
    rew, flags = np.zeros(T, np.float32), np.zeros((T, 4), np.int32)
    rc = lib().fsim_dense_replay(device, coef.ctypes.data, len(coef), subtasks.ctypes.data, nsub, int(n_pre), obs0.ctypes.data,
                                 obs.ctypes.data, ac.ctypes.data, ac.shape[1], connected.ctypes.data, T, rew.ctypes.data,
                                 flags.ctypes.data)
    if rc:
        raise FsimError("fsim_dense_replay rc=%d: %s" % (rc, lib().fsim_last_error().decode()))
    return rew, flags


def replay_is_aligned(p1, R1, p2, R2, nang, angles, pos_dist=0.1, rot_up=0.9, rot_fwd=0.9, proj_dist=0.3, device=0):
    """The device's _is_aligned on recorded site poses (include/fsim.h: fsim_replay_is_aligned) -> (ok [n] bool, target_quat [n, 4])."""
    f = lambda a, sh: np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(sh))
    n = len(nang)
    p1, R1, p2, R2, ang = f(p1, (n, 3)), f(R1, (n, 9)), f(p2, (n, 3)), f(R2, (n, 9)), f(angles, (n, 4))
    na = np.ascontiguousarray(np.asarray(nang, dtype=np.int32))
    ok, tq = np.zeros(n, dtype=np.int32), np.zeros((n, 4), dtype=np.float32)
    rc = lib().fsim_replay_is_aligned(int(device), pos_dist, rot_up, rot_fwd, proj_dist, n, p1.ctypes.data, R1.ctypes.data, p2.ctypes.data,
                                      R2.ctypes.data, na.ctypes.data, ang.ctypes.data, ok.ctypes.data, tq.ctypes.data)
    if rc != 0:
        raise FsimError("fsim_replay_is_aligned rc=%d: %s" % (rc, lib().fsim_last_error().decode()))
    return ok.astype(bool), tq


class FSim:
    """One handle = n_envs environments of one compiled model on one GPU."""

    def __init__(self, model, n_envs, device=0, config=None):
        import torch

        if not torch.cuda.is_available():
            raise FsimError("no ROCm device visible: furniture_amd's hot path has no CPU fallback")
        self.torch = torch
        self.cm = model
        self.n_envs = int(n_envs)
        self.device = torch.device("cuda", device)
        self.cfg = config if config is not None else default_config()
        blob = model.to_blob()
        h = ctypes.c_void_p()
        rc = lib().fsim_create(blob, len(blob), self.n_envs, device, ctypes.byref(self.cfg), ctypes.byref(h))
        if rc:
            raise FsimError("fsim_create rc=%d: %s" % (rc, lib().fsim_last_error().decode()))
        self._h = h
        d = [ctypes.c_int32() for _ in range(7)]
        self._chk(lib().fsim_dims(self._h, *[ctypes.byref(x) for x in d]))
        self.nq, self.nv, self.nu, self.dof_action, self.obs_dim, self.info_dim, self.stride = [x.value for x in d]
        self.max_contacts = lib().fsim_max_contacts(self._h)
        self.kernel_variant = lib().fsim_kernel_variant(self._h).decode()
        self.step_kernel = lib().fsim_step_kernel(self._h).decode()  # the launch structure fsim_config_t::multi_wave resolved to
        st = ctypes.c_void_p()
        self._chk(lib().fsim_stream(self._h, ctypes.byref(st)))
        # the handle's HIP stream as a torch stream: work enqueued under `with torch.cuda.stream(sim.torch_stream)` (an RCCL
        # collective, a copy) runs behind the step kernel without a host synchronisation
        self.torch_stream = torch.cuda.ExternalStream(st.value, device=self.device)
        self.env_block_words = lib().fsim_env_block_words(self._h)
        st = ctypes.c_void_p()
        self._chk(lib().fsim_stream(self._h, ctypes.byref(st)))
        self.stream_ptr = st.value

    def set_dense_reward(self, coef, subtasks):
        """Upload the dense-reward tables (furniture_amd.dense.pack_dense)."""
        coef = np.ascontiguousarray(coef, dtype=np.float32)
        subtasks = np.ascontiguousarray(subtasks, dtype=np.float32)
        self._chk(lib().fsim_set_dense_reward(self._h, coef.ctypes.data, len(coef), subtasks.ctypes.data, len(subtasks)))

    def set_preassembled(self, preassembled, num_connects=None, welds=False):
        """config.preassembled / set_subtask (furniture.py:163, 204-207): weld ids (furniture without a recipe) or recipe step
        indices (with one) that every following reset starts from; num_connects as in config.num_connects.  welds=True: the list
        holds weld ids whatever the furniture (config.assembled: all of them)."""
        if welds:  # weld ids as they are: no recipe lookup (a furniture may have more welds than recipe steps)
            ids, pairs, angles = np.ascontiguousarray(list(preassembled), dtype=np.int32), None, None
        else:
            ids, pairs, angles = preassembled_rows(self.cm, preassembled)
        self._chk(lib().fsim_set_preassembled(self._h, len(ids), ids.ctypes.data, pairs.ctypes.data if pairs is not None else None,
                                              angles.ctypes.data if angles is not None else None, -1 if num_connects is None else int(num_connects)))

    def _chk(self, rc):
        if rc:
            raise FsimError("fsim rc=%d: %s" % (rc, lib().fsim_last_error().decode()))

    # -- raw physics -------------------------------------------------------
    def physics_step(self, n=1):
        self._chk(lib().fsim_physics_step(self._h, int(n)))

    def physics_forward(self):
        self._chk(lib().fsim_physics_forward(self._h))

    def sync(self):
        self._chk(lib().fsim_sync(self._h))

    _shapes = None

    def _field_shapes(self):
        m = self.cm
        return dict(qpos=(m.nq, "f"), qvel=(m.nv, "f"), qacc_warmstart=(m.nv, "f"), qfrc_bias=(m.nv, "f"), ctrl=(m.nu, "f"),
                    qfrc_applied=(m.nv, "f"), xfrc_applied=(6 * m.nparts, "f"), eq_data=(7 * m.neq, "f"), eq_active=(m.neq, "i"),
                    geom_contype=(m.ngeom, "i"), geom_conaffinity=(m.ngeom, "i"), group=(m.nparts, "i"), qacc=(m.nv, "f"),
                    xpos=(3 * m.nbody, "f"), xquat=(4 * m.nbody, "f"), ncon=(1, "i"), contact_geoms=(2 * self.max_contacts, "i"),
                    solver_iters=(1, "i"), cursor=(8, "f"), dense=(DENSE_STATEW, "f"), env_block=(self.env_block_words, "i"))

    def get_state(self, *names):
        """dict name -> torch tensor [n_envs, dim] (device)."""
        torch = self.torch
        shapes = self._field_shapes()
        names = names or [n for n in shapes if (n != "cursor" or self.cm.meta.get("agent") == "Cursor")
                          and (n != "dense" or self.cfg.dense_reward)]
        out, p = {}, StatePtrs()
        for n in names:
            dim, kind = shapes[n]
            # zero-filled (geom_contype / geom_conaffinity only receive their colliding-geom entries); the fill runs on torch's
            # stream, the copies on the handle's: the synchronize below orders them
            t = torch.zeros((self.n_envs, dim), dtype=torch.float32 if kind == "f" else torch.int32, device=self.device)
            out[n] = t
            setattr(p, n, t.data_ptr() if dim else None)
        torch.cuda.current_stream(self.device).synchronize()
        self._chk(lib().fsim_get_state(self._h, ctypes.byref(p)))
        self.sync()
        return out

    def set_state(self, **fields):
        torch = self.torch
        shapes = self._field_shapes()
        p, keep = StatePtrs(), []
        for n, v in fields.items():
            dim, kind = shapes[n]
            t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
            t = t.to(device=self.device, dtype=torch.float32 if kind == "f" else torch.int32).reshape(-1, dim)
            if t.shape[0] == 1 and self.n_envs > 1:
                t = t.expand(self.n_envs, dim)
            t = t.contiguous()
            assert t.shape == (self.n_envs, dim), (n, t.shape)
            keep.append(t)
            setattr(p, n, t.data_ptr() if dim else None)
        torch.cuda.synchronize(self.device)
        self._chk(lib().fsim_set_state(self._h, ctypes.byref(p)))
        self.sync()

    # -- env hot path ------------------------------------------------------
    def set_reset_tables(self, part_qpos, robot_noise=None, mask=None):
        pq = np.ascontiguousarray(part_qpos, dtype=np.float32).reshape(self.n_envs, -1)
        rn = None
        n_noise = 0
        if robot_noise is not None:
            rn = np.ascontiguousarray(robot_noise, dtype=np.float32).reshape(self.n_envs, -1)
            n_noise = rn.shape[1] // max(1, len(self.cm.arm_qposadr))
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._chk(lib().fsim_set_reset_tables(self._h, None if mk is None else mk.ctypes.data, pq.ctypes.data,
                                              None if rn is None else rn.ctypes.data, n_noise))

    def set_attach_noise(self, noise, mask=None):
        """config.reset_robot_after_attach: the joint noise [n, narm joints] each env's NEXT attach re-poses the arm with
        (include/fsim.h: fsim_set_attach_noise)"""
        nz = np.ascontiguousarray(noise, dtype=np.float32).reshape(self.n_envs, -1)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._chk(lib().fsim_set_attach_noise(self._h, None if mk is None else mk.ctypes.data, nz.ctypes.data))

    def set_init_state(self, qpos=None, qvel=None, mask=None):
        """set_init_qpos for the masked envs (None = all); qpos None clears it (include/fsim.h: fsim_set_init_state)"""
        mk = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        if qpos is None:
            self._chk(lib().fsim_set_init_state(self._h, None if mk is None else mk.ctypes.data, None, None))
            return
        q = np.ascontiguousarray(np.broadcast_to(np.asarray(qpos, dtype=np.float32).reshape(-1, self.nq), (self.n_envs, self.nq)))
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(qvel, dtype=np.float32).reshape(-1, self.nv), (self.n_envs, self.nv)))
        self._chk(lib().fsim_set_init_state(self._h, None if mk is None else mk.ctypes.data, q.ctypes.data, v.ctypes.data))

    def reset(self, mask=None, obs=None):
        self._chk(lib().fsim_reset(self._h, None if mask is None else mask.data_ptr(), None if obs is None else obs.data_ptr()))

    def step(self, action, obs, reward, done, info):
        """Enqueue one FurnitureEnv.step() of every env on the handle's stream (asynchronous).  sync() before the next step():
        the counter tables_needed() reads lives in host memory and is cleared when a step is enqueued, so a second step enqueued
        without a sync in between loses the first one's reset notifications."""
        self._chk(lib().fsim_step(self._h, action.data_ptr(), obs.data_ptr(), reward.data_ptr(), done.data_ptr(), info.data_ptr()))

    def read_into(self, host_tensor, dev_tensor):
        """fsim_read: copy a device tensor into a (pinned) host tensor of the same size on the handle's transfer stream; complete on return"""
        nbytes = dev_tensor.numel() * dev_tensor.element_size()
        assert host_tensor.numel() * host_tensor.element_size() == nbytes and dev_tensor.is_contiguous() and host_tensor.is_contiguous()
        self._chk(lib().fsim_read(self._h, host_tensor.data_ptr(), dev_tensor.data_ptr(), nbytes))
        return host_tensor

    def tables_needed(self):
        """envs that consumed their reset table in the last step (valid after sync())"""
        return int(lib().fsim_tables_needed(self._h))

    def lookahead_stats(self):
        """fsim_lookahead_stats: dict(enabled, units (reset substeps run by look-ahead jobs), swapped, inline, jobs_per_launch, units_per_job) -- valid after sync()"""
        out = (ctypes.c_int64 * 6)()
        self._chk(lib().fsim_lookahead_stats(self._h, out))
        return dict(zip(("enabled", "units", "swapped", "inline", "jobs_per_launch", "units_per_job"), [int(x) for x in out]))

    def overflow_resteps(self):
        """fsim_overflow_resteps: env-steps repeated with a 64-slot layout because the step kernel's 48 contact slots did not hold them"""
        return int(lib().fsim_overflow_resteps(self._h))

    def set_max_episode_steps(self, n):
        self._chk(lib().fsim_set_max_episode_steps(self._h, int(n)))
        self.cfg.max_episode_steps = int(n)

    # -- env-logic replay hooks (include/fsim.h: fsim_replay_*) ------------------------------------------
    def replay_try_connect(self, part12, group, used, aligned, step_in, num_connect_steps):
        i32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int32))
        part12, group, used, step_in = i32(part12), i32(group), i32(used), i32(step_in)
        al = np.ascontiguousarray(np.asarray(aligned, dtype=np.uint8))
        n = len(step_in)
        out = np.zeros((n, 5), dtype=np.int32)
        self._chk(lib().fsim_replay_try_connect(self._h, n, int(num_connect_steps), part12.ctypes.data, group.ctypes.data, used.ctypes.data,
                                                al.ctypes.data, step_in.ctypes.data, out.ctypes.data))
        return out

    def replay_touch_scan(self, ncon, geoms, script):
        i32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int32))
        ncon, geoms = i32(ncon), i32(geoms)
        sc = np.ascontiguousarray(np.asarray(script, dtype=np.uint8))
        n, maxc = len(ncon), geoms.shape[1]
        masks, tried = np.zeros((n, 3), dtype=np.int32), np.zeros((n, 4), dtype=np.int32)
        self._chk(lib().fsim_replay_touch_scan(self._h, n, maxc, ncon.ctypes.data, geoms.ctypes.data, sc.ctypes.data, masks.ctypes.data, tried.ctypes.data))
        return masks, tried

    def kernel_time_ms(self):
        ms, n = ctypes.c_double(), ctypes.c_int32()
        self._chk(lib().fsim_kernel_time_ms(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            lib().fsim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
