"""bench.py -- env-steps/s of the FurnitureEnv.step() hot path on MI355X (BASELINE.json metric).

Workload = BASELINE config 2: FurnitureSawyerEnv + table_lack_0825, control_type=impedance, 4096 envs per GPU,
U(-1,1)^9 random actions (fps.py protocol), max_episode_steps=150 with in-kernel auto-reset, fp32 state.
One "step" = one FurnitureEnv.step() on every env = 50 physics substeps + connector logic + obs + reward.

    python bench.py                      # SURVEY 8(d) protocol: 100 warm-up + 1000 timed steps (>= 6 full-batch resets inside)
    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 ...         # launched plainly: re-executes itself as 8 ranks (torch.distributed.run, RCCL); refuses (rc 2)
                                         # if fewer than 8 GPUs are visible -- it never falls back to one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --config 4 --gpus 8  # BASELINE configs: 2 (default), 3 Sawyer + swivel_chair 8192 envs, 4 Baxter + desk_mikael,
                                         # 5 mixed-furniture batch (table_lack / chair_agne / shelf_ivar per lane)

Rank 0 prints ONE JSON line.  `value` is the whole-job aggregate; inputs are resident in HBM when the timed region
starts (actions are generated on the device).  `roofline` prices the fused step kernel against HBM peak with the
algorithmic bytes of SURVEY.md section 8(d); `cpu_baseline` times the CPU checker (oracle/, a C port of the same
pipeline) on the host cores with the same protocol -- a reported baseline, not the target.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
AGENT, FURNITURE = "Sawyer", "table_lack_0825"
MAX_EPISODE_STEPS = 150
SEED = int(os.environ.get("FSIM_BENCH_SEED", "123"))  # (development: workload-realisation noise; the benchmark seed is 123)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# SURVEY.md section 8(d), fused formulation, fp32: in = state 122 w + per-env mutable model 99 w + action 9 w,
# out = state 122 w + mutable model 99 w + obs 64 w + reward/done/info ~8 w  -> 523 words per env-step
ALGO_BYTES_PER_ENV_STEP = 523 * 4


def _cpu_worker(args):
    idx, seconds = args
    from furniture_amd.mjcf.model import load_compiled
    from oracle.oracle_env import FurnitureEnvOracle, OracleConfig
    m = load_compiled(AGENT, FURNITURE)
    env = FurnitureEnvOracle(m, OracleConfig(max_episode_steps=MAX_EPISODE_STEPS, seed=SEED + idx))
    rng = np.random.RandomState(SEED + idx)
    env.reset()
    # how much of the baseline is the C physics (osim_step) and how much the Python env logic around it: timed on worker 0 only
    # (two clock reads per substep would be 3 % of a worker's own time)
    tc = [0.0]
    if idx == 0:
        raw = env.sim.step

        def timed_step():
            t = time.perf_counter()
            r = raw()
            tc[0] += time.perf_counter() - t
            return r
        env.sim.step = timed_step
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        _, _, done, _ = env.step(rng.uniform(-1, 1, 9))
        n += 1
        if done:
            env.reset()
    dt = time.time() - t0
    return n, dt, (tc[0] / dt if idx == 0 else None)


def _cpu_native_worker(q, seconds, n):
    """(spawned with OMP_NUM_THREADS set): oracle/libfsim_cpu.so -- the SAME C-ABI as the product's (include/fsim.h) on host memory, env
    logic and physics in C, one env per OpenMP thread -- driven with the benchmark's own protocol: auto-reset at 150 steps, reset
    tables from the reference's RNG stream uploaded for the envs that consumed theirs."""
    import ctypes
    from furniture_amd.envs import ResetTableQueue, ResetTableSampler, make_config
    from furniture_amd.mjcf.model import load_compiled
    from furniture_amd.sim import FsimConfig, INFO_DIM, INFO_NEEDS_TABLE
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libfsim_cpu.so"))
    L.fsim_last_error.restype = ctypes.c_char_p
    m = load_compiled(AGENT, FURNITURE)
    ecfg = make_config(unity=False, record_vid=False, furniture_name=FURNITURE, max_episode_steps=MAX_EPISODE_STEPS, seed=SEED)
    cfg = FsimConfig()
    L.fsim_default_config(ctypes.byref(cfg))
    cfg.max_episode_steps, cfg.auto_reset = MAX_EPISODE_STEPS, 1
    h = ctypes.c_void_p()
    blob = m.to_blob()
    vp = ctypes.c_void_p

    def ck(rc):
        if rc != 0:
            raise RuntimeError(L.fsim_last_error().decode())
    ck(L.fsim_create(blob, ctypes.c_size_t(len(blob)), n, 0, ctypes.byref(cfg), ctypes.byref(h)))
    tables = ResetTableQueue(ResetTableSampler(m, ecfg, SEED, 0, n))

    def upload(mask=None):
        parts, noise = tables.take(mask)
        parts, noise = np.ascontiguousarray(parts, dtype=np.float32), np.ascontiguousarray(noise, dtype=np.float32)
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        ck(L.fsim_set_reset_tables(h, None if mk is None else vp(mk.ctypes.data), vp(parts.ctypes.data), vp(noise.ctypes.data), 101))
    upload()
    obs = np.zeros((n, 64), dtype=np.float32)
    rew, done, info = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.uint8), np.zeros((n, INFO_DIM), dtype=np.int32)
    ck(L.fsim_reset(h, None, vp(obs.ctypes.data)))
    upload()
    rng = np.random.RandomState(SEED)
    k, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        a = rng.uniform(-1, 1, (n, 9)).astype(np.float32)
        ck(L.fsim_step(h, vp(a.ctypes.data), vp(obs.ctypes.data), vp(rew.ctypes.data), vp(done.ctypes.data), vp(info.ctypes.data)))
        k += 1
        if L.fsim_tables_needed(h):
            upload(info[:, INFO_NEEDS_TABLE] > 0)
    dt = time.time() - t0
    tables.close()
    L.fsim_destroy(h)
    q.put((k * n, dt))


def cpu_baseline(seconds=10.0, kind="native"):
    cores = max(1, min(os.cpu_count() or 1, 64))
    ctx = mp.get_context("spawn")
    if kind == "native":
        # (the checker is built by __graft_entry__.build(); a tree without it -- or without a compiler -- falls back to the Python env logic, and says so)
        import subprocess
        so = os.path.join(ROOT, "oracle", "libfsim_cpu.so")
        if subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libfsim_cpu.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0 and not os.path.exists(so):
            out = cpu_baseline(seconds, "python")
            out["note"] = "oracle/libfsim_cpu.so could not be built here: the Python env logic over the C physics was timed instead"
            return out
        n = 16 * cores  # (sixteen envs per thread, dynamically scheduled: an env inside its 301-substep auto-reset or a slow contact step does not hold the batch)
        q = ctx.Queue()
        old = os.environ.get("OMP_NUM_THREADS")
        os.environ["OMP_NUM_THREADS"] = str(cores)
        try:
            p = ctx.Process(target=_cpu_native_worker, args=(q, seconds, n))
            p.start()
            import queue
            while True:  # (a worker that died -- library not built, ... -- must not leave the bench waiting)
                try:
                    steps, wall = q.get(timeout=1.0)
                    break
                except queue.Empty:
                    if not p.is_alive():
                        out = cpu_baseline(seconds, "python")
                        out["note"] = "the native worker (oracle/libfsim_cpu.so) exited with code %s: the Python env logic over the C physics was timed instead" % p.exitcode
                        return out
            p.join()
        finally:
            if old is None:
                os.environ.pop("OMP_NUM_THREADS", None)
            else:
                os.environ["OMP_NUM_THREADS"] = old
        return {"value": steps / wall, "unit": "env-steps/s", "cores": cores, "kind": "port", "impl": "native C, same C-ABI (oracle/libfsim_cpu.so)",
                "sample": "%d envs on %d OpenMP threads x %.0f s of FurnitureSawyerEnv+table_lack_0825 random-action steps incl. the auto-resets at 150 steps, "
                          "env logic and fp64 physics in C behind include/fsim.h's entry points on host memory; %.1f env-steps/s per core"
                          % (n, cores, wall, steps / wall / cores)}
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(i, seconds) for i in range(cores)])
    steps = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    share = res[0][2]
    return {"value": steps / wall, "unit": "env-steps/s", "cores": cores, "kind": "port", "impl": "Python env logic over the C physics (oracle/oracle_env.py)",
            "c_physics_share": round(share, 3) if share is not None else None,
            "sample": "%d envs (one per core) x %.0f s of FurnitureSawyerEnv+table_lack_0825 random-action steps incl. resets; "
                      "%.1f env-steps/s per core; %.0f %% of worker 0's wall time inside the C physics (osim_step), the rest is the Python env "
                      "logic of the oracle env" % (cores, seconds, steps / wall / cores, 100 * (share or 0))}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, selftest=False):
    """`python bench.py --gpus N` launched plainly (no RANK in the environment): run N ranks of this script under
    torch.distributed.run, one per GPU (the reference's fan-out is one worker process per env, furniture/env/base.py:55-80; here it
    is one process per GPU).  Returns the launcher's exit code; rank 0's JSON line goes to this process's stdout."""
    import subprocess
    if not selftest:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.stderr.write("bench.py: %d GPUs requested, %d visible -- refusing to run on fewer (no silent fallback)\n" % (n, have))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def launcher_selftest():
    """CPU stand-in for the N-rank launch path (tests/test_bench_launcher.py): the ranks meet over gloo, reduce their ranks, and
    rank 0 prints one JSON line.  No GPU, no physics -- it checks the spawning, the rendezvous and the one-line contract."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "rank_sum": float(t)}))
    dist.destroy_process_group()


def mixed_bench(args):
    """BASELINE config 5: every lane of the batch runs one of several furniture models (round robin over the global lane index),
    one FSim handle + HIP stream per model, padded observation slab, RCCL gather of the padded slab (furniture_amd/mixed.py)."""
    import torch
    import torch.distributed as dist
    from furniture_amd.envs import make_config
    from furniture_amd.mixed import FurnitureMixedBatchEnv
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d\n" % (args.gpus, world))
        sys.exit(2)
    names = args.furniture.split(",")
    n = args.envs_per_gpu // len(names) * len(names)
    os.environ.setdefault("FSIM_ALLOW_OVERFLOW", "1")  # (a benchmark run: an env that drops contacts is counted and reported, not fatal)
    env = FurnitureMixedBatchEnv(args.agent, names, n, device=local, first_env_index=rank * n,
                                 config=make_config(unity=False, record_vid=False, control_type="impedance", max_episode_steps=MAX_EPISODE_STEPS, seed=SEED))
    env.reset()
    g = torch.Generator(device=env.device)
    g.manual_seed(SEED + rank)
    acts = torch.empty((args.warmup + args.steps, n, env.dof), device=env.device).uniform_(-1, 1, generator=g)
    for t in range(args.warmup):
        env.step(acts[t]); env.gather()
    torch.cuda.synchronize(env.device)
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        env.step(acts[t]); env.gather()
    torch.cuda.synchronize(env.device)
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], device=env.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    if rank == 0:
        print(json.dumps({"metric": "env-steps/sec (whole node), EXPLORATION config 5: mixed batch %s, %d envs/GPU" % ("/".join(names), n),
                          "value": world * n * args.steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": "BASELINE config 5: Furniture%sEnv, lane i -> %s[i %% %d], padded observation slab %d wide" % (args.agent, names, len(names), env.obs_dim),
                                                          "envs_per_gpu": n, "global_envs": world * n, "obs_finite": bool(torch.isfinite(env.obs).all()),
                                                          "envs_that_dropped_contacts": int(sum(int((sub.sim.get_state("env_block")["env_block"].view(torch.int32)[:, 6] != 0).sum())
                                                                                                for sub in env.subs if sub is not None))}}))
    env.close()
    if distributed:
        dist.destroy_process_group()


CONFIGS = {  # BASELINE.json `configs` (1 is the CPU reference case: tests/test_oracle_env.py)
    2: dict(agent="Sawyer", furniture="table_lack_0825", envs=4096),
    3: dict(agent="Sawyer", furniture="swivel_chair_0700", envs=8192),
    4: dict(agent="Baxter", furniture="desk_mikael_1064", envs=4096),
    5: dict(agent="Sawyer", furniture="table_lack_0825,chair_agne_0007,shelf_ivar_0678", envs=4095),  # one model per lane, round robin
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config (2 = the headline benchmark)")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["native", "python"], default="native", help="native: oracle/libfsim_cpu.so (C env logic + physics behind the same C-ABI, OpenMP); python: oracle/oracle_env.py, one process per core")
    ap.add_argument("--agent", default=AGENT, help="(exploration only; the benchmark workload is the default)")
    ap.add_argument("--furniture", default=FURNITURE, help="(exploration only)")
    ap.add_argument("--dense", action="store_true", help="(exploration only) FurnitureSawyerDenseRewardEnv: 8-phase dense reward + its config overrides")
    ap.add_argument("--control-type", default="impedance", help="(exploration only) a torque-level arm controller, e.g. position_orientation")
    ap.add_argument("--episode-window", type=int, default=-1,
                    help="steps of a SECOND, un-headlined timed window run after the contract's (reported as config.episode_window_*): -1 = one full "
                         "episode + 10 steps when --steps is shorter than an episode, 0 = off")
    ap.add_argument("--obs-bf16", action="store_true", help="store the observation slab as bfloat16 (BASELINE config 2's narrow slab; state stays fp32)")
    ap.add_argument("--no-lookahead", action="store_true", help="(exploration only) resets run inside the terminal step's launch instead of ahead of time (fsim_config_t::lookahead_reset = 0)")
    ap.add_argument("--multi-wave", default="auto", choices=["auto", "off", "rule", "all"], help="(exploration only) fsim_config_t::multi_wave of every slab")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("FSIM_BENCH_GROUPS", "4")),
                    help="env groups (slabs) per GPU, each a handle of its own, stepped asynchronously (1 = one synchronous launch)")
    ap.add_argument("--threads", type=int, default=int(os.environ.get("FSIM_BENCH_THREADS", "1")), choices=[0, 1],
                    help="1 (default since round 6; single-process runs only): one host thread per slab -- each slab is re-stepped as soon as ITS step is "
                         "done (+1 %: 836 / 834 k against 826 / 828 k on one box, profiles/r06_b_*); 0: one thread, round robin (what a rank under torch.distributed.run does)")
    args = ap.parse_args()
    # the host-side reset-table sampler (libfsim_host.so) is OpenMP code; with one host thread per slab, G of its parallel regions start at
    # once when a batch-wide episode end reaches every slab: G x all-cores threads spinning on 64 cores took tens of seconds
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.groups * max(1, int(os.environ.get("WORLD_SIZE", "1")))))))
    if os.environ.get("FSIM_BENCH_WATCHDOG"):  # development: where every thread is if the run takes longer than this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FSIM_BENCH_WATCHDOG"]), exit=True)
    if "RANK" not in os.environ and args.gpus > 1:  # plain launch: become N ranks
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], selftest=args.launcher_selftest))
    if args.launcher_selftest:
        return launcher_selftest()
    if args.config != 2:
        preset = CONFIGS[args.config]
        args.agent, args.furniture, args.envs_per_gpu = preset["agent"], preset["furniture"], preset["envs"]
    if args.config == 5:
        return mixed_bench(args)

    import torch
    import torch.distributed as dist
    from furniture_amd.dist import gather_observations, shard_range, step_wait_and_gather
    from furniture_amd.envs import ResetTableQueue, ResetTableSampler, make_config
    from furniture_amd.mjcf.model import load_compiled
    from furniture_amd.sim import FSim, INFO_DIM, INFO_NEEDS_TABLE, MULTI_WAVE, default_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run (also with one rank)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d\n" % (args.gpus, world))
        sys.exit(2)
    n = args.envs_per_gpu
    lo, hi = shard_range(rank, world, n)

    m = load_compiled(args.agent, args.furniture, args.control_type)
    ecfg = make_config(unity=False, record_vid=False, control_type=args.control_type, furniture_name=args.furniture,
                       max_episode_steps=MAX_EPISODE_STEPS, seed=SEED)
    cfg = default_config()
    cfg.max_episode_steps = MAX_EPISODE_STEPS
    cfg.auto_reset = 1
    cfg.obs_bf16 = 1 if args.obs_bf16 else 0
    cfg.lookahead_reset = 0 if args.no_lookahead else 1
    # which step kernel: the library's "auto" looks at one handle's env count; the slabs of this rank share the chip, so the rule is
    # applied while they hold at most two rounds of envs together (8 env slots per CU: 4096 on MI355X) and the one-wave kernel
    # otherwise (config 3: four slabs of 2048 swivel-chair envs ran 953 k env-steps/s with the rule and 1.06 M without)
    slots = 8 * torch.cuda.get_device_properties(local).multi_processor_count
    cfg.multi_wave = MULTI_WAVE[args.multi_wave] if args.multi_wave != "auto" else (MULTI_WAVE["off"] if n > 2 * slots else MULTI_WAVE["auto"])
    if args.control_type != "impedance":
        from furniture_amd.envs import CONTROLLER_CODES
        cfg.control_type = CONTROLLER_CODES[args.control_type]
    if os.environ.get("FSIM_BENCH_TOL"):  # development: Newton tolerance sweep (the shipped default is fsim_default_config's)
        cfg.solver_tolerance = float(os.environ["FSIM_BENCH_TOL"])
    if args.dense:  # config/furniture_sawyer_dense.py:4-14
        from furniture_amd.dense import pack_dense
        cfg.dense_reward, cfg.auto_align = 1, 0
        cfg.alignment_pos_dist, cfg.alignment_rot_dist_up, cfg.alignment_rot_dist_forward, cfg.alignment_project_dist = 0.02, 0.99, 0.99, 0.0
    # The rank's envs are split into `groups` equal slabs, each with its own FSim handle and HIP stream.  A step of the
    # batch = one step of every slab; the slabs are software-pipelined (while slab A's long-tail envs finish, slab B's
    # kernel fills the CUs), which is how a learner double-buffers a vectorised env (VecEnv step_async/step_wait).
    G = max(1, args.groups)
    assert n % G == 0, "--envs-per-gpu must be divisible by --groups"
    ng = n // G

    class Slab:
        pass

    slabs = []
    for g in range(G):
        sl = Slab()
        sl.index = g
        sl.sim = FSim(m, ng, device=local, config=cfg)
        if args.dense:
            sl.sim.set_dense_reward(*pack_dense(m))
        dev = sl.sim.device
        sl.tables = ResetTableQueue(ResetTableSampler(m, ecfg, SEED, lo + g * ng, ng))  # reference RNG stream, drawn one reset ahead
        sl.sim.set_reset_tables(*sl.tables.take())
        sl.obs = torch.zeros((ng, sl.sim.obs_dim), device=dev, dtype=torch.bfloat16 if args.obs_bf16 else torch.float32)
        sl.rew = torch.zeros(ng, device=dev)
        sl.done = torch.zeros(ng, dtype=torch.uint8, device=dev)
        sl.info = torch.zeros((ng, INFO_DIM), dtype=torch.int32, device=dev)
        sl.info_host = torch.zeros((ng, INFO_DIM), dtype=torch.int32).pin_memory()
        sl.gen = torch.Generator(device=dev)
        sl.gen.manual_seed(SEED + rank * 64 + g)
        sl.inflight = False
        sl.resteps = 0
        sl.pg = dist.new_group() if distributed else None  # one RCCL communicator (= one internal stream) per slab
        sl.sim.reset(None, sl.obs)
        sl.sim.sync()
        sl.sim.set_reset_tables(*sl.tables.take())  # tables for the first auto-reset
        slabs.append(sl)
    dev = slabs[0].sim.device

    # When the all-gather of a slab-step is enqueued.  One process: right behind the step kernel on the handle's stream (no host
    # round trip between the two).  Under RCCL: after fsim_sync -- because fsim_sync may re-step an env
    # whose contacts overflowed the slots (1.6 per million env-steps) and rewrite its rows, and a gather chained behind the FIRST pass
    # would hand the learner that env's stale row on every rank (a second gather only on the rank that saw the re-step would not be entered
    # by the others).  One collective per slab-step on
    # every rank either way, entered in the same slab order.
    GATHER_AFTER_SYNC = distributed

    def wait(sl):
        if not sl.inflight:
            return
        if GATHER_AFTER_SYNC:  # (furniture_amd/dist.py: sync -- incl. the overflow re-step if one was needed --, then the collective)
            sl.gathered = step_wait_and_gather(sl.sim, sl.obs, sl.rew, sl.done, tag=sl.index, stream=sl.sim.torch_stream if distributed else None, group=sl.pg)
        else:
            sl.sim.sync()  # the step kernel and the gather chained behind it
        sl.inflight = False
        sl.resteps = sl.sim.overflow_resteps()
        if sl.sim.tables_needed():  # host-side reference RNG stream for the envs that just consumed their reset table
            t_h = time.perf_counter()
            # (the WHOLE contiguous info block: a plain DMA copy.  A column slice is a strided gather KERNEL first, which waits for a wave
            #  slot -- behind the other slabs' reset-step kernels that was 55 ms of an idle host)
            # into a pinned buffer allocated up front, on the slab's own stream: no allocation inside the loop (hipMalloc / hipFree -- also
            # the ones behind torch's allocators -- wait until NO kernel is running)
            need = sl.sim.read_into(sl.info_host, sl.info).numpy()[:, INFO_NEEDS_TABLE]
            mask = need > 0
            if (need > 1).any():  # an unstable env: the reference draws twice (reset inside step() + the worker's reset)
                sl.tables.take(need > 1)
            t_a = time.perf_counter()
            p, nz = sl.tables.take(mask)
            t_b = time.perf_counter()
            sl.sim.set_reset_tables(p, nz, mask=mask)
            if os.environ.get("FSIM_BENCH_TRACE") and mask.sum() > 100:
                sys.stderr.write("  tables for slab %d (%d envs): info %.1f ms, take %.1f ms, upload %.1f ms\n" % (sl.index, int(mask.sum()), (t_a - t_h) * 1e3, (t_b - t_a) * 1e3, (time.perf_counter() - t_b) * 1e3))

    # U(-1,1) actions of every step, generated on the device BEFORE the timed region (the contract: inputs resident in HBM when
    # the clock starts) -- one slab-step's actions are a [ng, dof] slice; no torch kernel is launched inside the loop, where it
    # would queue behind the step kernels' waves for up to a millisecond
    ew_steps = args.episode_window if args.episode_window >= 0 else (MAX_EPISODE_STEPS + 10 if args.steps < MAX_EPISODE_STEPS else 0)
    total_steps = args.warmup + args.steps + ew_steps
    for sl in slabs:
        sl.actions = torch.empty((total_steps, ng, sl.sim.dof_action), device=dev).uniform_(-1, 1, generator=sl.gen)
        sl.t = 0
    torch.cuda.synchronize(dev)

    def launch(sl):
        sl.sim.step(sl.actions[sl.t], sl.obs, sl.rew, sl.done, sl.info)
        sl.t += 1
        if not GATHER_AFTER_SYNC:  # (single process: gather_observations hands the inputs back -- there is nobody to gather from)
            sl.gathered = gather_observations(sl.obs, sl.rew, sl.done, tag=sl.index, stream=sl.sim.torch_stream, group=sl.pg)
        sl.inflight = True

    def one_step():
        for sl in slabs:
            wait(sl)
            launch(sl)

    def drain():
        for sl in slabs:
            wait(sl)

    # development (FSIM_BENCH_ASYNC=1): every slab still steps exactly k times, but is relaunched as soon as ITS step has finished
    # instead of in round-robin order.  Measured with 2 / 4 / 8 slabs over five env/action seeds: 4 slabs relaunched this way
    # reach 705-740 k env-steps/s on the default seed's first 100 steps, but 634 k on average over seeds against 672 k for the
    # default (2 slabs, round robin), and the 1000-step protocol is unchanged (631 k) -- not adopted.
    ASYNC = os.environ.get("FSIM_BENCH_ASYNC", "0") == "1" and not distributed

    TRACE = os.environ.get("FSIM_BENCH_TRACE")  # development: wall time of every block of 50 batched steps, on stderr

    THREADS = bool(args.threads) and not distributed
    TRACE = os.environ.get("FSIM_BENCH_TRACE")  # development: wall time of every block of 50 batched steps, on stderr

    def run_steps(k):
        if THREADS:  # one host thread per slab (ctypes releases the GIL inside fsim_sync): every slab cycles at its own pace
            import threading

            def loop(sl):
                tw = tl = 0.0
                for _ in range(k):
                    t_a = time.perf_counter()
                    wait(sl)
                    t_b = time.perf_counter()
                    launch(sl)
                    tl += time.perf_counter() - t_b
                    tw += t_b - t_a
                if TRACE and sl.index in (0, len(slabs) - 1):
                    sys.stderr.write("slab %d: %d steps, host blocked in wait %.3f ms/step, in launch %.3f ms/step\n" % (sl.index, k, tw / k * 1e3, tl / k * 1e3))
            th = [threading.Thread(target=loop, args=(sl,)) for sl in slabs]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            return
        if not ASYNC:
            tb = time.perf_counter()
            for i in range(k):
                one_step()
                if TRACE and i % 50 == 49:
                    sys.stderr.write("steps %4d..%4d: %.3f ms/step\n" % (i - 49, i, (time.perf_counter() - tb) / 50 * 1e3))
                    tb = time.perf_counter()
            return
        left = {sl.index: k for sl in slabs}
        while any(left.values()) or any(sl.inflight for sl in slabs):
            for sl in slabs:
                if sl.inflight and not sl.sim.torch_stream.query():
                    continue
                wait(sl)
                if left[sl.index]:
                    launch(sl)
                    left[sl.index] -= 1

    run_steps(args.warmup)
    drain()
    la0 = [sl.sim.lookahead_stats() for sl in slabs]
    for sl in slabs:
        sl.sim.kernel_time_ms()  # reset the accumulators
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_steps(args.steps)
    drain()
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    la_mid = [sl.sim.lookahead_stats() for sl in slabs]
    kt = [sl.sim.kernel_time_ms() for sl in slabs]  # (of the contract's timed region: read before the second window adds its launches)
    # Second window, NOT the headline: the contract's K steps after a reset contain no episode end (150-step episodes), so no reset and no
    # look-ahead reset work; this one spans a full episode right behind it -- every env resets once inside it -- on the same clock rules.
    ew = None
    if ew_steps > 0:
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        run_steps(ew_steps)
        drain()
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        dt_ew = time.perf_counter() - t1
        if distributed:
            t = torch.tensor([dt_ew], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ew = float(t)
        la_end = [sl.sim.lookahead_stats() for sl in slabs]
        d_ = {k: sum(b[k] - a[k] for a, b in zip(la_mid, la_end)) for k in ("units", "swapped", "inline")}
        reset_sub = 401 if m.meta.get("has_recipe") else 301
        ew = {"steps": ew_steps, "env_steps_per_s": world * n * ew_steps / dt_ew, "ms_per_step": dt_ew / ew_steps * 1e3,
              "resets": d_["swapped"] + d_["inline"], "resets_taken_from_lookahead": d_["swapped"], "reset_substeps": d_["units"] + d_["inline"] * reset_sub,
              "note": "a second timed window right behind the contract's, spanning one full episode (every env resets once inside it); per-rank counts"}
    la1 = la_mid
    # envs whose record carries the sticky contact-overflow word (fsim_model.hpp E_OVERFLOW): some launch since the handle was created
    # -- warm-up included -- needed more contact slots than the kernel's LDS image holds and dropped the rest for that substep
    from furniture_amd.sim import E_OVERFLOW
    dropped = int(sum(int((sl.sim.get_state("env_block")["env_block"].view(torch.int32)[:, E_OVERFLOW] != 0).sum()) for sl in slabs))
    la = {k: sum(b[k] - a[k] for a, b in zip(la0, la1)) for k in ("units", "swapped", "inline")}
    reset_substeps = 401 if m.meta.get("has_recipe") else 301  # sim.step() calls of one _reset (tests/golden/reset_trace.npz)
    klaunches = sum(k[1] for k in kt)
    kms = sum(k[0] * k[1] for k in kt) / max(1, klaunches)
    units_per_launch = ng  # env-steps one launch of the dominant kernel processes
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    obs = torch.cat([sl.obs for sl in slabs])
    finite = bool(torch.isfinite(obs).all())

    if rank == 0:
        total_env_steps = world * n * args.steps
        value = total_env_steps / dt
        achieved = ALGO_BYTES_PER_ENV_STEP * units_per_launch / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        # What the counters say about this kernel (rocprofv3 --pmc passes of the same workload, scripts/profile_round.sh; bench.py
        # cannot profile itself).  HBM traffic feeds the contract's roofline object; the rest says what actually binds.
        traffic, traffic_note, pmc = None, None, {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
                pmc = json.load(f)
            traffic = pmc["bytes_per_env_step"] * units_per_launch
            traffic_note = ("builder-lease PMC, not this run: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes, %s), %d B per env-step x %d env-steps per launch"
                            % (pmc.get("source", "profiles/"), pmc["bytes_per_env_step"], units_per_launch))
        except Exception:
            pass
        # chip-wide utilisation of THIS run: instructions per env-step (PMC, a property of the workload) x the measured env-step rate
        # over the issue slots of the chip (SIMDs x clock; one wave-instruction per SIMD per cycle at best)
        props = torch.cuda.get_device_properties(local)
        simds, clk = props.multi_processor_count * 4, float(getattr(props, "clock_rate", 2.4e6)) * 1e3  # (kHz; MI355X_MICROARCH.md: 2.4 GHz peak engine clock)
        per_gpu_rate = value / world
        util = lambda k: (pmc[k] * per_gpu_rate / (simds * clk)) if pmc.get(k) else None
        binding = {"bound": "per-wave instruction issue + LDS wait (not HBM, not MFMA); the step ends with its slowest env",
                   "issue_util_this_run": util("insts_per_env_step"), "valu_util_this_run": util("valu_insts_per_env_step"),
                   "util_def": "wave-instructions (all / VALU) issued per SIMD per cycle, chip-wide, over the timed region: PMC instructions per env-step x measured env-steps/s / (SIMDs x clock)",
                   "simds": simds, "clock_hz": clk,
                   "occupancy_this_run_waves_per_simd": (pmc["wave_cycles_per_env_step"] * per_gpu_rate / (simds * clk)) if pmc.get("wave_cycles_per_env_step") else None,
                   "waves_per_simd_limit": 2,
                   "wait_frac": pmc.get("wait_frac"), "issue_frac": pmc.get("issue_frac"),
                   "insts_per_env_step": pmc.get("insts_per_env_step"), "valu_insts_per_env_step": pmc.get("valu_insts_per_env_step"),
                   "wave_cycles_per_env_step": pmc.get("wave_cycles_per_env_step"),
                   "algorithmic_flops_per_env_step": [1.5e6, 5.5e6], "fp32_vector_peak_tflops": 157.3,
                   "achieved_tflops_algorithmic": [1.5e6 * per_gpu_rate / 1e12, 5.5e6 * per_gpu_rate / 1e12],
                   "pmc_source": "builder-lease PMC passes (per-env-step counts), not this run: %s" % pmc.get("source"),
                   # (builder-lease measurement of the benchmark's own slab 0, not this run: DESIGN.md section 6 item 6)
                   "slab_of_1024_alone_vs_one_of_four": {"ms_per_step_alone": 4.15, "ms_per_step_one_of_four": 4.90, "env_slowdown_when_shared": 1.08,
                                                         "launch_ends_with": "a second-round one-wave env that started ~1 ms late (31 of 35 steps)",
                                                         "source": "profiles/r06_w_slab_alone_against_one_of_four.txt, profiles/r06_w_same_envs_alone_and_shared_timeline.txt"}}
        line = {
            "metric": "env-steps/sec (whole node), Sawyer+table_lack 4096 envs/GPU" if (args.agent, args.furniture, n, args.dense, args.control_type) == (AGENT, FURNITURE, ENVS_PER_GPU, False, "impedance")
            else "env-steps/sec (whole node), EXPLORATION %s+%s%s %d envs/GPU" % (args.agent, args.furniture, (" dense-reward" if args.dense else "") + ("" if args.control_type == "impedance" else " control_type=" + args.control_type), n), "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Furniture%sEnv + %s, %s control, %d envs/GPU, U(-1,1)^%d actions, "
                                   "50 substeps/step, max_episode_steps=150 with in-kernel auto-reset" % (args.agent, args.furniture, args.control_type, n, slabs[0].sim.dof_action),
                       "envs_per_gpu": n, "global_envs": world * n,
                       "parallelism": "env-sharded x%d, RCCL obs all-gather; %d slab(s) of %d envs per GPU, %s, %s" % (
                           world, G, ng, "a scheduler + step launch per slab-step on the slab's own HIP stream",
                           "one host thread per slab" if THREADS else "one host thread, round robin"),
                       "rccl_world": dist.get_world_size() if distributed else 1,  # ranks RCCL's communicator saw (1 without torch.distributed.run)
                       # every episode end costs its reset (the reference's _reset: 401 sim.step() calls here).  They are executed INSIDE the
                       # timed region: ahead of the step that needs them (look-ahead jobs) or inside that step's launch
                       "resets_in_timed_region": la["swapped"] + la["inline"], "resets_taken_from_lookahead": la["swapped"], "resets_inside_step_launch": la["inline"],
                       "lookahead_reset_units_in_timed_region": la["units"],  # reset substeps run by look-ahead jobs (waves of the step launches that had no env left)
                       "reset_substeps_in_timed_region": la["units"] + la["inline"] * reset_substeps,
                       "envs_that_dropped_contacts": dropped, "contact_slots": slabs[0].sim.max_contacts,
                       # env-steps (warm-up included) the library repeated with a 64-slot layout because 48 slots did not hold their contacts
                       # (fsim_overflow_resteps: done inside fsim_sync, i.e. inside the timed region)
                       "overflow_resteps": int(sum(sl.sim.overflow_resteps() for sl in slabs)),
                       "episode_window": ew, "episode_window_env_steps_per_s": ew["env_steps_per_s"] if ew else None,
                       "physics_substeps_per_s": value * 50, "obs_finite": finite, "obs_dtype": "bf16" if args.obs_bf16 else "f32", "kernel_variant": slabs[0].sim.kernel_variant,
                       "reference_published_single_core_env_steps_per_s": 225},
            # `bound`: what the contract's two choices are priced against is HBM (BASELINE.json asks for the HBM fraction) and `frac` is
            # computed against it; what actually binds is neither -- `bound_actual` / `binding` say what, with numbers.  `traffic` and the
            # per-env-step instruction counts come from the builder's own rocprofv3 PMC passes of this workload (profiles/pmc_latest.json),
            # the utilisation figures below are recomputed FOR THIS RUN from them and this run's measured rate.
            "roofline": {"bound": "hbm", "bound_actual": "per-wave instruction issue + LDS latency, at a batch-synchronous step that ends with its slowest env (see binding)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": slabs[0].sim.step_kernel, "kernel_avg_ms": kms, "kernel_launches": klaunches,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * units_per_launch,
                         "env_steps_per_launch": units_per_launch,
                         # `achieved` / `frac` price ONE launch (the contract's definition) -- but G launches of G slabs overlap on the chip, so the
                         # chip-wide algorithmic rate over the timed region is the whole batch's bytes / the batched step's wall time
                         "concurrent_launches": G, "chip_wide_achieved": ALGO_BYTES_PER_ENV_STEP * n / (dt / args.steps) / 1e9,
                         "chip_wide_frac": ALGO_BYTES_PER_ENV_STEP * n / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "note": "fused 50-substep step keeps state in LDS: HBM fraction is ~0 by design; see `binding`",
                         # what binds instead (SURVEY 8d asked for VALU utilisation and occupancy): one wavefront = one env, and a
                         # wave issues at most one instruction per ~5 cycles (scripts/dev/micro: 5.0 cycles per dependent-distance-4
                         # v_fma at 1-2 waves per SIMD), so an env-step costs ~5 cycles x instructions + LDS waits
                         "binding": binding},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.cpu_baseline)
        else:  # (the contract times it on rank 0 at N = 1 only: the 1-GPU line of the same sweep carries it)
            line["cpu_baseline"] = {"skipped": "--no-cpu-baseline" if args.no_cpu_baseline else "n_gpus > 1: timed at N = 1 only (see the 1-GPU line)"}
        print(json.dumps(line))
    for sl in slabs:
        sl.tables.close()
        sl.sim.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
