// fsim.hip -- libfsim.so: kernels + host side + the C-ABI of include/fsim.h.
//
// One HIP workgroup = one wavefront (64 lanes) = one environment.  The env's state record is
// streamed HBM -> LDS once per launch, all n_substeps physics steps (and the env logic around
// them) run out of LDS, and the record is streamed back: compulsory traffic only.
// Workgroup b lands on XCD b % 8; envs are independent and the model tables are read-only, so
// every XCD keeps its own L2 copy of the (tens of KB) model and no cross-XCD coherence is needed.
#include "../../include/fsim.h"
#include "fsim_env.hpp"
#include "fsim_spec.hpp"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

static thread_local char g_err[512];
extern "C" const char *fsim_last_error(void) { return g_err; }
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return code; } while (0)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FAIL(FSIM_EHIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------------------------------ kernels
struct KParams {
  int n_envs, n_substeps, mode; // mode: 0 physics step, 1 forward only
  int newton_maxit;
  float newton_tol;
};

#ifdef FSIM_DBG_FILL
// development (scripts/dev/r5/lds_uninit.sh): LDS words [lo, hi) of every env image are set to a pattern before the env runs -- with the rest
// of the LDS poisoned by another kernel, the range whose filling changes the results contains a word that is read before it is written
__device__ int g_dbg_fill[3];
#endif
__device__ __forceinline__ void load_record(float *L, const float *rec, int n, int lane) {
  for (int i = lane; i < n; i += 64) L[i] = rec[i];
}
__device__ __forceinline__ void store_record(float *rec, const float *L, int n, int lane) {
  for (int i = lane; i < n; i += 64) rec[i] = L[i];
}

template <class Ctx> __global__ __launch_bounds__(64 * Ctx::NW, 2) void k_physics(const DModel *mp, const Layout *lp, KParams kp, float *state, float *aux) {
  extern __shared__ float L[];
  CModel &m = *(CModel *)mp;
  const int env = blockIdx.x;
  if (env >= kp.n_envs) return;
  const Ctx c(L, m, *(CLayout *)lp, (int)threadIdx.x, kp.newton_maxit, kp.newton_tol);
  const int lane = c.lane;
  if constexpr (Ctx::NW > 1) if (c.wave > 0) { mw_helper_loop(c); return; } // helper waves (multi-wave kernel)
  float *rec = state + (size_t)env * c.ly.stride;
  load_record(L, rec, c.ly.stride, lane);
  for (int i = lane; i < SC_WORDS; i += 64) reinterpret_cast<int *>(L + c.ly.scal)[i] = 0;
  if constexpr (Ctx::NW > 1) if (lane < FSIM_MWCW) c.I(c.ly.mwc)[lane] = 0;
  SYNC();
  fs_load_cache(c);
  if (kp.mode == 1) fs_substeps(c, 1, 1);
  else fs_substeps(c, kp.n_substeps, 0);
  if constexpr (Ctx::NW > 1) mw_post(c, MW_EXIT);
  // aux: [qacc nv][xpos 3nr][xquat 4nr][ncon, niter, overflow, bad][contact geoms 2*ncon_max]
  if (aux) {
    float *a = aux + (size_t)env * (c.D.nv + 7 * c.D.nr + 4 + 2 * c.ly.ncon_max);
    for (int d = lane; d < c.D.nv; d += 64) a[d] = L[c.ly.x + d];
    for (int i = lane; i < 3 * c.D.nr; i += 64) a[c.D.nv + i] = L[c.ly.xpos + i];
    for (int i = lane; i < 4 * c.D.nr; i += 64) a[c.D.nv + 3 * c.D.nr + i] = L[c.ly.xquat + i];
    int *ai = reinterpret_cast<int *>(a + c.D.nv + 7 * c.D.nr);
    const int *scal = reinterpret_cast<const int *>(L + c.ly.scal);
    if (lane == 0) { ai[0] = scal[SC_NCON]; ai[1] = scal[SC_NITER]; ai[2] = scal[SC_OVERFLOW]; ai[3] = scal[SC_BAD]; }
    // ordered list of (geom1, geom2) original ids of listed contacts
    if (lane == 0) {
      int n = 0, nslot = scal[SC_NSLOT];
      for (int s = 0; s < nslot; s++) {
        const int *ri = reinterpret_cast<const int *>(L + c.ly.con + FSIM_CONW * s);
        if (ri[C_ACTIVE]) { ai[4 + 2 * n] = m.cg_orig[ri[C_G1]]; ai[4 + 2 * n + 1] = m.cg_orig[ri[C_G2]]; n++; }
      }
      for (; n < c.ly.ncon_max; n++) { ai[4 + 2 * n] = -1; ai[4 + 2 * n + 1] = -1; }
    }
  }
  SYNC();
  env_cursor_keep_poses(c);
  store_record(rec, L, c.ly.stride, lane);
}

// What every flavour of the step kernel does for ONE env once it knows which env and which waves (fsim_env.hpp does the work).
struct StepArgs {
  EnvCfg cfg;
  float *state;
  const float *action;
  float *obs, *reward;
  uint8_t *done;
  int *info;
  const float *tab_parts, *tab_noise;
  const float *tab_attach; // [n][narmj]: joint noise of each env's next attach (config.reset_robot_after_attach), or null
  int n_noise;
  const uint8_t *reset_mask;
  int do_step;
  int *prof, *cost;
  const float *init_state;
  const uint8_t *init_mask;
  int *nreset;
  const EnvCfg *cfg_dev; // `cfg` again, in device memory (for the out-of-line env_reset: EnvResetIO)
  // look-ahead reset: shadow records [n][stride], shadow observation rows [n][obs_dim] (f32 | bf16), progress / serial words [n]; null = off
  float *sh_state;
  void *sh_obs;
  int *sh_prog, *sh_serial;
  const int *tab_serial; // [n] serial number of the reset table on the device (bumped by the host with every upload)
  const int *sh_jobs;    // look-ahead jobs of this launch (env indices, listed by k_schedule; q[4] of them), or null
  int la_chunk;          // reset units per look-ahead job
  // Contact-overflow re-step (round 4): every step launch leaves the pre-step record of each env in `prev` and lists the envs whose step
  // dropped contacts (ovf_list / *ovf_count, host-mapped); fsim_sync re-steps those from `prev` (state_in) with a layout of more slots.
  float *prev;
  const float *state_in;
  int *ovf_list, *ovf_count;
  int ovf_cap;
  int *stats; // host-mapped counters (fsim::h_nreset + 1): [0] resets taken from a shadow record, [1] resets executed in a step / reset launch, [2] reset units run by look-ahead jobs
};
enum { JOB_AUTO = 0 /* a.do_step decides */, JOB_RESET = 1 /* the deferred reset of a multi-wave workgroup's env */ };
// load_cache: the wave's LDS copy of the model tables is not there yet (a bundled wave steps several envs one after the other
// and loads it once)
// returns 1 if the env's reset was deferred (DEFER, see env_step): the caller runs it as a JOB_RESET
// a: the kernel's own argument block; cfg: the kernel's own EnvCfg
template <class Ctx, bool DEFER = false, class A> DEV int env_run(const Ctx &c, const EnvCfg &cfg, A &a, int env, long long t_entry, bool load_cache = true, int job = JOB_AUTO) {
  float *L = c.L;
  const int lane = c.lane;
  float *rec = a.state + (size_t)env * c.ly.stride;
#ifdef FSIM_TIMELINE
  const long long tw0_ = wall_clock64(); // (100 MHz, one counter for the whole device: clock64() has an offset per XCD)
#endif
#ifdef FSIM_DBG_FILL
  for (int i = g_dbg_fill[0] + lane; i < g_dbg_fill[1]; i += 64) reinterpret_cast<int *>(L)[i] = g_dbg_fill[2];
  SYNC();
#endif
  load_record(L, a.state_in ? a.state_in + (size_t)env * c.ly.stride : rec, c.ly.stride, lane);
  if (a.prev && job != JOB_RESET) store_record(a.prev + (size_t)env * c.ly.stride, L, c.ly.stride, lane); // (what a re-step starts from)
  const int twords = reinterpret_cast<int *>(L + c.ly.scal)[SC_TWORDS]; // (set by fs_load_cache: survives the per-env clearing)
  for (int i = lane; i < SC_WORDS; i += 64) reinterpret_cast<int *>(L + c.ly.scal)[i] = 0;
  if constexpr (Ctx::NW > 1) if (lane < FSIM_MWCW) c.I(c.ly.mwc)[lane] = 0;
  SYNC();
#if defined(FSIM_TIMELINE) && !defined(FSIM_PROFILE)
  const long long tlc0_ = wall_clock64();
#endif
  if (load_cache) fs_load_cache(c);
  else { if (lane == 0) reinterpret_cast<int *>(L + c.ly.scal)[SC_TWORDS] = twords; SYNC(); }
#if defined(FSIM_TIMELINE) && !defined(FSIM_PROFILE)
  const int tlc_ = (int)(wall_clock64() - tlc0_), tlr_ = (int)(tlc0_ - tw0_);
#endif
  EnvIO io;
  io.action = a.action ? a.action + (size_t)env * cfg.dof_action : nullptr;
  io.obs = a.obs ? reinterpret_cast<float *>(reinterpret_cast<char *>(a.obs) + (size_t)env * cfg.obs_dim * (cfg.obs_bf16 ? 2 : 4)) : nullptr;
  io.reward = a.reward ? a.reward + env : nullptr;
  io.done = a.done ? a.done + env : nullptr;
  io.info = a.info ? a.info + (size_t)env * FSIM_INFO_DIM : nullptr;
  io.tab_parts = a.tab_parts ? a.tab_parts + (size_t)env * 7 * c.D.nparts : nullptr;
  io.tab_noise = a.tab_noise ? a.tab_noise + (size_t)env * a.n_noise * c.D.narmj : nullptr;
  io.n_noise = a.n_noise;
  io.tab_attach = a.tab_attach ? a.tab_attach + (size_t)env * c.D.narmj : nullptr;
  io.nreset = a.nreset;
  io.init_state = (a.init_state && a.init_mask && a.init_mask[env]) ? a.init_state + (size_t)env * (c.D.nq + c.D.nv) : nullptr;
  io.cost = a.cost ? a.cost + env : nullptr;
  io.t0 = t_entry;
  io.cfg_dev = a.cfg_dev;
  io.sh_state = a.sh_state ? a.sh_state + (size_t)env * c.ly.stride : nullptr;
  io.sh_obs = a.sh_obs ? static_cast<const char *>(a.sh_obs) + (size_t)env * cfg.obs_dim * (cfg.obs_bf16 ? 2 : 4) : nullptr;
  io.sh_prog = a.sh_prog ? a.sh_prog + env : nullptr;
  // (a deferred reset was deferred BECAUSE the shadow was not ready when the env's step looked: it resets inline whatever a job has finished since)
  io.sh_prog0 = (a.sh_prog && job != JOB_RESET) ? __builtin_amdgcn_readfirstlane(a.sh_prog[env]) : -1;
  io.sh_serial = a.sh_prog ? a.sh_serial[env] : 0;
  io.tab_serial = a.sh_prog ? a.tab_serial[env] : 0;
  io.stats = a.stats;
  int deferred = 0;
  if (job == JOB_RESET) env_reset_or_swap(c, cfg, io);
  else if (a.do_step) deferred = env_step<Ctx, DEFER>(c, cfg, io);
  else if (!a.reset_mask || a.reset_mask[env]) env_reset_or_swap(c, cfg, io);
  if constexpr (Ctx::NW > 1) mw_post(c, MW_EXIT);
  SYNC();
  if (a.ovf_count && lane == 0 && reinterpret_cast<const int *>(L + c.ly.scal)[SC_OVERFLOW] != 0) { // this launch dropped contacts of this env
    const int k = __hip_atomic_fetch_add(a.ovf_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (k < a.ovf_cap) a.ovf_list[k] = env;
  }
#ifdef FSIM_PROFILE
#ifdef FSIM_TIMELINE
  // development: when and where this workgroup ran (scripts/dev/timeline.py: how many are resident at a time)
  if (lane == 0) {
    int *ps_ = reinterpret_cast<int *>(L + c.ly.scal);
    ps_[53] = (int)(tw0_ & 0x7fffffff); ps_[54] = (int)(wall_clock64() & 0x7fffffff);
    ps_[50] = (int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // HW_REG_HW_ID, 32 bits
    int xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); ps_[49] = xcc;
  }
  SYNC();
#endif
  // (the debug row: words [0, nv) are read back as `qacc`, the ones behind that as `contact_geoms` -- the fields whose read-back is a plain copy)
  if (a.prof && lane < 48) a.prof[(size_t)env * (c.D.nv + 7 * c.D.nr + 4 + 2 * c.ly.ncon_max) + (lane < c.D.nv ? lane : 7 * c.D.nr + 4 + lane)] = reinterpret_cast<int *>(L + c.ly.scal)[16 + lane];
#endif
#if defined(FSIM_TIMELINE) && !defined(FSIM_PROFILE)
  // development (scripts/dev/timeline_x.py): start / end tick only, into the same words of the debug row -- without the 48 profile words
  // per env in LDS, with which two bundles of k_env_step_x no longer fit a CU
  if (a.prof && lane == 0) {
    int *row = a.prof + (size_t)env * (c.D.nv + 7 * c.D.nr + 4 + 2 * c.ly.ncon_max);
    row[37] = (int)(tw0_ & 0x7fffffff); row[38] = (int)(wall_clock64() & 0x7fffffff);
    row[35] = tlr_; row[36] = load_cache ? tlc_ : -1; // (10 ns ticks: record load, model-cache build -- -1: the wave had it already)
  }
#endif
  env_cursor_keep_poses(c);
  store_record(rec, L, c.ly.stride, lane);
  return deferred;
}

// One look-ahead JOB: the next a.la_chunk units of the reset env's next episode will start from (env_reset_units), on the shadow
// record -- which carries the intermediate state from job to job -- from the reset table the host has already uploaded.  The last job
// adds what a reset inside a launch adds (env_post: scheduler words, observation -> the shadow observation row).  Run by the waves of a
// step launch that have no env left to step (k_env_step_x) / by extra workgroups (k_env_step); k_schedule lists the jobs.
// A shadow that belongs to another table (serial) is started over.
template <class Ctx, class A> DEV void env_shadow_job(const Ctx &c, const EnvCfg &cfg, A &a, int env, bool load_cache) {
  float *L = c.L;
  const int lane = c.lane;
  const int serial = a.tab_serial[env];
  int prog = a.sh_prog[env];
  if (a.sh_serial[env] != serial) prog = 0;
  float *rec = a.sh_state + (size_t)env * c.ly.stride;
  // (a reset overwrites every word of the record but the episode counter and the sticky overflow report, which the swap takes from the
  //  live record: the first job starts from zeros)
  if (prog > 0) load_record(L, rec, c.ly.stride, lane);
  else for (int i = lane; i < c.ly.stride; i += 64) L[i] = 0.0f;
  const int twords = reinterpret_cast<int *>(L + c.ly.scal)[SC_TWORDS];
  for (int i = lane; i < SC_WORDS; i += 64) reinterpret_cast<int *>(L + c.ly.scal)[i] = 0;
  SYNC();
  if (load_cache) fs_load_cache(c);
  else { if (lane == 0) reinterpret_cast<int *>(L + c.ly.scal)[SC_TWORDS] = twords; SYNC(); }
  EnvIO io{};
  io.obs = reinterpret_cast<float *>(static_cast<char *>(a.sh_obs) + (size_t)env * cfg.obs_dim * (cfg.obs_bf16 ? 2 : 4));
  io.tab_parts = a.tab_parts ? a.tab_parts + (size_t)env * 7 * c.D.nparts : nullptr;
  io.tab_noise = a.tab_noise ? a.tab_noise + (size_t)env * a.n_noise * c.D.narmj : nullptr;
  io.n_noise = a.n_noise;
  io.init_state = (a.init_state && a.init_mask && a.init_mask[env]) ? a.init_state + (size_t)env * (c.D.nq + c.D.nv) : nullptr;
  io.cfg_dev = a.cfg_dev;
  const int total = env_reset_total(cfg, io.init_state != nullptr), p1 = min(prog + a.la_chunk, total);
  env_reset_units(c, a.cfg_dev, env_reset_io(io), prog, p1);
  if (p1 == total) env_post(c, cfg, io, 0);
  SYNC();
  env_cursor_keep_poses(c);
  store_record(rec, L, c.ly.stride, lane);
  if (lane == 0) {
    // (a shadow whose reset dropped contacts is never valid -- progress beyond "consumed": the terminal step then resets inside its
    //  launch, where the overflow re-step can repeat it with more slots; the jobs stop being listed)
    const bool dropped = reinterpret_cast<const int *>(L + c.ly.env)[E_OVERFLOW] != 0;
    a.sh_prog[env] = dropped ? total + 1 : p1; a.sh_serial[env] = serial;
    if (a.stats) __hip_atomic_fetch_add(a.stats + 2, p1 - prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// One env per workgroup: the one-wave kernel (Ctx::NW == 1; workgroup b steps env order[b]) and the multi-wave kernel on its own
// (Ctx::NW == 4, FSIM_MW=all: every env gets four waves -- development and tests).
template <class Ctx> __global__ __launch_bounds__(64 * Ctx::NW, FSIM_WPE) void k_env_step(const DModel *mp, const Layout *lp, KParams kp, StepArgs a, const int *order, const int *q) {
  extern __shared__ float L[];
  CModel &m = *(CModel *)mp;
  long long t_entry = clock64();
  const Ctx c(L, m, *(CLayout *)lp, (int)threadIdx.x, kp.newton_maxit, kp.newton_tol);
  if ((int)blockIdx.x >= kp.n_envs) { // workgroups behind the envs: this launch's look-ahead jobs (one-wave kernel only), q[4] of them
    if constexpr (Ctx::NW == 1) { const int j = (int)blockIdx.x - kp.n_envs; if (a.sh_jobs && j < q[4]) env_shadow_job(c, a.cfg, a, a.sh_jobs[j], true); }
    return;
  }
  // workgroups are dispatched in blockIdx order: `order` lists the envs longest-predicted-job first (k_schedule)
  const int env = order ? order[blockIdx.x] : (int)blockIdx.x;
  if constexpr (Ctx::NW > 1) if (c.wave > 0) { mw_helper_loop(c); return; } // helper waves
  env_run(c, a.cfg, a, env, t_entry);
}

// The step kernel proper: ONE launch of persistent 4-wave workgroups.  Every workgroup first serves the MULTI-WAVE queue -- the envs
// the scheduler selected (mworder[0 .. *mwn)): four waves step one env, wave 0 as main, the others as its helpers -- and, once that
// queue is empty, turns into a BUNDLE: four independent one-wave workers (each wave its own LDS image, never synchronising again) that
// take envs of the longest-job-first list from a device-wide queue until it is empty (a workgroup's LDS is only released when its
// last wave ends, so fixed bundles would idle behind their slowest env).  Workgroups are dispatched in index order and the first
// thing any of them does is take a multi-wave env, so the long jobs start first; with the two kinds in separate launches the one-wave
// workgroups took every slot that freed up and the 4-wave ones starved.  There is no cap on the number of multi-wave envs: the
// selection is a function of each env's own state, whatever the rest of the batch does (a cap would make an env's arithmetic depend
// on its neighbours), and a launch with more of them than workgroups just serves them in turn.
// q: [0] multi-wave envs of the launch, [1] the others, [2] head of the bundle queue, [3] head of the multi-wave queue (k_schedule)
template <class CtxM, class CtxB> __global__ __launch_bounds__(256, FSIM_WPE) void k_env_step_x(const DModel *mp, const Layout *lp, const Layout *lp_mw, KParams kp, StepArgs a,
                                                                                           const int *order, const int *mworder, int *q, int *defer) {
  extern __shared__ float L[];
  CModel &m = *(CModel *)mp;
  long long t_entry = clock64();
  const int nmw = q[0];
  // Envs of this workgroup's multi-wave phase whose step ended the episode with no shadow record ready: their reset is ONE-wave work
  // (env_step DEFER) and runs in the bundle phase below, on wave 0, before it takes anything from the queue -- the same instantiation
  // a bundled env's reset runs, so a reset's bits do not depend on who stepped the env.  The list is the workgroup's row of `defer`
  // ([gridDim.x][n_envs] in device memory, written and read by wave 0 alone): it holds whatever number of deferrals the launch produces
  // (until round 5 it was four 16-bit indices in a register pair, and a workgroup whose list was full left the queue for good).
  int *const dlist = defer + (size_t)blockIdx.x * kp.n_envs;
  int ndef = 0;
  if (nmw > 0) {
    int *s_slot = reinterpret_cast<int *>(L); // (between two envs nothing in the workgroup's LDS is live)
    for (;;) {
      if (threadIdx.x == 0) *s_slot = atomicAdd(q + 3, 1);
      __syncthreads();
      const int slot = __builtin_amdgcn_readfirstlane(*s_slot);
      __syncthreads(); // (everybody has read the slot before wave 0 goes on and overwrites it)
      if (slot >= nmw) break;
      // (the thread index through an opaque copy, here and in the loops below: per-lane address arithmetic of the env code then cannot be
      //  hoisted out of the persistent loops to the kernel's entry, where it was spilled -- ~92 scratch stores per wave of every launch)
      int tid_ = (int)threadIdx.x;
      __asm__ volatile("" : "+v"(tid_));
      const CtxM c(L, m, *(CLayout *)lp_mw, tid_, kp.newton_maxit, kp.newton_tol);
      if (c.wave > 0) { mw_helper_fn(c); continue; }
      const int env = mworder[slot];
      if (env_run<CtxM, true>(c, a.cfg, a, env, t_entry)) { if (c.lane == 0) dlist[ndef] = env; ndef++; }
      t_entry = clock64();
    }
  }
  {
    const int nb = q[1];
    bool first = true;
    for (;; first = false) {
      int tid_ = (int)threadIdx.x;
      __asm__ volatile("" : "+v"(tid_));
      const CtxB c(L, m, *(CLayout *)lp, tid_, kp.newton_maxit, kp.newton_tol);
      int env, job = JOB_AUTO;
      if (ndef > 0) { ndef--; env = __builtin_amdgcn_readfirstlane(dlist[ndef]); job = JOB_RESET; } // (wave 0 only: ndef is 0 on the others)
      else {
        int slot = 0;
        if (c.lane == 0) slot = atomicAdd(q + 2, 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= nb) break;
        env = order[slot];
      }
      env_run(c, a.cfg, a, env, first ? t_entry : clock64(), first, job);
    }
    // nothing left to step: this wave would retire while the launch waits for its slowest env -- it runs look-ahead jobs instead
    // (the next reset of envs whose table is on the device already: env_shadow_job), q[4] of them listed by k_schedule
    if (a.sh_jobs) {
      const int nj = q[4];
      for (;; first = false) {
        int tid_ = (int)threadIdx.x;
        __asm__ volatile("" : "+v"(tid_));
        const CtxB c(L, m, *(CLayout *)lp, tid_, kp.newton_maxit, kp.newton_tol);
        int j = 0;
        if (c.lane == 0) j = atomicAdd(q + 5, 1);
        j = __builtin_amdgcn_readfirstlane(j);
        if (j >= nj) break;
        env_shadow_job(c, a.cfg, a, a.sh_jobs[j], first);
      }
    }
  }
}

// Longest-job-first launch order.  An env-step's cost varies 5x with its contact state (robot gripping a part =>
// coupled Newton systems) and 7x when the episode ends inside the launch (in-kernel reset), and the kernel's duration
// is the finish time of the last wave: sorting the grid by predicted cost keeps the long jobs off the tail.
// Key (written by env_step): -1 = will reset; bit 30 = a robot hand is near a part (likely to couple); low bits = shader
// cycles >> 10 of the step just taken.  One workgroup, bucket sort: [reset | near, by cost | the rest, by cost]; order
// inside a bucket is arbitrary -- it only affects timing, never results (envs are independent).
// ONE wavefront with a few registers: the kernel runs between two step kernels of its stream while the other slab's step kernel
// holds every SIMD's register file (2 x 256 VGPRs) -- a 1024-thread workgroup had to wait ~0.2 ms for a whole CU to drain
// before it could start (rocprofv3: 217 us average for 10 us of work), a single small wave takes the first slot that frees.
// Multi-wave selection (use_mw): env i goes to the multi-wave queue of this launch iff its last step took at least mw_k
// Newton iterations (E_NITER of its record) and it is not about to hit the time limit (key -1: its step ends in a reset, which is
// one-wave work) -- a function of the env's OWN state, never of timing or of the other envs of the batch, so an env's results do not
// depend on the schedule, the batch it is stepped in or the slab layout -> mworder / q[0]; `order` / q[1] then list the others.
// The kernel is one latency chain (it sits between two step kernels of its stream): the keys are fetched once, eight loads in
// flight per lane, and kept in LDS for the sorting passes (208 us -> ~20 us for 2048 envs).
#define FSIM_SCHED_SELECTED ((int)0x80000000)
#ifndef FSIM_SCHED_U
#define FSIM_SCHED_U 8 // envs per lane whose keys are in flight at once (the unrolled fetch: code size against loads in flight)
#endif
// Look-ahead jobs (la.sh_prog != null): env i gets one this launch iff a reset table is on the device (serial > 0), its shadow record
// does not yet hold that table's complete reset, and its episode is la.defer steps old -- the first la.maxjobs such envs in index
// order (an env keeps its job from launch to launch until its shadow is complete) -> la.jobs / q[4].  Device state only.
struct LaSched { const int *sh_prog, *sh_serial, *tab_serial; const uint8_t *init_mask; int *jobs; int maxjobs, defer, total, total_init, eplen_off; };
// The scheduler proper, run by ONE wave (tid = lane): keys[n] and hist[257] are LDS scratch of the caller; counts[0..2] receive the
// number of multi-wave envs, of the others, and of look-ahead jobs.  WSYNC orders the wave's LDS phases (a one-wave workgroup's
// __syncthreads in k_schedule; a compiler barrier when a wave of a bigger workgroup runs it).
template <class WS> __device__ __forceinline__ void sched_run(const int tid, int *keys, int *hist, const int *cost, int *order, int n, const int *state, int stride, int niter_off, int mw_k,
                                                              int use_mw, int *mworder, int *counts, const LaSched &la, WS wsync) {
  for (int b = tid; b < 257; b += 64) hist[b] = 0;
  int base = 0, lm = 0, jbase = 0;
  for (int i0 = 0; i0 < n; i0 += 64 * FSIM_SCHED_U) {
    int cv[FSIM_SCHED_U], v[FSIM_SCHED_U];
    bool jb[FSIM_SCHED_U];
#pragma unroll
    for (int u = 0; u < FSIM_SCHED_U; u++) {
      const int i = i0 + 64 * u + tid;
      cv[u] = i < n ? cost[i] : -1;
      v[u] = (use_mw && i < n) ? state[(size_t)i * stride + niter_off] : -0x7fffffff;
      jb[u] = false;
      if (la.sh_prog && i < n) {
        const int ts = la.tab_serial[i], done = (la.init_mask && la.init_mask[i]) ? la.total_init : la.total;
        jb[u] = ts > 0 && (la.sh_serial[i] != ts || la.sh_prog[i] < done) && state[(size_t)i * stride + la.eplen_off] >= la.defer;
      }
    }
#pragma unroll
    for (int u = 0; u < FSIM_SCHED_U; u++) {
      const int i = i0 + 64 * u + tid;
      const bool take = i < n && v[u] >= mw_k && cv[u] != -1;
      const unsigned long long mask = __ballot(take);
      const int idx = base + __popcll(mask & ((1ull << tid) - 1ull));
      if (take) mworder[idx] = i;
      base += __popcll(mask);
      if (i < n) keys[i] = take ? FSIM_SCHED_SELECTED : cv[u];
      if (i < n && !take && cv[u] >= 0) lm = max(lm, cv[u] & 0x3fffffff);
      if (la.sh_prog) {
        const unsigned long long jm = __ballot(jb[u]);
        const int jx = jbase + __popcll(jm & ((1ull << tid) - 1ull));
        if (jb[u] && jx < la.maxjobs) la.jobs[jx] = i;
        jbase += __popcll(jm);
      }
    }
  }
  const int nsel = base;
  if (tid == 0) { counts[0] = nsel; counts[1] = n - nsel; counts[2] = min(jbase, la.maxjobs); }
  for (int o = 32; o > 0; o >>= 1) lm = max(lm, __shfl_xor(lm, o, 64));
  wsync();
  // (a float multiply, not a 64-bit division: the inlined division routines were half of this kernel's code, and the code is never in
  //  the instruction cache when the kernel runs -- it spends its ~0.1 ms fetching instructions; buckets only order the launch)
  const float bscale = 128.0f / ((float)lm + 1.0f);
  auto bucket = [&](int cv) {
    if (cv < 0) return 0;
    const int base = (cv >> 30) & 1 ? 1 : 129;
    return base + 127 - min(127, (int)((float)(cv & 0x3fffffff) * bscale));
  };
  for (int i = tid; i < n; i += 64) { const int cv = keys[i]; if (cv != FSIM_SCHED_SELECTED) atomicAdd(&hist[bucket(cv)], 1); }
  wsync();
  { // exclusive prefix sum over the 257 buckets: five consecutive buckets per lane, wave scan of the lane totals
    int h[5], tot = 0;
    for (int k = 0; k < 5; k++) { const int b = 5 * tid + k; h[k] = b < 257 ? hist[b] : 0; tot += h[k]; }
    int inc = tot;
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o, 64); if (tid >= o) inc += up; }
    int acc = inc - tot;
    for (int k = 0; k < 5; k++) { const int b = 5 * tid + k; if (b < 257) hist[b] = acc; acc += h[k]; }
  }
  wsync();
  for (int i = tid; i < n; i += 64) { const int cv = keys[i]; if (cv != FSIM_SCHED_SELECTED) order[atomicAdd(&hist[bucket(cv)], 1)] = i; }
}
__global__ __launch_bounds__(64) void k_schedule(const int *cost, int *order, int n, const int *state, int stride, int niter_off, int mw_k, int use_mw,
                                                int *mworder, int *q, LaSched la) {
  extern __shared__ int keys[]; // [n]
  __shared__ int hist[257];
  __shared__ int counts[4];
  sched_run((int)threadIdx.x, keys, hist, cost, order, n, state, stride, niter_off, mw_k, use_mw, mworder, counts, la, [] { __syncthreads(); });
  __syncthreads();
  // (q[2], q[3], q[5]: heads of the bundle / multi-wave / look-ahead work queues)
  if (threadIdx.x == 0) { q[0] = counts[0]; q[1] = counts[1]; q[2] = 0; q[3] = 0; q[4] = counts[2]; q[5] = 0; }
}

// strided gather/scatter between the AoS env records and caller [n, dim] arrays
__global__ void k_copy_field(float *state, int stride, int off, int dim, float *ext, int n_envs, int to_state) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_envs * dim) return;
  int e = i / dim, k = i % dim;
  if (to_state) state[(size_t)e * stride + off + k] = ext[i];
  else ext[i] = state[(size_t)e * stride + off + k];
}
// Cursor agent: [pos0 pos1 sel0 sel1] <-> the EC_* block of the env record (selection stored as int part + 1)
__global__ void k_copy_cursor(float *state, int stride, int off, float *ext, int n_envs, int to_state) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  float *ec = state + (size_t)e * stride + off;
  int *eci = reinterpret_cast<int *>(ec);
  float *x = ext + (size_t)e * 8;
  if (to_state) {
    for (int k = 0; k < 6; k++) { ec[EC_POS + k] = x[k]; ec[EC_XPOS + k] = x[k]; }
    eci[EC_SEL] = (int)x[6]; eci[EC_SEL + 1] = (int)x[7];
  } else {
    for (int k = 0; k < 6; k++) x[k] = ec[EC_POS + k];
    x[6] = (float)eci[EC_SEL]; x[7] = (float)eci[EC_SEL + 1];
  }
}
// geom_contype/conaffinity live per colliding geom; the C-ABI speaks original geom ids
__global__ void k_copy_geommask(float *state, int stride, int off, int ncg, const int *cg_orig, int ngeom, int *ext, int n_envs, int to_state) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_envs * ncg) return;
  int e = i / ncg, k = i % ncg;
  int *st = reinterpret_cast<int *>(state + (size_t)e * stride + off) + k;
  if (to_state) *st = ext[(size_t)e * ngeom + cg_orig[k]];
  else ext[(size_t)e * ngeom + cg_orig[k]] = *st;
}
// body poses of ORIGINAL bodies from reduced-body poses
__global__ void k_expand_bodies(const float *aux, int auxstride, int nv, int nr, int nbody, const int *body_red, const float *relpos,
                                const float *relquat, float *xpos, float *xquat, int n_envs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_envs * nbody) return;
  int e = i / nbody, b = i % nbody;
  const float *a = aux + (size_t)e * auxstride;
  int r = body_red[b];
  V3 p = ldv3(a + nv + 3 * r);
  Q4 q = ldq(a + nv + 3 * nr + 4 * r);
  V3 pos = p + qrot(q, ldv3(relpos + 3 * b));
  Q4 qq = qmul(q, ldq(relquat + 4 * b));
  if (xpos) stv3(xpos + 3 * (size_t)i, pos);
  if (xquat) stq(xquat + 4 * (size_t)i, qq);
}

// ------------------------------------------------------------------------------------------ host
struct BlobEnt { char name[48]; int32_t code; int32_t pad; int64_t count; int64_t off; };

// ---- kernel variants: the generic kernels (run-time layout, any model) and the specialised ones of fsim_spec.hpp
typedef void (*PhysicsFn)(const DModel *, const Layout *, KParams, float *, float *);
typedef void (*EnvStepFn)(const DModel *, const Layout *, KParams, StepArgs, const int *, const int *);
typedef void (*EnvStepXFn)(const DModel *, const Layout *, const Layout *, KParams, StepArgs, const int *, const int *, int *, int *);
#define FSIM_MW_NW 4 // waves per env of the multi-wave kernels = one-wave envs per bundle
#define FSIM_LA_MAXJOBS 512 // look-ahead jobs per launch, at most
#define FSIM_OVF_CAP 1024    // envs of one step launch the overflow re-step lists, at most (the others keep their sticky report)
struct KernelSet { const char *name; PhysicsFn physics; EnvStepFn env_step; PhysicsFn physics_mw; EnvStepFn env_step_mw; EnvStepXFn env_step_x; };
enum { MW_OFF = 0, MW_RULE = 1, MW_ALL = 2 }; // fsim::mw_mode

struct fsim {
  int device = 0, n_envs = 0;
  bool has_ik = false; // the model carries the IK chain table (Sawyer)
  bool has_mesh = false; // some colliding geom is a convex mesh (the PLAIN specialisations are compiled without those branches)
  hipStream_t xfer = nullptr; // host -> device table uploads (must not queue behind a running step kernel)
  hipStream_t stream = nullptr;
  // multi-wave kernels (fsim_solver.hpp): MW_OFF one-wave kernel only, MW_RULE = a step is ONE launch of k_env_step_x -- the envs
  // the scheduler's rule picks get four waves, the others ride in bundles of four --, MW_ALL = every env gets four waves in every
  // launch (development and tests).  Decided ONCE, at fsim_create, from fsim_config_t::multi_wave (auto: from n_envs and the device's CU
  // count alone) and reported by fsim_step_kernel(): an env's arithmetic depends on the mode, so nothing else may change it.
  int mw_mode = MW_RULE, mw_k = 150;
  char step_kernel[96] = "";
  Layout ly_mw{};
  Layout *d_ly_mw = nullptr;
  int lds_bytes_mw = 0, lds_bytes_x = 0;
  int *d_defer = nullptr; // k_env_step_x: [x_grid][n_envs] deferred resets of each workgroup's multi-wave phase
  int *d_mworder = nullptr, *d_mwn = nullptr; // q: [0] multi-wave envs of the launch, [1] the others, [2] / [3] heads of the bundle / multi-wave queues, [4] look-ahead jobs, [5] head of their queue
  int x_resident = 0;                         // bundle workgroups that can be resident at once (2 per CU)
  int x_grid = 0;                             // workgroups of a k_env_step_x launch
  // contact-overflow re-step (StepArgs::prev ...): an env whose step needed more contact slots than the step kernel's LDS image holds is
  // stepped again from its pre-step record by a four-wave team with a 64-slot layout (the team has the LDS of four envs), at fsim_sync
  bool redo_on = false, redo_armed = false;
  float *d_prev = nullptr;
  int *d_ovf_list = nullptr;
  // the re-step LADDER: rung 0 for models on 48 slots = 64 slots on the generic four-wave kernel, then (round 6) 128 slots on the generic
  // one-wave kernel with two slots per lane for the envs that overflow 64 as well; models on 64 slots have the one rung 64 -> 128
  // (round 6, last rung: 512 slots -- eight per lane -- on the generic one-wave kernel `generic8`, which also takes islands of more than 64 dofs through
  //  the LDS-resident factorisation fs_chol_all_lds: the furniture whose reset starts with the planks inside each other -- 250 to 370 contacts in the
  //  first substeps, every part in one island)
  int redo_rungs = 0;
  Layout ly_r[3]{};
  Layout *d_ly_r[3] = {nullptr, nullptr, nullptr};
  int lds_bytes_r[3] = {0, 0, 0}, redo_block[3] = {0, 0, 0}, redo_slots[3] = {0, 0, 0};
  EnvStepFn redo_kernel[3] = {nullptr, nullptr, nullptr};
  int *d_ovf_list2 = nullptr;                 // envs a re-step rung lists for the next one
  int last_do_step = 0;
  int64_t n_redone = 0;
  struct { const float *action; void *obs; float *reward; uint8_t *done; int32_t *info; } last{};
  DModel m{};
  Layout ly{};
  fsim_config_t cfg{};
  EnvCfg ecfg{};
  EnvCfg ecfg_sent{};          // what d_ecfg holds
  EnvCfg *d_ecfg = nullptr;    // device copy of ecfg (StepArgs::cfg_dev), refreshed by launch_env when ecfg has changed
  void *d_model = nullptr;
  DModel *d_m = nullptr;
  Layout *d_ly = nullptr;
  float *d_state = nullptr, *d_aux = nullptr, *d_tab_parts = nullptr, *d_tab_noise = nullptr, *d_tab_attach = nullptr;
  int *d_cost = nullptr, *d_order = nullptr; // longest-job-first scheduling (k_schedule)
  int *d_nreset = nullptr, *h_nreset = nullptr; // envs that consumed their reset table in the last step launch (device counter, pinned host copy)
  float *d_init = nullptr;       // set_init_qpos: [n][nq + nv] state the masked envs' resets start from
  uint8_t *d_init_mask = nullptr;
  std::vector<uint8_t> h_init_mask; // host mirror of d_init_mask: "is any init state set" is asked by fsim_set_preassembled
  float *d_dense = nullptr; // dense-reward tables: DC_WORDS coefficients, then nsub rows of DS_WORDS
  int *d_pre = nullptr;     // pre-assembled starts: [n_pre][3] (EnvCfg::pre_tab)
  bool lpt = true;
  int n_noise = 0;
  int auxstride = 0, lds_bytes = 0;
  std::vector<char> blob;
  KernelSet ks{};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double acc_ms = 0;
  int acc_n = 0;
  bool timing_pending = false, timing = false; // HIP-event timing of the step kernel: off until fsim_kernel_time_ms is first called
  int nbody = 0, ngeom = 0;
  // ---- look-ahead reset (env_shadow_job): waves of the step launches that have run out of envs compute every env's NEXT reset, a
  // chunk per launch, from the reset table the host has already uploaded, into shadow records the terminal step copies in.  All of
  // it on the handle's stream, decided on the device (k_schedule lists the jobs); the host only numbers the tables.
  bool la_on = false;
  float *d_sh_state = nullptr;
  void *d_sh_obs = nullptr;
  int *d_sh_prog = nullptr, *d_sh_serial = nullptr, *d_tab_serial = nullptr, *d_sh_jobs = nullptr;
  std::vector<int> h_tab_serial;       // serial number of each env's reset table (bumped with every upload / change of what a reset starts from)
  int la_jobs = 0, la_defer = 0, la_chunk = 0; // jobs per launch, steps into the episode before an env's shadow is started, reset units per job
};

static void la_policy(fsim *s);
static int la_new_tables(fsim *s, const uint8_t *mask);

struct Arena {
  std::vector<char> host;
  std::vector<std::pair<const void **, size_t>> fix;
  template <class T> void add(const T **slot, const std::vector<T> &v) {
    size_t off = (host.size() + 15) / 16 * 16;
    host.resize(off + v.size() * sizeof(T) + 16);
    if (!v.empty()) memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
    fix.push_back({reinterpret_cast<const void **>(slot), off});
  }
};

static const BlobEnt *blob_entry(const std::vector<char> &blob, const char *name) {
  int32_t n;
  memcpy(&n, blob.data() + 12, 4);
  const BlobEnt *e = reinterpret_cast<const BlobEnt *>(blob.data() + 16);
  for (int i = 0; i < n; i++)
    if (strncmp(e[i].name, name, 48) == 0) return &e[i];
  return nullptr;
}
static bool blob_f(const std::vector<char> &blob, const char *name, std::vector<float> &out) {
  const BlobEnt *e = blob_entry(blob, name);
  if (!e || e->code != 0 || (size_t)e->off + (size_t)e->count * 8 > blob.size()) { snprintf(g_err, sizeof g_err, "model blob: bad/missing f64 entry '%s'", name); return false; }
  const double *p = reinterpret_cast<const double *>(blob.data() + e->off);
  out.resize(e->count);
  for (int64_t i = 0; i < e->count; i++) out[i] = (float)p[i];
  return true;
}
static bool blob_i(const std::vector<char> &blob, const char *name, std::vector<int> &out) {
  const BlobEnt *e = blob_entry(blob, name);
  if (!e || e->code != 1 || (size_t)e->off + (size_t)e->count * 4 > blob.size()) { snprintf(g_err, sizeof g_err, "model blob: bad/missing i32 entry '%s'", name); return false; }
  const int32_t *p = reinterpret_cast<const int32_t *>(blob.data() + e->off);
  out.assign(p, p + e->count);
  return true;
}

extern "C" void fsim_default_config(fsim_config_t *c) {
  memset(c, 0, sizeof *c);
  c->control_type = 0; c->n_substeps = 50; c->max_episode_steps = 2000; c->discrete_grip = 1; c->rescale_actions = 1;
  c->auto_align = 1; c->num_connect_steps = 0; c->auto_reset = 1; c->solver_iterations = 100; c->reset_robot_after_attach = 0;
  c->solver_tolerance = 1e-6f;
  c->multi_wave = 0; c->lookahead_reset = 1; c->overflow_restep = 1;
  c->alignment_pos_dist = 0.1f; c->alignment_rot_dist_up = 0.9f; c->alignment_rot_dist_forward = 0.9f; c->alignment_project_dist = 0.3f;
  c->ctrl_penalty_coef = 1e-3f; c->unstable_penalty_coef = 100.f; c->success_reward = 100.f; c->touch_reward = 10.f; c->pick_reward = 100.f;
  c->furn_xyz_rand = 0.02f; c->furn_rot_rand = 3.f; c->agent_xyz_rand = 0.001f;
  c->move_speed = 0.1f; c->rotate_speed = 22.5f; c->cursor_boundary = 1.5f;
}

#define LF(field, name) do { std::vector<float> v_; if (!blob_f(s->blob, name, v_)) return FSIM_EINVAL; ar.add(&s->m.field, v_); } while (0)
#define LI(field, name) do { std::vector<int> v_; if (!blob_i(s->blob, name, v_)) return FSIM_EINVAL; ar.add(&s->m.field, v_); } while (0)

static int build_model(fsim *s) {
  Arena ar;
  std::vector<int> dims, rdims;
  if (!blob_i(s->blob, "dims", dims) || !blob_i(s->blob, "rdims", rdims)) return FSIM_EINVAL;
  std::vector<float> opt, trace;
  if (!blob_f(s->blob, "opt", opt) || !blob_f(s->blob, "trace_M0", trace)) return FSIM_EINVAL;
  DModel &m = s->m;
  m.nq = dims[0]; m.nv = dims[1]; m.nu = dims[2]; m.nbody = dims[3]; m.ngeom = dims[5]; m.nsite = dims[6]; m.neq = dims[7];
  m.nM = dims[9]; m.nparts = dims[10]; m.narm = dims[12]; m.nconn = dims[13]; m.agent = dims[15];
  m.nr = rdims[0]; m.ntree = rdims[1]; m.ncg = rdims[2]; m.ncp = rdims[3]; m.maxdepth = rdims[4];
  s->nbody = m.nbody; s->ngeom = m.ngeom;
  m.timestep = opt[0]; m.gravity[0] = opt[1]; m.gravity[1] = opt[2]; m.gravity[2] = opt[3]; m.impratio = opt[4];
  m.meaninertia_scale = 1.0f / fmaxf(trace[0], 1e-15f);
  if (m.nv > 128) FAIL(FSIM_ENOMEM, "nv=%d > 128: the Newton factorisation maps the dofs on two passes of 64 solver lanes", m.nv);
  if (m.nr > 32) FAIL(FSIM_ENOMEM, "more than 31 moving bodies (the world body + 31: subtree masks are one 32-bit word)");
  if (m.ncp > 65535) FAIL(FSIM_ENOMEM, "too many candidate pairs");
  if (m.ncg > 255) FAIL(FSIM_ENOMEM, "more than 255 colliding geoms (broadphase records hold 8-bit geom indices)");
  LI(r_parent, "r_parent"); LI(r_jtype, "r_jtype"); LI(r_qposadr, "r_qposadr"); LI(r_dofadr, "r_dofadr"); LI(r_dofnum, "r_dofnum");
  LI(r_depth, "r_depth"); LI(r_tree, "r_tree"); LI(r_chainadr, "r_chainadr"); LI(r_chainlen, "r_chainlen"); LI(r_ancmask, "r_ancmask");
  LF(r_pos, "r_pos"); LF(r_quat, "r_quat"); LF(r_jaxis, "r_jaxis"); LF(r_jpos, "r_jpos"); LF(r_mass, "r_mass"); LF(r_ipos, "r_ipos");
  LF(r_inertia, "r_inertia"); LI(chain_dofs, "chain_dofs");
  LI(tree_dofadr, "tree_dofadr"); LI(tree_dofnum, "tree_dofnum"); LI(tree_bodyadr, "tree_bodyadr"); LI(tree_bodynum, "tree_bodynum");
  LI(dof_parent, "dof_parentid"); LI(dof_Madr, "dof_Madr"); LI(dof_rbody, "dof_rbody"); LI(dof_tree, "dof_tree"); LI(dof_qposadr, "dof_qposadr");
  LI(M_i, "M_i"); LI(M_j, "M_j");
  LF(dof_armature, "dof_armature"); LF(dof_damping, "dof_damping"); LF(dof_invweight0, "dof_invweight0");
  LI(lim_dof, "lim_dof"); LF(lim_range, "lim_range"); LF(lim_margin, "lim_margin"); LF(lim_solref, "lim_solref"); LF(lim_solimp, "lim_solimp");
  { std::vector<int> v; blob_i(s->blob, "lim_dof", v); m.nlim = (int)v.size(); }
  if (2 * m.nlim > 64) FAIL(FSIM_ENOMEM, "%d limited joints: the Newton solve keeps one joint-limit record per lane (2 per joint, <= 64)", m.nlim);
  LI(cg_body, "cg_body"); LI(cg_type, "cg_type"); LI(cg_condim, "cg_condim"); LI(cg_partid, "cg_partid"); LI(cg_fingerrole, "cg_fingerrole");
  LI(cg_isfloor, "cg_isfloor"); LI(cg_isrobot, "cg_isrobot"); LI(cg_ispartcol, "cg_ispartcol"); LI(cg_orig, "cg_orig");
  LI(cg_contype0, "cg_contype0"); LI(cg_conaffinity0, "cg_conaffinity0");
  LI(cg_cursor, "cg_cursor"); LI(cg_namepart, "cg_namepart"); LF(cursor_pos0, "cursor_pos0");
  LF(cg_pos, "cg_pos"); LF(cg_mat, "cg_mat"); LF(cg_size, "cg_size"); LF(cg_rbound, "cg_rbound"); LF(cg_friction, "cg_friction");
  LF(cg_solref, "cg_solref"); LF(cg_solimp, "cg_solimp"); LF(cg_margin, "cg_margin"); LF(cg_gap, "cg_gap"); LF(cg_solmix, "cg_solmix");
  LF(cg_invweight, "cg_invweight");
  LI(cp, "cp");
  {
    std::vector<int> cp_, ty_;
    std::vector<float> mg_, gp_, rb_, sz_;
    blob_i(s->blob, "cp", cp_); blob_i(s->blob, "cg_type", ty_); blob_f(s->blob, "cg_margin", mg_); blob_f(s->blob, "cg_gap", gp_);
    blob_f(s->blob, "cg_rbound", rb_); blob_f(s->blob, "cg_size", sz_);
    // convex-mesh colliders (tables compiled since round 5; absent = none)
    std::vector<int> ma_, mn_;
    std::vector<float> mv_;
    blob_i(s->blob, "cg_meshadr", ma_); blob_i(s->blob, "cg_meshnum", mn_); blob_f(s->blob, "mesh_vert", mv_);
    if (mv_.empty()) mv_.assign(4, 0.0f);
    if (mv_.size() / 3 > 65535) FAIL(FSIM_ENOMEM, "more than 65535 hull vertices of mesh colliders");
    for (int v : mn_) {
      if (v > 32767) FAIL(FSIM_ENOMEM, "a mesh collider's hull has %d vertices (the packed geom word holds 15 bits)", v);
      if (v > 0) s->has_mesh = true;
    }
    ar.add(&s->m.mesh_vert, mv_);
    std::vector<float> rec((size_t)16 * std::max(m.ncp, 1), 0.0f);
    for (int p = 0; p < m.ncp; p++) {
      int g1 = cp_[3 * p], g2 = cp_[3 * p + 1], w[4] = {g1, g2, cp_[3 * p + 2], ty_[g1] | (ty_[g2] << 8)};
      float *r = rec.data() + 16 * p;
      memcpy(r, w, 16);
      r[4] = std::max(mg_[g1], mg_[g2]); r[5] = std::max(gp_[g1], gp_[g2]); r[6] = rb_[g1]; r[7] = rb_[g2];
      for (int k = 0; k < 3; k++) { r[8 + k] = sz_[3 * g1 + k]; r[12 + k] = sz_[3 * g2 + k]; }
      const int gs[2] = {g1, g2};
      for (int q = 0; q < 2; q++) {
        const int g = gs[q];
        int packed = 0;
        if (ty_[g] == GT_MESH) {
          if ((size_t)g >= ma_.size() || ma_[g] < 0 || mn_[g] <= 0) FAIL(FSIM_EINVAL, "mesh collider without hull vertices (model compiled before round 5?)");
          packed = ma_[g] | (mn_[g] << 16);
        }
        memcpy(r + 11 + 4 * q, &packed, 4);
      }
    }
    ar.add(&s->m.pair_rec, rec);
    std::vector<int> bp((size_t)2 * std::max(m.ncp, 1), 0);
    for (int p = 0; p < m.ncp; p++) {
      int g1 = cp_[3 * p], g2 = cp_[3 * p + 1];
      const bool plane = ty_[g1] == GT_PLANE;
      const float margin = std::max(mg_[g1], mg_[g2]), bound = plane ? rb_[g2] + margin : rb_[g1] + rb_[g2] + margin;
      bp[2 * p] = g1 | (g2 << 8) | ((plane ? 1 : 0) << 16);
      memcpy(&bp[2 * p + 1], &bound, 4);
    }
    ar.add(&s->m.pair_bp, bp);
  }
  LI(s_body, "s_body"); LF(s_pos, "s_pos"); LF(s_quat, "s_quat");
  {
    // actuator -> dof / qpos addresses
    std::vector<int> aj, jd, jq;
    if (!blob_i(s->blob, "actuator_jntid", aj) || !blob_i(s->blob, "jnt_dofadr", jd) || !blob_i(s->blob, "jnt_qposadr", jq)) return FSIM_EINVAL;
    std::vector<int> ad(aj.size()), aq(aj.size());
    for (size_t i = 0; i < aj.size(); i++) { ad[i] = jd[aj[i]]; aq[i] = jq[aj[i]]; }
    ar.add(&m.act_dof, ad); ar.add(&m.act_qpos, aq);
  }
  LI(act_ctrllimited, "actuator_ctrllimited"); LI(act_forcelimited, "actuator_forcelimited");
  LF(act_gain, "actuator_gain"); LF(act_bias, "actuator_bias"); LF(act_ctrlrange, "actuator_ctrlrange"); LF(act_forcerange, "actuator_forcerange");
  LF(act_gear, "actuator_gear"); LF(ctrl_bias, "ctrl_bias"); LF(ctrl_weight, "ctrl_weight");
  LI(eq_rbody1, "eq_rbody1"); LI(eq_rbody2, "eq_rbody2"); LI(eq_part1, "eq_part1"); LI(eq_part2, "eq_part2");
  LF(eq_solref, "eq_solref"); LF(eq_solimp, "eq_solimp"); LF(eq_invweight, "eq_invweight"); LF(eq_data0, "eq_data0"); LI(eq_active0, "eq_active0");
  LI(part_rbody, "part_rbody"); LI(part_qposadr, "part_qposadr"); LI(part_dofadr, "part_dofadr"); LI(body_red, "body_red");
  LF(body_relpos, "body_relpos"); LF(body_relquat, "body_relquat"); LF(part_mass, "part_mass");
  LI(arm_qposadr, "arm_qposadr"); LI(arm_dofadr, "arm_dofadr"); LI(grip_qposadr, "grip_qposadr"); LI(grip_dofadr, "grip_dofadr");
  LI(eef_siteid, "eef_siteid"); LI(hand_body, "hand_bodyid");
  LF(arm_initqpos, "arm_initqpos"); LF(grip_initqpos, "grip_initqpos"); LF(qpos0, "qpos0");
  { // IK chain table(s): IKT_ARM floats per arm + IKT_TAIL (fsim_ik.hpp); models without one get zeros and refuse control_type 7 / 8
    std::vector<float> v_; blob_f(s->blob, "ik_table", v_);
    const size_t want = (size_t)IKT_ARM * std::max(m.narm, 1) + IKT_TAIL;
    s->has_ik = m.narm > 0 && v_.size() == want;
    if (!s->has_ik) v_.assign(want, 0.0f);
    ar.add(&s->m.ik_tab, v_);
  }
  { std::vector<int> v; blob_i(s->blob, "arm_qposadr", v); m.narmj = (int)v.size(); blob_i(s->blob, "grip_qposadr", v); m.ngripj = (int)v.size(); }
  LI(conn_siteid, "conn_siteid"); LI(conn_partid, "conn_partid"); LI(conn_keya, "conn_keya"); LI(conn_keyb, "conn_keyb"); LI(conn_nangle, "conn_nangle");
  LF(conn_angles, "conn_angles");
  LI(part_site_adr, "part_site_adr"); LI(part_site_num, "part_site_num"); LI(part_sites, "part_sites");
  HIPCHK(hipMalloc(&s->d_model, ar.host.size()));
  HIPCHK(hipMemcpy(s->d_model, ar.host.data(), ar.host.size(), hipMemcpyHostToDevice));
  for (auto &f : ar.fix) *f.first = static_cast<char *>(s->d_model) + f.second;
  return FSIM_OK;
}

static LayoutIn layout_in(const fsim *s, int ncon_max) {
  const DModel &m = s->m;
  LayoutIn in{};
  in.nq = m.nq; in.nv = m.nv; in.nu = m.nu; in.nr = m.nr; in.ntree = m.ntree; in.ncg = m.ncg; in.nparts = m.nparts; in.neq = m.neq;
  in.nlim = m.nlim; in.nM = m.nM;
  std::vector<int> tn, ca, cl;
  blob_i(s->blob, "tree_dofnum", tn); blob_i(s->blob, "r_chainadr", ca); blob_i(s->blob, "r_chainlen", cl);
  for (int n_ : tn) in.Mwords += n_ * (n_ + 1) / 2;
  for (size_t b = 0; b < ca.size(); b++) in.nchain = std::max(in.nchain, ca[b] + cl[b]);
  in.env_words = E_FIXED_WORDS + m.nparts + env_extra_words(m, s->cfg);
  in.eik_rel = s->cfg.dense_reward ? ED_WORDS : 0;
  in.ncon_max = ncon_max;
  // furniture with many long parts (bookcase planks lying side by side): more part-part pairs survive the broadphase
  in.maxsurv = ncon_max > 128 ? 1024 : (ncon_max > 64 ? 192 : (m.nparts > 8 ? 128 : FSIM_MAXSURV));
  return in;
}

static bool same_dims(const Dims &a, const Dims &b) { return memcmp(&a, &b, sizeof(Dims)) == 0; }
static bool same_in(const LayoutIn &a, const LayoutIn &b) { return memcmp(&a, &b, sizeof(LayoutIn)) == 0; }
static KernelSet pick_kernels(const Dims &d, const LayoutIn &in, bool plain_cfg) { // plain_cfg: no controller / IK / dense reward, no convex-mesh collider (SpecCtx::PLAIN compiles those branches out)
  if (!getenv("FSIM_GENERIC")) { // (development / tests: force the generic kernels)
#define FS_TRY(S) { const Dims sd = S::D; const LayoutIn si = S::in; if (same_dims(d, sd) && same_in(in, si) && (plain_cfg || !S::plain)) return KernelSet{S::name, k_physics<SpecCtx<S>>, k_env_step<SpecCtx<S>>, k_physics<SpecCtx<S, FSIM_MW_NW>>, k_env_step<SpecCtx<S, FSIM_MW_NW>>, k_env_step_x<SpecCtx<S, FSIM_MW_NW>, SpecCtx<S, 1, true>>}; }
    FSIM_SPEC_LIST(FS_TRY)
#undef FS_TRY
  }
  // (more than 64 contact slots: two slot sets per lane in the Newton solve -- one-wave kernels only)
  if (in.ncon_max > 128) return KernelSet{"generic8", k_physics<GenCtxT<1, false, 8>>, k_env_step<GenCtxT<1, false, 8>>, nullptr, nullptr, nullptr};
  if (in.ncon_max > 64) return KernelSet{"generic2", k_physics<GenCtxT<1, false, 2>>, k_env_step<GenCtxT<1, false, 2>>, nullptr, nullptr, nullptr};
  return KernelSet{"generic", k_physics<GenCtx>, k_env_step<GenCtx>, k_physics<GenCtxT<FSIM_MW_NW>>, k_env_step<GenCtxT<FSIM_MW_NW>>, k_env_step_x<GenCtxT<FSIM_MW_NW>, GenCtxT<1, true>>};
}

extern "C" int fsim_create(const void *model_blob, size_t nbytes, int n_envs, int device, const fsim_config_t *cfg, fsim_t **out) {
  if (!model_blob || nbytes < 64 || n_envs <= 0 || !out) FAIL(FSIM_EINVAL, "fsim_create: bad arguments");
  if (memcmp(model_blob, "FSIMBLOB", 8) != 0) FAIL(FSIM_EINVAL, "fsim_create: not an FSIMBLOB");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FAIL(FSIM_ENODEV, "fsim_create: no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) FAIL(FSIM_EINVAL, "fsim_create: device %d out of range (%d devices)", device, ndev);
  HIPCHK(hipSetDevice(device));
  fsim *s = new fsim();
  s->device = device; s->n_envs = n_envs;
  s->blob.assign(static_cast<const char *>(model_blob), static_cast<const char *>(model_blob) + nbytes);
  if (cfg) s->cfg = *cfg; else fsim_default_config(&s->cfg);
  int rc = build_model(s);
  if (rc) { delete s; return rc; }
  // contact slots: 48 by default (Sawyer + table_lack holds 21 at rest); 64 for furniture with eight or nine parts or many collision
  // primitives (>= 34 colliding geoms: chairs, table_torsby; >= 31 with five parts and more: table_klubbo_0740 passes through 50-62
  // contacts while its reset settles the overlapping parts, counted with the oracle); 128 -- two slots per lane in the Newton solve,
  // Ctx::NS == 2 -- for ten parts and more (4 part-floor contacts per part at rest plus the part-part ones)
  int ncon_max = s->m.nparts >= 10 ? 128 : ((s->m.nparts >= 8 || s->m.ncg >= 34 || (s->m.nparts >= 5 && s->m.ncg >= 31)) ? 64 : 48);
  if (const char *e = getenv("FSIM_NCON_MAX")) ncon_max = atoi(e);
  if (ncon_max < 8 || ncon_max > 512) { delete s; FAIL(FSIM_EINVAL, "FSIM_NCON_MAX must be in [8, 512] (the Newton solve keeps one, two or eight contact slots per lane)"); }
  if (s->cfg.dense_reward && s->m.agent != 0) { delete s; FAIL(FSIM_EINVAL, "dense_reward exists for the Sawyer agent only (FurnitureSawyerDenseRewardEnv)"); }
  if ((s->cfg.control_type == 7 || s->cfg.control_type == 8) && (s->m.agent == 2 || !s->has_ik)) { delete s; FAIL(FSIM_EINVAL, "control_type 7 / 8 (ik / ik_quaternion) is built for the Sawyer and Baxter agents, on a model compiled with the IK chain table"); }
  if (s->cfg.control_type == 1 || s->cfg.control_type < 0 || s->cfg.control_type > 8) { delete s; FAIL(FSIM_EINVAL, "control_type %d: 0 (impedance; also the reference's 'torque', which is the impedance flow on the motor-actuated model), 2..6 (arm controllers), 7 (ik) and 8 (ik_quaternion) are built", cfg ? cfg->control_type : 0); }
  if (env_controller_kind(s->cfg)) {
    if (s->m.agent != 0 || s->cfg.dense_reward) { delete s; FAIL(FSIM_EINVAL, "arm controllers (control_type 2..6) are built for the Sawyer agent, sparse reward"); }
    std::vector<float> ag; blob_f(s->blob, "actuator_gain", ag);
    if (s->m.nu != 9 || ag.size() != 9 || ag[0] != 1.0f) { delete s; FAIL(FSIM_EINVAL, "arm controllers need the motor-actuated model (compiled with a torque-level control_type, robot_torque.xml)"); }
  }
  if (s->m.nr > 32) { int nr_ = s->m.nr; delete s; FAIL(FSIM_EINVAL, "model has %d moving bodies; this build supports <= 32 (body bitmasks)", nr_); }
  { // (fs_body_spatial's packed chain words: 25 bits of dofs relative to the tree's first dof)
    std::vector<int> tn_;
    blob_i(s->blob, "tree_dofnum", tn_);
    for (int n_ : tn_) if (n_ > 25) { delete s; FAIL(FSIM_EINVAL, "a kinematic tree of the model has %d dofs; this build supports <= 25 per tree", n_); }
  }
  if (s->m.ntree > 16 || s->m.nv > 128) { int nt_ = s->m.ntree, nv_ = s->m.nv; delete s; FAIL(FSIM_EINVAL, "model has %d trees / %d dofs; this build supports <= 16 trees and <= 128 dofs", nt_, nv_); }
  const LayoutIn lin = layout_in(s, ncon_max);
  s->ly = make_layout(lin);
  s->ks = pick_kernels(s->m, lin, !s->cfg.dense_reward && !env_controller_kind(s->cfg) && s->cfg.control_type != 7 && s->cfg.control_type != 8 && !s->has_mesh);
  s->lds_bytes = s->ly.lds_words * 4;
  if (const char *e = getenv("FSIM_LDS_PAD")) s->lds_bytes += atoi(e); // development: lower the occupancy on purpose
  if (s->lds_bytes > 160 * 1024) { int w = s->ly.lds_words; delete s; FAIL(FSIM_ENOMEM, "per-env LDS image %d words exceeds 160 KiB", w); }
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(s->ks.physics), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(s->ks.env_step), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes));
  HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&s->xfer, hipStreamNonBlocking));
  { // multi-wave kernels
    s->ly_mw = make_layout(lin, FSIM_MW_NW);
    s->lds_bytes_mw = s->ly_mw.lds_words * 4;
    s->lds_bytes_x = std::max(s->lds_bytes_mw, FSIM_MW_NW * 4 * FSIM_BUNDLE_STRIDE(s->ly.lds_words));
    // fsim_config_t::multi_wave: 0 auto, 1 off, 2 rule, 3 all.  (FSIM_MW=0 / 1 / all in the environment overrides it: development)
    int want = s->cfg.multi_wave;
    if (const char *e = getenv("FSIM_MW")) want = !strcmp(e, "all") ? 3 : (atoi(e) ? 2 : 1);
    if (want < 0 || want > 3) { delete s; FAIL(FSIM_EINVAL, "fsim_config_t::multi_wave must be 0 (auto), 1 (off), 2 (rule) or 3 (all)"); }
    if (const char *e = getenv("FSIM_MW_K")) s->mw_k = atoi(e);
    hipDeviceProp_t pr;
    HIPCHK(hipGetDeviceProperties(&pr, device));
    const bool can_all = s->ks.env_step_mw && s->lds_bytes_mw <= 160 * 1024;
    // bundles of four keep the one-wave kernel's occupancy only while two of them fit a CU's LDS; bigger models stay on the one-wave kernel
    const bool can_rule = can_all && s->ks.env_step_x && 2 * s->lds_bytes_x <= 160 * 1024 && !getenv("FSIM_NO_LPT") && n_envs <= 32768; // (the scheduler keeps one key per env in LDS)
    if (want == 3) { if (!can_all) { delete s; FAIL(FSIM_EINVAL, "multi_wave = all: this model has no multi-wave kernel (more than 64 contact slots, or its LDS image does not fit)"); } s->mw_mode = MW_ALL; }
    else if (want == 2) { if (!can_rule) { delete s; FAIL(FSIM_EINVAL, "multi_wave = rule: this model / batch cannot run k_env_step_x (LDS image, contact slots or batch size)"); } s->mw_mode = MW_RULE; }
    else if (want == 1) s->mw_mode = MW_OFF;
    else {
      // auto: a launch with more envs than the chip has wave slots (8 per CU) is bound by throughput, not by its slowest env, and four
      // waves per env only cost slots there (Sawyer + swivel_chair at 8192 envs: 1.03 M env-steps/s on the one-wave kernel, 0.95 M with
      // the rule).  A function of n_envs and the device alone -- callers that split a big batch into slabs say multi_wave = off themselves
      // (bench.py --config 3); nothing about OTHER handles or the process is looked at.
      s->mw_mode = (can_rule && n_envs <= 8 * pr.multiProcessorCount) ? MW_RULE : MW_OFF;
    }
    s->x_resident = 2 * pr.multiProcessorCount;
    // workgroups of a launch: the one-wave envs in bundles of four plus a sixteenth of the batch for four-wave teams (2-4 % of the envs are
    // multi-wave; more of them than teams are served in turn), at most what is resident at once.  Every wave of the launch pays ~92
    // scratch stores at entry (the argument block does not fit the SGPRs of the persistent loop): n/4 + n/8 workgroups measured the same
    // throughput with 20 % more HBM writes (round 4, FSIM_X_GRID sweep: 256 .. 512 flat)
    s->x_grid = std::min((n_envs + FSIM_MW_NW - 1) / FSIM_MW_NW + std::max(1, n_envs / 16), s->x_resident);
    if (const char *e = getenv("FSIM_X_GRID")) s->x_grid = std::max(1, std::min(atoi(e), s->x_resident)); // development: workgroups of a k_env_step_x launch
    // overflow re-step: models on the default 48 slots (the benchmark's LDS budget), stepped again with 64 slots and longer broadphase lists
    if ((ncon_max == 48 || ncon_max == 64 || ncon_max == 128) && s->cfg.overflow_restep != 0 && !getenv("FSIM_NO_OVERFLOW_REDO") && !env_controller_kind(s->cfg)) {
      for (int slots = ncon_max == 48 ? 64 : (ncon_max == 64 ? 128 : 512); slots <= 512 && s->redo_rungs < 3; slots = slots == 64 ? 128 : (slots == 128 ? 512 : 1024)) {
        if (slots == 64 && s->m.nv > 64) continue; // (the four-wave kernel's solve knows one row pass)
        LayoutIn lr = lin;
        const bool big = slots >= 128; // (128 / 512 slots: two / eight per lane, the one-wave kernels of the `generic2` / `generic8` sets; the workgroup has the CU's LDS to itself)
        lr.ncon_max = slots; lr.maxsurv = std::max(lin.maxsurv, slots == 512 ? 1024 : (big ? 192 : 128));
        const Layout ly = make_layout(lr, big ? 1 : FSIM_MW_NW);
        if (ly.lds_words * 4 > 160 * 1024 || ly.stride != s->ly.stride) break;
        const int r = s->redo_rungs++;
        s->ly_r[r] = ly;
        s->lds_bytes_r[r] = ly.lds_words * 4;
        s->redo_kernel[r] = slots == 512 ? static_cast<EnvStepFn>(k_env_step<GenCtxT<1, false, 8>>)
                                         : (big ? static_cast<EnvStepFn>(k_env_step<GenCtxT<1, false, 2>>) : static_cast<EnvStepFn>(k_env_step<GenCtxT<FSIM_MW_NW>>));
        s->redo_block[r] = big ? 64 : 64 * FSIM_MW_NW;
        s->redo_slots[r] = slots;
      }
      s->redo_on = s->redo_rungs > 0;
    }
    snprintf(s->step_kernel, sizeof s->step_kernel, "%s", s->mw_mode == MW_RULE ? "k_env_step_x (multi-wave rule + bundles)" : (s->mw_mode == MW_ALL ? "k_env_step (four waves per env)" : "k_env_step (one wave per env)"));
    if (s->mw_mode) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(s->ks.physics_mw), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes_mw));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(s->ks.env_step_mw), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes_mw));
      if (s->mw_mode == MW_RULE) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(s->ks.env_step_x), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes_x));
    }
  }
  HIPCHK(hipEventCreate(&s->ev0)); HIPCHK(hipEventCreate(&s->ev1));
  size_t sbytes = (size_t)n_envs * s->ly.stride * 4;
  HIPCHK(hipMalloc(&s->d_state, sbytes));
  s->auxstride = s->m.nv + 7 * s->m.nr + 4 + 2 * s->ly.ncon_max;
  HIPCHK(hipMalloc(&s->d_aux, (size_t)n_envs * s->auxstride * 4));
  HIPCHK(hipMemsetAsync(s->d_aux, 0, (size_t)n_envs * s->auxstride * 4, s->stream));
  HIPCHK(hipMalloc(&s->d_cost, (size_t)n_envs * 4)); HIPCHK(hipMalloc(&s->d_order, (size_t)n_envs * 4));
  HIPCHK(hipMemsetAsync(s->d_cost, 0, (size_t)n_envs * 4, s->stream));
  // counter in pinned host memory the kernel increments in place (system-scope atomic, only when an episode ends): no copy or
  // memset kernel queues behind the step kernel's waves
  // (words 1, 2: look-ahead statistics -- resets taken from a shadow record / executed inside a step or reset launch; never cleared)
  HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->h_nreset), 64, hipHostMallocMapped));
  memset(s->h_nreset, 0, 64);
  HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->d_nreset), s->h_nreset, 0));
  s->lpt = !getenv("FSIM_NO_LPT");
  HIPCHK(hipMalloc(&s->d_mworder, (size_t)n_envs * 4)); HIPCHK(hipMalloc(&s->d_mwn, 32));
  HIPCHK(hipMemset(s->d_mwn, 0, 32));
  if (s->mw_mode == MW_RULE) HIPCHK(hipMalloc(&s->d_defer, (size_t)std::max(1, s->x_grid) * n_envs * 4));
  if ((size_t)n_envs * 4 > 150 * 1024) s->lpt = false; // (the scheduler keeps one key per env in LDS)
  else if ((size_t)n_envs * 4 > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_schedule), hipFuncAttributeMaxDynamicSharedMemorySize, n_envs * 4));
  // look-ahead reset: off for what carries state across a reset or draws inside it (arm controllers keep their ramps, the IK controller
  // its target, reset_robot_after_attach resets from the host) and for MW_ALL (every reset then runs on four waves: other bits)
  s->la_on = s->cfg.lookahead_reset != 0 && !getenv("FSIM_NO_LOOKAHEAD") && s->mw_mode != MW_ALL && !env_controller_kind(s->cfg) && s->cfg.control_type == 0 &&
             !s->cfg.reset_robot_after_attach && s->lpt; // (s->lpt: the look-ahead jobs are listed by the scheduler kernel)
  // initial record: qpos0, default masks, weld data, env block zero
  {
    std::vector<float> rec(s->ly.stride, 0.0f), q0, ed;
    std::vector<int> ea, ct, ca;
    blob_f(s->blob, "qpos0", q0); blob_f(s->blob, "eq_data0", ed); blob_i(s->blob, "eq_active0", ea);
    blob_i(s->blob, "cg_contype0", ct); blob_i(s->blob, "cg_conaffinity0", ca);
    memcpy(rec.data() + s->ly.qpos, q0.data(), q0.size() * 4);
    if (!ed.empty()) memcpy(rec.data() + s->ly.eqdata, ed.data(), ed.size() * 4);
    if (!ea.empty()) memcpy(rec.data() + s->ly.eqactive, ea.data(), ea.size() * 4);
    memcpy(rec.data() + s->ly.contype, ct.data(), ct.size() * 4);
    memcpy(rec.data() + s->ly.conaff, ca.data(), ca.size() * 4);
    int *envw = reinterpret_cast<int *>(rec.data() + s->ly.env);
    for (int p = 0; p < s->m.nparts; p++) envw[E_GROUP + p] = p;
    if (int ck = env_controller_kind(s->cfg)) { // Controller.reset() state (arm_controller.py:570-575): identity orientations
      float *K = rec.data() + s->ly.env + E_GROUP + s->m.nparts;
      reinterpret_cast<int *>(K)[EK_KIND] = ck;
      if (ck <= CK_POS) for (int i = 0; i < 3; i++) K[EK_LGO + 4 * i] = K[EK_OINIT + 4 * i] = K[EK_GORI + 4 * i] = 1.0f;
    }
    std::vector<float> all((size_t)n_envs * s->ly.stride);
    for (int e = 0; e < n_envs; e++) memcpy(all.data() + (size_t)e * s->ly.stride, rec.data(), s->ly.stride * 4);
    HIPCHK(hipMemcpy(s->d_state, all.data(), sbytes, hipMemcpyHostToDevice));
  }
#ifdef FSIM_DBG_FILL
  {
    int v[3] = {0, 0, 0};
    if (const char *e = getenv("FSIM_DBG_FILL")) sscanf(e, "%d,%d,%x", &v[0], &v[1], (unsigned *)&v[2]);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_fill), v, sizeof v));
    const Layout &y = s->ly;
    fprintf(stderr, "[fsim dbg] stride %d xpos %d cvel %d cinert(H) %d cdof %d M %d LD %d smooth %d x %d grad %d gpos %d surv %d con %d weld %d lim %d W %d G %d scal %d hmap %d hA %d hP %d pitem %d k_begin %d k_end %d lds_words %d\n",
            y.stride, y.xpos, y.cvel, y.cinert, y.cdof, y.M, y.LD, y.smooth, y.x, y.grad, y.gpos, y.surv, y.con, y.weld, y.lim, y.W, y.G, y.scal, y.hmap, y.hA, y.hP, y.pitem, y.k_begin, y.k_end, y.lds_words);
  }
#endif
  if (getenv("FSIM_VERBOSE")) {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(s->ks.env_step), 64, s->lds_bytes);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(s->ks.env_step));
    fprintf(stderr, "[fsim] kernel=%s lds_bytes=%d stride_words=%d occupancy(blocks/CU)=%d regs=%d localmem(scratch)=%zu ncon_max=%d\n", s->ks.name, s->lds_bytes, s->ly.stride, nb,
            fa.numRegs, (size_t)fa.localSizeBytes, s->ly.ncon_max);
    if (s->mw_mode) {
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(s->ks.env_step_mw), 64 * FSIM_MW_NW, s->lds_bytes_mw);
      hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(s->ks.env_step_mw));
      fprintf(stderr, "[fsim] multi-wave kernel (%d waves per env, mode %d, k %d): lds_bytes=%d occupancy(blocks/CU)=%d regs=%d localmem(scratch)=%zu\n", FSIM_MW_NW, s->mw_mode,
              s->mw_k, s->lds_bytes_mw, nb, fa.numRegs, (size_t)fa.localSizeBytes);
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(s->ks.env_step_x), 64 * FSIM_MW_NW, s->lds_bytes_x);
      hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(s->ks.env_step_x));
      fprintf(stderr, "[fsim] step kernel (multi-wave workgroups + bundles of %d one-wave envs): lds_bytes=%d occupancy(blocks/CU)=%d regs=%d localmem(scratch)=%zu\n", FSIM_MW_NW,
              s->lds_bytes_x, nb, fa.numRegs, (size_t)fa.localSizeBytes);
    }
  }
  env_fill_cfg(s->ecfg, s->cfg, s->m);
  if (s->la_on) {
    HIPCHK(hipMalloc(&s->d_sh_state, sbytes));
    HIPCHK(hipMalloc(&s->d_sh_obs, (size_t)n_envs * s->ecfg.obs_dim * 4 + 16));
    HIPCHK(hipMalloc(&s->d_sh_prog, (size_t)n_envs * 4)); HIPCHK(hipMalloc(&s->d_sh_serial, (size_t)n_envs * 4)); HIPCHK(hipMalloc(&s->d_tab_serial, (size_t)n_envs * 4));
    HIPCHK(hipMemset(s->d_sh_prog, 0, (size_t)n_envs * 4)); HIPCHK(hipMemset(s->d_sh_serial, 0, (size_t)n_envs * 4)); HIPCHK(hipMemset(s->d_tab_serial, 0, (size_t)n_envs * 4));
    HIPCHK(hipMalloc(&s->d_sh_jobs, (size_t)FSIM_LA_MAXJOBS * 4));
    s->h_tab_serial.assign(n_envs, 0);
    la_policy(s);
  }
  if (s->ecfg.obs_dim > 36 * FSIM_NPAIR + 3 * FSIM_NPAIR + 4) { int od = s->ecfg.obs_dim; delete s; FAIL(FSIM_ENOMEM, "obs_dim %d exceeds the LDS staging area of the observation (%d words)", od, 36 * FSIM_NPAIR + 3 * FSIM_NPAIR + 4); }
  { std::vector<int> fl; if (blob_i(s->blob, "flags", fl) && !fl.empty()) s->ecfg.has_recipe = fl[0]; }
  HIPCHK(hipMalloc(&s->d_m, sizeof(DModel))); HIPCHK(hipMalloc(&s->d_ly, sizeof(Layout))); HIPCHK(hipMalloc(&s->d_ly_mw, sizeof(Layout)));
  HIPCHK(hipMalloc(&s->d_ecfg, sizeof(EnvCfg))); memset(&s->ecfg_sent, 0xff, sizeof(EnvCfg)); // (allocated here, not at the first launch: no allocation on the step path)
  HIPCHK(hipMemcpy(s->d_m, &s->m, sizeof(DModel), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->d_ly, &s->ly, sizeof(Layout), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->d_ly_mw, &s->ly_mw, sizeof(Layout), hipMemcpyHostToDevice));
  if (s->redo_on) {
    for (int r = 0; r < s->redo_rungs; r++) {
      HIPCHK(hipMalloc(&s->d_ly_r[r], sizeof(Layout)));
      HIPCHK(hipMemcpy(s->d_ly_r[r], &s->ly_r[r], sizeof(Layout), hipMemcpyHostToDevice));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(s->redo_kernel[r]), hipFuncAttributeMaxDynamicSharedMemorySize, std::max(s->lds_bytes_r[r], std::max(s->lds_bytes_mw, s->lds_bytes))));
    }
    HIPCHK(hipMalloc(&s->d_prev, (size_t)n_envs * s->ly.stride * 4));
    HIPCHK(hipMalloc(&s->d_ovf_list, FSIM_OVF_CAP * 4));
    HIPCHK(hipMalloc(&s->d_ovf_list2, FSIM_OVF_CAP * 4));
  }
  *out = s;
  return FSIM_OK;
}

extern "C" void fsim_destroy(fsim_t *s) {
  if (!s) return;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  hipFree(s->d_sh_state); hipFree(s->d_sh_obs); hipFree(s->d_sh_prog); hipFree(s->d_sh_serial); hipFree(s->d_tab_serial); hipFree(s->d_sh_jobs);
  if (s->xfer) { hipStreamSynchronize(s->xfer); hipStreamDestroy(s->xfer); }
  hipFree(s->d_ly_r[0]); hipFree(s->d_ly_r[1]); hipFree(s->d_ly_r[2]); hipFree(s->d_prev); hipFree(s->d_ovf_list); hipFree(s->d_ovf_list2);
  hipFree(s->d_ly_mw); hipFree(s->d_defer); hipFree(s->d_mworder); hipFree(s->d_mwn); hipFree(s->d_ecfg);
  hipFree(s->d_m); hipFree(s->d_ly); hipFree(s->d_model); hipFree(s->d_state); hipFree(s->d_aux); hipFree(s->d_tab_parts); hipFree(s->d_tab_noise); hipFree(s->d_tab_attach); hipFree(s->d_cost); hipFree(s->d_order); hipFree(s->d_dense); hipFree(s->d_pre); hipFree(s->d_init); hipFree(s->d_init_mask); if (s->h_nreset) hipHostFree(s->h_nreset);
  if (s->ev0) hipEventDestroy(s->ev0);
  if (s->ev1) hipEventDestroy(s->ev1);
  if (s->stream) hipStreamDestroy(s->stream);
  delete s;
}

extern "C" int fsim_dims(const fsim_t *s, int32_t *nq, int32_t *nv, int32_t *nu, int32_t *dof_action, int32_t *obs_dim, int32_t *info_dim,
                         int32_t *stride) {
  if (!s) FAIL(FSIM_EINVAL, "null handle");
  if (nq) *nq = s->m.nq; if (nv) *nv = s->m.nv; if (nu) *nu = s->m.nu;
  if (dof_action) *dof_action = s->ecfg.dof_action; if (obs_dim) *obs_dim = s->ecfg.obs_dim;
  if (info_dim) *info_dim = FSIM_INFO_DIM; if (stride) *stride = s->ly.stride;
  return FSIM_OK;
}
extern "C" int fsim_tables_needed(const fsim_t *s) { return s && s->h_nreset ? *s->h_nreset : 0; }
extern "C" int fsim_max_contacts(const fsim_t *s) { return s ? s->ly.ncon_max : 0; }
extern "C" const char *fsim_kernel_variant(const fsim_t *s) { return s && s->ks.name ? s->ks.name : ""; }
extern "C" const char *fsim_step_kernel(const fsim_t *s) { return s ? s->step_kernel : ""; }
extern "C" int fsim_env_block_words(const fsim_t *s) { return s ? E_FIXED_WORDS + s->m.nparts + env_extra_words(s->m, s->cfg) : 0; }
extern "C" int fsim_stream(fsim_t *s, void **st) { if (!s || !st) FAIL(FSIM_EINVAL, "null"); *st = s->stream; return FSIM_OK; }
static int redo_overflowed(fsim *s);
// After a step launch and before anything else reads or advances the records: wait for it and re-step the envs that dropped contacts
// (a caller that queues two steps without fsim_sync in between gets the wait here: the second step must start from corrected records).
static int settle(fsim *s) {
  if (!s->redo_armed) return FSIM_OK;
  s->redo_armed = false;
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->h_nreset[4] > 0) return redo_overflowed(s);
  return FSIM_OK;
}
extern "C" int fsim_sync(fsim_t *s) {
  if (!s) FAIL(FSIM_EINVAL, "null");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  return settle(s);
}

static KParams kparams(const fsim *s, int nsub, int mode) {
  KParams kp;
  kp.n_envs = s->n_envs; kp.n_substeps = nsub; kp.mode = mode; kp.newton_maxit = s->cfg.solver_iterations; kp.newton_tol = s->cfg.solver_tolerance;
  return kp;
}
static void timing_begin(fsim *s) { hipEventRecord(s->ev0, s->stream); }
static void timing_end(fsim *s) { hipEventRecord(s->ev1, s->stream); s->timing_pending = true; }
static void timing_collect(fsim *s) {
  if (!s->timing_pending) return;
  hipEventSynchronize(s->ev1);
  float ms = 0;
  if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) { s->acc_ms += ms; s->acc_n++; }
  s->timing_pending = false;
}

extern "C" int fsim_physics_step(fsim_t *s, int nsub) {
  if (!s || nsub < 0) FAIL(FSIM_EINVAL, "bad args");
  HIPCHK(hipSetDevice(s->device));
  { int rc_ = settle(s); if (rc_) return rc_; }
  if (s->mw_mode == 2) hipLaunchKernelGGL(s->ks.physics_mw, dim3(s->n_envs), dim3(64 * FSIM_MW_NW), s->lds_bytes_mw, s->stream, s->d_m, s->d_ly_mw, kparams(s, nsub, 0), s->d_state, s->d_aux);
  else hipLaunchKernelGGL(s->ks.physics, dim3(s->n_envs), dim3(64), s->lds_bytes, s->stream, s->d_m, s->d_ly, kparams(s, nsub, 0), s->d_state, s->d_aux);
  HIPCHK(hipGetLastError());
  return FSIM_OK;
}
extern "C" int fsim_physics_forward(fsim_t *s) {
  if (!s) FAIL(FSIM_EINVAL, "bad args");
  HIPCHK(hipSetDevice(s->device));
  { int rc_ = settle(s); if (rc_) return rc_; }
  if (s->mw_mode == 2) hipLaunchKernelGGL(s->ks.physics_mw, dim3(s->n_envs), dim3(64 * FSIM_MW_NW), s->lds_bytes_mw, s->stream, s->d_m, s->d_ly_mw, kparams(s, 0, 1), s->d_state, s->d_aux);
  else hipLaunchKernelGGL(s->ks.physics, dim3(s->n_envs), dim3(64), s->lds_bytes, s->stream, s->d_m, s->d_ly, kparams(s, 0, 1), s->d_state, s->d_aux);
  HIPCHK(hipGetLastError());
  return FSIM_OK;
}

static int copy_field(fsim *s, int off, int dim, void *ext, int to_state) {
  if (!ext || dim == 0) return FSIM_OK;
  int n = s->n_envs * dim;
  hipLaunchKernelGGL(k_copy_field, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_state, s->ly.stride, off, dim, static_cast<float *>(ext), s->n_envs, to_state);
  HIPCHK(hipGetLastError());
  return FSIM_OK;
}
static int xfer_state(fsim *s, const fsim_state_ptrs_t *p, int to_state) {
  if (!s || !p) FAIL(FSIM_EINVAL, "null");
  HIPCHK(hipSetDevice(s->device));
  { int rc_ = settle(s); if (rc_) return rc_; }
  const DModel &m = s->m;
  const Layout &ly = s->ly;
  int rc;
  if ((rc = copy_field(s, ly.qpos, m.nq, p->qpos, to_state))) return rc;
  if ((rc = copy_field(s, ly.qvel, m.nv, p->qvel, to_state))) return rc;
  if ((rc = copy_field(s, ly.qaccws, m.nv, p->qacc_warmstart, to_state))) return rc;
  if ((rc = copy_field(s, ly.qfrcbias, m.nv, p->qfrc_bias, to_state))) return rc;
  if ((rc = copy_field(s, ly.ctrl, m.nu, p->ctrl, to_state))) return rc;
  if ((rc = copy_field(s, ly.qfrcapp, m.nv, p->qfrc_applied, to_state))) return rc;
  if ((rc = copy_field(s, ly.xfrc, 6 * m.nparts, p->xfrc_applied, to_state))) return rc;
  if ((rc = copy_field(s, ly.eqdata, 7 * m.neq, p->eq_data, to_state))) return rc;
  if ((rc = copy_field(s, ly.eqactive, m.neq, p->eq_active, to_state))) return rc;
  if ((rc = copy_field(s, ly.env, E_FIXED_WORDS + m.nparts + env_extra_words(m, s->cfg), p->env_block, to_state))) return rc; // (first: the named fields below win)
  if ((rc = copy_field(s, ly.env + E_GROUP, m.nparts, p->group, to_state))) return rc;
  if (p->dense) {
    if (!s->cfg.dense_reward) FAIL(FSIM_EINVAL, "state field 'dense' exists for dense_reward handles only");
    if ((rc = copy_field(s, ly.env + E_GROUP + m.nparts, ED_WORDS, p->dense, to_state))) return rc;
  }
  if (p->cursor) {
    if (m.agent != 2) FAIL(FSIM_EINVAL, "state field 'cursor' exists for the Cursor agent only");
    int n = s->n_envs;
    hipLaunchKernelGGL(k_copy_cursor, dim3((n + 63) / 64), dim3(64), 0, s->stream, s->d_state, ly.stride, ly.env + E_GROUP + m.nparts, p->cursor, n, to_state);
    HIPCHK(hipGetLastError());
  }
  for (int k = 0; k < 2; k++) {
    int32_t *ext = k == 0 ? p->geom_contype : p->geom_conaffinity;
    if (!ext) continue;
    int n = s->n_envs * m.ncg;
    hipLaunchKernelGGL(k_copy_geommask, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_state, ly.stride, k == 0 ? ly.contype : ly.conaff, m.ncg,
                       m.cg_orig, s->ngeom, ext, s->n_envs, to_state);
    HIPCHK(hipGetLastError());
  }
  if (!to_state) {
    int as = s->auxstride;
    if (p->qacc) HIPCHK(hipMemcpy2DAsync(p->qacc, m.nv * 4, s->d_aux, as * 4, m.nv * 4, s->n_envs, hipMemcpyDeviceToDevice, s->stream));
    if (p->xpos || p->xquat) {
      int n = s->n_envs * m.nbody;
      hipLaunchKernelGGL(k_expand_bodies, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d_aux, as, m.nv, m.nr, m.nbody, m.body_red, m.body_relpos,
                         m.body_relquat, p->xpos, p->xquat, s->n_envs);
      HIPCHK(hipGetLastError());
    }
    int ioff = m.nv + 7 * m.nr;
    if (p->ncon) HIPCHK(hipMemcpy2DAsync(p->ncon, 4, s->d_aux + ioff, as * 4, 4, s->n_envs, hipMemcpyDeviceToDevice, s->stream));
    if (p->solver_iters) HIPCHK(hipMemcpy2DAsync(p->solver_iters, 4, s->d_aux + ioff + 1, as * 4, 4, s->n_envs, hipMemcpyDeviceToDevice, s->stream));
    if (p->contact_geoms)
      HIPCHK(hipMemcpy2DAsync(p->contact_geoms, 2 * ly.ncon_max * 4, s->d_aux + ioff + 4, as * 4, 2 * ly.ncon_max * 4, s->n_envs, hipMemcpyDeviceToDevice, s->stream));
  }
  return FSIM_OK;
}
extern "C" int fsim_get_state(fsim_t *s, const fsim_state_ptrs_t *dst) { return xfer_state(s, dst, 0); }
extern "C" int fsim_set_state(fsim_t *s, const fsim_state_ptrs_t *src) { return xfer_state(s, src, 1); }

extern "C" int fsim_set_reset_tables(fsim_t *s, const uint8_t *mask, const float *part_qpos, const float *robot_noise, int n_noise) {
  if (!s || !part_qpos) FAIL(FSIM_EINVAL, "bad args");
  HIPCHK(hipSetDevice(s->device));
  // (a step of this handle still in flight reads the tables -- its terminal envs' resets, its look-ahead jobs: it completes first.
  //  fsim.h says so since round 5; the repo's own callers always called between fsim_sync and the next step)
  { int rc_ = settle(s); if (rc_) return rc_; }
  HIPCHK(hipStreamSynchronize(s->stream));
  const DModel &m = s->m;
  size_t pw = (size_t)7 * m.nparts, nw = (size_t)n_noise * m.narmj;
  if (!s->d_tab_parts) HIPCHK(hipMalloc(&s->d_tab_parts, (size_t)s->n_envs * pw * 4 + 16));
  if (robot_noise && (!s->d_tab_noise || s->n_noise != n_noise)) {
    if (s->d_tab_noise) { HIPCHK(hipStreamSynchronize(s->stream)); hipFree(s->d_tab_noise); s->d_tab_noise = nullptr; }
    HIPCHK(hipMalloc(&s->d_tab_noise, (size_t)s->n_envs * nw * 4 + 16));
    s->n_noise = n_noise;
  }
  if (!s->xfer) HIPCHK(hipStreamCreateWithFlags(&s->xfer, hipStreamNonBlocking));
  if (int rc = la_new_tables(s, mask)) return rc; // (new serial numbers: shadows computed from the old rows no longer count)
  // (no launch of this handle is in flight here: see the settle above; other handles' kernels do not read these rows)
  if (!mask) {
    HIPCHK(hipMemcpyAsync(s->d_tab_parts, part_qpos, (size_t)s->n_envs * pw * 4, hipMemcpyHostToDevice, s->xfer));
    if (robot_noise) HIPCHK(hipMemcpyAsync(s->d_tab_noise, robot_noise, (size_t)s->n_envs * nw * 4, hipMemcpyHostToDevice, s->xfer));
  } else {
    for (int e = 0; e < s->n_envs;) { // one copy per run of consecutive masked envs (a full-batch reset is a single run)
      if (!mask[e]) { e++; continue; }
      int e1 = e;
      while (e1 < s->n_envs && mask[e1]) e1++;
      size_t cnt = (size_t)(e1 - e);
      HIPCHK(hipMemcpyAsync(s->d_tab_parts + e * pw, part_qpos + e * pw, cnt * pw * 4, hipMemcpyHostToDevice, s->xfer));
      if (robot_noise) HIPCHK(hipMemcpyAsync(s->d_tab_noise + e * nw, robot_noise + e * nw, cnt * nw * 4, hipMemcpyHostToDevice, s->xfer));
      e = e1;
    }
  }
  HIPCHK(hipStreamSynchronize(s->xfer)); // host buffers may be reused by the caller right away
  return FSIM_OK;
}

extern "C" int fsim_set_attach_noise(fsim_t *s, const uint8_t *mask, const float *noise) {
  if (!s || !noise) FAIL(FSIM_EINVAL, "bad args");
  if (!s->ecfg.reset_robot_after_attach) FAIL(FSIM_EINVAL, "fsim_set_attach_noise: the handle was not created with reset_robot_after_attach = 1");
  HIPCHK(hipSetDevice(s->device));
  const size_t w = (size_t)s->m.narmj;
  if (w == 0) return FSIM_OK; // (the Cursor agent: _initialize_robot_pos draws nothing)
  if (!s->xfer) HIPCHK(hipStreamCreateWithFlags(&s->xfer, hipStreamNonBlocking));
  if (!s->d_tab_attach) { // (cleared on the SAME stream the rows are then copied on: a null-stream memset is not ordered against it)
    HIPCHK(hipMalloc(&s->d_tab_attach, (size_t)s->n_envs * w * 4 + 16));
    HIPCHK(hipMemsetAsync(s->d_tab_attach, 0, (size_t)s->n_envs * w * 4, s->xfer));
  }
  for (int e = 0; e < s->n_envs;) { // one copy per run of consecutive masked envs
    if (mask && !mask[e]) { e++; continue; }
    int e1 = e;
    while (e1 < s->n_envs && (!mask || mask[e1])) e1++;
    HIPCHK(hipMemcpyAsync(s->d_tab_attach + e * w, noise + e * w, (size_t)(e1 - e) * w * 4, hipMemcpyHostToDevice, s->xfer));
    e = e1;
  }
  HIPCHK(hipStreamSynchronize(s->xfer));
  return FSIM_OK;
}

extern "C" int fsim_set_init_state(fsim_t *s, const uint8_t *mask, const float *qpos, const float *qvel) {
  if (!s) FAIL(FSIM_EINVAL, "null");
  if (s->ecfg.n_pre > 0 && qpos) FAIL(FSIM_EINVAL, "fsim_set_init_state: not combined with pre-assembled starts (fsim_set_preassembled)");
  HIPCHK(hipSetDevice(s->device));
  { int rc_ = settle(s); if (rc_) return rc_; } // (a step still in flight -- and its overflow re-step -- run under the configuration they were launched with)
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->la_on && (s->d_init || qpos)) { // the resets of the masked envs start elsewhere from now on: their shadow records are void
    if (int rc = la_new_tables(s, mask)) return rc;
    HIPCHK(hipStreamSynchronize(s->xfer));
  }
  const int n = s->n_envs, nq = s->m.nq, nv = s->m.nv, w = nq + nv;
  if (!qpos) { // set_init_qpos(None): clears; allocates nothing
    if (!s->d_init) return FSIM_OK;
    for (int e = 0; e < n; e++) if (!mask || mask[e]) s->h_init_mask[e] = 0;
    HIPCHK(hipMemcpy(s->d_init_mask, s->h_init_mask.data(), n, hipMemcpyHostToDevice));
    return FSIM_OK;
  }
  if (!qvel) FAIL(FSIM_EINVAL, "fsim_set_init_state: qvel missing");
  if (!s->d_init) {
    HIPCHK(hipMalloc(&s->d_init, (size_t)n * w * 4)); HIPCHK(hipMalloc(&s->d_init_mask, n));
    HIPCHK(hipMemset(s->d_init_mask, 0, n));
    s->h_init_mask.assign(n, 0);
  }
  std::vector<float> row(w);
  for (int e = 0; e < n; e++) {
    if (mask && !mask[e]) continue;
    memcpy(row.data(), qpos + (size_t)e * nq, (size_t)nq * 4); memcpy(row.data() + nq, qvel + (size_t)e * nv, (size_t)nv * 4);
    HIPCHK(hipMemcpy(s->d_init + (size_t)e * w, row.data(), (size_t)w * 4, hipMemcpyHostToDevice));
    s->h_init_mask[e] = 1;
  }
  HIPCHK(hipMemcpy(s->d_init_mask, s->h_init_mask.data(), n, hipMemcpyHostToDevice));
  return FSIM_OK;
}

// ---- look-ahead reset, host side: the policy numbers and the table serials -- everything else happens on the device.
static void la_policy(fsim *s) {
  const int T = std::max(1, s->cfg.max_episode_steps);
  // reset units per job: about what a cheap env-step costs (50 substeps), so that a job fits the idle tail of a launch
  s->la_chunk = 51;
  // an env's shadow is started la_defer steps into its episode: the first steps of an episode are the ones with the most robot-part
  // contacts, i.e. the slowest, and the jobs of a batch that ended together are spread over the rest of the episode
  s->la_defer = T / 5;
  if (const char *e = getenv("FSIM_LA_CHUNK")) s->la_chunk = std::max(1, atoi(e));  // (development / tests)
  if (const char *e = getenv("FSIM_LA_DEFER")) s->la_defer = std::max(0, atoi(e));
  // jobs per launch: every env's reset (at most 401 units) done within ~half of what is left of the episode
  const int per_env = (401 + s->la_chunk - 1) / s->la_chunk;
  s->la_jobs = (int)std::ceil((double)s->n_envs * per_env / std::max(1.0, 0.5 * (T - s->la_defer)));
  if (const char *e = getenv("FSIM_LA_JOBS")) s->la_jobs = atoi(e);
  s->la_jobs = std::max(1, std::min(s->la_jobs, FSIM_LA_MAXJOBS));
}
// The envs in mask (host, null = all) get a new reset table, or what their reset starts from changes: a new serial number (a shadow
// computed from the old one no longer counts and is started over).  Copied on s->xfer; the caller synchronises it.
static int la_new_tables(fsim *s, const uint8_t *mask) {
  if (!s->la_on) return FSIM_OK;
  if (!s->xfer) HIPCHK(hipStreamCreateWithFlags(&s->xfer, hipStreamNonBlocking));
  for (int e = 0; e < s->n_envs;) {
    if (mask && !mask[e]) { e++; continue; }
    int e1 = e;
    while (e1 < s->n_envs && (!mask || mask[e1])) { s->h_tab_serial[e1] = (s->h_tab_serial[e1] & 0x3fffffff) + 1; e1++; }
    HIPCHK(hipMemcpyAsync(s->d_tab_serial + e, s->h_tab_serial.data() + e, (size_t)(e1 - e) * 4, hipMemcpyHostToDevice, s->xfer));
    e = e1;
  }
  return FSIM_OK;
}
extern "C" int64_t fsim_overflow_resteps(const fsim_t *s) { return s ? s->n_redone : 0; }
extern "C" int fsim_lookahead_stats(fsim_t *s, int64_t *out) {
  if (!s || !out) FAIL(FSIM_EINVAL, "null");
  out[0] = s->la_on ? 1 : 0; out[1] = s->h_nreset[3]; out[2] = s->h_nreset[1]; out[3] = s->h_nreset[2]; out[4] = s->la_jobs; out[5] = s->la_chunk;
  return FSIM_OK;
}



static int launch_env(fsim *s, const float *action, float *obs, float *reward, uint8_t *done, int32_t *info, const uint8_t *mask, int do_step) {
  HIPCHK(hipSetDevice(s->device));
  { int rc_ = settle(s); if (rc_) return rc_; }
  if (s->timing) timing_collect(s);
  bool sched = do_step && s->lpt;
  if (do_step) *s->h_nreset = 0; // (host-resident counter: no launch of this handle is in flight once the caller has synchronised)
  const bool mw_rule = sched && s->mw_mode == MW_RULE, mw_all = s->mw_mode == MW_ALL;
  if (sched) {
    LaSched la{};
    if (s->la_on && s->d_tab_parts) {
      la.sh_prog = s->d_sh_prog; la.sh_serial = s->d_sh_serial; la.tab_serial = s->d_tab_serial; la.init_mask = s->d_init_mask; la.jobs = s->d_sh_jobs;
      la.maxjobs = s->la_jobs; la.defer = std::min(s->la_defer, std::max(0, s->cfg.max_episode_steps - 1)); la.eplen_off = s->ly.env + E_EPISODE_LENGTH;
      la.total = 100 + (s->ecfg.has_recipe ? 100 : 0) + 201; la.total_init = 100; // (env_reset_total)
    }
    hipLaunchKernelGGL(k_schedule, dim3(1), dim3(64), (size_t)s->n_envs * 4, s->stream, s->d_cost, s->d_order, s->n_envs, reinterpret_cast<const int *>(s->d_state), s->ly.stride,
                       s->ly.env + E_NITER, s->mw_k, mw_rule ? 1 : 0, s->d_mworder, s->d_mwn, la);
  }
  if (s->timing) timing_begin(s);
  const KParams kp = kparams(s, s->cfg.n_substeps, 0);
  StepArgs a;
  a.cfg = s->ecfg; a.state = s->d_state; a.action = action; a.obs = obs; a.reward = reward; a.done = done; a.info = info;
  a.tab_parts = s->d_tab_parts; a.tab_noise = s->d_tab_noise; a.tab_attach = s->d_tab_attach; a.n_noise = s->n_noise; a.reset_mask = mask; a.do_step = do_step;
  a.prof = reinterpret_cast<int *>(s->d_aux); a.cost = do_step ? s->d_cost : nullptr; a.init_state = s->d_init; a.init_mask = s->d_init_mask;
  a.nreset = do_step ? s->d_nreset : nullptr;
  a.stats = s->d_nreset + 1;
  a.prev = nullptr; a.state_in = nullptr; a.ovf_list = nullptr; a.ovf_count = nullptr; a.ovf_cap = 0;
  if (s->redo_on) { // (step AND reset launches: a reset that drops contacts is repeated the same way)
    a.prev = s->d_prev; a.ovf_list = s->d_ovf_list; a.ovf_count = s->d_nreset + 4; a.ovf_cap = FSIM_OVF_CAP;
    s->h_nreset[4] = 0;
    s->last = {action, obs, reward, done, info};
    s->last_do_step = do_step;
  }
  s->redo_armed = s->redo_on;
  a.sh_state = nullptr; a.sh_obs = nullptr; a.sh_prog = nullptr; a.sh_serial = nullptr; a.tab_serial = nullptr; a.sh_jobs = nullptr; a.la_chunk = s->la_chunk;
  if (s->la_on) { a.sh_state = s->d_sh_state; a.sh_obs = s->d_sh_obs; a.sh_prog = s->d_sh_prog; a.sh_serial = s->d_sh_serial; a.tab_serial = s->d_tab_serial; }
  const bool jobs = sched && s->la_on && s->d_tab_parts && !mw_all; // (k_schedule has listed them)
  if (jobs) a.sh_jobs = s->d_sh_jobs;
  if (memcmp(&s->ecfg_sent, &s->ecfg, sizeof(EnvCfg)) != 0) { // (rare: max_episode_steps, dense tables, pre-assembled starts)
    HIPCHK(hipMemcpyAsync(s->d_ecfg, &s->ecfg, sizeof(EnvCfg), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream)); // (pageable source: the copy must have left the host struct before it can change again)
    memcpy(&s->ecfg_sent, &s->ecfg, sizeof(EnvCfg));
  }
  a.cfg_dev = s->d_ecfg;
  if (mw_all)
    hipLaunchKernelGGL(s->ks.env_step_mw, dim3(s->n_envs), dim3(64 * FSIM_MW_NW), s->lds_bytes_mw, s->stream, s->d_m, s->d_ly_mw, kp, a, sched ? s->d_order : nullptr, s->d_mwn);
  else if (mw_rule) // persistent workgroups: as many as the one-wave envs need in bundles of four plus an eighth of the batch for multi-wave envs, at most what is resident at once
    hipLaunchKernelGGL(s->ks.env_step_x, dim3(s->x_grid), dim3(64 * FSIM_MW_NW), s->lds_bytes_x, s->stream,
                       s->d_m, s->d_ly, s->d_ly_mw, kp, a, s->d_order, s->d_mworder, s->d_mwn, s->d_defer);
  else
    hipLaunchKernelGGL(s->ks.env_step, dim3(s->n_envs + (jobs ? s->la_jobs : 0)), dim3(64), s->lds_bytes, s->stream, s->d_m, s->d_ly, kp, a, sched ? s->d_order : nullptr, s->d_mwn);
  hipError_t e = hipGetLastError();
  if (s->timing) timing_end(s);
  if (e != hipSuccess) FAIL(FSIM_EHIP, "k_env_step launch: %s", hipGetErrorString(e));
  return FSIM_OK;
}
// The envs the last step launch listed (StepArgs::ovf_list: their step needed more contact slots / longer broadphase lists than the
// step kernel's LDS image holds, and dropped the rest) are stepped AGAIN from their pre-step records by the generic four-wave kernel
// with a 64-slot layout -- one workgroup per env, the whole 25 KB image in the team's LDS; for models on 64 slots: the generic one-wave kernel with 128 -- before fsim_sync returns: record,
// observation, reward, done and info rows of those envs are overwritten.  Rare (Sawyer + table_lack_0825: 1.6 per million env-steps),
// so its cost is a second small launch on those steps.  A deterministic function of the env's own pre-step record and action; the
// counters the first pass advanced (tables needed, reset statistics) are not advanced again; an env that overflows 64 slots too keeps
// its sticky report.  Consumers that read the step's outputs in stream order WITHOUT fsim_sync (an RCCL gather enqueued behind the step
// kernel) see the first pass's rows for such an env.
static int redo_overflowed(fsim *s) {
  int cnt = std::min(s->h_nreset[4], FSIM_OVF_CAP);
  std::vector<int> list(cnt);
  HIPCHK(hipMemcpy(list.data(), s->d_ovf_list, (size_t)cnt * 4, hipMemcpyDeviceToHost));
  int *cur = s->d_ovf_list, *nxt = s->d_ovf_list2;
  for (int r = 0; r < s->redo_rungs && cnt > 0; r++) {
    std::sort(list.begin(), list.end());
    list.erase(std::unique(list.begin(), list.end()), list.end()); // (a deferred reset is a second pass over the same env)
    cnt = (int)list.size();
    HIPCHK(hipMemcpy(cur, list.data(), (size_t)cnt * 4, hipMemcpyHostToDevice));
    KParams kp = kparams(s, s->cfg.n_substeps, 0);
    kp.n_envs = cnt; // (workgroups beyond the list do nothing)
    StepArgs a;
    a.cfg = s->ecfg; a.state = s->d_state; a.action = s->last.action; a.obs = reinterpret_cast<float *>(s->last.obs); a.reward = s->last.reward; a.done = s->last.done; a.info = s->last.info;
    a.tab_parts = s->d_tab_parts; a.tab_noise = s->d_tab_noise; a.tab_attach = s->d_tab_attach; a.n_noise = s->n_noise; a.reset_mask = nullptr; a.do_step = s->last_do_step;
    a.prof = reinterpret_cast<int *>(s->d_aux); a.cost = s->last_do_step ? s->d_cost : nullptr; a.init_state = s->d_init; a.init_mask = s->d_init_mask;
    a.nreset = nullptr; a.stats = nullptr;
    a.prev = nullptr; a.state_in = s->d_prev; a.ovf_list = nullptr; a.ovf_count = nullptr; a.ovf_cap = 0;
    const bool more = r + 1 < s->redo_rungs; // the envs that drop contacts on this rung's layout too are listed for the next one
    if (more) { a.ovf_list = nxt; a.ovf_count = s->d_nreset + 5; a.ovf_cap = FSIM_OVF_CAP; s->h_nreset[5] = 0; }
    a.sh_state = nullptr; a.sh_obs = nullptr; a.sh_prog = nullptr; a.sh_serial = nullptr; a.tab_serial = nullptr; a.sh_jobs = nullptr; a.la_chunk = s->la_chunk;
    a.cfg_dev = s->d_ecfg;
    hipLaunchKernelGGL(s->redo_kernel[r], dim3(cnt), dim3(s->redo_block[r]), s->lds_bytes_r[r], s->stream, s->d_m, s->d_ly_r[r], kp, a, cur, s->d_mwn);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(FSIM_EHIP, "overflow re-step launch: %s", hipGetErrorString(e));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->n_redone += cnt;
    cnt = more ? std::min(s->h_nreset[5], FSIM_OVF_CAP) : 0;
    if (cnt > 0) { list.resize(cnt); HIPCHK(hipMemcpy(list.data(), nxt, (size_t)cnt * 4, hipMemcpyDeviceToHost)); std::swap(cur, nxt); }
  }
  return FSIM_OK;
}
// ---- dense-reward env
static_assert(DC_WORDS == FSIM_DENSE_NCOEF && DS_WORDS == FSIM_DENSE_SUBW && ED_WORDS == FSIM_DENSE_STATEW, "fsim.h dense sizes");
static int dense_check(const float *coef, int ncoef, const float *sub, int nsub, int nsite, int nparts, int nconn) {
  if (!coef || !sub || ncoef != DC_WORDS || nsub < 1 || nsub > 16) FAIL(FSIM_EINVAL, "dense reward: need %d coefficients and 1..16 subtasks", DC_WORDS);
  if (nsite >= 0) {
    auto bad = [&](float v, int n) { return !(v >= 0 && v < n && v == (float)(int)v); };
    if (bad(coef[DC_GRIPTIP_SITE], nsite) || bad(coef[DC_GRIP_SITE], nsite)) FAIL(FSIM_EINVAL, "dense reward: bad gripper site id");
    for (int i = 0; i < nsub; i++) {
      const float *t = sub + DS_WORDS * i;
      if (bad(t[DS_LEG_PART], nparts) || bad(t[DS_TABLE_PART], nparts) || bad(t[DS_LEG_SITE], nsite) || bad(t[DS_TABLE_SITE], nsite) ||
          bad(t[DS_GL_SITE], nsite) || bad(t[DS_GR_SITE], nsite) || bad(t[DS_K_LEG], nconn) || bad(t[DS_K_TABLE], nconn))
        FAIL(FSIM_EINVAL, "dense reward: subtask %d refers to a part/site/connector that does not exist", i);
    }
  }
  return FSIM_OK;
}
extern "C" int fsim_set_dense_reward(fsim_t *s, const float *coef, int ncoef, const float *sub, int nsub) {
  if (!s) FAIL(FSIM_EINVAL, "null");
  if (!s->cfg.dense_reward) FAIL(FSIM_EINVAL, "fsim_set_dense_reward: the handle was not created with dense_reward = 1");
  if (int rc = dense_check(coef, ncoef, sub, nsub, s->m.nsite, s->m.nparts, s->m.nconn)) return rc;
  HIPCHK(hipSetDevice(s->device));
  { int rc_ = settle(s); if (rc_) return rc_; }
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->la_on) { if (int rc = la_new_tables(s, nullptr)) return rc; HIPCHK(hipStreamSynchronize(s->xfer)); }
  if (s->d_dense) { hipFree(s->d_dense); s->d_dense = nullptr; }
  HIPCHK(hipMalloc(&s->d_dense, (size_t)(DC_WORDS + DS_WORDS * nsub) * 4));
  HIPCHK(hipMemcpy(s->d_dense, coef, DC_WORDS * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(s->d_dense + DC_WORDS, sub, (size_t)DS_WORDS * nsub * 4, hipMemcpyHostToDevice));
  s->ecfg.dense_coef = s->d_dense; s->ecfg.dense_sub = s->d_dense + DC_WORDS; s->ecfg.dense_nsub = nsub;
  return FSIM_OK;
}

struct DenseReplayP { // sensor values from recorded arrays ([nsub][FSIM_DENSE_OBSW], oracle/dense_reward.py O_*)
  const float *o;
  DEV void obs(int st, DObs &d) const {
    const float *p = o + FSIM_DENSE_OBSW * st;
    d.eef = ldv3(p); d.gl = ldv3(p + 3); d.gr = ldv3(p + 6); d.leg = ldv3(p + 9); d.legsite = ldv3(p + 12); d.tablesite = ldv3(p + 15);
    d.legup = ldv3(p + 18); d.tableup = ldv3(p + 21); d.legfwd = ldv3(p + 24); d.tablefwd = ldv3(p + 27); d.gripup = ldv3(p + 30);
    d.gripfwd = ldv3(p + 33); d.touch_l = p[36] != 0.0f; d.touch_r = p[37] != 0.0f;
  }
  DEV bool aligned(int st) const { return o[FSIM_DENSE_OBSW * st + 38] != 0.0f; }
};
__global__ void k_dense_replay(const float *coef, const float *sub, int nsub, int n_pre, const float *obs0, const float *obs, const float *ac,
                               int dof, const uint8_t *connected, int T, float *out_reward, int *out_flags) {
  if (threadIdx.x || blockIdx.x) return;
  float S[ED_WORDS];
  DenseReplayP p{obs0};
  dense_reset(S, coef, sub, p, n_pre);
  for (int t = 0; t < T; t++) {
    p.o = obs + (size_t)t * nsub * FSIM_DENSE_OBSW;
    DenseOut d = dense_compute(S, coef, sub, nsub, p, ac + (size_t)t * dof, dof, connected[t] != 0);
    out_reward[t] = d.reward;
    out_flags[4 * t] = d.done; out_flags[4 * t + 1] = d.success; out_flags[4 * t + 2] = (int)S[ED_PHASE]; out_flags[4 * t + 3] = (int)S[ED_SUBTASK];
  }
}
extern "C" int fsim_dense_replay(int device, const float *coef, int ncoef, const float *sub, int nsub, int n_pre, const float *obs0,
                                 const float *obs, const float *ac, int dof, const uint8_t *connected, int T, float *out_reward,
                                 int32_t *out_flags) {
  if (!obs0 || !obs || !ac || !connected || !out_reward || !out_flags || T < 1 || dof < 3 || n_pre < 0 || n_pre >= nsub) FAIL(FSIM_EINVAL, "fsim_dense_replay: bad args");
  if (int rc = dense_check(coef, ncoef, sub, nsub, -1, 0, 0)) return rc;
  HIPCHK(hipSetDevice(device));
  size_t nc = DC_WORDS, ns = (size_t)DS_WORDS * nsub, no0 = (size_t)nsub * FSIM_DENSE_OBSW, no = no0 * T, na = (size_t)T * dof;
  float *d = nullptr;
  uint8_t *dc = nullptr;
  HIPCHK(hipMalloc(&d, (nc + ns + no0 + no + na + T + 4 * (size_t)T) * 4));
  HIPCHK(hipMalloc(&dc, T));
  float *d_c = d, *d_s = d_c + nc, *d_o0 = d_s + ns, *d_o = d_o0 + no0, *d_a = d_o + no, *d_r = d_a + na;
  int *d_f = reinterpret_cast<int *>(d_r + T);
  hipMemcpy(d_c, coef, nc * 4, hipMemcpyHostToDevice); hipMemcpy(d_s, sub, ns * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_o0, obs0, no0 * 4, hipMemcpyHostToDevice); hipMemcpy(d_o, obs, no * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_a, ac, na * 4, hipMemcpyHostToDevice); hipMemcpy(dc, connected, T, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_dense_replay, dim3(1), dim3(64), 0, 0, d_c, d_s, nsub, n_pre, d_o0, d_o, d_a, dof, dc, T, d_r, d_f);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out_reward, d_r, (size_t)T * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_flags, d_f, (size_t)T * 16, hipMemcpyDeviceToHost);
  hipFree(d); hipFree(dc);
  if (e != hipSuccess) FAIL(FSIM_EHIP, "fsim_dense_replay: %s", hipGetErrorString(e));
  return FSIM_OK;
}

// ---- env-logic replay hooks: the DEVICE functions of the connector state machine fed with recorded inputs, so that the
// golden vectors the reference's own methods produced (tests/golden/env_logic.npz, step_scan.npz) are checked against the
// code that runs inside fsim_step -- not only against the CPU restatement.
__global__ void k_replay_is_aligned(AlignCfg cfg, int n, const float *p1, const float *R1, const float *p2, const float *R2, const int *nang,
                                    const float *angles, int *out_ok, float *out_tq) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float tq[4] = {__int_as_float(0x7fc00000), __int_as_float(0x7fc00000), __int_as_float(0x7fc00000), __int_as_float(0x7fc00000)};
  bool ok = env_is_aligned_core(ldv3(p1 + 3 * i), ldm3(R1 + 9 * i), ldv3(p2 + 3 * i), ldm3(R2 + 9 * i), nang[i], angles + 4 * i, cfg, tq);
  out_ok[i] = ok ? 1 : 0;
  for (int k = 0; k < 4; k++) out_tq[4 * i + k] = tq[k];
}
extern "C" int fsim_replay_is_aligned(int device, float pos_dist, float rot_up, float rot_fwd, float proj_dist, int n, const float *p1, const float *R1,
                                      const float *p2, const float *R2, const int32_t *nang, const float *angles, int32_t *out_ok, float *out_tq) {
  if (n < 1 || !p1 || !R1 || !p2 || !R2 || !nang || !angles || !out_ok || !out_tq) FAIL(FSIM_EINVAL, "fsim_replay_is_aligned: bad args");
  HIPCHK(hipSetDevice(device));
  float *d = nullptr;
  const size_t w = (size_t)n * (3 + 9 + 3 + 9 + 1 + 4 + 1 + 4);
  HIPCHK(hipMalloc(&d, w * 4));
  float *dp1 = d, *dR1 = dp1 + 3 * n, *dp2 = dR1 + 9 * n, *dR2 = dp2 + 3 * n, *dang = dR2 + 9 * n, *dtq = dang + 4 * n;
  int *dn = reinterpret_cast<int *>(dtq + 4 * n), *dok = dn + n;
  hipMemcpy(dp1, p1, 12 * n, hipMemcpyHostToDevice); hipMemcpy(dR1, R1, 36 * n, hipMemcpyHostToDevice);
  hipMemcpy(dp2, p2, 12 * n, hipMemcpyHostToDevice); hipMemcpy(dR2, R2, 36 * n, hipMemcpyHostToDevice);
  hipMemcpy(dang, angles, 16 * n, hipMemcpyHostToDevice); hipMemcpy(dn, nang, 4 * n, hipMemcpyHostToDevice);
  AlignCfg ac; ac.pos_dist = pos_dist; ac.rot_up = rot_up; ac.rot_fwd = rot_fwd; ac.proj_dist = proj_dist;
  hipLaunchKernelGGL(k_replay_is_aligned, dim3((n + 63) / 64), dim3(64), 0, 0, ac, n, dp1, dR1, dp2, dR2, dn, dang, dok, dtq);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out_ok, dok, 4 * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_tq, dtq, 16 * n, hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) FAIL(FSIM_EHIP, "fsim_replay_is_aligned: %s", hipGetErrorString(e));
  return FSIM_OK;
}

// one workgroup per trial, on the handle's model and record layout: the env block (group table, used-site bits, connect step)
// comes from the caller, the alignment test is a recorded truth table [nconn][nconn]
__global__ __launch_bounds__(64) void k_replay_try_connect(const DModel *mp, const Layout *lp, int num_connect_steps, const int *part12, const int *group,
                                                         const int *used, const unsigned char *aligned, const int *step_in, int *out) {
  extern __shared__ float L[];
  CModel &m = *(CModel *)mp;
  const GenCtx c(L, m, *(CLayout *)lp, threadIdx.x, 0, 0.0f);
  const int t = blockIdx.x, nconn = c.D.nconn;
  int *E = c.I(c.ly.env);
  for (int i = threadIdx.x; i < E_FIXED_WORDS + c.D.nparts; i += 64) E[i] = 0;
  SYNC();
  if (threadIdx.x == 0) {
    for (int p = 0; p < c.D.nparts; p++) E[E_GROUP + p] = group[t * c.D.nparts + p];
    for (int k = 0; k < nconn; k++) if (used[t * nconn + k]) E[E_CONNSITES0 + (k >> 5)] |= 1 << (k & 31);
    E[E_CONNECT_STEP] = step_in[t];
    const unsigned char *al = aligned + (size_t)t * nconn * nconn;
    int f1, f2;
    const bool searched = env_connect_search(c, part12[2 * t], part12[2 * t + 1], [&](int k1, int k2) { return al[k1 * nconn + k2] != 0; }, &f1, &f2);
    const int approach = env_connect_decide(E, num_connect_steps, searched, f1);
    out[5 * t] = f1; out[5 * t + 1] = f2; out[5 * t + 2] = (!approach && f1 >= 0) ? 1 : 0; // returns True only when it connects
    out[5 * t + 3] = E[E_CONNECT_STEP]; out[5 * t + 4] = (approach && f2 >= 0) ? m.conn_partid[f2] : -1;
  }
}
extern "C" int fsim_replay_try_connect(fsim_t *s, int n, int num_connect_steps, const int32_t *part12, const int32_t *group, const int32_t *used,
                                       const uint8_t *aligned, const int32_t *step_in, int32_t *out) {
  if (!s || n < 1 || !part12 || !group || !used || !aligned || !step_in || !out) FAIL(FSIM_EINVAL, "fsim_replay_try_connect: bad args");
  HIPCHK(hipSetDevice(s->device));
  const int np = s->m.nparts, nc = s->m.nconn;
  int *d = nullptr; unsigned char *da = nullptr;
  HIPCHK(hipMalloc(&d, (size_t)n * (2 + np + nc + 1 + 5) * 4)); HIPCHK(hipMalloc(&da, (size_t)n * nc * nc + 4));
  int *d12 = d, *dg = d12 + 2 * n, *du = dg + (size_t)n * np, *ds = du + (size_t)n * nc, *dout = ds + n;
  hipMemcpy(d12, part12, 8 * n, hipMemcpyHostToDevice); hipMemcpy(dg, group, (size_t)4 * n * np, hipMemcpyHostToDevice);
  hipMemcpy(du, used, (size_t)4 * n * nc, hipMemcpyHostToDevice); hipMemcpy(ds, step_in, 4 * n, hipMemcpyHostToDevice);
  hipMemcpy(da, aligned, (size_t)n * nc * nc, hipMemcpyHostToDevice);
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_replay_try_connect), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes));
  hipLaunchKernelGGL(k_replay_try_connect, dim3(n), dim3(64), s->lds_bytes, s->stream, s->d_m, s->d_ly, num_connect_steps, d12, dg, du, da, ds, dout);
  hipError_t e = hipStreamSynchronize(s->stream);
  if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)20 * n, hipMemcpyDeviceToHost);
  hipFree(d); hipFree(da);
  if (e != hipSuccess) FAIL(FSIM_EHIP, "fsim_replay_try_connect: %s", hipGetErrorString(e));
  return FSIM_OK;
}

// contact lists (colliding-geom indices) -> fs_touch_flags -> env_finger_scan with scripted try_connect outcomes: which parts
// are tried, in which order
__global__ __launch_bounds__(64) void k_replay_touch_scan(const DModel *mp, const Layout *lp, int maxc, const int *ncon, const int *geoms, const unsigned char *script,
                                                        int *out_masks, int *out_tried) {
  extern __shared__ float L[];
  CModel &m = *(CModel *)mp;
  const GenCtx c(L, m, *(CLayout *)lp, threadIdx.x, 0, 0.0f);
  const int t = blockIdx.x;
  int *scal = c.I(c.ly.scal);
  const int n = min(ncon[t], c.ly.ncon_max);
  for (int sl = threadIdx.x; sl < n; sl += 64) {
    int *ri = c.I(c.ly.con + FSIM_CONW * sl);
    ri[C_ACTIVE] = 1; ri[C_G1] = geoms[(t * maxc + sl) * 2]; ri[C_G2] = geoms[(t * maxc + sl) * 2 + 1];
  }
  if (threadIdx.x == 0) scal[SC_NSLOT] = n;
  SYNC();
  fs_touch_flags(c);
  int ntried = 0;
  env_finger_scan(c.D.narm, scal, [&](int part) {
    if (threadIdx.x == 0) out_tried[4 * t + ntried] = part;
    int r = script[4 * t + ntried] ? 1 : 0;
    ntried++;
    return r;
  });
  if (threadIdx.x == 0) {
    for (int k = ntried; k < 4; k++) out_tried[4 * t + k] = -1;
    out_masks[3 * t] = scal[SC_TOUCHL]; out_masks[3 * t + 1] = scal[SC_TOUCHR]; out_masks[3 * t + 2] = scal[SC_TOUCHF];
  }
}
extern "C" int fsim_replay_touch_scan(fsim_t *s, int n, int maxc, const int32_t *ncon, const int32_t *geoms, const uint8_t *script, int32_t *out_masks,
                                      int32_t *out_tried) {
  if (!s || n < 1 || maxc < 1 || !ncon || !geoms || !script || !out_masks || !out_tried) FAIL(FSIM_EINVAL, "fsim_replay_touch_scan: bad args");
  HIPCHK(hipSetDevice(s->device));
  for (int t = 0; t < n; t++) {
    if (ncon[t] < 0 || ncon[t] > maxc) FAIL(FSIM_EINVAL, "fsim_replay_touch_scan: ncon[%d] out of range", t);
    for (int k = 0; k < 2 * ncon[t]; k++) if (geoms[(size_t)t * maxc * 2 + k] < 0 || geoms[(size_t)t * maxc * 2 + k] >= s->m.ncg) FAIL(FSIM_EINVAL, "fsim_replay_touch_scan: geom index out of range (colliding-geom indices expected)");
  }
  int *d = nullptr; unsigned char *dsc = nullptr;
  HIPCHK(hipMalloc(&d, (size_t)n * (1 + 2 * maxc + 3 + 4) * 4)); HIPCHK(hipMalloc(&dsc, (size_t)4 * n));
  int *dn = d, *dg = dn + n, *dm = dg + (size_t)2 * n * maxc, *dt = dm + 3 * n;
  hipMemcpy(dn, ncon, 4 * n, hipMemcpyHostToDevice); hipMemcpy(dg, geoms, (size_t)8 * n * maxc, hipMemcpyHostToDevice);
  hipMemcpy(dsc, script, (size_t)4 * n, hipMemcpyHostToDevice);
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_replay_touch_scan), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes));
  hipLaunchKernelGGL(k_replay_touch_scan, dim3(n), dim3(64), s->lds_bytes, s->stream, s->d_m, s->d_ly, maxc, dn, dg, dsc, dm, dt);
  hipError_t e = hipStreamSynchronize(s->stream);
  if (e == hipSuccess) e = hipMemcpy(out_masks, dm, (size_t)12 * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_tried, dt, (size_t)16 * n, hipMemcpyDeviceToHost);
  hipFree(d); hipFree(dsc);
  if (e != hipSuccess) FAIL(FSIM_EHIP, "fsim_replay_touch_scan: %s", hipGetErrorString(e));
  return FSIM_OK;
}

extern "C" int fsim_reset(fsim_t *s, const uint8_t *mask_dev, void *obs_dev) {
  if (!s) FAIL(FSIM_EINVAL, "null");
  if (!s->d_tab_parts) FAIL(FSIM_EINVAL, "fsim_reset: call fsim_set_reset_tables first");
  if (s->cfg.dense_reward && !s->d_dense) FAIL(FSIM_EINVAL, "fsim_reset: dense_reward needs fsim_set_dense_reward first");
  return launch_env(s, nullptr, static_cast<float *>(obs_dev), nullptr, nullptr, nullptr, mask_dev, 0);
}
extern "C" int fsim_step(fsim_t *s, const float *action, void *obs, float *reward, uint8_t *done, int32_t *info) {
  if (!s || !action) FAIL(FSIM_EINVAL, "fsim_step: null handle/action");
  if (s->cfg.auto_reset && !s->d_tab_parts) FAIL(FSIM_EINVAL, "fsim_step: auto_reset needs fsim_set_reset_tables");
  if (s->cfg.dense_reward && !s->d_dense) FAIL(FSIM_EINVAL, "fsim_step: dense_reward needs fsim_set_dense_reward first");
  return launch_env(s, action, static_cast<float *>(obs), reward, done, info, nullptr, 1);
}
extern "C" int fsim_set_preassembled(fsim_t *s, int n_pre, const int32_t *ids, const int32_t *conn_pairs, const float *angles, int num_connects) {
  if (!s || n_pre < 0 || n_pre > 16 || (n_pre > 0 && !ids)) FAIL(FSIM_EINVAL, "fsim_set_preassembled: bad arguments");
  const bool recipe = s->ecfg.has_recipe != 0 && conn_pairs != nullptr; // (no connector pairs: the list holds weld ids -- config.assembled)
  if (n_pre > 0 && std::any_of(s->h_init_mask.begin(), s->h_init_mask.end(), [](uint8_t v) { return v != 0; }))
    FAIL(FSIM_EINVAL, "fsim_set_preassembled: not combined with fsim_set_init_state (an env still has an init state set; clear it with qpos = NULL)");
  if (n_pre > 0 && recipe && !angles) FAIL(FSIM_EINVAL, "fsim_set_preassembled: recipe steps need their angles next to the connector pairs");
  HIPCHK(hipSetDevice(s->device));
  { int rc_ = settle(s); if (rc_) return rc_; }
  std::vector<int> tab(3 * (size_t)n_pre, 0);
  for (int i = 0; i < n_pre; i++) {
    if (recipe) {
      const int k1 = conn_pairs[2 * i], k2 = conn_pairs[2 * i + 1];
      if (k1 < 0 || k1 >= s->m.nconn || k2 < 0 || k2 >= s->m.nconn) FAIL(FSIM_EINVAL, "fsim_set_preassembled: connector index out of range in row %d", i);
      tab[3 * i] = k1; tab[3 * i + 1] = k2; memcpy(&tab[3 * i + 2], &angles[i], 4);
    } else {
      if (ids[i] < 0 || ids[i] >= s->m.neq) FAIL(FSIM_EINVAL, "fsim_set_preassembled: weld id %d out of range (%d welds)", ids[i], s->m.neq);
      tab[3 * i] = ids[i];
    }
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->la_on) { if (int rc = la_new_tables(s, nullptr)) return rc; HIPCHK(hipStreamSynchronize(s->xfer)); }
  if (s->d_pre) { hipFree(s->d_pre); s->d_pre = nullptr; }
  if (n_pre > 0) {
    HIPCHK(hipMalloc(&s->d_pre, tab.size() * 4));
    HIPCHK(hipMemcpy(s->d_pre, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  }
  s->ecfg.n_pre = n_pre; s->ecfg.pre_mode = recipe ? 1 : 0; s->ecfg.pre_tab = s->d_pre;
  // _success_num_conn (furniture.py:1476-1481)
  s->ecfg.success_num_conn = num_connects >= 0 ? num_connects + n_pre : s->m.nparts - 1;
  return FSIM_OK;
}
extern "C" int fsim_set_max_episode_steps(fsim_t *s, int n) {
  if (!s || n <= 0) FAIL(FSIM_EINVAL, "fsim_set_max_episode_steps: bad arguments");
  HIPCHK(hipSetDevice(s->device));
  { int rc_ = settle(s); if (rc_) return rc_; }
  s->cfg.max_episode_steps = n; s->ecfg.max_episode_steps = n; // EnvCfg is passed by value with every launch
  if (s->la_on) la_policy(s);
  return FSIM_OK;
}
// Device -> host copy of caller memory on the handle's transfer stream (pinned destination: a DMA transfer, no kernel, no allocation).
// Complete on return.  The handle's in-flight step is waited for first (its overflow re-step included): the rows read are final.
extern "C" int fsim_read(fsim_t *s, void *host_dst, const void *dev_src, size_t nbytes) {
  if (!s || !host_dst || !dev_src) FAIL(FSIM_EINVAL, "null");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  { int rc_ = settle(s); if (rc_) return rc_; }
  if (!s->xfer) HIPCHK(hipStreamCreateWithFlags(&s->xfer, hipStreamNonBlocking));
  HIPCHK(hipMemcpyAsync(host_dst, dev_src, nbytes, hipMemcpyDeviceToHost, s->xfer));
  HIPCHK(hipStreamSynchronize(s->xfer));
  return FSIM_OK;
}
extern "C" int fsim_kernel_time_ms(fsim_t *s, double *avg_ms, int32_t *n) {
  if (!s) FAIL(FSIM_EINVAL, "null");
  s->timing = true;
  timing_collect(s);
  if (avg_ms) *avg_ms = s->acc_n ? s->acc_ms / s->acc_n : 0.0;
  if (n) *n = s->acc_n;
  s->acc_ms = 0; s->acc_n = 0;
  return FSIM_OK;
}
