// fsim_host.c -- libfsim_host.so: host-side helper of the env layer (plain C, no GPU): the reference's reset-time RNG
// stream for MANY envs per call.
//
// Every env i owns `np.random.RandomState(seed + i)` (furniture/env/base.py:77, furniture.py:72) and each reset consumes, in this
// order, the draws of UniformRandomSampler.sample() (tasks/placement_sampler.py:138-190: per part, rejection-sampled x / y
// offsets, then one draw for the -- constant -- rotation) and 101 x n_arm_joints uniform draws of _initialize_robot_pos
// (furniture.py:1761-1779).  ResetTableSampler (furniture_amd/envs.py) replays that stream per env in Python: 94 us per env,
// GIL-bound, i.e. 0.39 s for a 4096-env batch-wide reset.  This file restates the two NumPy primitives involved --
// MT19937 `genrand_int32` and legacy `random_sample` = (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53, uniform(low, high) =
// low + (high - low) * random_sample() -- over an array of generator states, so the whole batch is one call (~1 ms).
// The state layout is RandomState.get_state()'s: 624 key words + the position.  Pinned against the reference's own sampler
// through tests/golden/placement_sampler.npz (tests/test_sampler_golden.py runs both paths).
#include <math.h>
#include <omp.h>
#include <stdint.h>

#define MT_N 624
#define MT_M 397
#define FSIM_MT_WORDS 625 /* key[624], pos */

static void mt_refill(uint32_t *mt) {
  int kk;
  uint32_t y;
  for (kk = 0; kk < MT_N - MT_M; kk++) {
    y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
    mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  for (; kk < MT_N - 1; kk++) {
    y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
    mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
  mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
static uint32_t mt_next(uint32_t *st) {
  uint32_t y;
  if (st[MT_N] >= MT_N) { mt_refill(st); st[MT_N] = 0; }
  y = st[st[MT_N]++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
static double mt_double(uint32_t *st) {
  const uint32_t a = mt_next(st) >> 5, b = mt_next(st) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
static double mt_uniform(uint32_t *st, double low, double high) { return low + (high - low) * mt_double(st); }

// states [n][625]; draw_mask / place_mask [n] (null = all): envs that draw at all / that take the placement draws (config.fix_init
// envs keep their first placement: later resets take only the joint-noise draws).  base_xy [nparts][2], radius [nparts];
// out_xy [n][nparts][2] (written for placing envs), noise [n][n_noise] float32, uniform(-a, a).  Returns 0, or 1 + the index of
// an env whose parts could not be placed in 10000 tries per part (the reference raises RandomizationError there): that env stops
// at the failing part, as the reference's sampler does (no further part, no joint noise), and status[e] (null = not wanted) says
// which envs failed; the caller rolls the generators back (furniture_amd/envs.py: a failed draw consumes nothing).
// nthreads: OpenMP threads of this call (<= 0: the runtime's default) -- N ranks of one node hit their batch-wide reset on the same
// step, each must stay on its share of the host cores.
// (compiled with -ffp-contract=off: `low + (high - low) * u` must round twice, as NumPy's does, also where the target has an FMA)
int fsim_host_reset_draw(uint32_t *states, int n, const uint8_t *draw_mask, const uint8_t *place_mask, int nparts, const double *base_xy,
                         const double *radius, double lo, double hi, double rot_hi, int n_noise, double a, double *out_xy, float *noise,
                         uint8_t *status, int nthreads) {
  int e, failed = 0;
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) reduction(max : failed) num_threads(nthreads)
  for (e = 0; e < n; e++) {
    int i, j, t, env_failed = 0;
    uint32_t *st = states + (long)e * FSIM_MT_WORDS;
    if (status) status[e] = 0;
    if (draw_mask && !draw_mask[e]) continue;
    if (!place_mask || place_mask[e]) {
      double *xy = out_xy + (long)e * nparts * 2;
      for (i = 0; i < nparts && !env_failed; i++) {
        int ok = 0;
        for (t = 0; t < 10000 && !ok; t++) {
          const double x = base_xy[2 * i] + mt_uniform(st, lo, hi), y = base_xy[2 * i + 1] + mt_uniform(st, lo, hi);
          ok = 1;
          for (j = 0; j < i; j++)
            if (!(hypot(x - xy[2 * j], y - xy[2 * j + 1]) > radius[j] + radius[i])) { ok = 0; break; }
          if (ok) { (void)mt_uniform(st, rot_hi, rot_hi); xy[2 * i] = x; xy[2 * i + 1] = y; }
        }
        if (!ok) { env_failed = 1; if (failed < 1 + e) failed = 1 + e; }
      }
    }
    if (env_failed) { if (status) status[e] = 1; continue; }
    if (n_noise > 0) {
      float *z = noise + (long)e * n_noise;
      for (i = 0; i < n_noise; i++) z[i] = (float)mt_uniform(st, -a, a);
    }
  }
  return failed;
}

// n successive random_sample() values of one generator (tests)
void fsim_host_random_sample(uint32_t *state, int n, double *out) {
  int i;
  for (i = 0; i < n; i++) out[i] = mt_double(state);
}

// RandomState(seed) for n integer seeds: init_genrand (the legacy seeding NumPy applies to an int), position at the end of the block
void fsim_host_seed(uint32_t *states, int n, const uint32_t *seeds) {
  int e;
#pragma omp parallel for schedule(static)
  for (e = 0; e < n; e++) {
    uint32_t *mt = states + (long)e * FSIM_MT_WORDS;
    int k;
    mt[0] = seeds[e];
    for (k = 1; k < MT_N; k++) mt[k] = 1812433253u * (mt[k - 1] ^ (mt[k - 1] >> 30)) + (uint32_t)k;
    mt[MT_N] = MT_N;
  }
}
