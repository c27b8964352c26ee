/* examples/c_host.c -- a host program in plain C on top of include/fsim.h: no Python, no torch.  What a maintainer's cgo / JNI / N-API stub
 * does, written out: device buffers from the HIP runtime, the model blob and the reset tables from files, fsim_create / fsim_set_reset_tables /
 * fsim_reset / fsim_step / fsim_sync, results copied back and printed as checksums.  tests/test_c_host_gpu.py builds it
 * (hipcc examples/c_host.c -Iinclude -Lfurniture_amd/csrc -lfsim), runs it and compares every number with the same calls made through ctypes.
 *
 *   c_host <model.blob> <tables.bin> <n_envs> <steps>
 *   tables.bin: int32 nparts7, int32 nnoise_words, then float32 part_qpos[n][nparts7], float32 robot_noise[n][nnoise_words]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fsim.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define FS(x) do { if ((x) != FSIM_OK) { fprintf(stderr, "%s: %s\n", #x, fsim_last_error()); return 3; } } while (0)

static void *slurp(const char *path, size_t *n) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
  void *p = malloc(*n);
  if (fread(p, 1, *n, f) != *n) { fclose(f); free(p); return NULL; }
  fclose(f);
  return p;
}
/* a checksum that does not depend on the summation order of a parallel reduction: 64-bit sum of the float bit patterns */
static unsigned long long bits_sum(const void *p, size_t words) { const uint32_t *w = (const uint32_t *)p; unsigned long long s = 0; for (size_t i = 0; i < words; i++) s += w[i]; return s; }

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s model.blob tables.bin n_envs steps\n", argv[0]); return 1; }
  size_t nb, nt;
  void *blob = slurp(argv[1], &nb), *tab = slurp(argv[2], &nt);
  const int n = atoi(argv[3]), steps = atoi(argv[4]);
  if (!blob || !tab || n <= 0) { fprintf(stderr, "cannot read the inputs\n"); return 1; }
  const int32_t nparts7 = ((int32_t *)tab)[0], nnoise = ((int32_t *)tab)[1];
  const float *parts = (const float *)((int32_t *)tab + 2), *noise = parts + (size_t)n * nparts7;
  if (nt < 8 + ((size_t)n * nparts7 + (size_t)n * nnoise) * 4) { fprintf(stderr, "tables.bin too short\n"); return 1; }

  fsim_config_t cfg;
  fsim_default_config(&cfg);
  cfg.max_episode_steps = 5; /* an auto-reset inside the run */
  cfg.auto_reset = 1;
  fsim_t *h;
  FS(fsim_create(blob, nb, n, 0, &cfg, &h));
  int32_t nq, nv, nu, dof, obs_dim, info_dim, stride;
  FS(fsim_dims(h, &nq, &nv, &nu, &dof, &obs_dim, &info_dim, &stride));
  printf("kernel %s | nq %d nv %d nu %d dof %d obs %d info %d\n", fsim_kernel_variant(h), nq, nv, nu, dof, obs_dim, info_dim);

  float *d_act, *d_obs, *d_rew; uint8_t *d_done; int32_t *d_info;
  HIP(hipMalloc((void **)&d_act, (size_t)n * dof * 4)); HIP(hipMalloc((void **)&d_obs, (size_t)n * obs_dim * 4)); HIP(hipMalloc((void **)&d_rew, (size_t)n * 4));
  HIP(hipMalloc((void **)&d_done, (size_t)n)); HIP(hipMalloc((void **)&d_info, (size_t)n * info_dim * 4));
  float *act = (float *)calloc((size_t)n * dof, 4), *obs = (float *)malloc((size_t)n * obs_dim * 4), *rew = (float *)malloc((size_t)n * 4);
  uint8_t *done = (uint8_t *)malloc((size_t)n); int32_t *info = (int32_t *)malloc((size_t)n * info_dim * 4);

  FS(fsim_set_reset_tables(h, NULL, parts, noise, 101)); /* 101 joint-noise rows per reset (furniture.py:1580, 1606-1611) */
  FS(fsim_reset(h, NULL, d_obs));
  FS(fsim_sync(h));
  FS(fsim_set_reset_tables(h, NULL, parts, noise, 101)); /* the table of the first auto-reset: the same placement again */
  HIP(hipMemcpy(obs, d_obs, (size_t)n * obs_dim * 4, hipMemcpyDeviceToHost));
  printf("reset obs %llu\n", bits_sum(obs, (size_t)n * obs_dim));
  for (int t = 0; t < steps; t++) {
    for (int e = 0; e < n; e++) for (int k = 0; k < dof; k++) act[(size_t)e * dof + k] = (float)(((e * 31 + k * 17 + t * 7) % 21) - 10) / 10.0f; /* a fixed pattern in [-1, 1] */
    HIP(hipMemcpy(d_act, act, (size_t)n * dof * 4, hipMemcpyHostToDevice));
    FS(fsim_step(h, d_act, d_obs, d_rew, d_done, d_info));
    FS(fsim_sync(h));
    HIP(hipMemcpy(obs, d_obs, (size_t)n * obs_dim * 4, hipMemcpyDeviceToHost)); HIP(hipMemcpy(rew, d_rew, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(done, d_done, (size_t)n, hipMemcpyDeviceToHost)); HIP(hipMemcpy(info, d_info, (size_t)n * info_dim * 4, hipMemcpyDeviceToHost));
    int nd = 0, need = 0;
    for (int e = 0; e < n; e++) { nd += done[e]; need += info[(size_t)e * info_dim + FSIM_INFO_NEEDS_TABLE] != 0; }
    printf("step %d obs %llu reward %llu done %d needs_table %d (fsim_tables_needed %d)\n", t, bits_sum(obs, (size_t)n * obs_dim), bits_sum(rew, (size_t)n), nd, need, fsim_tables_needed(h));
    if (need) FS(fsim_set_reset_tables(h, NULL, parts, noise, 101));
  }
  fsim_destroy(h);
  hipFree(d_act); hipFree(d_obs); hipFree(d_rew); hipFree(d_done); hipFree(d_info);
  free(act); free(obs); free(rew); free(done); free(info); free(blob); free(tab);
  return 0;
}
