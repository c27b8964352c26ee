// Development micro-benchmark (not part of the product): wave-level dense Cholesky solve variants for the
// Newton step, n ~ 39, one wavefront per workgroup, LDS footprint padded to the real kernel's (occupancy 6/CU).
//   hipcc --offload-arch=gfx950 -O3 -o bench_chol scripts/dev/bench_chol.hip && ./bench_chol
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define DEV __device__ __forceinline__
#define SYNC() __syncthreads()
DEV float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
#define FS_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)
DEV float wave_sum(float v) {
  int x;
#define FS_STEP(ctrl, rmask) x = FS_DPP(0, __float_as_int(v), ctrl, rmask); v += __int_as_float(x)
  FS_STEP(0xB1, 0xf); FS_STEP(0x4E, 0xf); FS_STEP(0x141, 0xf); FS_STEP(0x140, 0xf); FS_STEP(0x142, 0xa); FS_STEP(0x143, 0xc);
#undef FS_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- variant A: lane = row, left-looking, factor in LDS (dense single island), rsqrt pivots
DEV void solveA(float *H, const float *g, float *p, int n, int lane) {
  const int i = lane; const bool row = i < n; const int ri = i * (i + 1) / 2;
  float mydinv = 0;
  for (int j = 0; j < n; j++) {
    const int rj = j * (j + 1) / 2;
    float s = 0;
    if (row && i >= j) {
      const float *Hi = H + ri, *Hj = H + rj;
      float s0 = Hi[j], s1 = 0;
      for (int k = 0; k < j; k += 6) {
        float hi[6], hj[6];
#pragma unroll
        for (int q = 0; q < 6; q++) { int kk = min(k + q, j - 1); hi[q] = Hi[kk]; hj[q] = Hj[kk]; }
#pragma unroll
        for (int q = 0; q < 6; q++) { float pr = (k + q < j) ? hi[q] * hj[q] : 0.0f; if (q & 1) s1 -= pr; else s0 -= pr; }
      }
      s = s0 + s1;
    }
    float djj = rl(s, j);
    float rinv = rsqrtf(fmaxf(djj, 1e-30f));
    if (row) { if (i == j) { H[rj + j] = djj * rinv; mydinv = rinv; } else if (i > j) H[ri + j] = s * rinv; }
    SYNC();
  }
  float b = row ? -g[i] : 0.0f;
  for (int j = 0; j < n; j++) { float yj = rl(b * mydinv, j); if (i == j) b = yj; else if (i > j && row) b -= H[ri + j] * yj; }
  for (int j = n - 1; j >= 0; j--) { float pj = rl(b * mydinv, j); if (i == j) b = pj; else if (i < j) b -= H[j * (j + 1) / 2 + i] * pj; }
  if (row) p[i] = b;
  SYNC();
}

// ---- variant C: lane = row, RIGHT-looking, whole factor in registers, readlane broadcasts, DPP back-substitution
template <int NMAX>
DEV void solveC(const float *H, const float *g, float *p, int n, int lane) {
  const int i = lane; const bool row = i < n; const int ri = i * (i + 1) / 2;
  float Lr[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; k++) Lr[k] = (row && k <= i) ? H[ri + k] : 0.0f;
  float mydinv = 0;
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n) {
      float d = rl(Lr[j], j);
      float rinv = rsqrtf(fmaxf(d, 1e-30f));
      float lij = Lr[j] * rinv; // column j of L, lane i holds L[i][j] (lanes i < j hold zeros * rinv = 0)
      Lr[j] = lij;
      if (i == j) mydinv = rinv;
#pragma unroll
      for (int k = j + 1; k < NMAX; k++) Lr[k] -= lij * rl(lij, k); // harmless garbage above the diagonal (k > i)
    }
  }
  float b = row ? -g[i] : 0.0f;
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n) {
      float yj = rl(b * mydinv, j);
      b = (i == j) ? yj : (i > j ? b - Lr[j] * yj : b);
    }
  }
#pragma unroll
  for (int jj = 0; jj < NMAX; jj++) {
    const int j = NMAX - 1 - jj;
    if (j < n) {
      float t = (i > j && row) ? Lr[j] * b : 0.0f; // b holds p_i for i > j already
      float s = wave_sum(t);
      if (i == j) b = (b - s) * mydinv;
    }
  }
  if (row) p[i] = b;
  SYNC();
}

// ---- variant B: lane = row, left-looking in registers (round-1 version)
template <int NMAX>
DEV void solveB(float *H, const float *g, float *p, int n, int lane) {
  const int i = lane; const bool row = i < n; const int ri = i * (i + 1) / 2;
  float Lr[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; k++) Lr[k] = (row && k <= i) ? H[ri + k] : 0.0f;
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n) {
      float s = Lr[j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= Lr[k] * rl(Lr[k], j);
      float djj = rl(s, j);
      float ljj = sqrtf(fmaxf(djj, 1e-30f));
      Lr[j] = (i == j) ? ljj : s / ljj;
    }
  }
  float b = row ? -g[i] : 0.0f;
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n) { float yj = rl(b, j) / rl(Lr[j], j); b = (i == j) ? yj : (i > j ? b - Lr[j] * yj : b); }
  }
#pragma unroll
  for (int k = 0; k < NMAX; k++) if (row && k <= i) H[ri + k] = Lr[k];
  SYNC();
  for (int j = n - 1; j >= 0; j--) { const int rj = j * (j + 1) / 2; float pj = rl(b, j) / H[rj + j]; if (i == j) b = pj; else if (i < j) b -= H[rj + i] * pj; }
  if (row) p[i] = b;
  SYNC();
}

template <int V>
__global__ __launch_bounds__(64, 2) void k_bench(const float *H0, const float *g0, float *pout, long long *cyc, int n, int reps) {
  extern __shared__ float L[];
  float *H = L, *g = L + 2100, *p = L + 2200;
  int lane = threadIdx.x, nH = n * (n + 1) / 2;
  long long tot = 0;
  for (int r = 0; r < reps; r++) {
    for (int e = lane; e < nH; e += 64) H[e] = H0[e];
    if (lane < n) g[lane] = g0[lane] * (1.0f + 0.001f * r);
    SYNC();
    long long t0 = clock64();
    if (V == 0) solveA(H, g, p, n, lane);
    else if (V == 1) solveB<48>(H, g, p, n, lane);
    else if (V == 2) solveC<48>(H, g, p, n, lane);
    else solveC<64>(H, g, p, n, lane);
    tot += clock64() - t0;
  }
  if (lane < n) pout[blockIdx.x * 64 + lane] = p[lane];
  if (lane == 0) cyc[blockIdx.x] = tot / reps;
}

int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 39, blocks = 4096, reps = 20;
  int nH = n * (n + 1) / 2;
  std::vector<double> A(n * n), Hd(n * n, 0.0);
  srand(1);
  for (auto &a : A) a = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = (i == j) ? 0.5 : 0; for (int k = 0; k < n; k++) s += A[i * n + k] * A[j * n + k]; Hd[i * n + j] = s; }
  std::vector<float> Hp(nH), g(n);
  for (int i = 0; i < n; i++) { g[i] = (float)(rand() / (double)RAND_MAX - 0.5); for (int j = 0; j <= i; j++) Hp[i * (i + 1) / 2 + j] = (float)Hd[i * n + j]; }
  // reference solve in double
  std::vector<double> Ld(Hd), x(n);
  for (int j = 0; j < n; j++) { for (int k = 0; k < j; k++) for (int i = j; i < n; i++) Ld[i * n + j] -= Ld[i * n + k] * Ld[j * n + k]; double d = sqrt(Ld[j * n + j]); for (int i = j; i < n; i++) Ld[i * n + j] /= d; }
  for (int i = 0; i < n; i++) { double s = -g[i]; for (int k = 0; k < i; k++) s -= Ld[i * n + k] * x[k]; x[i] = s / Ld[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= Ld[k * n + i] * x[k]; x[i] = s / Ld[i * n + i]; }
  float *dH, *dg, *dp; long long *dc;
  hipMalloc(&dH, nH * 4); hipMalloc(&dg, n * 4); hipMalloc(&dp, blocks * 64 * 4); hipMalloc(&dc, blocks * 8);
  hipMemcpy(dH, Hp.data(), nH * 4, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), n * 4, hipMemcpyHostToDevice);
  const char *names[4] = {"A lds left-looking", "B regs left-looking<48>", "C regs right-looking<48>", "C regs right-looking<64>"};
  int lds = 24000;
  for (int v = 0; v < 4; v++) {
    if ((v == 1 || v == 2) && n > 48) continue;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(k_bench<0>, dim3(blocks), dim3(64), lds, 0, dH, dg, dp, dc, n, reps);
      if (v == 1) hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(64), lds, 0, dH, dg, dp, dc, n, reps);
      if (v == 2) hipLaunchKernelGGL(k_bench<2>, dim3(blocks), dim3(64), lds, 0, dH, dg, dp, dc, n, reps);
      if (v == 3) hipLaunchKernelGGL(k_bench<3>, dim3(blocks), dim3(64), lds, 0, dH, dg, dp, dc, n, reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(blocks); std::vector<float> p(64);
    hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(p.data(), dp, 64 * 4, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x_ : c) mean += x_; mean /= blocks;
    double err = 0, nrm = 0; for (int i = 0; i < n; i++) { err = fmax(err, fabs(p[i] / (1.0 + 0.001 * (reps - 1)) - x[i])); nrm = fmax(nrm, fabs(x[i])); }
    printf("n=%d %-28s mean cycles/solve %.0f  kernel %.3f ms (%d blocks x %d reps => %.2f us per solve-slot)  rel err %.2e\n", n, names[v], mean, ms, blocks, reps, ms * 1e3 / reps / (blocks / 1536.0), err / nrm);
  }
  return 0;
}
