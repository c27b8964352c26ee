// microbenchmark: is a wave running straight-line code bound by instruction fetch once the loop body exceeds the I-cache?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define KERNEL(N)                                                                                                      \
  __global__ __launch_bounds__(64) void k##N(float *out, long long *cyc, int reps, float kk) {                         \
    float a = threadIdx.x, b = 1, c = 2, d = 3;                                                                        \
    long long t0 = clock64();                                                                                          \
    asm volatile("s_mov_b32 s20, %5\n s_getpc_b64 s[22:23]\n"                                                          \
                 ".rept " #N "\n v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n" \
                 " v_fma_f32 %0, %0, %4, 1.0\n v_fma_f32 %1, %1, %4, 2.0\n v_fma_f32 %2, %2, %4, 0.5\n v_fma_f32 %3, %3, %4, 4.0\n .endr\n"      \
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc0 1f\n s_setpc_b64 s[22:23]\n 1:\n"       \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(kk), "s"(reps) : "s20", "s22", "s23", "scc");              \
    long long t1 = clock64();                                                                                          \
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;                                                                \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                   \
  }
KERNEL(64) KERNEL(256) KERNEL(512) KERNEL(768) KERNEL(1024) KERNEL(1536) KERNEL(2048) KERNEL(4096)
template <class F> void bench(F kern, int N, int blocks) {
  float *out; long long *cyc;
  hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
  const int reps = 50;
  for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc, reps, 0.999f); hipDeviceSynchronize(); }
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  double wall = 0; printf("body %6d instr (%4d KB) blocks %5d: %.2f cycles/instr\n", 8 * N, 8 * N * 8 / 1024, blocks, s / blocks / reps / (8.0 * N));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {1024, 2048, 3072, 4096, 8192}) {
    bench(k256, 256, blocks);
  }
  return 0;
}
