// microbenchmark for the multi-wave (4 waves = 1 env) kernel variant:
//   (1) what an s_barrier between the 4 waves of a workgroup costs (all busy / one busy + three parked);
//   (2) when 4-wave workgroups get placed while a kernel of 1-wave, 256-VGPR, 20 KB-LDS workgroups fills the chip
//       (two streams; launch order; stream priority; a third kernel already resident).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mw scripts/dev/micro/mw.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ __launch_bounds__(256) void k_barrier(long long *out, int reps, int work) {
  extern __shared__ float L[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = lane;
  __syncthreads();
  long long t0 = clock64();
  for (int r = 0; r < reps; r++) {
    // wave 0 works `work` dependent LDS round trips, the others go straight to the barrier
    if (wave == 0) for (int k = 0; k < work; k++) { L[lane] = acc; acc = L[(lane + 1) & 63] + 1.0f; }
    __syncthreads();
  }
  long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
  if (acc == -1.0f) out[0] = 0;
}

// spin for `us` microseconds of the 100 MHz wall clock; record start / end
__global__ __launch_bounds__(64, 2) __attribute__((amdgpu_num_vgpr(256))) void k_bulk(long long *ts, int us) {
  extern __shared__ float L[];
  long long t0 = wall_clock64();
  L[threadIdx.x] = 1.0f;
  while (wall_clock64() - t0 < 100LL * us) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { ts[2 * blockIdx.x] = t0; ts[2 * blockIdx.x + 1] = wall_clock64(); }
}
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(256))) void k_mw(long long *ts, int us) {
  extern __shared__ float L[];
  long long t0 = wall_clock64();
  L[threadIdx.x] = 1.0f;
  while (wall_clock64() - t0 < 100LL * us) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) { ts[2 * blockIdx.x] = t0; ts[2 * blockIdx.x + 1] = wall_clock64(); }
}

static void stats(const char *name, std::vector<long long> &h, int n, long long t_ref) {
  std::vector<double> st(n), en(n);
  for (int i = 0; i < n; i++) { st[i] = (h[2 * i] - t_ref) / 100.0; en[i] = (h[2 * i + 1] - t_ref) / 100.0; }
  std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
  printf("    %-6s n=%5d start us: min %8.1f p50 %8.1f p90 %8.1f max %8.1f | end max %8.1f\n", name, n, st[0], st[n / 2], st[n * 9 / 10], st[n - 1], en[n - 1]);
}

int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("%s CUs %d\n", pr.name, pr.multiProcessorCount);
  { // (1)
    long long *d; hipMalloc(&d, 8 * 4 * 1024);
    for (int blocks : {1, 256, 512}) for (int work : {0, 4, 32}) {
      const int reps = 2000;
      hipLaunchKernelGGL(k_barrier, dim3(blocks), dim3(256), 1024, 0, d, reps, work); hipDeviceSynchronize();
      hipLaunchKernelGGL(k_barrier, dim3(blocks), dim3(256), 1024, 0, d, reps, work); hipDeviceSynchronize();
      std::vector<long long> h(4 * blocks); hipMemcpy(h.data(), d, 32 * blocks, hipMemcpyDeviceToHost);
      double s = 0; for (int b = 0; b < blocks; b++) s += h[4 * b];
      printf("barrier: blocks %4d work %2d round trips on wave 0: %.1f cycles per iteration (wave 0)\n", blocks, work, s / blocks / reps);
    }
    hipFree(d);
  }
  // (2)
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_bulk), hipFuncAttributeMaxDynamicSharedMemorySize, 20204);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_mw), hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 1024);
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  printf("stream priority range: least %d greatest %d\n", lo, hi);
  hipStream_t sA, sB, sC, sHi;
  hipStreamCreateWithFlags(&sA, hipStreamNonBlocking); hipStreamCreateWithFlags(&sB, hipStreamNonBlocking); hipStreamCreateWithFlags(&sC, hipStreamNonBlocking);
  hipStreamCreateWithPriority(&sHi, hipStreamNonBlocking, hi);
  const int NB = 2048, NM = 128;
  long long *dB, *dM, *dP; hipMalloc(&dB, 16 * NB); hipMalloc(&dM, 16 * NM); hipMalloc(&dP, 16 * NB);
  std::vector<long long> hB(2 * NB), hM(2 * NM), hP(2 * NB);
  for (int variant = 0; variant < 6; variant++) {
    for (int rep = 0; rep < 2; rep++) {
      hipDeviceSynchronize();
      const char *what = "";
      bool prior = variant >= 3; // a kernel of 2048 x 1-wave workgroups (1.2 ms) is already resident when the step is launched
      int v = variant % 3;
      if (prior) { hipLaunchKernelGGL(k_bulk, dim3(NB), dim3(64), 20204, sC, dP, 1200); std::this_thread::sleep_for(std::chrono::microseconds(600)); }
      hipStream_t sm = v == 2 ? sHi : sA;
      if (v == 1) { what = "bulk first, then mw"; hipLaunchKernelGGL(k_bulk, dim3(NB), dim3(64), 20204, sB, dB, 1200); hipLaunchKernelGGL(k_mw, dim3(NM), dim3(256), 24 * 1024, sm, dM, 3000); }
      else { what = v == 2 ? "mw first (high-priority stream), then bulk" : "mw first, then bulk"; hipLaunchKernelGGL(k_mw, dim3(NM), dim3(256), 24 * 1024, sm, dM, 3000); hipLaunchKernelGGL(k_bulk, dim3(NB), dim3(64), 20204, sB, dB, 1200); }
      hipDeviceSynchronize();
      hipMemcpy(hB.data(), dB, 16 * NB, hipMemcpyDeviceToHost); hipMemcpy(hM.data(), dM, 16 * NM, hipMemcpyDeviceToHost);
      long long t_ref = std::min(*std::min_element(hB.begin(), hB.end()), *std::min_element(hM.begin(), hM.end()));
      if (prior) { hipMemcpy(hP.data(), dP, 16 * NB, hipMemcpyDeviceToHost); }
      if (rep == 1) {
        printf("variant %d: %s%s\n", variant, what, prior ? " | 2048 1-wave workgroups (1.2 ms) launched 0.6 ms earlier" : "");
        stats("mw", hM, NM, t_ref); stats("bulk", hB, NB, t_ref);
        if (prior) stats("prior", hP, NB, t_ref);
      }
    }
  }
  return 0;
}
