// development (round 5): the platform primitives the shared work pool relies on, checked on the GPU box before building on them:
// (1) hipStreamWaitValue32 / hipStreamWriteValue32 on host-mapped pinned memory, (2) a resident kernel that sees a CPU store to
// host-mapped memory, (3) a device system-scope atomic the CPU sees while the kernel is still running, (4) a stream that waits
// on a value set by a kernel of ANOTHER stream, with a kernel chained behind the wait.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ long long wall64() { return (long long)wall_clock64(); }
__global__ void k_spin(volatile int *post, int *ack, long long budget) {
  const long long t0 = wall64();
  while (__hip_atomic_load((int *)post, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
    __builtin_amdgcn_s_sleep(32);
    if (wall64() - t0 > budget) { __hip_atomic_store(ack, -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); return; }
  }
  __hip_atomic_store(ack, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // stay resident a little longer: the CPU must see the ack while this kernel is still running
  const long long t1 = wall64();
  while (wall64() - t1 < 20000000) __builtin_amdgcn_s_sleep(64); // 0.2 s at 100 MHz
  __hip_atomic_store(ack + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_set(int *flag, int v, int *data) { data[0] = 42; __threadfence_system(); __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_read(int *data, int *out) { out[0] = data[0]; }
int main() {
  int *h, *d;
  CK(hipHostMalloc((void **)&h, 4096, hipHostMallocMapped));
  for (int i = 0; i < 1024; i++) h[i] = 0;
  CK(hipHostGetDevicePointer((void **)&d, h, 0));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  // (2) + (3)
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, (volatile int *)d, d + 1, 300000000LL);
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  auto t0 = std::chrono::steady_clock::now();
  __atomic_store_n(h, 1, __ATOMIC_SEQ_CST);
  int seen = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.15) { if (__atomic_load_n(h + 1, __ATOMIC_ACQUIRE)) { seen = __atomic_load_n(h + 1, __ATOMIC_ACQUIRE); break; } }
  double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
  printf("cpu store -> resident kernel -> cpu sees ack=%d after %.1f us, kernel still running=%d\n", seen, us, __atomic_load_n(h + 2, __ATOMIC_ACQUIRE) == 0);
  CK(hipStreamSynchronize(s1));
  // (1) + (4): s2 waits for a value a kernel on s1 sets; a kernel behind the wait reads what the setter wrote to device memory
  int *dd, *dout;
  CK(hipMalloc((void **)&dd, 64)); CK(hipMalloc((void **)&dout, 64)); CK(hipMemset(dd, 0, 64)); CK(hipMemset(dout, 0, 64));
  hipError_t e = hipStreamWaitValue32(s2, d + 8, 7, hipStreamWaitValueGte, 0xffffffffu);
  printf("hipStreamWaitValue32 on host-mapped memory: %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_read, dim3(1), dim3(1), 0, s2, dd, dout);
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    printf("  before the value is set: stream s2 query = %s (want not ready)\n", hipGetErrorString(hipStreamQuery(s2)));
    hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s1, d + 8, 7, dd);
    CK(hipStreamSynchronize(s2));
    int out = 0; CK(hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost));
    printf("  after: chained kernel read %d (want 42)\n", out);
  }
  e = hipStreamWriteValue32(s1, d + 9, 5, 0);
  printf("hipStreamWriteValue32: %s", hipGetErrorString(e));
  if (e == hipSuccess) { CK(hipStreamSynchronize(s1)); printf(" -> host reads %d (want 5)", h[9]); }
  printf("\n");
  // WaitValue on DEVICE memory
  int *dflag; CK(hipMalloc((void **)&dflag, 64)); CK(hipMemset(dflag, 0, 64));
  e = hipStreamWaitValue32(s2, dflag, 3, hipStreamWaitValueGte, 0xffffffffu);
  printf("hipStreamWaitValue32 on device memory: %s\n", hipGetErrorString(e));
  if (e == hipSuccess) { hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s1, dflag, 3, dd); CK(hipStreamSynchronize(s2)); printf("  released\n"); }
  printf("done\n");
  return 0;
}
