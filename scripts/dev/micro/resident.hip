// development (round 5): what can the host still do while a kernel is RESIDENT on (almost) every CU?  A spinning kernel of 496 workgroups x
// 256 threads with 80 KB of LDS each (the work pool's footprint: two per CU, 16 slots spare) sits on a non-blocking stream; the host
// then times small copies and launches issued in the ways the repo and torch issue them.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256, 2) void k_resident(volatile int *stop, long long budget, float *sink) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = (long long)wall_clock64();
  while (__hip_atomic_load((int *)stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && (long long)wall_clock64() - t0 < budget) __builtin_amdgcn_s_sleep(100);
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = lds[5];
}
__global__ void k_small(int *p) { p[threadIdx.x] = threadIdx.x; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 496;
  int *h, *d;
  CK(hipHostMalloc((void **)&h, 4096, hipHostMallocMapped)); h[0] = 0;
  CK(hipHostGetDevicePointer((void **)&d, h, 0));
  hipStream_t sr, s2;
  CK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float *sink; CK(hipMalloc((void **)&sink, 64));
  int *dbuf; CK(hipMalloc((void **)&dbuf, 1 << 20)); CK(hipMemset(dbuf, 1, 1 << 20));
  int *pinned; CK(hipHostMalloc((void **)&pinned, 1 << 20, 0));
  static int pageable[1 << 18];
  CK(hipFuncSetAttribute((const void *)k_resident, hipFuncAttributeMaxDynamicSharedMemorySize, 80832));
  hipLaunchKernelGGL(k_resident, dim3(wgs), dim3(256), 80832, sr, (volatile int *)d, 300000000LL /* 3 s */, sink);
  std::this_thread::sleep_for(std::chrono::milliseconds(50));
  printf("resident kernel: %d workgroups; stream query: %s\n", wgs, hipGetErrorString(hipStreamQuery(sr)));
  double t;
#define TIME(label, ...) t = now(); __VA_ARGS__; printf("  %-70s %8.3f ms\n", label, (now() - t) * 1e3); fflush(stdout);
  TIME("hipMemcpyAsync D2H 4 KB -> pinned on a non-blocking stream + sync", CK(hipMemcpyAsync(pinned, dbuf, 4096, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)));
  TIME("hipMemcpyAsync D2H 68 KB -> pinned on a non-blocking stream + sync", CK(hipMemcpyAsync(pinned, dbuf, 69632, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)));
  TIME("hipMemcpyAsync H2D 140 KB pinned on a non-blocking stream + sync", CK(hipMemcpyAsync(dbuf, pinned, 143360, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(s2)));
  TIME("hipMemcpyAsync H2D 140 KB PAGEABLE on a non-blocking stream + sync", CK(hipMemcpyAsync(dbuf, pageable, 143360, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(s2)));
  TIME("small kernel on a non-blocking stream + sync", hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s2, dbuf); CK(hipStreamSynchronize(s2)));
  TIME("hipMemcpyAsync D2H 68 KB -> pinned on the NULL stream + hipStreamSynchronize(0)", CK(hipMemcpyAsync(pinned, dbuf, 69632, hipMemcpyDeviceToHost, 0)); CK(hipStreamSynchronize(0)));
  TIME("hipMemcpy (synchronous) D2H 68 KB -> pageable", CK(hipMemcpy(pageable, dbuf, 69632, hipMemcpyDeviceToHost)));
  TIME("hipMemcpy (synchronous) H2D 4 KB pageable", CK(hipMemcpy(dbuf, pageable, 4096, hipMemcpyHostToDevice)));
  TIME("small kernel on the NULL stream + hipStreamSynchronize(0)", hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, 0, dbuf); CK(hipStreamSynchronize(0)));
  TIME("hipMalloc + hipFree 1 MB", { void *q; CK(hipMalloc(&q, 1 << 20)); CK(hipFree(q)); });
  __atomic_store_n(h, 1, __ATOMIC_SEQ_CST);
  TIME("stop flag -> resident kernel gone", CK(hipStreamSynchronize(sr)));
  return 0;
}
