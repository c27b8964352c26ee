// test tool (tests/test_lds_clean_gpu.py; built by __graft_entry__.build() into tests/liblds_poison.so): fill the LDS of every CU with a bit pattern.
// A kernel that reads LDS words it never wrote then shows up as a result that depends on the pattern.
#include <hip/hip_runtime.h>
__global__ void k_poison(unsigned pattern, int words) {
  extern __shared__ unsigned P[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) P[i] = pattern;
  __syncthreads();
  if (P[(threadIdx.x * 7) % words] != pattern) __builtin_trap();
}
extern "C" int lds_poison(unsigned pattern) {
  const int bytes = 64 * 1024;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_poison, dim3(256 * 8), dim3(256), bytes, 0, pattern, bytes / 4);
  return (int)hipDeviceSynchronize();
}
