// development micro-benchmark (round 6): do CU-masked HIP streams give a launch CUs of its own on MI355X (SPX, 8 XCDs)?
// Mask bit i belongs to XCD i % 8 (the driver deals the bits round-robin over the XCDs), so a bit range [8 a, 8 b) is b - a CUs in
// every XCD.  Stream A: bits [0, 8 r); stream B: bits [32, 256).  Every workgroup records its (xcc, se, sh, cu) and spins; printed:
// the CUs each stream's workgroups ran on, whether the two sets are disjoint, and whether the two kernels overlap in time.
//   hipcc --offload-arch=gfx950 -O2 -o cumask cumask.hip && ./cumask [r]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_where(unsigned *out, long long spin) {
  const unsigned hw = __builtin_amdgcn_s_getreg(63492);  // HW_REG_HW_ID, 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg(63508); // HW_REG_XCC_ID
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static int where(unsigned hw, unsigned xcc) { return ((xcc & 15) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15); }

int main(int argc, char **argv) {
  const int r = argc > 1 ? atoi(argv[1]) : 1;
  hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
  printf("device %s, %d CUs\n", pr.name, pr.multiProcessorCount);
  std::vector<uint32_t> ma(8, 0), mb(8, 0);
  for (int i = 0; i < 8 * r; i++) ma[i / 32] |= 1u << (i % 32);
  for (int i = 32; i < 256; i++) mb[i / 32] |= 1u << (i % 32);
  hipStream_t sa, sb;
  CHK(hipExtStreamCreateWithCUMask(&sa, 8, ma.data()));
  CHK(hipExtStreamCreateWithCUMask(&sb, 8, mb.data()));
  const int na = 64, nb = 2048;
  unsigned *da, *db; CHK(hipMalloc(&da, na * 8)); CHK(hipMalloc(&db, nb * 8));
  hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
  for (int rep = 0; rep < 2; rep++) {
    CHK(hipEventRecord(b0, sb)); hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, sb, db, 20000LL /* 200 us */); CHK(hipEventRecord(b1, sb));
    CHK(hipEventRecord(a0, sa)); hipLaunchKernelGGL(k_where, dim3(na), dim3(256), 0, sa, da, 20000LL); CHK(hipEventRecord(a1, sa));
    CHK(hipDeviceSynchronize());
  }
  float ta, tb, tab; hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb, b0, b1); hipEventElapsedTime(&tab, b0, a1);
  std::vector<unsigned> ha(2 * na), hb(2 * nb);
  CHK(hipMemcpy(ha.data(), da, na * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hb.data(), db, nb * 8, hipMemcpyDeviceToHost));
  std::set<int> A, B; int perx_a[16] = {0}, perx_b[16] = {0};
  for (int i = 0; i < na; i++) A.insert(where(ha[2 * i], ha[2 * i + 1]));
  for (int i = 0; i < nb; i++) B.insert(where(hb[2 * i], hb[2 * i + 1]));
  int common = 0; for (int x : A) { common += B.count(x); perx_a[x >> 12]++; } for (int x : B) perx_b[x >> 12]++;
  printf("stream A (mask bits [0, %d)): %d workgroups of 200 us ran on %zu CUs in %.3f ms; per XCD:", 8 * r, na, A.size(), ta); for (int x = 0; x < 8; x++) printf(" %d", perx_a[x]); printf("\n");
  printf("stream B (mask bits [32, 256)): %d workgroups ran on %zu CUs in %.3f ms; per XCD:", nb, B.size(), tb); for (int x = 0; x < 8; x++) printf(" %d", perx_b[x]); printf("\n");
  printf("CUs in both sets: %d; B launched first, A right behind it: A ended %.3f ms after B began (B alone lasts %.3f: A did %s wait for B)\n", common, tab, tb, tab < 0.8f * tb ? "NOT" : "");
  return 0;
}
