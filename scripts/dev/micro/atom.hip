// microbenchmark: what does a run of LDS float atomics (ds_add_f32, no return) cost one wave, as a function of how many lanes are
// active and how many of them hit the SAME word (the Hessian's body blocks: 21 atomics per contact lane, the contacts of one body
// all on the same 21 words)?  Beside it: the same values stored without conflict (ds_write_b32, odd stride) and a quad
// pre-reduction by DPP before the atomic.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/dev/micro/atom scripts/dev/micro/atom.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE> __global__ __launch_bounds__(64) void k_atom(long long *out, int reps, int nact, int nbody) {
  __shared__ float L[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) L[i] = 0;
  __syncthreads();
  float v[21];
  for (int k = 0; k < 21; k++) v[k] = 1.0f + 0.001f * (lane + k);
  const bool act = lane < nact;
  // contacts of a body sit in adjacent lanes (manifolds of up to 4 points), nbody distinct targets
  const int per = (nact + nbody - 1) / nbody;
  const int b = lane / per;
  float *A = L + 21 * b;
  float *P = L + 1024 + 21 * lane;
  long long t0 = clock64();
  for (int r = 0; r < reps; r++) {
    if (MODE == 0) {
      if (act)
#pragma unroll
        for (int k = 0; k < 21; k++) atomicAdd(A + k, v[k]);
    } else if (MODE == 1) {
      if (act)
#pragma unroll
        for (int k = 0; k < 21; k++) P[k] = v[k];
    } else {
      // quad pre-reduction: lanes 4q..4q+3 are assumed to share the target (exact when per is a multiple of 4)
#pragma unroll
      for (int k = 0; k < 21; k++) {
        float x = act ? v[k] : 0.0f;
        x += __shfl_xor(x, 1, 64);
        x += __shfl_xor(x, 2, 64);
        if (act && (lane & 3) == 0) atomicAdd(A + k, x);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
  long long t1 = clock64();
  if (lane == 0) out[0] = t1 - t0;
  if (L[lane] == -1.0f) out[1] = 0;
}

int main() {
  long long *d, h[2];
  hipMalloc(&d, 16);
  const int reps = 200;
  for (int nact : {8, 16, 25, 48, 64})
    for (int nbody : {1, 2, 4, 6, 12, 25, 64}) {
      if (nbody > nact) continue;
      double c[3];
      for (int mode = 0; mode < 3; mode++) {
        for (int w = 0; w < 2; w++) {
          if (mode == 0) hipLaunchKernelGGL(k_atom<0>, dim3(1), dim3(64), 0, 0, d, reps, nact, nbody);
          if (mode == 1) hipLaunchKernelGGL(k_atom<1>, dim3(1), dim3(64), 0, 0, d, reps, nact, nbody);
          if (mode == 2) hipLaunchKernelGGL(k_atom<2>, dim3(1), dim3(64), 0, 0, d, reps, nact, nbody);
          hipDeviceSynchronize();
        }
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        c[mode] = (double)h[0] / reps;
      }
      printf("active lanes %2d on %2d bodies (%2d per word): 21 atomics %7.0f cycles (%5.1f each) | 21 plain stores %6.0f | quad pre-reduction + atomics %7.0f\n", nact, nbody,
             (nact + nbody - 1) / nbody, c[0], c[0] / 21, c[1], c[2]);
    }
  return 0;
}
