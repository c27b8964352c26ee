// fsim_math.hpp -- small fp32 vector / quaternion / spatial-algebra helpers (device).
#pragma once
#include <hip/hip_runtime.h>

#define FS_MINVAL 1e-15f
#define DEV __device__ __forceinline__
// waves per SIMD the step kernels and their out-of-line callees are register-allocated for (2: 256 VGPRs, 3: 168, 4: 128)
#ifndef FSIM_WPE
#define FSIM_WPE 2
#endif
#define FSIM_OUTLINE __device__ __noinline__

// Model tables live in global memory, but a pointer loaded from a struct that was itself reached through a pointer is
// "flat" to the compiler: every such load becomes flat_load_dword, which also ticks the LDS counter (lgkmcnt), so LDS
// waits get stuck behind HBM/L2 latencies.  GP() launders the pointer through address space 1 => global_load_dword.
template <class T> __device__ __forceinline__ const __attribute__((address_space(1))) T *GP(const T *p) {
  return (const __attribute__((address_space(1))) T *)p;
}
template <class U, class T> __device__ __forceinline__ const __attribute__((address_space(1))) U *GPC(const T *p) { // + element type change
  return (const __attribute__((address_space(1))) U *)p;
}

typedef float f4_t __attribute__((ext_vector_type(4))); // plain vector types: loadable through any address space
typedef int i4_t __attribute__((ext_vector_type(4)));

struct V3 { float x, y, z; };
DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
template <class P> DEV V3 ldv3(P p) { return v3(p[0], p[1], p[2]); } // P: generic, LDS or global (GP) float pointer
DEV void stv3(float *p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
DEV V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float norm(V3 a) { return sqrtf(dot(a, a)); }
DEV V3 normalized(V3 a, float *len = nullptr) {
  float n = norm(a);
  if (len) *len = n;
  if (n < 1e-30f) return v3(1, 0, 0);
  return a * (1.0f / n);
}
DEV float comp(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

struct Q4 { float w, x, y, z; };
DEV Q4 q4(float w, float x, float y, float z) { Q4 q; q.w = w; q.x = x; q.y = y; q.z = z; return q; }
template <class P> DEV Q4 ldq(P p) { return q4(p[0], p[1], p[2], p[3]); }
DEV void stq(float *p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
DEV Q4 qmul(Q4 a, Q4 b) {
  return q4(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w);
}
DEV Q4 qconj(Q4 a) { return q4(a.w, -a.x, -a.y, -a.z); }
DEV Q4 qnormalized(Q4 q) {
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < 1e-30f) return q4(1, 0, 0, 0);
  float s = 1.0f / n;
  return q4(q.w * s, q.x * s, q.y * s, q.z * s);
}
DEV Q4 axisangle(V3 ax, float ang) {
  float s, c;
  sincosf(0.5f * ang, &s, &c);
  return q4(c, ax.x * s, ax.y * s, ax.z * s);
}
DEV V3 qrot(Q4 q, V3 v) { // rotate v by unit quaternion
  V3 u = v3(q.x, q.y, q.z);
  V3 t = 2.0f * cross(u, v);
  return v + q.w * t + cross(u, t);
}

// row-major 3x3
struct M3 { float m[9]; };
DEV M3 q2m(Q4 q) {
  M3 R;
  float w = q.w, x = q.x, y = q.y, z = q.z;
  R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - w * z); R.m[2] = 2 * (x * z + w * y);
  R.m[3] = 2 * (x * y + w * z); R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - w * x);
  R.m[6] = 2 * (x * z - w * y); R.m[7] = 2 * (y * z + w * x); R.m[8] = 1 - 2 * (x * x + y * y);
  return R;
}
template <class P> DEV M3 ldm3(P p) { M3 R; for (int i = 0; i < 9; i++) R.m[i] = p[i]; return R; }
DEV void stm3(float *p, const M3 &R) { for (int i = 0; i < 9; i++) p[i] = R.m[i]; }
DEV V3 mulv(const M3 &R, V3 v) {
  return v3(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z, R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z);
}
DEV V3 multv(const M3 &R, V3 v) {
  return v3(R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z, R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z);
}
DEV V3 colv(const M3 &R, int k) { return v3(R.m[k], R.m[3 + k], R.m[6 + k]); }
DEV M3 mulm(const M3 &A, const M3 &B) {
  M3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}

// spatial vectors [ang; lin] referenced at a tree's centre of mass
struct S6 { V3 a, l; };
DEV S6 lds6(const float *p) { S6 s; s.a = ldv3(p); s.l = ldv3(p + 3); return s; }
DEV void sts6(float *p, S6 s) { stv3(p, s.a); stv3(p + 3, s.l); }
DEV S6 s6zero() { S6 s; s.a = v3(0, 0, 0); s.l = v3(0, 0, 0); return s; }
DEV S6 operator+(S6 x, S6 y) { S6 s; s.a = x.a + y.a; s.l = x.l + y.l; return s; }
DEV S6 operator*(S6 x, float k) { S6 s; s.a = x.a * k; s.l = x.l * k; return s; }
DEV float dot6(S6 x, S6 y) { return dot(x.a, y.a) + dot(x.l, y.l); }
DEV S6 cross_motion(S6 v, S6 m) { S6 r; r.a = cross(v.a, m.a); r.l = cross(v.a, m.l) + cross(v.l, m.a); return r; }
DEV S6 cross_force(S6 v, S6 f) { S6 r; r.a = cross(v.a, f.a) + cross(v.l, f.l); r.l = cross(v.a, f.l); return r; }
// inertia record: Ixx Iyy Izz Ixy Ixz Iyz hx hy hz m  (about the reference point, h = m*(c - ref))
DEV S6 inert_mul(const float *I, S6 v) {
  V3 h = v3(I[6], I[7], I[8]);
  S6 f;
  V3 Iw = v3(I[0] * v.a.x + I[3] * v.a.y + I[4] * v.a.z, I[3] * v.a.x + I[1] * v.a.y + I[5] * v.a.z, I[4] * v.a.x + I[5] * v.a.y + I[2] * v.a.z);
  f.a = Iw + cross(h, v.l);
  f.l = I[9] * v.l - cross(h, v.a);
  return f;
}

// Wave-wide reductions on the DPP network (no LDS crossbar round trips): two quad permutes, half-row mirror and row
// mirror give every lane its 16-lane row total; row_bcast15 / row_bcast31 chain the four rows into lane 63, whose value
// is broadcast through an SGPR.  ~8 VALU instructions instead of six dependent ds_bpermute shuffles.  All 64 lanes
// must be active at the call site.
#define FS_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)
DEV float wave_sum(float v) {
  int x;
#define FS_STEP(ctrl, rmask) x = FS_DPP(0, __float_as_int(v), ctrl, rmask); v += __int_as_float(x)
  FS_STEP(0xB1, 0xf);  // quad_perm [1,0,3,2]
  FS_STEP(0x4E, 0xf);  // quad_perm [2,3,0,1]
  FS_STEP(0x141, 0xf); // row_half_mirror
  FS_STEP(0x140, 0xf); // row_mirror
  FS_STEP(0x142, 0xa); // row_bcast15 -> rows 1, 3
  FS_STEP(0x143, 0xc); // row_bcast31 -> rows 2, 3
#undef FS_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
DEV float wave_max(float v) {
  int x;
#define FS_STEP(ctrl, rmask) x = FS_DPP(__float_as_int(v), __float_as_int(v), ctrl, rmask); v = fmaxf(v, __int_as_float(x))
  FS_STEP(0xB1, 0xf);
  FS_STEP(0x4E, 0xf);
  FS_STEP(0x141, 0xf);
  FS_STEP(0x140, 0xf);
  FS_STEP(0x142, 0xa);
  FS_STEP(0x143, 0xc);
#undef FS_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
DEV int wave_or(int v) {
#define FS_STEP(ctrl, rmask) v |= FS_DPP(0, v, ctrl, rmask)
  FS_STEP(0xB1, 0xf);
  FS_STEP(0x4E, 0xf);
  FS_STEP(0x141, 0xf);
  FS_STEP(0x140, 0xf);
  FS_STEP(0x142, 0xa);
  FS_STEP(0x143, 0xc);
#undef FS_STEP
  return __builtin_amdgcn_readlane(v, 63);
}
