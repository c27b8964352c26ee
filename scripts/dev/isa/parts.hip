// development: one kernel per solver routine, so that the ISA of each can be read and counted on its own
// (hipcc -S --cuda-device-only; scripts/dev/isa/count.py)
#include "../../../furniture_amd/csrc/fsim_solver.hpp"
#include "../../../furniture_amd/csrc/fsim_spec.hpp"
typedef SpecCtx<Spec_sawyer_table_lack_0825> C;
#define K(name, ...) extern "C" __global__ __launch_bounds__(64) void name(const DModel *mp, int a, int b, float *out) { \
  extern __shared__ float lds[]; CModel &m = *(CModel *)mp; C c(lds, m, (int)threadIdx.x, 100, 1e-8f); __VA_ARGS__ }
K(isa_chol12, { out[c.lane] = fs_chol_phase<12>(c, c.ly.hmap, a + c.lane < 39 ? c.lane : -1, c.lane & 15, b, RowBcast()); })
K(isa_chol16, { out[c.lane] = fs_chol_phase<16>(c, c.ly.hmap, a + c.lane < 39 ? c.lane : -1, c.lane & 15, b, RowBcast()); })
K(isa_cholsolve, { out[c.lane] = fs_chol_solve(c, c.ly.hmap); })
K(isa_grad, { SolSlot S = fs_load_slots(c); SlotK sk = fs_gradient(c, S); out[c.lane] = sk.K[0] + sk.K[1] + sk.K[2] + sk.K[3] + sk.K[4] + sk.K[5] + sk.zone + sk.on; })
K(isa_hess, { SolSlot S = fs_load_slots(c); SlotK sk; sk.on = a > c.lane; sk.zone = b; for (int q = 0; q < 6; q++) sk.K[q] = out[64 * q + c.lane]; fs_hessian(c, sk, S); })
K(isa_mulM, { fs_mulM(c, c.ly.Mp, c.ly.p); })
K(isa_bodyspatial, { fs_body_spatial(c, c.ly.p); })
K(isa_jdot, { SolSlot S = fs_load_slots(c); fs_jdot(c, S, c.ly.p, false); out[c.lane] = S.jp[0] + S.jp[1] + S.jp[2] + S.ljp; })
K(isa_lineeval, { SolSlot S = fs_load_slots(c); float d1, d2; bool nq; fs_line_eval(c, S, out[0], &d1, &d2, a, &nq); out[c.lane] = d1 + d2 + nq; })
K(isa_loadslots, { SolSlot S = fs_load_slots(c); out[c.lane] = S.jar[0] + S.dn + S.mu + S.fx.x + S.fy.y + S.fz.z + S.r1.x + S.r2.y + S.bt1 + S.bt2 + S.ljar + S.ld + S.aref[1] + S.aref[2] + S.tb + S.dt + S.lsign + S.laref + S.ldof + S.anyweld; })
K(isa_solve, { fs_solve(c, a); })
K(isa_kin, { fs_kinematics(c); })
K(isa_com, { fs_com_inertia(c); })
K(isa_crb, { fs_crb_factor(c); })
K(isa_vel, { fs_velocity_bias(c); })
K(isa_collide, { fs_collide(c); })
K(isa_constraints, { fs_make_constraints(c); })
#define KE(name, ...) extern "C" __global__ __launch_bounds__(64) void name(const DModel *mp, int a, int b, float *out) { \
  extern __shared__ float lds[]; CModel &m = *(CModel *)mp; C c(lds, m, (int)threadIdx.x, 100, 1e-8f); Emit<C> e(c, 8, a, b, out[1], out[2]); \
  V3 p1 = ldv3(out + 3 * c.lane), p2 = ldv3(out + 300 + 3 * c.lane), s1 = ldv3(out + 600 + 3 * c.lane), s2 = ldv3(out + 900 + 3 * c.lane); \
  M3 R1 = ldm3(out + 1200 + 9 * c.lane), R2 = ldm3(out + 2400 + 9 * c.lane); __VA_ARGS__ }
KE(isa_boxbox, { np_box_box(e, p1, R1, s1, p2, R2, s2); })
KE(isa_planebox, { np_plane_box(e, p1, R1, p2, R2, s2); })
KE(isa_cylboxsep, { out[c.lane] = np_cyl_box_separated(p1, R1, s1, p2, R2, s2, out[5]); })
KE(isa_mpr, { Shape A, B; A.type = a; A.pos = p1; A.R = R1; A.size = s1; B.type = b; B.pos = p2; B.R = R2; B.size = s2; A.verts = nullptr; A.nvert = 0; B.verts = nullptr; B.nvert = 0; np_mpr(e, A, B); })
