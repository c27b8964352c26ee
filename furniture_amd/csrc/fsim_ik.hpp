// fsim_ik.hpp -- control_type "ik" (SURVEY f3): the end-effector command -> joint-velocity path of FurnitureEnv._do_ik_step
// (furniture.py:2899-2991) with a batched damped-least-squares solver in place of pybullet.calculateInverseKinematics
// (controllers/sawyer_ik_controller.py:51-88, 177-269).  PARITY UNPINNED for the solver itself (Bullet's iteration is not
// reproducible without its source, and the arm is redundant); everything around it is restated: the URDF chain the
// reference's IK runs on (m.ik_tab, parsed by furniture_amd/mjcf/urdf_chain.py), target bookkeeping in the robot base frame,
// user_sensitivity, the Rz(-90 deg) end-effector convention, the P controller and the three closed-loop repeats.
// CPU statement of the same algorithm: oracle/ik.py (the checker).
//
// One wave = one env: the solve is ~12 iterations x (7-joint forward kinematics + 6x7 Jacobian + 6x6 Cholesky), about 20 k
// scalar-like instructions against ~2 M for the 150 physics substeps of an "ik" step, so every lane runs it redundantly
// (no LDS traffic, no synchronisation) and lane 0 stores the result.
#pragma once

// per-env block after the group table (env_extra_words)
enum { EI_TARGET = 0 /* ik_robot_target_pos, base frame */, EI_IQUAT = 3 /* _initial_right_hand_quat, raw 4 numbers */,
       EI_QCMD = 7 /* commanded_joint_positions */, EI_ACT = 14 /* scaled+permuted d_pos (3), then ik: rotation action (3), grip | ik_quaternion: quaternion wxyz (4), grip */, EI_HPOS = 22 /* right_hand world position of the last forward pass (what the next step's _bounded_d_pos reads) */,
       EI_WORDS = 26 };
// m.ik_tab layout (floats), per arm (stride IKT_ARM): joint_pos 7x3 | joint_quat 7x4 (wxyz) | eef_pos 3 | eef_quat 4 | rest 7 | lower 7 |
// upper 7; after the last arm: base_pos 3 | base_quat 4 | user_sensitivity | P gain | rest mode (0 table, 1 current joints) | Rz(-90) flag
enum { IKT_JPOS = 0, IKT_JQUAT = 21, IKT_EEF = 49, IKT_EEFQ = 52, IKT_REST = 56, IKT_LOWER = 63, IKT_UPPER = 70, IKT_ARM = 77,
       IKT_BPOS = 0, IKT_BQUAT = 3, IKT_SENS = 7, IKT_GAIN = 8, IKT_RESTMODE = 9, IKT_RZ = 10, IKT_TAIL = 11 };
#define IK_ITERS 12
#define IK_TAIL 4
#define IK_DAMP2 (0.05f * 0.05f)
#define IK_NULL_GAIN 0.01f

struct IkFk { V3 p; M3 R; V3 o[7], z[7]; };
// forward kinematics of the URDF chain: child = parent . Trans(xyz) . Rot(rpy) . Rz(q_i); end effector = CoM frame of link 6
template <class TP> DEV void ik_fk(TP tab, const float *q, IkFk &f) {
  M3 R; for (int i = 0; i < 9; i++) R.m[i] = (i & 3) == 0 ? 1.0f : 0.0f;
  V3 p = v3(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 7; i++) {
    p = p + mulv(R, ldv3(tab + IKT_JPOS + 3 * i));
    R = mulm(R, q2m(qnormalized(ldq(tab + IKT_JQUAT + 4 * i))));
    f.o[i] = p; f.z[i] = colv(R, 2);
    float s, cq;
    sincosf(q[i], &s, &cq);
    M3 Z; Z.m[0] = cq; Z.m[1] = -s; Z.m[2] = 0; Z.m[3] = s; Z.m[4] = cq; Z.m[5] = 0; Z.m[6] = 0; Z.m[7] = 0; Z.m[8] = 1;
    R = mulm(R, Z);
  }
  f.p = p + mulv(R, ldv3(tab + IKT_EEF));
  f.R = mulm(R, q2m(qnormalized(ldq(tab + IKT_EEFQ)))); // (identity for Sawyer: the product is then exact)
}
// axis * angle of a rotation matrix
DEV V3 ik_rotvec(const M3 &R) {
  V3 v = v3(R.m[7] - R.m[5], R.m[2] - R.m[6], R.m[3] - R.m[1]) * 0.5f;
  float s = norm(v), cth = 0.5f * (R.m[0] + R.m[4] + R.m[8] - 1.0f);
  if (s < 1e-12f) return v;
  return v * (atan2f(s, cth) / s);
}
// 6x6 SPD solve in place (packed lower triangle A[21], rhs b[6] -> x)
DEV void ik_chol6(float *A) {
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float d = A[j * (j + 1) / 2 + j];
#pragma unroll
    for (int p = 0; p < j; p++) d -= A[j * (j + 1) / 2 + p] * A[j * (j + 1) / 2 + p];
    const float dinv = rsqrtf(fmaxf(d, 1e-30f));
    A[j * (j + 1) / 2 + j] = dinv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float s = A[i * (i + 1) / 2 + j];
#pragma unroll
      for (int p = 0; p < j; p++) s -= A[i * (i + 1) / 2 + p] * A[j * (j + 1) / 2 + p];
      A[i * (i + 1) / 2 + j] = s * dinv;
    }
  }
}
DEV void ik_solve6(const float *A, float *b) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float s = b[i];
#pragma unroll
    for (int p = 0; p < i; p++) s -= A[i * (i + 1) / 2 + p] * b[p];
    b[i] = s * A[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float s = b[i];
#pragma unroll
    for (int p = i + 1; p < 6; p++) s -= A[p * (p + 1) / 2 + i] * b[p];
    b[i] = s * A[i * (i + 1) / 2 + i];
  }
}
// hand (right_hand body) pose of the last forward pass
template <class Ctx> DEV void ik_hand_world(const Ctx &c, int arm, V3 *pos, M3 *R) {
  CModel &m = c.m;
  const int hb = GP(m.hand_body)[arm], rb = GP(m.body_red)[hb];
  const M3 Rb = ldm3(c.L + c.ly.xmat + 9 * rb);
  *pos = ldv3(c.L + c.ly.xpos + 3 * rb) + mulv(Rb, ldv3(GP(m.body_relpos) + 3 * hb));
  *R = mulm(Rb, q2m(qnormalized(ldq(GP(m.body_relquat) + 4 * hb))));
}
// transform_utils.mat2quat (xyzw, w >= 0): the unit quaternion of a rotation matrix
DEV void ik_mat2quat_xyzw(const M3 &R, float *q) {
  float tr = R.m[0] + R.m[4] + R.m[8], w, x, y, z;
  if (tr > 0) { float s = sqrtf(tr + 1.0f) * 2; w = 0.25f * s; x = (R.m[7] - R.m[5]) / s; y = (R.m[2] - R.m[6]) / s; z = (R.m[3] - R.m[1]) / s; }
  else if (R.m[0] > R.m[4] && R.m[0] > R.m[8]) { float s = sqrtf(1.0f + R.m[0] - R.m[4] - R.m[8]) * 2; w = (R.m[7] - R.m[5]) / s; x = 0.25f * s; y = (R.m[1] + R.m[3]) / s; z = (R.m[2] + R.m[6]) / s; }
  else if (R.m[4] > R.m[8]) { float s = sqrtf(1.0f + R.m[4] - R.m[0] - R.m[8]) * 2; w = (R.m[2] - R.m[6]) / s; x = (R.m[1] + R.m[3]) / s; y = 0.25f * s; z = (R.m[5] + R.m[7]) / s; }
  else { float s = sqrtf(1.0f + R.m[8] - R.m[0] - R.m[4]) * 2; w = (R.m[3] - R.m[1]) / s; x = (R.m[2] + R.m[6]) / s; y = (R.m[5] + R.m[7]) / s; z = 0.25f * s; }
  if (w < 0) { w = -w; x = -x; y = -y; z = -z; }
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

// end of _reset (furniture.py:1643-1650): _initial_<arm>_hand_quat = _<arm>_hand_quat; controller.sync_state()
// target_only: controller.sync_state() alone (what _connect calls after config.reset_robot_after_attach re-posed the arm, furniture.py:921-924)
template <class Ctx> DEV void env_ik_sync(const Ctx &c, const bool target_only = false) {
  CModel &m = c.m;
  const auto tail = GP(m.ik_tab) + IKT_ARM * c.D.narm;
  const M3 RbT = ck_transpose(q2m(qnormalized(ldq(tail + IKT_BQUAT))));
  for (int arm = 0; arm < c.D.narm; arm++) {
    float *K = c.L + c.ly.eik + EI_WORDS * arm;
    V3 hp; M3 hR;
    ik_hand_world(c, arm, &hp, &hR);
    float iq[4];
    ik_mat2quat_xyzw(mulm(RbT, hR), iq); // hand orientation in the frame of body "base" (furniture.py:3380-3457)
    float q[7];
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = c.L[c.ly.qpos + GP(m.arm_qposadr)[7 * arm + i]];
    IkFk f;
    ik_fk(GP(m.ik_tab) + IKT_ARM * arm, q, f);
    if (c.lane == 0) {
      stv3(K + EI_TARGET, f.p); // ik_robot_target_pos := the IK chain's own end-effector position (sync_state)
      if (!target_only) {
        for (int i = 0; i < 4; i++) K[EI_IQUAT + i] = iq[i];
        stv3(K + EI_HPOS, hp);
      }
    }
  }
  SYNC();
}

// End of a step: what the NEXT _do_ik_step will read from sim.data before any new forward pass -- the hand position (for
// _bounded_d_pos) and, for ik_quaternion, _right_hand_quat -- i.e. the poses of the last forward pass, one integration old.
// LDS does not survive the launch, so they are kept in the env record.
template <class Ctx> DEV void env_ik_remember(const Ctx &c, int mode) {
  CModel &m = c.m;
  const auto tail = GP(m.ik_tab) + IKT_ARM * c.D.narm;
  const M3 RbT = ck_transpose(q2m(qnormalized(ldq(tail + IKT_BQUAT))));
  for (int arm = 0; arm < c.D.narm; arm++) {
    float *K = c.L + c.ly.eik + EI_WORDS * arm;
    V3 hp; M3 hR;
    ik_hand_world(c, arm, &hp, &hR);
    float rh[4];
    ik_mat2quat_xyzw(mulm(RbT, hR), rh);
    if (c.lane == 0) {
      stv3(K + EI_HPOS, hp);
      if (mode == 2) for (int i = 0; i < 4; i++) K[EI_IQUAT + i] = rh[i];
    }
  }
  SYNC();
}

// get_control(dpos, rotation) (sawyer_ik_controller.py:51-88): new target, solve, store commanded_joint_positions
template <class Ctx> FSIM_OUTLINE void env_ik(Ctx cv, float rotate_speed, int mode) {
  FS_REBUILD_CTX(cv);
  CModel &m = c.m;
  const auto tail = GP(m.ik_tab) + IKT_ARM * c.D.narm;
  const float sens = tail[IKT_SENS];
  const bool rest_current = tail[IKT_RESTMODE] != 0.0f, rz = tail[IKT_RZ] != 0.0f;
#pragma unroll 1
  for (int arm = 0; arm < c.D.narm; arm++) {
  const auto tab = GP(m.ik_tab) + IKT_ARM * arm;
  float *K = c.L + c.ly.eik + EI_WORDS * arm;
  const V3 hp = ldv3(K + EI_HPOS);
  // _bounded_d_pos (furniture.py:1252-1258, limits :170-171)
  V3 a = ldv3(K + EI_ACT);
  V3 dpos = v3(fminf(fmaxf(a.x, -1.5f - hp.x), 1.5f - hp.x), fminf(fmaxf(a.y, -1.5f - hp.y), 1.5f - hp.y), fminf(fmaxf(a.z, 0.0f - hp.z), 1.5f - hp.z));
  Q4 qi = q4(K[EI_IQUAT], K[EI_IQUAT + 1], K[EI_IQUAT + 2], K[EI_IQUAT + 3]);
  M3 rot;
  if (mode == 2) {
    // control_type "ik_quaternion" (furniture.py:2994-3030): the action carries a quaternion (wxyz, convert_quat -> xyzw) that
    // _make_input composes with the CURRENT hand orientation: rotation = quat2mat(right_hand_quat (x) action_quat)
    const float *rh = K + EI_IQUAT; // _right_hand_quat (xyzw) of the last forward pass (env_ik_remember)
    Q4 q = qmul(q4(rh[3], rh[0], rh[1], rh[2]), q4(K[EI_ACT + 3], K[EI_ACT + 4], K[EI_ACT + 5], K[EI_ACT + 6]));
    float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (n2 < 8.8817842e-16f) { for (int i = 0; i < 9; i++) rot.m[i] = (i & 3) == 0 ? 1.0f : 0.0f; } // quat2mat: n < eps * 4 -> identity
    else rot = q2m(qnormalized(q));
  } else {
  // _initial_right_hand_quat = euler_to_quat(action[3:6] * rotate_speed, _initial_right_hand_quat): pyquaternion reads the stored
  // 4 numbers (an xyzw quaternion from mat2quat) as wxyz -- reproduced as is (furniture.py:2917-2919, transform_utils.py:617-630)
  const float d2r = 0.017453292519943295f;
  Q4 qx = axisangle(v3(1, 0, 0), K[EI_ACT + 3] * rotate_speed * d2r), qy = axisangle(v3(0, 1, 0), K[EI_ACT + 4] * rotate_speed * d2r),
     qz = axisangle(v3(0, 0, 1), K[EI_ACT + 5] * rotate_speed * d2r);
  qi = qmul(qi, qmul(qz, qmul(qy, qx)));
  // rotation = quat2mat(right_hand_quat (x) (right_hand_quat^-1 (x) initial)) = quat2mat(initial), the 4 numbers read as xyzw
  rot = q2m(qnormalized(q4(qi.z, qi.w, qi.x, qi.y))); // raw (r0, r1, r2, r3) as xyzw -> w = r3 (= qi.z), x = r0 (= qi.w) ...
  }
  // joint_positions_for_eef_command (:227-269): target += dpos * user_sensitivity; orientation . Rz(-90 deg)
  const V3 tp = ldv3(K + EI_TARGET) + dpos * sens;
  M3 tR = rot;
  if (rz) { // Sawyer: the commanded hand orientation . Rz(-90 deg) is the target of link right_l6 (sawyer_ik_controller.py:248-254)
    M3 Zm; Zm.m[0] = 0; Zm.m[1] = 1; Zm.m[2] = 0; Zm.m[3] = -1; Zm.m[4] = 0; Zm.m[5] = 0; Zm.m[6] = 0; Zm.m[7] = 0; Zm.m[8] = 1;
    tR = mulm(rot, Zm);
  }
  float q[7], q0[7], lo[7], hi[7];
#pragma unroll
  for (int i = 0; i < 7; i++) { q[i] = q0[i] = c.L[c.ly.qpos + GP(m.arm_qposadr)[7 * arm + i]]; lo[i] = tab[IKT_LOWER + i]; hi[i] = tab[IKT_UPPER + i]; }
#pragma unroll 1
  for (int it = 0; it < IK_ITERS; it++) {
    IkFk f;
    ik_fk(tab, q, f);
    V3 ep = tp - f.p, er = ik_rotvec(mulm(tR, ck_transpose(f.R)));
    float J[6][7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
      V3 jp = cross(f.z[i], f.p - f.o[i]);
      J[0][i] = jp.x; J[1][i] = jp.y; J[2][i] = jp.z; J[3][i] = f.z[i].x; J[4][i] = f.z[i].y; J[5][i] = f.z[i].z;
    }
    float A[21];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int s = 0; s <= r; s++) {
        float v = r == s ? IK_DAMP2 : 0.0f;
#pragma unroll
        for (int i = 0; i < 7; i++) v += J[r][i] * J[s][i];
        A[r * (r + 1) / 2 + s] = v;
      }
    ik_chol6(A);
    float y[6] = {ep.x, ep.y, ep.z, er.x, er.y, er.z};
    ik_solve6(A, y);
    float dq[7];
#pragma unroll
    for (int i = 0; i < 7; i++) { float s = 0;
#pragma unroll
      for (int r = 0; r < 6; r++) s += J[r][i] * y[r];
      dq[i] = s; }
    if (it < IK_ITERS - IK_TAIL) { // null-space pull towards rest_poses (:263), projected with the same damped inverse
      float n[7], y2[6];
#pragma unroll
      for (int i = 0; i < 7; i++) n[i] = IK_NULL_GAIN * ((rest_current ? q0[i] : tab[IKT_REST + i]) - q[i]); // Baxter rests at the current joints (:321)
#pragma unroll
      for (int r = 0; r < 6; r++) { float s = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) s += J[r][i] * n[i];
        y2[r] = s; }
      ik_solve6(A, y2);
#pragma unroll
      for (int i = 0; i < 7; i++) { float s = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) s += J[r][i] * y2[r];
        dq[i] += n[i] - s; }
    }
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = fminf(fmaxf(q[i] + dq[i], lo[i]), hi[i]);
  }
  if (c.lane == 0) {
    stv3(K + EI_TARGET, tp);
    if (mode != 2) { K[EI_IQUAT] = qi.w; K[EI_IQUAT + 1] = qi.x; K[EI_IQUAT + 2] = qi.y; K[EI_IQUAT + 3] = qi.z; }
#pragma unroll
    for (int i = 0; i < 7; i++) K[EI_QCMD + i] = q[i];
  }
  } // arm
  SYNC();
}
