// fsim_ctrl.hpp -- the reference's torque-level arm controllers as a per-substep stage of the fused step kernel.
//
// Reference: furniture/env/controllers/arm_controller.py (PositionOrientationController :488-797, PositionController :870-951,
// JointImpedanceController :368-486, JointVelocityController :309-366, JointTorqueController :246-306), driven by
// FurnitureEnv._do_controller_step / _pre_action (furniture.py:3065-3093, 1706-1759): one torque update before EVERY physics
// substep, ctrl[arm] = qfrc_bias[arm] + torques.  Only what the env can construct is built: the parameters of
// controllers/controller_config.hjson without overrides (furniture.py:94-101, 1665-1704) -- linear interpolation,
// impedance_flag false, no nullspace posture, no limits.  CPU restatement pinned to the reference classes: oracle/controllers.py.
//
// What update_model (arm_controller.py:109-136) reads is MuJoCo's memory as sim.step() left it: hand pose, body Jacobian,
// mass matrix and qfrc_bias of the forward pass BEFORE the last integration, qpos/qvel after it, and mujoco_py's
// body_xvelp/xvelr = jac . qvel.  That is exactly what LDS holds at the top of a substep (the previous substep's forward
// results were not overwritten by the integrator), so the stage reads LDS only.
//
// Work split: lanes 0..6 = arm joints (Jacobian columns, ramp state of the joint-space kinds, final J'w), lanes 0..5 = rows of
// J for the six M^-1 J' solves (each lane factors the 7x7 arm block of M redundantly -- same instructions for all lanes),
// lane 0 = the serial cartesian part (ramp state, orientation error, two thresholded 3x3 inverses).  Scratch = the Newton
// Hessian area, dead between substeps.
#pragma once

// per-env controller block, placed after the group table (env_extra_words); word offsets
enum { EK_KIND = 0, EK_STEP, EK_LIVE, EK_GOSET, EK_ACT = 4 /* 7 arm command entries + gripper */, EK_S = 12,
       // position / position_orientation
       EK_LGP = EK_S, EK_LGO = EK_LGP + 3, EK_LBASE = EK_LGO + 9, EK_LDELTA = EK_LBASE + 3, EK_ODELTA = EK_LDELTA + 3,
       EK_OINIT = EK_ODELTA + 3, EK_GORI = EK_OINIT + 9, EK_WORDS = EK_GORI + 9 + 1,
       // joint-space kinds (alias)
       EK_JLAST = EK_S, EK_JBASE = EK_JLAST + 7, EK_JDELTA = EK_JBASE + 7 };
enum { CK_NONE = 0, CK_POS_ORI = 1, CK_POS = 2, CK_JOINT_IMP = 3, CK_JOINT_VEL = 4, CK_JOINT_TORQUE = 5 };
#define CK_NJ 7

// transform_utils.py:360-380 (euler2mat)
DEV M3 ck_euler2mat(V3 e) {
  float si, ci, sj, cj, sk, ck;
  sincosf(-e.z, &si, &ci); sincosf(-e.y, &sj, &cj); sincosf(-e.x, &sk, &ck);
  float cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  M3 R;
  R.m[0] = cj * ci; R.m[1] = cj * si; R.m[2] = -sj;
  R.m[3] = sj * cs - sc; R.m[4] = sj * ss + cc; R.m[5] = cj * sk;
  R.m[6] = sj * cc + ss; R.m[7] = sj * sc - cs; R.m[8] = cj * ck;
  return R;
}
DEV M3 ck_transpose(const M3 &A) { M3 T; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T.m[3 * i + j] = A.m[3 * j + i]; return T; }
// arm_controller.py:180-201: half the sum of the column cross products
DEV V3 ck_ori_err(const M3 &desired, const M3 &current) {
  V3 e = cross(colv(current, 0), colv(desired, 0)) + cross(colv(current, 1), colv(desired, 1)) + cross(colv(current, 2), colv(desired, 2));
  return e * 0.5f;
}
// arm_controller.py:781-790: inverse through the SVD with singular values below 0.00025 zeroed.  The argument is J M^-1 J'
// (symmetric positive semi-definite), so the SVD is the eigen-decomposition: cyclic Jacobi, fully unrolled (no indexed arrays).
DEV M3 ck_pinv_sym3(const M3 &A) {
  float a00 = A.m[0], a01 = A.m[1], a02 = A.m[2], a11 = A.m[4], a12 = A.m[5], a22 = A.m[8];
  float v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
#define CK_ROT(app, aqq, apq, arp, arq, v0p, v0q, v1p, v1q, v2p, v2q)                                      \
  if (fabsf(apq) > 1e-30f) {                                                                                \
    float th = (aqq - app) / (2.0f * apq);                                                                   \
    float t = (th >= 0 ? 1.0f : -1.0f) / (fabsf(th) + sqrtf(th * th + 1.0f));                                \
    float cs = 1.0f / sqrtf(t * t + 1.0f), sn = t * cs;                                                      \
    app -= t * apq; aqq += t * apq; apq = 0;                                                                 \
    float x_ = arp, y_ = arq; arp = cs * x_ - sn * y_; arq = sn * x_ + cs * y_;                              \
    x_ = v0p; y_ = v0q; v0p = cs * x_ - sn * y_; v0q = sn * x_ + cs * y_;                                    \
    x_ = v1p; y_ = v1q; v1p = cs * x_ - sn * y_; v1q = sn * x_ + cs * y_;                                    \
    x_ = v2p; y_ = v2q; v2p = cs * x_ - sn * y_; v2q = sn * x_ + cs * y_;                                    \
  }
#pragma unroll 1
  for (int sweep = 0; sweep < 6; sweep++) {
    CK_ROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21) // (p, q) = (0, 1), r = 2
    CK_ROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22) // (0, 2), r = 1
    CK_ROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22) // (1, 2), r = 0
  }
#undef CK_ROT
  const float thr = 0.00025f;
  float s0 = fabsf(a00) < thr ? 0.0f : 1.0f / a00, s1 = fabsf(a11) < thr ? 0.0f : 1.0f / a11, s2 = fabsf(a22) < thr ? 0.0f : 1.0f / a22;
  M3 P;
  P.m[0] = v00 * v00 * s0 + v01 * v01 * s1 + v02 * v02 * s2; P.m[1] = v00 * v10 * s0 + v01 * v11 * s1 + v02 * v12 * s2;
  P.m[2] = v00 * v20 * s0 + v01 * v21 * s1 + v02 * v22 * s2; P.m[4] = v10 * v10 * s0 + v11 * v11 * s1 + v12 * v12 * s2;
  P.m[5] = v10 * v20 * s0 + v11 * v21 * s1 + v12 * v22 * s2; P.m[8] = v20 * v20 * s0 + v21 * v21 * s1 + v22 * v22 * s2;
  P.m[3] = P.m[1]; P.m[6] = P.m[2]; P.m[7] = P.m[5];
  return P;
}

// entry (i, j) of the arm block of M (tree-packed lower triangle, k_tmap)
template <class Ctx> DEV float ck_M(const Ctx &c, int i, int j) {
  int di = GP(c.m.arm_dofadr)[i], dj = GP(c.m.arm_dofadr)[j];
  return c.L[c.ly.M + fs_hidx(c, c.ly.k_tmap, max(di, dj), min(di, dj))];
}

// _pre_action for the (single, Sawyer) arm: writes ctrl[0..6] (motors) and ctrl[7..8] (fingers)
template <class Ctx> DEV void fs_controller(const Ctx &c, int policy_step) {
  CModel &m = c.m;
  float *L = c.L;
  float *K = L + c.ly.env + E_GROUP + c.D.nparts;
  int *Ki = reinterpret_cast<int *>(K);
  const int kind = Ki[EK_KIND];
  float *S = L + c.ly.H; // scratch: [0,42) J rows, [42,84) M^-1 J' rows, [84,102) the two 3x3 blocks, [102,108) wrench, [108,115) joint torques
  const float N = floorf(0.2f * 20.0f / c.D.timestep); // ramp_ratio * control_freq / timestep (arm_controller.py:114): 2000
  const int k = c.lane;
  const bool jl = k < CK_NJ;
  const int dk = jl ? GP(m.arm_dofadr)[k] : 0;
  const float qk = jl ? L[c.ly.qpos + GP(m.arm_qposadr)[k]] : 0.0f;
  float qdk = jl ? L[c.ly.qvel + dk] : 0.0f;
  const int step0 = policy_step ? 0 : Ki[EK_STEP];
  float tau = 0.0f;
  if (kind >= CK_JOINT_IMP) {
    // ---- joint-space kinds: every arm-joint lane owns one component of the ramp
    const float range = kind == CK_JOINT_IMP ? 0.2f : (kind == CK_JOINT_VEL ? 1.0f : (k < 3 ? 0.5f : (k < 5 ? 0.2f : 0.1f)));
    const float a = jl ? fminf(fmaxf(K[EK_ACT + k], -1.0f), 1.0f) * range : 0.0f; // transform_action (:99-107)
    if (policy_step) {
      float last = jl ? K[EK_JLAST + k] : 0.0f, goal = a;
      if (kind == CK_JOINT_IMP) {
        bool zero = true; // np.linalg.norm(last_goal_joint) == 0 -> start the ramp from the current joints (:446-447)
        for (int j = 0; j < CK_NJ; j++) zero = zero && (K[EK_JLAST + j] == 0.0f);
        if (zero) last = qk;
        goal = qk + a;
      }
      SYNC();
      if (jl) { K[EK_JBASE + k] = last; K[EK_JDELTA + k] = (goal - last) / N; }
    }
    SYNC();
    const float lg = jl ? K[EK_JBASE + k] + (float)(step0 + 1) * K[EK_JDELTA + k] : 0.0f; // linear[step] = last + (step + 1) * delta
    if (jl) K[EK_JLAST + k] = lg;
    if (kind == CK_JOINT_TORQUE) tau = lg;
    else if (kind == CK_JOINT_VEL) {
      const float kv = k == 0 ? 8.0f : k == 1 ? 7.0f : k == 2 ? 6.0f : k == 3 ? 4.0f : k == 4 ? 2.0f : k == 5 ? 0.5f : 0.1f;
      tau = kv * (lg - qdk);
    } else {
      const float kp = k < 4 ? 55.0f : (k == 4 ? 30.0f : (k == 5 ? 15.5f : 5.5f)); // (kp_max + kp_min) / 2, damping (2 + 0) / 2 = 1
      const float kv = 2.0f * sqrtf(kp);
      float n2 = 0;
      for (int j = 0; j < CK_NJ; j++) { float v = L[c.ly.qvel + GP(m.arm_dofadr)[j]]; n2 += v * v; }
      const float nrm = sqrtf(n2);
      if (nrm > 7.0f) qdk = qdk / (nrm * 7.0f); // :485-487
      if (jl) S[108 + k] = kp * (lg - qk) - kv * qdk;
      SYNC();
      if (jl) for (int j = 0; j < CK_NJ; j++) tau += ck_M(c, k, j) * S[108 + j]; // decoupled_torques = M . torques (:492)
    }
  } else {
    // ---- cartesian kinds
    const int hb = GP(m.hand_body)[0], rb = GP(m.body_red)[hb];
    const M3 Rb = ldm3(L + c.ly.xmat + 9 * rb);
    const V3 pos = ldv3(L + c.ly.xpos + 3 * rb) + mulv(Rb, ldv3(GP(m.body_relpos) + 3 * hb));
    const M3 R = mulm(Rb, q2m(qnormalized(ldq(GP(m.body_relquat) + 4 * hb))));
    if (jl) {
      V3 jp = fs_col(c, dk, pos), jr = ldv3(L + c.ly.cdof + 6 * dk);
      S[0 * 7 + k] = jp.x; S[1 * 7 + k] = jp.y; S[2 * 7 + k] = jp.z; S[3 * 7 + k] = jr.x; S[4 * 7 + k] = jr.y; S[5 * 7 + k] = jr.z;
    }
    SYNC();
    if (k < 6) {
      // Cholesky of the arm block (row-major packed lower triangle in registers), then M y = J[k, :]'
      float G[28];
#pragma unroll
      for (int i = 0; i < CK_NJ; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) G[i * (i + 1) / 2 + j] = ck_M(c, i, j);
#pragma unroll
      for (int j = 0; j < CK_NJ; j++) {
        float d = G[j * (j + 1) / 2 + j];
#pragma unroll
        for (int p = 0; p < j; p++) d -= G[j * (j + 1) / 2 + p] * G[j * (j + 1) / 2 + p];
        const float dinv = rsqrtf(fmaxf(d, 1e-30f));
        G[j * (j + 1) / 2 + j] = dinv; // the diagonal holds 1 / L_jj
#pragma unroll
        for (int i = j + 1; i < CK_NJ; i++) {
          float s = G[i * (i + 1) / 2 + j];
#pragma unroll
          for (int p = 0; p < j; p++) s -= G[i * (i + 1) / 2 + p] * G[j * (j + 1) / 2 + p];
          G[i * (i + 1) / 2 + j] = s * dinv;
        }
      }
      float y[CK_NJ];
#pragma unroll
      for (int i = 0; i < CK_NJ; i++) {
        float s = S[k * 7 + i];
#pragma unroll
        for (int p = 0; p < i; p++) s -= G[i * (i + 1) / 2 + p] * y[p];
        y[i] = s * G[i * (i + 1) / 2 + i];
      }
#pragma unroll
      for (int i = CK_NJ - 1; i >= 0; i--) {
        float s = y[i];
#pragma unroll
        for (int p = i + 1; p < CK_NJ; p++) s -= G[p * (p + 1) / 2 + i] * y[p];
        y[i] = s * G[i * (i + 1) / 2 + i];
      }
#pragma unroll
      for (int i = 0; i < CK_NJ; i++) S[42 + k * 7 + i] = y[i];
      // row k of its block: (Jx M^-1 Jx') for k < 3, (Jr M^-1 Jr') for k >= 3
      const int blk = k < 3 ? 0 : 3;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        float s = 0;
#pragma unroll
        for (int i = 0; i < CK_NJ; i++) s += S[(blk + j) * 7 + i] * y[i];
        S[84 + k * 3 + j] = s;
      }
    }
    SYNC();
    if (k == 0) {
      V3 velp = v3(0, 0, 0), velr = v3(0, 0, 0); // body_xvelp / body_xvelr = jac . qvel
      for (int j = 0; j < CK_NJ; j++) {
        float v = L[c.ly.qvel + GP(m.arm_dofadr)[j]];
        velp = velp + v3(S[j], S[7 + j], S[14 + j]) * v;
        velr = velr + v3(S[21 + j], S[28 + j], S[35 + j]) * v;
      }
      if (policy_step) {
        V3 a = v3(fminf(fmaxf(K[EK_ACT], -1.0f), 1.0f), fminf(fmaxf(K[EK_ACT + 1], -1.0f), 1.0f), fminf(fmaxf(K[EK_ACT + 2], -1.0f), 1.0f)) * 0.05f;
        V3 goal_pos = pos + a; // set_goal_position (:794-802), no position limits
        if (kind == CK_POS_ORI) {
          V3 ao = v3(fminf(fmaxf(K[EK_ACT + 3], -1.0f), 1.0f), fminf(fmaxf(K[EK_ACT + 4], -1.0f), 1.0f), fminf(fmaxf(K[EK_ACT + 5], -1.0f), 1.0f)) * 0.2f;
          stm3(K + EK_GORI, mulm(ck_transpose(ck_euler2mat(-ao)), R)); // set_goal_orientation (:808-810)
        } else if (!Ki[EK_GOSET]) { stm3(K + EK_GORI, R); Ki[EK_GOSET] = 1; } // PositionController (:934-939): captured once
        if (K[EK_LGP] == 0.0f && K[EK_LGP + 1] == 0.0f && K[EK_LGP + 2] == 0.0f) stv3(K + EK_LGP, pos); // norm(last_goal_position) == 0
        // last_goal_orientation == eye: it becomes current_orientation_mat, a VIEW into sim.data.body_xmat that
        // orientation_initial_interpolation then aliases (:679-680, :635) -- live until the next policy step
        bool eye = true;
        for (int i = 0; i < 9; i++) eye = eye && (K[EK_LGO + i] == ((i & 3) == 0 ? 1.0f : 0.0f));
        Ki[EK_LIVE] = eye ? 1 : 0;
        if (eye) stm3(K + EK_LGO, R);
        V3 lgp = ldv3(K + EK_LGP);
        stv3(K + EK_LBASE, lgp);
        // (the reference divides: (goal - last) / N; N = 2000 is not a power of two, so divide too)
        K[EK_LDELTA] = (goal_pos.x - lgp.x) / N; K[EK_LDELTA + 1] = (goal_pos.y - lgp.y) / N; K[EK_LDELTA + 2] = (goal_pos.z - lgp.z) / N;
        M3 lgo = ldm3(K + EK_LGO);
        V3 oe = ck_ori_err(ldm3(K + EK_GORI), lgo);
        K[EK_ODELTA] = oe.x / N; K[EK_ODELTA + 1] = oe.y / N; K[EK_ODELTA + 2] = oe.z / N;
        stm3(K + EK_OINIT, lgo);
      }
      const float f1 = (float)(step0 + 1);
      V3 lgp = ldv3(K + EK_LBASE) + ldv3(K + EK_LDELTA) * f1;
      stv3(K + EK_LGP, lgp);
      V3 god = ldv3(K + EK_ODELTA) * f1;
      if (Ki[EK_LIVE]) stm3(K + EK_OINIT, R);
      M3 lgo = mulm(ck_transpose(ck_euler2mat(-god)), ldm3(K + EK_OINIT));
      stm3(K + EK_LGO, lgo);
      const float kp = 150.0f, kv = 2.0f * sqrtf(150.0f) * 1.0f; // initial_impedance_pos / ori, initial_damping (hjson :10-12)
      V3 f = (lgp - pos) * kp - velp * kv;
      V3 t = ck_ori_err(lgo, R) * kp - velr * kv;
      M3 Ax, Ar;
      for (int i = 0; i < 9; i++) { Ax.m[i] = S[84 + i]; Ar.m[i] = S[93 + i]; }
      // symmetrise (the two triangles come from different lanes' dot products)
      Ax.m[1] = Ax.m[3] = 0.5f * (Ax.m[1] + Ax.m[3]); Ax.m[2] = Ax.m[6] = 0.5f * (Ax.m[2] + Ax.m[6]); Ax.m[5] = Ax.m[7] = 0.5f * (Ax.m[5] + Ax.m[7]);
      Ar.m[1] = Ar.m[3] = 0.5f * (Ar.m[1] + Ar.m[3]); Ar.m[2] = Ar.m[6] = 0.5f * (Ar.m[2] + Ar.m[6]); Ar.m[5] = Ar.m[7] = 0.5f * (Ar.m[5] + Ar.m[7]);
      stv3(S + 102, mulv(ck_pinv_sym3(Ax), f)); // decoupled_force = lambda_x . desired_force (:722-726)
      stv3(S + 105, mulv(ck_pinv_sym3(Ar), t));
    }
    SYNC();
    if (jl) for (int r = 0; r < 6; r++) tau += S[r * 7 + k] * S[102 + r]; // J_full' . wrench (:731)
  }
  SYNC();
  if (k == 0) Ki[EK_STEP] = ((float)step0 < N - 1.0f) ? step0 + 1 : step0;
  // ctrl[arm] = qfrc_bias[arm] + torques (furniture.py:1756-1758; the bias of the same stale forward pass)
  if (jl) L[c.ly.ctrl + k] = L[c.ly.qfrcbias + dk] + tau;
  // gripper: format_action 1 -> [g, -g], bias + weight * a from actuator_ctrlrange, not clipped (furniture.py:1722-1737)
  if (k >= CK_NJ && k < CK_NJ + 2) {
    float g = K[EK_ACT + 7];
    L[c.ly.ctrl + k] = GP(m.ctrl_bias)[k] + GP(m.ctrl_weight)[k] * (k == CK_NJ ? g : -g);
  }
  SYNC();
}
