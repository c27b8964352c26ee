// fsim_dense.hpp -- the 8-phase dense reward of FurnitureSawyerDenseRewardEnv as a per-env state machine
// (furniture/env/furniture_sawyer_dense.py):
//   _reset_reward_variables/_set_next_subtask/_update_reward_variables :128-216     _collect_values :222-271
//   _compute_reward :273-577      phase rewards :579-944      _stable_grip_reward/_gripper_penalty/_ctrl_penalty :946-1019
// Scalar code run by lane 0 after the physics of a step; the state lives in the env block of the record (ED_*).  The code is
// templated on a provider P of the sensor values ({obs(subtask, DObs&), aligned(subtask)}): the env kernel reads them from the
// LDS poses, the replay kernel (fsim_dense_replay, the parity hook for the reference's golden vectors) from arrays.
// Only diff_rew=True is implemented: with diff_rew=False the reference raises AttributeError in grasp_leg (:668).
#pragma once
#include "fsim_math.hpp"

// coefficient table (order of furniture_amd/dense.py DENSE_COEF_DEFAULTS)
enum {
  DC_PHASE_BONUS = 0, DC_EEF_FWD, DC_EEF_UP, DC_EEF_ROT_THR, DC_GRIPPER_PEN, DC_MOVE_OTHER, DC_DROP_PEN, DC_EARLY_TERM, DC_INIT_EEF,
  DC_MOVE_EEF, DC_LOWER_EEF, DC_GRASP, DC_LIFT_Z, DC_LIFT_XY, DC_LIFT_Z_THR, DC_LIFT_XY_THR, DC_ALIGN_POS, DC_ALIGN_ROT,
  DC_ALIGN_POS_THR, DC_ALIGN_ROT_THR, DC_MOVE_POS, DC_MOVE_ROT, DC_MOVE_POS_THR, DC_MOVE_ROT_THR, DC_FINE_EXP, DC_FINE_POS,
  DC_FINE_ROT, DC_ALIGNED_BONUS, DC_CTRL_PEN, DC_RESET_ROBOT, DC_Z_FINEDIST, DC_GRIPTIP_SITE, DC_GRIP_SITE, DC_PHASE_OB, DC_WORDS
};
// per-subtask row
enum { DS_LEG_PART = 0, DS_TABLE_PART, DS_LEG_SITE, DS_TABLE_SITE, DS_GL_SITE, DS_GR_SITE, DS_ANGLE, DS_HAS_ANGLES, DS_WAYPOINT_Z,
       DS_GRIP_INIT_N, DS_GRIP_INIT0, DS_K_LEG = 14, DS_K_TABLE, DS_WORDS };
// per-env state (floats; the small integers are exact in fp32)
enum { ED_SUBTASK = 0, ED_PHASE, ED_FLAGS /* 1 leg_dropped, 2 table_moved, 4 leg_lift */, ED_FINE_ALIGNED, ED_INIT_TABLE_SITE,
       ED_INIT_LIFT_LEG = ED_INIT_TABLE_SITE + 3, ED_LIFT_LEG = ED_INIT_LIFT_LEG + 3, ED_INIT_EEF = ED_LIFT_LEG + 3,
       ED_PREV_INIT_EEF = ED_INIT_EEF + 3, ED_PREV_ABOVE, ED_PREV_EEF_LEG, ED_PREV_GRASP, ED_PREV_LIFT_Z, ED_PREV_LIFT_XY,
       ED_PREV_MOVE_POS, ED_PREV_UP, ED_PREV_FWD, ED_PREV_PROJ_T, ED_PREV_PROJ_L, ED_WORDS };

struct DObs {
  V3 eef, gl, gr, leg, legsite, tablesite, legup, tableup, legfwd, tablefwd, gripup, gripfwd;
  bool touch_l, touch_r;
};
struct DenseOut { float reward, phase_bonus; int done, success, phase_info; };

DEV float dn_cos(V3 a, V3 b) { return dot(a, b) / norm(a) / norm(b); }

// _update_reward_variables (:149-216) for the subtask in S[ED_SUBTASK]
template <class P> DEV void dense_update(float *S, const float *C_, const float *T_, P &p) {
  auto C = GP(C_);
  int st = (int)S[ED_SUBTASK];
  auto T = GP(T_) + DS_WORDS * st;
  DObs o;
  p.obs(st, o);
  S[ED_FLAGS] = 0; S[ED_FINE_ALIGNED] = 0;
  stv3(S + ED_INIT_TABLE_SITE, o.tablesite);
  stv3(S + ED_INIT_LIFT_LEG, o.leg);
  stv3(S + ED_LIFT_LEG, o.leg + v3(0, 0, T[DS_WAYPOINT_Z]));
  int phase = C[DC_RESET_ROBOT] != 0.0f ? 1 : 0;
  int ngi = (int)T[DS_GRIP_INIT_N];
  if (ngi > 0) {
    V3 ie = o.eef + v3(T[DS_GRIP_INIT0], T[DS_GRIP_INIT0 + 1], T[DS_GRIP_INIT0 + 2]);
    if (ngi == 4) ie.z = T[DS_GRIP_INIT0 + 3] - 0.085f; // distance between grip_base and griptip
    stv3(S + ED_INIT_EEF, ie);
  } else phase = 1;
  S[ED_PHASE] = (float)phase;
  if (phase == 1) S[ED_PREV_ABOVE] = norm(o.eef - (0.5f * (o.gl + o.gr) + v3(0, 0, 0.05f)));
  else S[ED_PREV_INIT_EEF] = norm(o.eef - ldv3(S + ED_INIT_EEF));
  S[ED_PREV_GRASP] = -1.0f;
  S[ED_PREV_LIFT_Z] = T[DS_WAYPOINT_Z];
  S[ED_PREV_LIFT_XY] = 0.0f;
}

// _reset_reward_variables (:128-139); n_pre = len(preassembled)
template <class P> DEV void dense_reset(float *S, const float *C, const float *T, P &p, int n_pre) {
  for (int i = 0; i < ED_WORDS; i++) S[i] = 0.0f;
  S[ED_SUBTASK] = (float)n_pre;
  dense_update(S, C, T, p);
}

// _set_next_subtask (:141-147)
template <class P> DEV bool dense_next(float *S, const float *C, const float *T, int nsub, P &p) {
  S[ED_SUBTASK] += 1.0f;
  if ((int)S[ED_SUBTASK] == nsub) return true;
  dense_update(S, C, T, p);
  return false;
}

DEV float dn_min0(float x, bool touched) { return touched ? x : fminf(x, 0.0f); }

// _compute_reward (:273-577).  ac: the RAW action (dof floats; [-2] gripper, [-1] connect), connected: self._connected.
template <class P> DEV DenseOut dense_compute(float *S, const float *C_, const float *T_, int nsub, P &p, const float *ac, int dof, bool connected) {
  auto C = GP(C_);
  int st = (int)S[ED_SUBTASK];
  DenseOut out;
  out.reward = 0; out.phase_bonus = 0; out.done = 0; out.success = 0; out.phase_info = 0;
  if (st >= nsub) { out.done = 1; return out; } // (the reference would index past the recipe: unreachable, the episode ended)
  auto T = GP(T_) + DS_WORDS * st;
  DObs o;
  p.obs(st, o);
  const float bonus = C[DC_PHASE_BONUS];
  const bool early = C[DC_EARLY_TERM] != 0.0f;
  // ---- _collect_values
  bool touched = o.touch_l && o.touch_r;
  V3 fr = o.legfwd;
  if (T[DS_HAS_ANGLES] != 0.0f) { // _project_connector_forward (furniture.py:1178-1199)
    V3 k = normalized(o.legup);
    float ang = T[DS_ANGLE];
    if (ang != ang) { // None: the rotation about leg_up that best matches table_forward
      float cs = dn_cos(o.legfwd, o.tablefwd), sn = sqrtf(1.0f - cs * cs);
      V3 rp = cs * o.legfwd + sn * cross(k, o.legfwd), rn = cs * o.legfwd - sn * cross(k, o.legfwd);
      fr = dn_cos(rp, o.tablefwd) > dn_cos(rn, o.tablefwd) ? rp : rn;
    } else {
      float a = ang / 180.0f * 3.14159265358979f;
      fr = cosf(a) * o.legfwd + sinf(a) * cross(k, o.legfwd);
    }
  }
  V3 grasp = 0.5f * (o.gl + o.gr);
  bool safe_grasp = touched && o.eef.z < grasp.z;
  float move_pos_dist = norm(o.tablesite - o.legsite);
  float move_above_dist = norm(o.tablesite + v3(0, 0, C[DC_Z_FINEDIST]) - o.legsite);
  float up_ang = dn_cos(o.legup, o.tableup), fwd_ang = dn_cos(fr, o.tablefwd);
  float proj_t = dn_cos(-o.tableup, o.legsite - o.tablesite), proj_l = dn_cos(o.legup, o.tablesite - o.legsite);
  float table_disp = norm(o.tablesite - ldv3(S + ED_INIT_TABLE_SITE));
  // ---- common terms
  float s2 = 0;
  for (int k = 0; k < dof - 2; k++) s2 += ac[k] * ac[k];
  float ctrl_pen = -C[DC_CTRL_PEN] * sqrtf(s2);
  float grip_a = ac[dof - 2], conn_a = ac[dof - 1];
  float up_d = dn_cos(o.gripup, v3(0, 0, -1));
  V3 gv = o.gr - o.gl;
  float fd = fmaxf(dn_cos(o.gripfwd, gv), dn_cos(-o.gripfwd, gv));
  float move_pen = -C[DC_MOVE_OTHER] * table_disp;
  bool table_moved = table_disp > 0.1f;
  int phase = (int)S[ED_PHASE];
  int flags = (int)S[ED_FLAGS];
  auto stable_succ = [&](int ph) {
    bool s = true;
    if (ph <= 4) s = s && up_d > C[DC_EEF_ROT_THR];
    if (ph >= 1 && ph <= 4) s = s && fd > C[DC_EEF_ROT_THR];
    return s;
  };
  // early picking / early fine alignment: only with phase_ob False (furniture_sawyer_dense.py:306-345)
  const bool skips = C[DC_PHASE_OB] == 0.0f;
  if (skips && safe_grasp && stable_succ(phase) && phase < 3) phase = 4;
  if (skips && touched && (phase == 4 || phase == 5)) {
    if ((move_pos_dist < C[DC_MOVE_POS_THR] || move_above_dist < C[DC_MOVE_POS_THR]) && up_ang > C[DC_MOVE_ROT_THR] &&
        fwd_ang > C[DC_MOVE_ROT_THR]) {
      phase = 7;
      S[ED_PREV_MOVE_POS] = move_pos_dist; S[ED_PREV_UP] = up_ang; S[ED_PREV_FWD] = fwd_ang;
      S[ED_PREV_PROJ_T] = proj_t; S[ED_PREV_PROJ_L] = proj_l;
    }
  }
  S[ED_PHASE] = (float)phase;
  out.phase_info = phase + 8 * st;
  // _stable_grip_reward with the (possibly advanced) phase
  float sg_rew = 0;
  if (phase <= 4) sg_rew += C[DC_EEF_UP] * (up_d - 1.0f);
  if (phase >= 1 && phase <= 4) sg_rew += (fabsf(fd) - 1.0f) * C[DC_EEF_FWD];
  bool sg_succ = stable_succ(phase);
  // _gripper_penalty
  bool open_phase = phase <= 2;
  bool grip_succ = open_phase ? grip_a < 0 : grip_a > 0;
  float grip_pen = (open_phase ? -grip_a : grip_a) * C[DC_GRIPPER_PEN];
  float phase_reward = 0, phase_bonus = 0;
  int done = 0, success = 0;
  auto lower_eef = [&](bool *succ) { // _lower_eef_reward (:621-653)
    V3 leg = grasp + v3(0, 0, -0.015f);
    float xy = sqrtf((o.eef.x - leg.x) * (o.eef.x - leg.x) + (o.eef.y - leg.y) * (o.eef.y - leg.y)), z = fabsf(o.eef.z - leg.z);
    float d = norm(o.eef - leg);
    float r = (fminf(S[ED_PREV_EEF_LEG], 0.2f) - fminf(d, 0.2f)) * C[DC_LOWER_EEF] * 10.0f;
    S[ED_PREV_EEF_LEG] = d;
    *succ = xy < 0.02f && z < 0.015f;
    return r;
  };
  auto dropped_or_moved = [&]() { // leg dropped / table moved during lift, align, move
    if (!touched) flags |= 1; else flags |= 2;
    done = early;
    if (early) phase_bonus -= bonus / 2;
  };
  auto next_subtask = [&]() {
    phase_bonus += bonus * 2;
    phase_bonus -= S[ED_FINE_ALIGNED] * C[DC_ALIGNED_BONUS]; // discourage staying in aligned mode
    S[ED_FLAGS] = (float)flags;
    bool fin = dense_next(S, C_, T_, nsub, p);
    if (fin) S[ED_PHASE] = 0.0f;
    flags = (int)S[ED_FLAGS];
    phase = (int)S[ED_PHASE];
    done = success = fin;
  };
  if (phase != 7 && connected) {
    bool correct = p.aligned(st);
    if (table_moved) { flags |= 2; done = early; if (early) phase_bonus -= bonus; }
    else if (correct) next_subtask();
    else { success = 0; done = 1; }
  } else if (phase == 0) {
    float d = norm(o.eef - ldv3(S + ED_INIT_EEF));
    phase_reward = (__expf(-10.0f * fminf(d, 0.5f)) - __expf(-10.0f * fminf(S[ED_PREV_INIT_EEF], 0.5f))) * C[DC_INIT_EEF] * 10.0f;
    S[ED_PREV_INIT_EEF] = d;
    if (d < 0.03f && sg_succ && grip_succ) {
      phase = 1; phase_bonus += bonus;
      S[ED_PREV_ABOVE] = norm(o.eef - (grasp + v3(0, 0, 0.05f)));
    }
  } else if (phase == 1) {
    float d = norm(o.eef - (grasp + v3(0, 0, 0.05f)));
    phase_reward = (fminf(S[ED_PREV_ABOVE], 1.0f) - fminf(d, 1.0f)) * C[DC_MOVE_EEF] * 10.0f;
    S[ED_PREV_ABOVE] = d;
    if (d < 0.03f && sg_succ && grip_succ) {
      phase = 2; phase_bonus += bonus;
      S[ED_PREV_EEF_LEG] = norm(o.eef - (grasp + v3(0, 0, -0.015f)));
    }
  } else if (phase == 2) {
    bool succ;
    phase_reward = lower_eef(&succ);
    if (succ && sg_succ && grip_succ) { phase_bonus += bonus; phase = 3; }
  } else if (phase == 3) {
    bool dummy;
    phase_reward = lower_eef(&dummy);
    phase_reward += (grip_a - S[ED_PREV_GRASP]) * C[DC_GRASP];
    S[ED_PREV_GRASP] = grip_a;
    if (touched && safe_grasp && sg_succ) { phase = 4; phase_bonus += bonus; }
  } else if (phase == 4) {
    V3 lift = ldv3(S + ED_LIFT_LEG);
    float xy = sqrtf((lift.x - o.leg.x) * (lift.x - o.leg.x) + (lift.y - o.leg.y) * (lift.y - o.leg.y)), z = fabsf(lift.z - o.leg.z);
    float zr = (fminf(S[ED_PREV_LIFT_Z], 0.5f) - fminf(z, 0.5f)) * C[DC_LIFT_Z] * 10.0f;
    S[ED_PREV_LIFT_Z] = z;
    float xr = (fminf(S[ED_PREV_LIFT_XY], 0.8f) - fminf(xy, 0.8f)) * C[DC_LIFT_XY] * 10.0f;
    S[ED_PREV_LIFT_XY] = xy;
    float r = xr + zr;
    bool lifted = o.leg.z > S[ED_INIT_LIFT_LEG + 2] + 0.01f;
    if (touched && lifted && safe_grasp && !(flags & 4)) { flags |= 4; r += bonus / 2; }
    if (!touched) r = fminf(r, 0.0f);
    phase_reward = r;
    bool succ = xy < C[DC_LIFT_XY_THR] && z < C[DC_LIFT_Z_THR];
    if (!touched || table_moved) dropped_or_moved();
    else if (succ) {
      phase = 5; phase_bonus += bonus;
      S[ED_PREV_MOVE_POS] = 0.0f; S[ED_PREV_UP] = up_ang; S[ED_PREV_FWD] = fwd_ang;
    }
  } else if (phase == 5) {
    float d = norm(ldv3(S + ED_LIFT_LEG) - o.leg);
    float pr = (fminf(S[ED_PREV_MOVE_POS], 0.4f) - fminf(d, 0.4f)) * C[DC_ALIGN_POS] * 10.0f;
    S[ED_PREV_MOVE_POS] = d;
    float ur = (up_ang - S[ED_PREV_UP]) * C[DC_ALIGN_ROT] * 10.0f;
    S[ED_PREV_UP] = up_ang;
    float fwr = (fwd_ang - S[ED_PREV_FWD]) * C[DC_ALIGN_ROT] * 10.0f;
    S[ED_PREV_FWD] = fwd_ang;
    phase_reward = dn_min0(pr, touched) + dn_min0(ur, touched) + dn_min0(fwr, touched);
    bool succ = d < C[DC_ALIGN_POS_THR] && up_ang > C[DC_ALIGN_ROT_THR] && fwd_ang > C[DC_ALIGN_ROT_THR] && touched;
    if (!touched || table_moved) dropped_or_moved();
    else if (succ) { phase = 6; phase_bonus += bonus * 2; S[ED_PREV_MOVE_POS] = move_above_dist; }
  } else if (phase == 6) {
    float pr = (fminf(S[ED_PREV_MOVE_POS], 0.5f) - fminf(move_above_dist, 0.5f)) * C[DC_MOVE_POS] * 10.0f;
    S[ED_PREV_MOVE_POS] = move_above_dist;
    float ur = (fmaxf(up_ang, 0.0f) - fmaxf(S[ED_PREV_UP], 0.0f)) * C[DC_MOVE_ROT] * 10.0f;
    S[ED_PREV_UP] = up_ang;
    float fwr = (fmaxf(fwd_ang, 0.0f) - fmaxf(S[ED_PREV_FWD], 0.0f)) * C[DC_MOVE_ROT] * 10.0f;
    S[ED_PREV_FWD] = fwd_ang;
    phase_reward = dn_min0(pr, touched) + dn_min0(ur, touched) + dn_min0(fwr, touched);
    bool succ = (move_above_dist < C[DC_MOVE_POS_THR] || move_pos_dist < C[DC_MOVE_POS_THR]) && up_ang > C[DC_MOVE_ROT_THR] &&
                fwd_ang > C[DC_MOVE_ROT_THR] && touched;
    if (!touched || table_moved) dropped_or_moved();
    else if (succ) {
      phase = 7; phase_bonus += bonus * 2;
      S[ED_PREV_MOVE_POS] = move_pos_dist; S[ED_PREV_PROJ_T] = proj_t; S[ED_PREV_PROJ_L] = proj_l;
    }
  } else { // move_leg_fine
    float ke = C[DC_FINE_EXP], rc = C[DC_FINE_ROT], thr = C[DC_MOVE_ROT_THR] - 0.1f;
    float pr = (__expf(ke * move_pos_dist) - __expf(ke * S[ED_PREV_MOVE_POS])) * C[DC_FINE_POS] * 10.0f;
    S[ED_PREV_MOVE_POS] = move_pos_dist;
    auto f = [&](float x) { return __expf(-2.0f * (1.0f - fmaxf(x, thr))); };
    auto g = [&](float x) { return __expf(-3.0f * (1.0f - fmaxf(fabsf(x), 0.5f))); };
    float ur = (f(up_ang) - f(S[ED_PREV_UP])) * rc * 10.0f;
    S[ED_PREV_UP] = up_ang;
    float fwr = (f(fwd_ang) - f(S[ED_PREV_FWD])) * rc * 10.0f;
    S[ED_PREV_FWD] = fwd_ang;
    float tr = (g(proj_t) - g(S[ED_PREV_PROJ_T])) * rc * 5.0f;
    S[ED_PREV_PROJ_T] = proj_t;
    float lr = (g(proj_l) - g(S[ED_PREV_PROJ_L])) * rc * 5.0f;
    S[ED_PREV_PROJ_L] = proj_l;
    bool fine = p.aligned(st);
    bool connect_succ = connected && fine;
    float r = dn_min0(pr, touched) + dn_min0(ur, touched) + dn_min0(fwr, touched) + dn_min0(tr, touched) + dn_min0(lr, touched);
    if (fine) { S[ED_FINE_ALIGNED] += 1.0f; r += (conn_a + 1.0f) * C[DC_ALIGNED_BONUS]; }
    phase_reward = connected ? 0.0f : r;
    bool advanced = false;
    if (table_moved) { flags |= 2; done = early; if (early) phase_bonus -= bonus; }
    else if (connected && fine) { next_subtask(); advanced = true; }
    else if (connected) { done = 1; success = 0; }
    if (!touched && !connect_succ) {
      // (after next_subtask() the flags are those of the NEW subtask, as in the reference, where _leg_dropped was just reset;
      //  unreachable there anyway: connect_succ is true whenever the subtask advanced)
      flags |= 1; done = early;
      if (early) phase_bonus -= bonus;
    }
    (void)advanced;
  }
  S[ED_PHASE] = (float)phase;
  S[ED_FLAGS] = (float)flags;
  float reward = ctrl_pen + phase_reward + sg_rew;
  reward += grip_pen + phase_bonus + move_pen;
  if ((flags & 1) && !early) reward -= C[DC_DROP_PEN];
  out.reward = reward; out.phase_bonus = phase_bonus; out.done = done; out.success = success;
  return out;
}
