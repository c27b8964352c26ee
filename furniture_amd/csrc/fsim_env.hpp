// fsim_env.hpp -- FurnitureEnv.step()/reset() for ONE env on ONE wavefront (device side of the gym surface).
//
// Restates the env-level control flow of the reference on top of the physics in fsim_physics/solver:
//   step            furniture/env/furniture.py:364-385, 405-449 ; furniture_sawyer.py:66-84 ; furniture_baxter.py:66-80
//   _setup_action   furniture.py:3332-3379        _do_simulation furniture.py:2857-2897
//   finger scan     furniture.py:1290-1330        _try_connect   furniture.py:926-1042
//   _is_aligned     furniture.py:1057-1153        _connect       furniture.py:847-924
//   _activate_weld  furniture.py:2761-2776        union-find     furniture.py:2738-2759
//   _get_obs        furniture.py:1344-1387 ; furniture_sawyer.py:103-155 ; furniture_baxter.py:98-165
//   _compute_reward furniture.py:482-541          _after_step    furniture.py:451-480
//   _reset          furniture.py:1406-1663
// Because a wave IS an env, every data-dependent branch of the state machine is wave-uniform: the decision
// is taken by lane 0 on LDS state, published through LDS, and the physics sub-steps it triggers are executed
// by all 64 lanes.
#pragma once
#include "fsim_solver.hpp"
#include "fsim_dense.hpp"
#include "fsim_ctrl.hpp"
#include "fsim_ik.hpp"

struct EnvCfg {
  int dof_action, obs_dim, n_substeps, max_episode_steps, discrete_grip, rescale_actions, auto_align, auto_reset, has_recipe, agent;
  float pos_dist, rot_up, rot_fwd, proj_dist, ctrl_penalty_coef, unstable_penalty_coef, success_reward, touch_reward, pick_reward;
  // Cursor agent (furniture_cursor.py:28-32, config/furniture.py:84-90)
  int num_connect_steps, gravity_comp;
  float move_speed, rotate_speed, cursor_boundary;
  // dense-reward env (fsim_dense.hpp): tables uploaded by fsim_set_dense_reward
  int dense, dense_nsub;
  const float *dense_coef, *dense_sub;
  int controller; // CK_* (fsim_ctrl.hpp): torque-level arm controller run before every physics substep, 0 = none
  int ik;         // 1: control_type "ik", 2: "ik_quaternion" (fsim_ik.hpp)
  int obs_bf16;   // the caller's observation slab is bfloat16 (fsim_config_t::obs_bf16)
  // pre-assembled starts (furniture.py:163, 204-207, 1476-1503, 1542-1566; fsim_set_preassembled)
  int n_pre, pre_mode;      // pre_mode 0: pre_tab rows = {weld id, -, -} (no recipe: welds switched on, groups merged);
                            //          1: rows = {connector of the recipe's site2, connector of its site1, angle bits or NaN} (_connect per row)
  const int *pre_tab;       // [n_pre][3], device memory
  int success_num_conn;     // _success_num_conn (furniture.py:1476-1481)
  int reset_robot_after_attach; // config.reset_robot_after_attach (furniture.py:919-925): _connect ends with _initialize_robot_pos()
};
struct EnvIO {
  const float *action;
  float *obs, *reward;
  uint8_t *done;
  int *info;
  const float *tab_parts, *tab_noise;
  const float *init_state; // set_init_qpos (furniture.py:315-316): [nq + nv] state this env's resets start from, or null
  int n_noise;
  int *nreset;    // device counter: envs that consumed their reset table in this launch (host reads it instead of scanning info)
  int *cost;      // scheduler key written by env_step (shader cycles >> 10 of the step just taken, -1 = will time out next step)
  long long t0;   // shader clock at kernel entry
  const float *tab_attach; // [narmj] joint noise of this env's NEXT attach (fsim_set_attach_noise), or null
  const EnvCfg *cfg_dev; // the handle's copy of the EnvCfg in device memory: what the out-of-line env_reset is given (see EnvResetIO)
  // look-ahead reset (fsim.hip "look-ahead reset"): this env's shadow record / observation row / progress words, or null
  const float *sh_state;
  const void *sh_obs;
  int *sh_prog;   // reset units of the shadow record done so far (env_reset_units); total + 1 = taken by a reset
  int sh_prog0;   // *sh_prog when this env's launch took the env: "is the shadow ready" is decided on THIS value, never on what a look-ahead
                  // job of the same launch writes later (no launch both produces and consumes a shadow; which path a reset takes is not timing)
  int sh_serial;  // serial number of the reset table the shadow record was (is being) computed from
  int tab_serial; // serial number of the table on the device (the host bumps it with every upload)
  int *stats;     // host-mapped counters: [0] resets taken from a shadow record, [1] resets executed inside a step / reset launch, [2] reset units run by look-ahead jobs
};
// What env_reset reads of an env's EnvIO, passed BY VALUE (registers).  env_reset is a real function: handing it the addresses of the
// kernel's EnvCfg (a kernel argument) and EnvIO made the compiler keep both in scratch -- a private copy per LANE, ~100 dwords
// written by every lane of every launch.
struct EnvResetIO {
  const float *tab_parts, *tab_noise, *init_state;
  int n_noise;
};
DEV EnvResetIO env_reset_io(const EnvIO &io) { return EnvResetIO{io.tab_parts, io.tab_noise, io.init_state, io.n_noise}; }

static inline int env_controller_kind(const fsim_config_t &c) { return c.control_type >= 2 && c.control_type <= 6 ? c.control_type - 1 : 0; }
static inline int env_extra_words(const DModel &m, const fsim_config_t &c) {
  const int ik = (c.control_type == 7 || c.control_type == 8) ? EI_WORDS * m.narm : 0;  // the dense-reward env may run under IK control: [dense | ik]
  return m.agent == 2 ? EC_WORDS + 7 * m.nr /* + env_cursor_keep_poses */ : (c.dense_reward ? ED_WORDS + ik : (env_controller_kind(c) ? EK_WORDS : ik));
}

static inline void env_fill_cfg(EnvCfg &e, const fsim_config_t &c, const DModel &m) {
  e.reset_robot_after_attach = c.reset_robot_after_attach ? 1 : 0;
  e.agent = m.agent;
  e.dof_action = m.agent == 0 ? 9 : (m.agent == 1 ? 17 : 15);
  e.obs_dim = 7 * m.nparts + (m.agent == 2 ? 8 : 29 * m.narm);
  e.num_connect_steps = m.agent == 2 ? (c.num_connect_steps > 0 ? c.num_connect_steps : 10) : c.num_connect_steps;
  e.gravity_comp = m.agent == 2;
  e.move_speed = c.move_speed; e.rotate_speed = c.rotate_speed; e.cursor_boundary = c.cursor_boundary;
  e.n_substeps = c.n_substeps; e.max_episode_steps = c.max_episode_steps; e.discrete_grip = c.discrete_grip;
  e.rescale_actions = c.rescale_actions; e.auto_align = c.auto_align; e.auto_reset = c.auto_reset;
  e.has_recipe = 0;
  e.pos_dist = c.alignment_pos_dist; e.rot_up = c.alignment_rot_dist_up; e.rot_fwd = c.alignment_rot_dist_forward; e.proj_dist = c.alignment_project_dist;
  e.ctrl_penalty_coef = c.ctrl_penalty_coef; e.unstable_penalty_coef = c.unstable_penalty_coef; e.success_reward = c.success_reward;
  e.touch_reward = c.touch_reward; e.pick_reward = c.pick_reward;
  e.dense = c.dense_reward; e.dense_nsub = 0; e.dense_coef = nullptr; e.dense_sub = nullptr;
  e.controller = env_controller_kind(c);
  if (e.controller) { // action = [arm command, grip, connect]; robot_ob without joint_pos / joint_vel (furniture_sawyer.py:112-153)
    e.dof_action = (e.controller == CK_POS_ORI ? 6 : (e.controller == CK_POS ? 3 : 7)) + 2;
    e.obs_dim = 7 * m.nparts + 15 * m.narm;
  }
  e.ik = c.control_type == 7 ? 1 : (c.control_type == 8 ? 2 : 0);
  e.obs_bf16 = c.obs_bf16 ? 1 : 0;
  e.n_pre = 0; e.pre_mode = 0; e.pre_tab = nullptr; e.success_num_conn = m.nparts - 1;
  if (e.ik) { // [per arm: dpos 3, rotation 3 | quaternion 4] + one grip per arm + connect (furniture_sawyer.py:52-64, furniture_baxter.py:26-37)
    e.dof_action = m.narm * (3 + (e.ik == 1 ? 3 : 4)) + m.narm + 1;
    e.obs_dim = 7 * m.nparts + 15 * m.narm;
  }
}

// ---------------------------------------------------------------------------------------------------- physics wrappers
template <class Ctx> DEV void fs_touch_flags(const Ctx &c) {
  // who touches whom, from the contact list of this forward pass (data.contact[0:ncon])
  CModel &m = c.m;
  int *scal = c.I(c.ly.scal);
  if (c.lane == 0) { scal[SC_TOUCHL] = 0; scal[SC_TOUCHR] = 0; scal[SC_TOUCHF] = 0; }
  SYNC();
  int nslot = scal[SC_NSLOT];
  for (int s = c.lane; s < nslot; s += 64) {
    const int *ri = c.I(c.ly.con + FSIM_CONW * s);
    if (!ri[C_ACTIVE]) continue;
    int g1 = ri[C_G1], g2 = ri[C_G2];
    for (int k = 0; k < 2; k++) {
      int ga = k ? g2 : g1, gb = k ? g1 : g2; // geom ga against the body of gb
      int part = m.cg_partid[gb];
      if (part < 0) continue;
      int role = m.cg_fingerrole[ga];
      // bits: arm*16 + part  (nparts <= 16, narm <= 2)
      for (int arm = 0; arm < c.D.narm; arm++) {
        if (role & (1 << (2 * arm))) atomicOr(&scal[SC_TOUCHL], 1 << (16 * arm + part));
        if (role & (1 << (2 * arm + 1))) atomicOr(&scal[SC_TOUCHR], 1 << (16 * arm + part));
      }
      if (m.cg_isfloor[ga]) atomicOr(&scal[SC_TOUCHF], 1 << part);
    }
  }
  SYNC();
}

// One forward pass (inlined exactly once, into fs_substeps below).
template <class Ctx> DEV void fs_forward_body(const Ctx &c) {
#ifdef FSIM_PROFILE
  // per-phase shader-clock accounting (development builds only): scal[16..] = cycles of
  // {kinematics+inertia+crb+factor, collide, velocity+smooth, constraints, solve}, then counters
  long long t0_ = clock64(), t1_;
#define FS_PROF(slot) do { t1_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[16 + slot] += (int)((t1_ - t0_) >> 4); t0_ = t1_; } while (0)
#else
#define FS_PROF(slot) do { } while (0)
#endif
  fs_kinematics(c);
  if (c.D.agent == 2 && c.lane < 6) { // data.xpos of the cursor bodies follows model.body_pos at every forward pass
    float *ec = c.L + c.ly.env + E_GROUP + c.D.nparts;
    ec[EC_XPOS + c.lane] = ec[EC_POS + c.lane];
  }
  FS_PROF(16);
  fs_com_inertia(c);
  FS_PROF(17);
  if constexpr (Ctx::NW > 1) {
    // multi-wave: helper wave 1 runs the collision pipeline (it reads the body poses only) beside the dynamics passes here
    mw_post(c, MW_COLLIDE);
    fs_crb_factor(c);
    FS_PROF(18);
    fs_velocity_bias(c);
    FS_PROF(20);
    fs_smooth(c);
    FS_PROF(21);
    mw_post(c, MW_IDLE);
    FS_PROF(1);
  } else {
  fs_crb_factor(c);
  FS_PROF(18);
  fs_collide(c);
  FS_PROF(1);
  fs_velocity_bias(c);
  FS_PROF(20);
  fs_smooth(c);
  FS_PROF(21);
  }
  int coupled = fs_make_constraints(c);
  FS_PROF(3);
#ifdef FSIM_PROFILE
  {
    int bad_ = 0;
    for (int d = c.lane; d < c.D.nv; d += 64) bad_ |= !isfinite(c.L[c.ly.asmooth + d]) | (!isfinite(c.L[c.ly.qfrcbias + d]) << 1);
    int ns_ = c.I(c.ly.scal)[SC_NSLOT];
    for (int s_ = c.lane; s_ < ns_; s_ += 64) { const float *r_ = c.L + c.ly.con + FSIM_CONW * s_; if (c.I(c.ly.con + FSIM_CONW * s_)[C_ACTIVE] == 1) bad_ |= (!isfinite(r_[C_AREF] + r_[C_AREF + 1] + r_[C_AREF + 2] + r_[C_DN]) << 2) | ((fabsf(r_[C_POS]) + fabsf(r_[C_POS + 1]) + fabsf(r_[C_POS + 2]) > 100.f) << 3); }
    bad_ = wave_or(bad_);
    if (bad_ && c.lane == 0 && !c.I(c.ly.scal)[27]) { c.I(c.ly.scal)[27] = 1000 + bad_; c.I(c.ly.scal)[28] = c.I(c.ly.scal)[21]; }
  }
#endif
  fs_solve(c, coupled);
  FS_PROF(4);
#ifdef FSIM_PROFILE
  if (c.lane == 0) {
    int *sc_ = c.I(c.ly.scal);
    sc_[21] += 1; sc_[22] += sc_[SC_NITER]; sc_[23] += coupled; sc_[24] += sc_[SC_NSURV]; sc_[25] += sc_[SC_NSLOT];
    if (sc_[SC_NITER] > sc_[26]) sc_[26] = sc_[SC_NITER];
  }
#endif
  if (c.D.agent == 2) {
    // parts named in a contact with cursor K (on_collision, furniture.py:3290-3310): kept in the env block because the
    // env reads data.contact of the PREVIOUS step when it selects (and LDS does not survive the launch)
    int t0m = 0, t1m = 0;
    int ns = c.I(c.ly.scal)[SC_NSLOT];
    for (int s_ = c.lane; s_ < ns; s_ += 64) {
      const int *ri = c.I(c.ly.con + FSIM_CONW * s_);
      if (!ri[C_ACTIVE]) continue;
      int cm = c.m.cg_cursor[ri[C_G1]] | c.m.cg_cursor[ri[C_G2]], pm = c.m.cg_namepart[ri[C_G1]] | c.m.cg_namepart[ri[C_G2]];
      if (cm & 1) t0m |= pm;
      if (cm & 2) t1m |= pm;
    }
    t0m = wave_or(t0m); t1m = wave_or(t1m);
    if (c.lane == 0) { int *ec = c.I(c.ly.env + E_GROUP + c.D.nparts); ec[EC_TOUCH] = t0m; ec[EC_TOUCH + 1] = t1m; }
  }
#if defined(FSIM_TIMELINE) && defined(FSIM_PROFILE)
  long long tg0_ = clock64();
#endif
  // instability guard (mj_checkPos / mj_checkVel / mj_checkAcc: NaN, Inf or a value beyond 1e10 in qpos, qvel or qacc -- the
  // warnings mujoco_py turns into the MujocoException that _do_simulation catches, furniture.py:2889-2897)
  int bad = 0;
  for (int d = c.lane; d < c.D.nv; d += 64) { float a = c.L[c.ly.x + d], v = c.L[c.ly.qvel + d]; bad |= !(fabsf(a) < 1e10f) | !(fabsf(v) < 1e10f); }
  for (int d = c.lane; d < c.D.nq; d += 64) { float q = c.L[c.ly.qpos + d]; bad |= !(fabsf(q) < 1e10f); }
  bad = wave_or(bad);
  if (bad && c.lane == 0) c.I(c.ly.scal)[SC_BAD] |= 2;
  SYNC();
#if defined(FSIM_TIMELINE) && defined(FSIM_PROFILE)
  { long long tg1_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[18] += (int)((tg1_ - tg0_) >> 4); }
#endif
}

// The ONE out-of-line physics routine: n x (forward [+ finger-touch scan on the last pass] + integrate), or a single
// forward pass.  fs_substeps and env_reset are real (non-inlined) functions because they are large and called from many
// places in the env state machine; the substep LOOP lives inside the callee so that the callee-saved register
// save/restore to scratch (tens of dwords per lane per call) is paid once per env step instead of once per substep --
// with per-substep calls it was 2.6 GB of HBM writes per 4096-env step (rocprofv3 WRITE_SIZE), 300x the state traffic.
//   mode bit 0: forward only (no integration);  mode bit 1: run fs_touch_flags after the last forward pass
//   CTRL (separate instantiation, so the default path's code and register allocation are untouched): the torque-level arm
//   controller runs before every substep (_do_controller_step, furniture.py:3065-3093); pass -1 is the sim.forward() that
//   precedes the loop, whose results the first _pre_action reads.
template <bool CTRL, class Ctx> static FSIM_OUTLINE void fs_substeps_t(Ctx cv, int n_, int mode_) {
#ifdef FSIM_OPAQUE_LANE
  // (development: the lane index is re-read through an opaque copy at the top of every substep, so that the per-lane address
  //  arithmetic of the pass cannot be hoisted out of the loop -- hoisted, those values live across the whole loop and some are spilled)
  extern __shared__ float fs_lds_[];
  Ctx c0_ = fs_rebuild(cv, fs_lds_);
#define FS_LOOP_CTX() { int ln_ = c0_.lane; asm volatile("" : "+v"(ln_)); c0_.lane = ln_; }
  const Ctx &c = c0_;
#else
  FS_REBUILD_CTX(cv);
#define FS_LOOP_CTX() do { } while (0)
#endif
  const int n = __builtin_amdgcn_readfirstlane(n_), mode = __builtin_amdgcn_readfirstlane(mode_);
  // (ONE inlined copy of the forward pass: the forward-only mode leaves the loop after its first pass -- with a call site of its
  //  own the 18 k-instruction body sat in this function twice)
  const bool fwd_only = mode & 1;
#pragma unroll 1
  for (int s = CTRL ? -1 : 0; fwd_only || s < n; s++) {
    FS_LOOP_CTX();
    if (CTRL && s >= 0 && !fwd_only) fs_controller(c, s == 0);
    fs_forward_body(c);
    if (fwd_only) {
      if (mode & 2) fs_touch_flags(c);
      return;
    }
    if (CTRL && s < 0) continue;
    if ((mode & 2) && s == n - 1) fs_touch_flags(c);
#if defined(FSIM_TIMELINE) && defined(FSIM_PROFILE)
    long long ti0_ = clock64();
#endif
    fs_integrate_body(c);
#if defined(FSIM_TIMELINE) && defined(FSIM_PROFILE)
    { long long ti1_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[16] += (int)((ti1_ - ti0_) >> 4); }
#endif
  }
}
template <class Ctx> DEV void fs_substeps(const Ctx &c, int n, int mode) { fs_substeps_t<false>(c, n, mode); }
template <class Ctx> DEV void fs_forward(const Ctx &c) { fs_substeps(c, 1, 1); }
template <class Ctx> DEV void fs_step(const Ctx &c) { fs_substeps(c, 1, 0); }

// ---------------------------------------------------------------------------------------------------- helpers (lane-0 scalar code)
DEV int env_find(int *grp, int i) {
  int r = i;
  while (grp[r] != r) r = grp[r];
  while (grp[i] != r) { int n = grp[i]; grp[i] = r; i = n; }
  return r;
}
template <class Ctx> DEV void env_site_pose(const Ctx &c, int site, V3 *pos, Q4 *quat, M3 *mat) {
  CModel &m = c.m;
  int b = GP(m.s_body)[site];
  Q4 qb = ldq(c.L + c.ly.xquat + 4 * b);
  M3 Rb = ldm3(c.L + c.ly.xmat + 9 * b);
  *pos = ldv3(c.L + c.ly.xpos + 3 * b) + mulv(Rb, ldv3(GP(m.s_pos) + 3 * site));
  Q4 q = qmul(qb, ldq(GP(m.s_quat) + 4 * site));
  if (quat) *quat = q;
  if (mat) *mat = q2m(qnormalized(q));
}
DEV float env_cos(V3 a, V3 b) { return dot(a, b) / norm(a) / norm(b); }
// lookat_to_quat(forward=f, up=u) -> wxyz   (ref transform_utils.py:457-512)
DEV Q4 env_lookat(V3 fwd, V3 up) {
  V3 f = normalized(fwd);
  V3 s = normalized(cross(normalized(up), f));
  V3 u = cross(f, s);
  float m00 = s.x, m01 = s.y, m02 = s.z, m10 = u.x, m11 = u.y, m12 = u.z, m20 = f.x, m21 = f.y, m22 = f.z;
  float tr = (m00 + m11) + m22, x, y, z, w;
  if (tr > 0) { float n = sqrtf(tr + 1); w = n * 0.5f; n = 0.5f / n; x = (m12 - m21) * n; y = (m20 - m02) * n; z = (m01 - m10) * n; }
  else if (m00 >= m11 && m00 >= m22) { float n = sqrtf(((1 + m00) - m11) - m22), k = 0.5f / n; x = 0.5f * n; y = (m01 + m10) * k; z = (m02 + m20) * k; w = (m12 - m21) * k; }
  else if (m11 > m22) { float n = sqrtf(((1 + m11) - m00) - m22), k = 0.5f / n; x = (m10 + m01) * k; y = 0.5f * n; z = (m21 + m12) * k; w = (m20 - m02) * k; }
  else { float n = sqrtf(((1 + m22) - m00) - m11), k = 0.5f / n; x = (m20 + m02) * k; y = (m21 + m12) * k; z = 0.5f * n; w = (m01 - m10) * k; }
  return q4(w, x, y, z);
}
DEV Q4 env_qinv(Q4 q) { float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; Q4 c_ = qconj(q); float s = 1.0f / n2; return q4(c_.w * s, c_.x * s, c_.y * s, c_.z * s); }
// transform_to_target_quat (ref transform_utils.py:641-664); rotate() normalises its quaternion first
DEV void env_ttq(V3 bp, Q4 bq, V3 p, Q4 q, Q4 target, V3 *np_, Q4 *nq) {
  Q4 rel = qmul(target, env_qinv(bq));
  *np_ = qrot(qnormalized(rel), p - bp) + bp;
  *nq = qmul(rel, q);
}
template <class Ctx> DEV void env_stop_part(const Ctx &c, int part, float gravity) {
  CModel &m = c.m;
  float *x = c.L + c.ly.xfrc + 6 * part;
  x[0] = 0; x[1] = 0; x[2] = -gravity * c.D.gravity[2] * GP(m.part_mass)[part]; x[3] = 0; x[4] = 0; x[5] = 0;
  int d = GP(m.part_dofadr)[part];
  for (int k = 0; k < 6; k++) { c.L[c.ly.qvel + d + k] = 0; c.L[c.ly.qfrcapp + d + k] = 0; }
}
// _move_objects_translation_quat (furniture.py:1163-1176): rigidly re-pose the whole weld group of `part`
template <class Ctx> DEV void env_move_group(const Ctx &c, int part, V3 translation, Q4 target, float gravity) {
  CModel &m = c.m;
  int *grp = c.I(c.ly.env + E_GROUP);
  float *qp = c.L + c.ly.qpos;
  int a0 = GP(m.part_qposadr)[part];
  V3 bp = ldv3(qp + a0);
  Q4 bq = ldq(qp + a0 + 3);
  int g = env_find(grp, part);
  for (int i = 0; i < c.D.nparts; i++) {
    if (env_find(grp, i) != g) continue;
    int a = GP(m.part_qposadr)[i];
    V3 np_; Q4 nq;
    env_ttq(bp, bq, ldv3(qp + a), ldq(qp + a + 3), target, &np_, &nq);
    stv3(qp + a, np_ + translation);
    stq(qp + a + 3, nq);
    env_stop_part(c, i, gravity);
  }
}
// site bounding box of a weld group, min/max initialised with 0 (quirk Q1, furniture.py:747-769)
template <class Ctx> DEV void env_bbox(const Ctx &c, int part, V3 *mn, V3 *mx) {
  CModel &m = c.m;
  int *grp = c.I(c.ly.env + E_GROUP);
  int g = env_find(grp, part);
  V3 lo = v3(0, 0, 0), hi = v3(0, 0, 0);
  for (int i = 0; i < c.D.nparts; i++) {
    if (env_find(grp, i) != g) continue;
    for (int k = 0; k < GP(m.part_site_num)[i]; k++) {
      V3 p; env_site_pose(c, GP(m.part_sites)[GP(m.part_site_adr)[i] + k], &p, nullptr, nullptr);
      lo = v3(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z));
      hi = v3(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
    }
  }
  *mn = lo; *mx = hi;
}

template <class Ctx> DEV void env_next_subtask(const Ctx &c) {
  CModel &m = c.m;
  int *E = c.I(c.ly.env);
  int *grp = E + E_GROUP;
  E[E_SUBTASK1] = -1; E[E_SUBTASK2] = -1;
  for (int i = 0; i < c.D.neq; i++) {
    int p1 = GP(m.eq_part1)[i], p2 = GP(m.eq_part2)[i];
    if (env_find(grp, p1) != env_find(grp, p2)) { E[E_SUBTASK1] = p1; E[E_SUBTASK2] = p2; return; }
  }
}

// _is_aligned (furniture.py:1057-1153) on two site poses (world position, rotation matrix), the allowed forward angles of the
// first site (na == 0: any) and the thresholds; writes the target quaternion on the same paths as the reference.  Separate
// from the pose gathering so that the reference's golden vectors can be fed to it directly (fsim_replay_is_aligned).
struct AlignCfg { float pos_dist, rot_up, rot_fwd, proj_dist; };
template <class AngP> DEV bool env_is_aligned_core(V3 p1, const M3 &R1, V3 p2, const M3 &R2, int na, AngP angles, const AlignCfg &cfg, float *tq) {
  V3 up1 = colv(R1, 2), up2 = colv(R2, 2), f1 = colv(R1, 1), f2 = colv(R2, 1);
  float pos_dist = norm(p1 - p2);
  float rot_up = env_cos(up1, up2);
  float proj12 = dot(up1, (p2 - p1) * (1.0f / norm(p2 - p1)));
  float proj21 = dot(up2, (p1 - p2) * (1.0f / norm(p1 - p2)));
  bool fwd_ok;
  if (na == 0) {
    fwd_ok = true;
    float cs = env_cos(f1, f2);
    V3 k = normalized(up1);
    float sn = sqrtf(1 - cs * cs);
    V3 rp = cs * f1 + sn * cross(k, f1), rn = cs * f1 - sn * cross(k, f1);
    V3 fr = env_cos(rp, f2) > env_cos(rn, f2) ? rp : rn;
    stq(tq, env_lookat(up1, fr));
  } else {
    fwd_ok = false;
    V3 k = normalized(up1);
    for (int a = 0; a < na; a++) {
      float ang = angles[a] / 180.0f * 3.14159265358979f;
      V3 fr = cosf(ang) * f1 + sinf(ang) * cross(k, f1);
      if (env_cos(fr, f2) > cfg.rot_fwd) { fwd_ok = true; stq(tq, env_lookat(up1, fr)); break; }
    }
  }
  if (pos_dist < cfg.pos_dist && rot_up > cfg.rot_up && fwd_ok && fabsf(proj12) > cfg.proj_dist && fabsf(proj21) > cfg.proj_dist) return true;
  if (pos_dist < cfg.pos_dist / 2 && rot_up > cfg.rot_up && fwd_ok) return true;
  return false;
}
// for connector indices k1, k2 of the env's model, on the poses of the last forward pass
template <class Ctx> DEV bool env_is_aligned(const Ctx &c, const EnvCfg &cfg, int k1, int k2) {
  CModel &m = c.m;
  V3 p1, p2; M3 R1, R2;
  env_site_pose(c, GP(m.conn_siteid)[k1], &p1, nullptr, &R1);
  env_site_pose(c, GP(m.conn_siteid)[k2], &p2, nullptr, &R2);
  AlignCfg ac;
  ac.pos_dist = cfg.pos_dist; ac.rot_up = cfg.rot_up; ac.rot_fwd = cfg.rot_fwd; ac.proj_dist = cfg.proj_dist;
  return env_is_aligned_core(p1, R1, p2, R2, GP(m.conn_nangle)[k1], GP(m.conn_angles) + FSIM_MAXANG * k1, ac, c.L + c.ly.env + E_TARGET_QUAT);
}

// sensor values of the dense reward from the poses of the last forward pass (furniture_sawyer_dense.py:222-271)
template <class Ctx> struct DenseSimP {
  const Ctx &c;
  const EnvCfg &cfg;
  DEV void obs(int st, DObs &o) const {
    CModel &m = c.m;
    auto C = GP(cfg.dense_coef);
    auto T = GP(cfg.dense_sub) + DS_WORDS * st;
    M3 R;
    env_site_pose(c, (int)C[DC_GRIPTIP_SITE], &o.eef, nullptr, nullptr);
    V3 dummy;
    env_site_pose(c, (int)C[DC_GRIP_SITE], &dummy, nullptr, &R);
    o.gripup = colv(R, 2); o.gripfwd = colv(R, 1);
    env_site_pose(c, (int)T[DS_GL_SITE], &o.gl, nullptr, nullptr);
    env_site_pose(c, (int)T[DS_GR_SITE], &o.gr, nullptr, nullptr);
    int leg = (int)T[DS_LEG_PART];
    o.leg = ldv3(c.L + c.ly.xpos + 3 * GP(m.part_rbody)[leg]);
    env_site_pose(c, (int)T[DS_LEG_SITE], &o.legsite, nullptr, &R);
    o.legup = colv(R, 2); o.legfwd = colv(R, 1);
    env_site_pose(c, (int)T[DS_TABLE_SITE], &o.tablesite, nullptr, &R);
    o.tableup = colv(R, 2); o.tablefwd = colv(R, 1);
    // _finger_contact(leg) of the (single) arm: furniture_sawyer.py:220-245
    const int *scal = c.I(c.ly.scal);
    o.touch_l = (scal[SC_TOUCHL] >> leg) & 1; o.touch_r = (scal[SC_TOUCHR] >> leg) & 1;
  }
  DEV bool aligned(int st) const {
    auto T = GP(cfg.dense_sub) + DS_WORDS * st;
    return env_is_aligned(c, cfg, (int)T[DS_K_LEG], (int)T[DS_K_TABLE]);
  }
};
template <class Ctx> DEV float *env_edense(const Ctx &c) { return c.L + c.ly.env + E_GROUP + c.D.nparts; }

// ---------------------------------------------------------------------------------------------------- connect
template <class Ctx> DEV int env_ecur(const Ctx &c) { return c.ly.env + E_GROUP + c.D.nparts; }
// Cursor agent: data.xpos / data.xquat as the last forward pass left them, kept in the record behind the EC_* block.  _step_discrete runs BEFORE the
// step's first forward pass and reads site poses there (_try_connect -> _is_aligned, the approach target: furniture.py:926-1042) -- in the reference what the
// previous step's last forward pass computed, one integration older than qpos.  The pose arrays live in LDS and do not survive the launch, so every launch
// that ends with the record stored keeps them (env_run, env_shadow_job, k_physics) and the Cursor step restores them before _step_discrete.
template <class Ctx> DEV void env_cursor_keep_poses(const Ctx &c) {
  if (c.D.agent != 2) return;
  float *st = c.L + env_ecur(c) + EC_WORDS;
  for (int i = c.lane; i < 3 * c.D.nr; i += 64) st[i] = c.L[c.ly.xpos + i];
  for (int i = c.lane; i < 4 * c.D.nr; i += 64) st[3 * c.D.nr + i] = c.L[c.ly.xquat + i];
  SYNC();
}
template <class Ctx> DEV void env_cursor_restore_poses(const Ctx &c) {
  const float *st = c.L + env_ecur(c) + EC_WORDS;
  for (int i = c.lane; i < 3 * c.D.nr; i += 64) c.L[c.ly.xpos + i] = st[i];
  for (int i = c.lane; i < 4 * c.D.nr; i += 64) c.L[c.ly.xquat + i] = st[3 * c.D.nr + i];
  SYNC();
  for (int b = c.lane; b < c.D.nr; b += 64) stm3(c.L + c.ly.xmat + 9 * b, q2m(qnormalized(ldq(c.L + c.ly.xquat + 4 * b))));
  SYNC();
}

// euler_to_quat(rotation_deg, quat) = quat * (qz * qy * qx)   (transform_utils.py:617-630)
DEV Q4 env_euler_quat(V3 deg, Q4 q) {
  const float k = 3.14159265358979f / 180.0f;
  Q4 qx = axisangle(v3(1, 0, 0), deg.x * k), qy = axisangle(v3(0, 1, 0), deg.y * k), qz = axisangle(v3(0, 0, 1), deg.z * k);
  return qmul(q, qmul(qz, qmul(qy, qx)));
}
// quat_slerp (transform_utils.py:122-160), shortest path, no spin
DEV Q4 env_slerp(Q4 q0, Q4 q1, float fraction) {
  Q4 a = qnormalized(q0), b = qnormalized(q1);
  if (fraction == 0.0f) return a;
  if (fraction == 1.0f) return b;
  float d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
  const float eps = 8.8817841970012523e-16f; // numpy float64 eps * 4, as in the reference
  if (fabsf(fabsf(d) - 1.0f) < eps) return a;
  if (d < 0.0f) { d = -d; b = q4(-b.w, -b.x, -b.y, -b.z); }
  d = fminf(d, 1.0f);
  float ang = acosf(d);
  if (fabsf(ang) < eps) return a;
  float isin = 1.0f / sinf(ang), sa = sinf((1.0f - fraction) * ang) * isin, sb = sinf(fraction * ang) * isin;
  return q4(a.w * sa + b.w * sb, a.x * sa + b.x * sb, a.y * sa + b.y * sb, a.z * sa + b.z * sb);
}

// _stop_selected_objects (furniture.py:771-779): every part in a selected group is frozen with gravity compensation
template <class Ctx> DEV void env_stop_selected(const Ctx &c, float gravity) {
  CModel &m = c.m;
  if (c.lane == 0) {
    int *grp = c.I(c.ly.env + E_GROUP);
    const int *ec = c.I(env_ecur(c));
    for (int i = 0; i < c.D.nparts; i++) {
      int g = env_find(grp, i);
      for (int k = 0; k < 2; k++)
        if (ec[EC_SEL + k] && env_find(grp, ec[EC_SEL + k] - 1) == g) { env_stop_part(c, i, gravity); break; }
    }
  }
  SYNC();
}

// _move_rotate_object (furniture.py:708-745): rigidly rotate the weld group of `part` by rot_deg about the part and shift
// it by `move`, validate with one forward+step and the site bounding box (_is_inside), undo the poses if it left the
// workspace.  returns (wave-uniform) 1 if the move was kept.
template <class Ctx> DEV int env_move_rotate(const Ctx &c, int part, V3 move, V3 rot_deg, float bnd) {
  CModel &m = c.m;
  int *grp = c.I(c.ly.env + E_GROUP);
  int *scal = c.I(c.ly.scal);
  // old part poses stay in registers (lane i keeps word i of the [nparts][7] pose table) so the move can be undone
  float keep[2] = {0, 0};
  for (int r = 0; r < 2; r++) { int i = c.lane + 64 * r; if (i < 7 * c.D.nparts) keep[r] = c.L[c.ly.qpos + GP(m.part_qposadr)[i / 7] + i % 7]; }
  SYNC();
  if (c.lane == 0) {
    int g = env_find(grp, part);
    int a0 = GP(m.part_qposadr)[part];
    Q4 bq = ldq(c.L + c.ly.qpos + a0 + 3);
    V3 bp = ldv3(c.L + c.ly.qpos + a0);
    Q4 target = env_euler_quat(rot_deg, bq);
    for (int i = 0; i < c.D.nparts; i++) {
      if (env_find(grp, i) != g) continue;
      int a = GP(m.part_qposadr)[i];
      V3 np_; Q4 nq;
      env_ttq(bp, bq, ldv3(c.L + c.ly.qpos + a), ldq(c.L + c.ly.qpos + a + 3), target, &np_, &nq);
      stv3(c.L + c.ly.qpos + a, np_ + move);
      stq(c.L + c.ly.qpos + a + 3, nq);
    }
  }
  SYNC();
  fs_step(c);
  if (c.lane == 0) {
    V3 mn, mx; env_bbox(c, part, &mn, &mx);
    int inside = !(mn.x < -bnd || mn.y < -bnd || mn.z < -0.05f || mx.x > bnd || mx.y > bnd || mx.z > bnd);
    scal[12] = inside;
    scal[13] = env_find(grp, part);
  }
  SYNC();
  int inside = scal[12];
  if (!inside) {
    int g = scal[13];
    for (int r = 0; r < 2; r++) {
      int i = c.lane + 64 * r;
      if (i < 7 * c.D.nparts) { int pi = i / 7; if (env_find(grp, pi) == g) c.L[c.ly.qpos + GP(m.part_qposadr)[pi] + i % 7] = keep[r]; }
    }
    SYNC();
  }
  return inside;
}

// The search of _try_connect (furniture.py:946-991), lane-0 scalar code: the first (site1 on group(part1), site2 on group(part2)
// or on any part) in connector order that is unused, name-matched and aligned.  `aligned(k1, k2)` is the alignment test --
// env_is_aligned in the env, a recorded truth table in fsim_replay_try_connect.
// returns false where the reference returns early (furniture.py:964-973: a group without connector sites, or no <weld> between
// the two groups) -- those paths leave _connect_step untouched
template <class Ctx, class AlignedFn> DEV bool env_connect_search(const Ctx &c, int part1, int part2, AlignedFn aligned, int *f1, int *f2) {
  CModel &m = c.m;
  int *E = c.I(c.ly.env);
  int *grp = E + E_GROUP;
  int found1 = -1, found2 = -1;
  *f1 = -1; *f2 = -1;
  int g1 = env_find(grp, part1), g2 = part2 >= 0 ? env_find(grp, part2) : -1;
  int n1 = 0, n2 = 0;
  for (int k = 0; k < c.D.nconn; k++) {
    int gk = env_find(grp, GP(m.conn_partid)[k]);
    n1 += gk == g1; n2 += g2 < 0 || gk == g2;
  }
  if (n1 == 0 || n2 == 0) return false;
  bool weld_ok = c.D.neq > 0;
  if (weld_ok && part2 >= 0) { // some <weld> must join two bodies of group(part1) U group(part2) (activity is not checked)
    weld_ok = false;
    for (int i = 0; i < c.D.neq && !weld_ok; i++) {
      int ga = env_find(grp, GP(m.eq_part1)[i]), gb = env_find(grp, GP(m.eq_part2)[i]);
      weld_ok = (ga == g1 || ga == g2) && (gb == g1 || gb == g2);
    }
  }
  if (!weld_ok) return false;
  {
    for (int k1 = 0; k1 < c.D.nconn && found1 < 0; k1++) {
      if (env_find(grp, GP(m.conn_partid)[k1]) != g1) continue;
      for (int k2 = 0; k2 < c.D.nconn; k2++) {
        if (g2 >= 0 && env_find(grp, GP(m.conn_partid)[k2]) != g2) continue;
        if ((E[E_CONNSITES0 + (k1 >> 5)] >> (k1 & 31)) & 1) continue;
        if ((E[E_CONNSITES0 + (k2 >> 5)] >> (k2 & 31)) & 1) continue;
        int a1 = GP(m.conn_keya)[k1], b1 = GP(m.conn_keyb)[k1], a2 = GP(m.conn_keya)[k2], b2 = GP(m.conn_keyb)[k2];
        bool match = (b1 < 0 || b2 < 0) ? (b1 < 0 && b2 < 0 && a1 == a2) : (a1 == b2 && b1 == a2);
        if (!match) continue;
        if (aligned(k1, k2)) { found1 = k1; found2 = k2; break; }
      }
    }
  }
  *f1 = found1; *f2 = found2;
  return true;
}
// _connect_step bookkeeping (furniture.py:993-1040): 1 = approach the aligned pose by one increment (found pair, counter below
// num_connect_steps; the counter is advanced), 0 = connect now if a pair was found (the counter restarts)
DEV int env_connect_decide(int *E, int num_connect_steps, bool searched, int found1) {
  if (!searched) return 0; // the reference returned before touching the counter
  if (found1 >= 0 && E[E_CONNECT_STEP] < num_connect_steps) { E[E_CONNECT_STEP] += 1; return 1; }
  E[E_CONNECT_STEP] = 0;
  return 0;
}
// finger-touch connect scan of _step_continuous (furniture.py:1290-1330) on the touch masks of the last forward pass: per arm the
// first part (in part order) touched by BOTH fingers is tried; a successful attempt ends the scan (quirk Q4).  try_connect(part)
// returns wave-uniform 1 if a connection was made.
template <class TryFn> DEV void env_finger_scan(int narm, const int *scal, TryFn try_connect) {
  int done_connect = 0;
  for (int arm = 0; arm < narm && !done_connect; arm++) {
    int both = (scal[SC_TOUCHL] >> (16 * arm)) & (scal[SC_TOUCHR] >> (16 * arm)) & 0xffff;
    if (!both) continue;
    int part = __ffs(both) - 1;
    done_connect = try_connect(part); // break after the first pinched part either way
  }
}

// _try_connect(part1, part2) (furniture.py:926-1042).  part2 < 0: any part (the arm agents).  returns (wave-uniform) 1 if a
// connection was made; with num_connect_steps > 0 (Cursor) an aligned pair is first approached over that many calls.
// _connect(site1, site2) (furniture.py:847-925) for connector indices k1, k2; the target orientation is in E_TARGET_QUAT
template <class Ctx> DEV void env_connect(const Ctx &c, const EnvCfg &cfg, int k1, int k2, bool auto_align) {
  CModel &m = c.m;
  int *E = c.I(c.ly.env);
  int *grp = E + E_GROUP;
  int *scal = c.I(c.ly.scal);
  int pA = GP(m.conn_partid)[k1], pB = GP(m.conn_partid)[k2];
  if (c.lane == 0) {
    E[E_CONNSITES0 + (k1 >> 5)] |= 1 << (k1 & 31);
    E[E_CONNSITES0 + (k2 >> 5)] |= 1 << (k2 & 31);
    E[E_SITE1] = GP(m.conn_siteid)[k1]; E[E_SITE2] = GP(m.conn_siteid)[k2];
    int gA = env_find(grp, pA), gB = env_find(grp, pB);
    int *ct = c.I(c.ly.contype), *ca = c.I(c.ly.conaff);
    for (int g = 0; g < c.D.ncg; g++) {
      int p = m.cg_partid[g];
      if (p < 0) continue;
      int gp = env_find(grp, p);
      if ((gp == gA || gp == gB) && ct[g] != 0) { ct[g] = (1 << 30) - 1 - (1 << (gA + 1)); ca[g] = 1 << (gA + 1); }
    }
    if (auto_align) {
      // _align_connectors -> _move_site_to_target(site2, [site1 pos, target quat])
      V3 s1p, s2p; Q4 s2q;
      env_site_pose(c, GP(m.conn_siteid)[k1], &s1p, nullptr, nullptr);
      env_site_pose(c, GP(m.conn_siteid)[k2], &s2p, &s2q, nullptr);
      Q4 target = ldq(c.L + c.ly.env + E_TARGET_QUAT);
      int a = GP(m.part_qposadr)[pB];
      V3 bp = ldv3(c.L + c.ly.qpos + a); Q4 bq = ldq(c.L + c.ly.qpos + a + 3);
      V3 npos; Q4 nquat;
      env_ttq(s2p, s2q, bp, bq, target, &npos, &nquat);
      V3 nsp; Q4 nsq;
      env_ttq(bp, bq, s2p, s2q, nquat, &nsp, &nsq);
      env_move_group(c, pB, s1p - nsp, nquat, cfg.gravity_comp ? 1.0f : 0.0f);
    }
  }
  SYNC();
  if (c.D.agent == 2) env_stop_selected(c, 1.0f);
  fs_step(c);
  if (c.lane == 0) {
    V3 mn1, mx1, mn2, mx2;
    env_bbox(c, pA, &mn1, &mx1); env_bbox(c, pB, &mn2, &mx2);
    float mz = fminf(mn1.z, mn2.z);
    reinterpret_cast<float *>(scal)[11] = mz;
  }
  SYNC();
  float mz = reinterpret_cast<float *>(scal)[11];
  if (mz < 0) {
    // _move_rotate_object(body, [0,0,-min z], [0,0,0]) for both bodies; each validates with forward+step (_is_inside)
    env_move_rotate(c, pA, v3(0, 0, -mz), v3(0, 0, 0), cfg.cursor_boundary);
    env_move_rotate(c, pB, v3(0, 0, -mz), v3(0, 0, 0), cfg.cursor_boundary);
  }
  if (c.D.agent == 2) env_stop_selected(c, 1.0f);
  fs_step(c);
  if (c.lane == 0) {
    // _activate_weld(body1, body2)
    for (int i = 0; i < c.D.neq; i++) {
      int p1 = GP(m.eq_part1)[i], p2 = GP(m.eq_part2)[i];
      if ((p1 == pA || p1 == pB) && (p2 == pA || p2 == pB)) {
        int a1 = GP(m.part_qposadr)[p1], a2 = GP(m.part_qposadr)[p2];
        Q4 q1i = env_qinv(ldq(c.L + c.ly.qpos + a1 + 3));
        Q4 rq = qmul(q1i, ldq(c.L + c.ly.qpos + a2 + 3));
        V3 rp = qrot(qnormalized(q1i), ldv3(c.L + c.ly.qpos + a2) - ldv3(c.L + c.ly.qpos + a1));
        float *ed = c.L + c.ly.eqdata + 7 * i;
        stv3(ed, rp); stq(ed + 3, rq);
        c.I(c.ly.eqactive)[i] = 1;
        int r1 = env_find(grp, pA), r2 = env_find(grp, pB);
        grp[r1] = r2;
      }
    }
    if (c.D.agent == 2) c.I(env_ecur(c))[EC_SEL + 1] = 0; // furniture.py:914-915
    E[E_NUM_CONNECTED] += 1;
    E[E_CONNECTED_THIS_STEP] = 1;
    E[E_CONNBODY1] = pA + 1;
    int a = GP(m.part_qposadr)[pA];
    for (int k = 0; k < 7; k++) c.L[c.ly.env + E_CB1_POS + k] = c.L[c.ly.qpos + a + k];
    env_next_subtask(c);
  }
  SYNC();
}

// _project_connector_quat(connector1, connector2, angle) (furniture.py:1201-1222): connector2's orientation when aligned
// with connector1 -- the target of _is_aligned without its thresholds.  Lane 0, on the poses of the last forward pass.
template <class Ctx> DEV void env_project_connector_quat(const Ctx &c, int k1, int k2, bool has_angle, float angle_deg) {
  CModel &m = c.m;
  V3 p1, p2; M3 R1, R2;
  env_site_pose(c, GP(m.conn_siteid)[k1], &p1, nullptr, &R1);
  env_site_pose(c, GP(m.conn_siteid)[k2], &p2, nullptr, &R2);
  const V3 up1 = colv(R1, 2), f1 = colv(R1, 1), f2 = colv(R2, 1), k = normalized(up1);
  V3 fr;
  if (!has_angle) {
    const float cs = env_cos(f1, f2), sn = sqrtf(1 - cs * cs);
    const V3 rp = cs * f1 + sn * cross(k, f1), rn = cs * f1 - sn * cross(k, f1);
    fr = env_cos(rp, f2) > env_cos(rn, f2) ? rp : rn;
  } else {
    const float ang = angle_deg / 180.0f * 3.14159265358979f;
    fr = cosf(ang) * f1 + sinf(ang) * cross(k, f1);
  }
  stq(c.L + c.ly.env + E_TARGET_QUAT, env_lookat(up1, fr));
}

template <class Ctx> DEV int env_try_connect(const Ctx &c, const EnvCfg &cfg, int part1, int part2) {
  CModel &m = c.m;
  int *E = c.I(c.ly.env);
  int *grp = E + E_GROUP;
  int *scal = c.I(c.ly.scal);
  // ---- lane 0: search the first aligned (site1, site2) pair in site-id order
  if (c.lane == 0) {
    int found1, found2;
    const bool searched = env_connect_search(c, part1, part2, [&](int k1, int k2) { return env_is_aligned(c, cfg, k1, k2); }, &found1, &found2);
    if (env_connect_decide(E, cfg.num_connect_steps, searched, found1) == 1) {
      // approach phase (furniture.py:993-1034): slerp / lerp part2's group towards the aligned pose, one increment per call
      const int n = cfg.num_connect_steps, step = E[E_CONNECT_STEP] - 1; // (already advanced by env_connect_decide)
      float *ec = c.L + env_ecur(c);
      int p2 = GP(m.conn_partid)[found2], a = GP(m.part_qposadr)[p2];
      V3 p2p = ldv3(c.L + c.ly.qpos + a); Q4 p2q = ldq(c.L + c.ly.qpos + a + 3);
      if (step == 0) {
        V3 s1p, s2p; Q4 s2q;
        env_site_pose(c, GP(m.conn_siteid)[found1], &s1p, nullptr, nullptr);
        env_site_pose(c, GP(m.conn_siteid)[found2], &s2p, &s2q, nullptr);
        V3 bpos; Q4 brot;
        env_ttq(s2p, s2q, p2p, p2q, ldq(c.L + c.ly.env + E_TARGET_QUAT), &bpos, &brot);
        bpos = bpos + (s1p - s2p);
        stv3(ec + EC_P2Q0, p2p); stq(ec + EC_P2Q0 + 3, p2q); stv3(ec + EC_BODY_POS, bpos); stq(ec + EC_BODY_ROT, brot);
      }
      V3 p0 = ldv3(ec + EC_P2Q0), bpos = ldv3(ec + EC_BODY_POS);
      Q4 q0 = ldq(ec + EC_P2Q0 + 3), brot = ldq(ec + EC_BODY_ROT);
      float lo = 1.0f / n, x = n > 1 ? lo + (0.9f - lo) * step / (n - 1) : lo; // np.linspace(1/n, 0.9, n)[step]
      V3 npos = p0 + (bpos - p0) * x;
      Q4 nrot = env_slerp(q0, brot, (float)(step + 1) / n);
      env_move_group(c, p2, npos - p2p, nrot, 1.0f);
      found1 = -1;
    }
    scal[9] = found1; scal[10] = found2;
  }
  SYNC();
  int k1 = scal[9], k2 = scal[10];
  if (k1 < 0) return 0;
  env_connect(c, cfg, k1, k2, cfg.auto_align != 0);
  return 1;
}

// ---------------------------------------------------------------------------------------------------- Cursor agent
// _step_discrete (furniture.py:800-845) + helpers _move_cursor / _select_object (furniture.py:700-798, 3290-3310).
template <class Ctx> DEV void env_cursor_discrete(const Ctx &c, const EnvCfg &cfg, const float *a) {
  CModel &m = c.m;
  int *E = c.I(c.ly.env);
  int *grp = E + E_GROUP;
  int *scal = c.I(c.ly.scal);
  int *eci = c.I(env_ecur(c));
  float *ecf = c.L + env_ecur(c);
  const float b = cfg.cursor_boundary;
  for (int k = 0; k < 2; k++) {
    V3 move = v3(a[7 * k], a[7 * k + 1], a[7 * k + 2]) * cfg.move_speed;
    V3 rot = v3(a[7 * k + 3], a[7 * k + 4], a[7 * k + 5]) * cfg.rotate_speed;
    bool select = a[7 * k + 6] > 0;
    if (c.lane == 0) {
      if (!select) eci[EC_SEL + k] = 0;
      V3 pos = ldv3(ecf + EC_XPOS + 3 * k) + move; // _cursor_pos() reads data.xpos
      int ok = fabsf(pos.x) < b && fabsf(pos.y) < b && fabsf(pos.z) < b && pos.z >= cfg.move_speed * 0.45f;
      if (ok) stv3(ecf + EC_POS + 3 * k, pos);
      scal[9] = ok; scal[10] = eci[EC_SEL + k];
    }
    SYNC();
    if (!scal[9]) continue;
    int sel = scal[10];
    if (sel) {
      if (!env_move_rotate(c, sel - 1, move, rot, b)) {
        if (c.lane == 0) { // _move_cursor(k, -move): data.xpos already reflects the moved cursor (forward ran inside _is_inside)
          V3 pos = ldv3(ecf + EC_XPOS + 3 * k) - move;
          if (fabsf(pos.x) < b && fabsf(pos.y) < b && fabsf(pos.z) < b && pos.z >= cfg.move_speed * 0.45f) stv3(ecf + EC_POS + 3 * k, pos);
        }
        SYNC();
        continue;
      }
    }
    if (c.lane == 0 && select && eci[EC_SEL + k] == 0) {
      // _select_object: first part (in part order) not in an already selected group that touches this cursor
      int hit = 0;
      for (int i = 0; i < c.D.nparts && !hit; i++) {
        int g = env_find(grp, i);
        bool taken = false;
        for (int q = 0; q < 2; q++) if (eci[EC_SEL + q] && env_find(grp, eci[EC_SEL + q] - 1) == g) taken = true;
        if (taken) continue;
        if ((eci[EC_TOUCH + k] >> i) & 1) hit = i + 1;
      }
      eci[EC_SEL + k] = hit;
    }
    SYNC();
  }
  float connect = a[14];
  int s0 = eci[EC_SEL], s1 = eci[EC_SEL + 1];
  if (connect > 0 && s0 && s1) env_try_connect(c, cfg, s0 - 1, s1 - 1);
  else { if (c.lane == 0 && E[E_CONNECT_STEP] > 0) E[E_CONNECT_STEP] = 0; SYNC(); }
}

// ---------------------------------------------------------------------------------------------------- observation / reward
template <class Ctx> DEV void env_write_obs(const Ctx &c, const EnvCfg &cfg, const EnvIO &io) {
  const int cfg_ik = Ctx::PLAIN ? 0 : cfg.ik, cfg_controller = Ctx::PLAIN ? 0 : cfg.controller; // (SpecCtx::PLAIN)
  if (!io.obs) return;
  CModel &m = c.m;
  const float *L = c.L;
  // the observation is assembled in LDS (the Hessian pair-block area: dead outside fs_hessian) and stored in one coalesced pass,
  // as float32 or -- fsim_config_t::obs_bf16 -- as bfloat16
  float *ob = c.L + c.ly.hP;
  if (io.info && c.lane == 0) { // the subtask of the state being observed (after an in-kernel auto-reset: the new episode's)
    io.info[FSIM_INFO_SUBTASK1] = c.I(c.ly.env)[E_SUBTASK1];
    io.info[FSIM_INFO_SUBTASK2] = c.I(c.ly.env)[E_SUBTASK2];
  }
  // object_ob: body xpos/xquat of every part as left by the last forward pass
  for (int i = c.lane; i < 7 * c.D.nparts; i += 64) {
    int p = i / 7, k = i % 7, b = GP(m.part_rbody)[p];
    ob[i] = k < 3 ? L[c.ly.xpos + 3 * b + k] : L[c.ly.xquat + 4 * b + k - 3];
  }
  int base = 7 * c.D.nparts;
  // data.site_xvelp / site_xvelr in mujoco_py are jac(site) . qvel: the Jacobian of the LAST forward pass (one integration
  // old after sim.step()) times the CURRENT qvel -- not mj_objectVelocity's cvel of that pass.
  if (c.D.narm) fs_body_spatial(c, c.ly.qvel);
  if (c.D.agent == 2) { // furniture_cursor.py:88-109: [cursor0 pos, cursor1 pos, selected0, selected1]
    const float *ec = L + c.ly.env + E_GROUP + c.D.nparts;
    if (c.lane < 6) ob[base + c.lane] = ec[EC_XPOS + c.lane];
    if (c.lane < 2) ob[base + 6 + c.lane] = reinterpret_cast<const int *>(ec)[EC_SEL + c.lane] ? 1.0f : 0.0f;
  }
  for (int arm = 0; arm < c.D.narm; arm++) {
    const int njm = c.D.narmj / c.D.narm;
    // joint_pos / joint_vel are part of robot_ob for impedance / torque only (furniture_sawyer.py:112-124)
    const int nj = (cfg_controller || cfg_ik) ? 0 : njm;
    float *o = ob + base + (2 * nj + 15) * arm;
    for (int k = c.lane; k < nj; k += 64) {
      o[k] = L[c.ly.qpos + GP(m.arm_qposadr)[arm * njm + k]];
      o[nj + k] = L[c.ly.qvel + GP(m.arm_dofadr)[arm * njm + k]];
    }
    if (c.lane < 2) o[2 * nj + c.lane] = L[c.ly.qpos + GP(m.grip_qposadr)[2 * arm + c.lane]];
    if (c.lane == 0) {
      int site = GP(m.eef_siteid)[arm];
      V3 sp; env_site_pose(c, site, &sp, nullptr, nullptr);
      stv3(o + 2 * nj + 2, sp);
      int hb = GP(m.hand_body)[arm], rb = GP(m.body_red)[hb];
      Q4 q = qmul(ldq(L + c.ly.xquat + 4 * rb), ldq(GP(m.body_relquat) + 4 * hb));
      o[2 * nj + 5] = q.x; o[2 * nj + 6] = q.y; o[2 * nj + 7] = q.z; o[2 * nj + 8] = q.w;
      int sb = GP(m.s_body)[site];
      S6 v = lds6(L + c.ly.W + 6 * sb);
      V3 vp = sb ? v.l + cross(v.a, sp - ldv3(L + c.ly.com + 3 * KI(r_tree, sb))) : v3(0, 0, 0);
      stv3(o + 2 * nj + 9, vp);
      stv3(o + 2 * nj + 12, sb ? v.a : v3(0, 0, 0));
    }
  }
  SYNC();
  if (cfg.obs_bf16) {
    unsigned short *o16 = reinterpret_cast<unsigned short *>(io.obs);
    for (int i = c.lane; i < cfg.obs_dim; i += 64) {
      const unsigned u = __float_as_uint(ob[i]);
      o16[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); // round to nearest even (finite values; NaN stays NaN)
    }
  } else
    for (int i = c.lane; i < cfg.obs_dim; i += 64) io.obs[i] = ob[i];
}

// ---------------------------------------------------------------------------------------------------- reset
template <class Ctx> DEV void env_gravity_comp(const Ctx &c) {
  CModel &m = c.m;
  for (int k = c.lane; k < c.D.narmj; k += 64) c.L[c.ly.qfrcapp + GP(m.arm_dofadr)[k]] = c.L[c.ly.qfrcbias + GP(m.arm_dofadr)[k]];
  for (int k = c.lane; k < c.D.ngripj; k += 64) c.L[c.ly.qfrcapp + GP(m.grip_dofadr)[k]] = c.L[c.ly.qfrcbias + GP(m.grip_dofadr)[k]];
  SYNC();
}
template <class Ctx> DEV void env_init_robot(const Ctx &c, const EnvResetIO &io, int draw, float move_speed) {
  CModel &m = c.m;
  if (c.D.agent == 2 && c.lane < 2) { // furniture.py:1763-1768: cursors at x = -+0.2, half a move step above the floor
    float *p = c.L + c.ly.env + E_GROUP + c.D.nparts + EC_POS + 3 * c.lane;
    p[0] = c.lane ? 0.2f : -0.2f; p[1] = 0.0f; p[2] = move_speed * 0.5f;
  }
  for (int k = c.lane; k < c.D.narmj; k += 64) {
    float noise = io.tab_noise ? io.tab_noise[(size_t)min(draw, io.n_noise - 1) * c.D.narmj + k] : 0.0f;
    c.L[c.ly.qpos + GP(m.arm_qposadr)[k]] = GP(m.arm_initqpos)[k] + noise;
  }
  for (int k = c.lane; k < c.D.ngripj; k += 64) c.L[c.ly.qpos + GP(m.grip_qposadr)[k]] = GP(m.grip_initqpos)[k];
  SYNC();
}
// The reset as a sequence of UNITS, one physics substep each (the reference's _reset calls sim.step() 401 times with a recipe, 301
// without, 100 from set_init_qpos: tests/golden/reset_trace.npz), so that it can be run whole -- env_reset -- or in pieces whose
// intermediate state travels in the env record (the look-ahead reset computes an env's NEXT reset a few dozen units per launch in
// waves the step kernel would otherwise retire, fsim.hip env_shadow_job).  Unit u of the normal path, NS = 100 (+ 100 with a recipe):
//   u = 0           before it: sim.reset(), masks, env block, part placement
//   u < NS          settling: every tenth unit starts by stopping the parts; step; _slow_objects.  u = 100 (recipe) starts with the
//                   pre-assembled connects
//   u = NS          gravity compensation, robot at its initial pose (draw 0), step, robot collision on, gravity compensation
//   u <= NS + 100   robot at its initial pose (draw u - NS), step
//   u >  NS + 100   the common tail: (first unit: controls cleared, forward pass, gravity compensation) step
// set_init_qpos path: 100 tail units.  After the last unit: IK sync, subtask, dense-reward variables.
// The SAME loop serves both uses, so running it in pieces executes the same instructions on the same values (every derived quantity
// is rebuilt by the forward pass of each step; what persists is the record): bit-identical, tests/test_lookahead_gpu.py.
DEV int env_reset_total(const EnvCfg &cfg, bool init) { return init ? 100 : (100 + (cfg.has_recipe ? 100 : 0) + 201); }
template <class Ctx0> static FSIM_OUTLINE void env_reset_units(Ctx0 cv, const EnvCfg *cfgp, const EnvResetIO io, int p0_, int p1_) {
  extern __shared__ float fs_lds_[];
  typedef FsIn<Ctx0> Ctx; // (see FsIn: the physics routine called from here is not the one the step calls)
  const Ctx c(fs_rebuild(cv, fs_lds_));
  const EnvCfg &cfg = *static_cast<const EnvCfg *>(fs_uniform_ptr(cfgp)); // (device memory: EnvIO::cfg_dev)
  const int cfg_ik = Ctx::PLAIN ? 0 : cfg.ik, cfg_dense = Ctx::PLAIN ? 0 : cfg.dense; // (SpecCtx::PLAIN)
  CModel &m = c.m;
  float *L = c.L;
  int *E = c.I(c.ly.env);
  const bool init = io.init_state != nullptr;
  const int NS = init ? 0 : 100 + (cfg.has_recipe ? 100 : 0), total = env_reset_total(cfg, init);
  const int p0 = __builtin_amdgcn_readfirstlane(p0_), p1 = min(__builtin_amdgcn_readfirstlane(p1_), total);
#pragma unroll 1
  for (int u = p0; u < p1; u++) {
    if (u == 0) {
      // sim.reset()
      for (int i = c.lane; i < c.D.nq; i += 64) L[c.ly.qpos + i] = GP(m.qpos0)[i];
      for (int i = c.lane; i < c.D.nv; i += 64) { L[c.ly.qvel + i] = 0; L[c.ly.qaccws + i] = 0; L[c.ly.qfrcbias + i] = 0; L[c.ly.qfrcapp + i] = 0; }
      for (int i = c.lane; i < c.D.nu; i += 64) L[c.ly.ctrl + i] = 0;
      for (int i = c.lane; i < 6 * c.D.nparts; i += 64) L[c.ly.xfrc + i] = 0;
      // robot collision off, part colliders on (furniture.py:1441-1461)
      for (int g = c.lane; g < c.D.ncg; g += 64) {
        int ct = m.cg_contype0[g], ca = m.cg_conaffinity0[g];
        if (m.cg_isrobot[g]) { ct = 0; ca = 0; }
        if (m.cg_ispartcol[g]) { ct = 1; ca = 1; }
        c.I(c.ly.contype)[g] = ct; c.I(c.ly.conaff)[g] = ca;
      }
      for (int e = c.lane; e < c.D.neq; e += 64) { c.I(c.ly.eqactive)[e] = 0; for (int k = 0; k < 7; k++) L[c.ly.eqdata + 7 * e + k] = GP(m.eq_data0)[7 * e + k]; }
      int episodes = E[E_EPISODE_COUNT], sticky = E[E_OVERFLOW];
      SYNC();
      for (int i = c.lane; i < E_FIXED_WORDS; i += 64) E[i] = 0;
      for (int p = c.lane; p < c.D.nparts; p += 64) E[E_GROUP + p] = p;
      if (c.D.agent == 2) for (int i = c.lane; i < EC_WORDS; i += 64) E[E_GROUP + c.D.nparts + i] = 0;
      SYNC();
      if (c.lane == 0) {
        E[E_EPISODE_COUNT] = episodes + 1; E[E_SITE1] = -1; E[E_SITE2] = -1; E[E_OVERFLOW] = sticky;
        if (cfg.n_pre > 0 && cfg.pre_mode == 0) // no recipe: the listed welds are switched on, their groups merged (furniture.py:1493-1501)
          for (int i = 0; i < cfg.n_pre; i++) {
            const int e = GP(cfg.pre_tab)[3 * i];
            c.I(c.ly.eqactive)[e] = 1;
            int *grp = E + E_GROUP;
            const int r1 = env_find(grp, GP(m.eq_part1)[e]), r2 = env_find(grp, GP(m.eq_part2)[e]);
            grp[r1] = r2;
          }
      }
      SYNC();
      if (init) {
        // set_init_qpos (furniture.py:1505-1519, 1568-1569, 1617-1618): set_env_state(given state) replaces placement, settling and the
        // robot initialisation (no RNG draw is consumed); robot collision on; the reference's forward passes in between do not change
        // qpos / qvel, the common tail below starts with one
        if (c.lane == 0) for (int p = 0; p < c.D.nparts; p++) env_stop_part(c, p, 0.0f);
        SYNC();
        for (int i = c.lane; i < c.D.nq; i += 64) L[c.ly.qpos + i] = io.init_state[i];
        for (int i = c.lane; i < c.D.nv; i += 64) L[c.ly.qvel + i] = io.init_state[c.D.nq + i];
        for (int g = c.lane; g < c.D.ncg; g += 64)
          if (m.cg_isrobot[g]) { c.I(c.ly.contype)[g] = m.cg_contype0[g]; c.I(c.ly.conaff)[g] = m.cg_conaffinity0[g]; }
        SYNC();
      } else {
        // place parts (host ran the reference's sampler; tasks/placement_sampler.py:138-190)
        for (int i = c.lane; i < 7 * c.D.nparts; i += 64) {
          int p = i / 7, k = i % 7;
          if (io.tab_parts) L[c.ly.qpos + GP(m.part_qposadr)[p] + k] = io.tab_parts[i];
        }
        SYNC();
      }
    }
    if (u < NS) { // ---- settling (env_settle: 10 x [stop the parts, 10 x (step, _slow_objects)]), twice with a recipe
      if (u == 100 && cfg.pre_mode == 1) // pre-assembled recipe steps (furniture.py:1542-1557): _connect(site2, site1) with the recipe's angle, latches cleared
        for (int i = 0; i < cfg.n_pre; i++) {
          const int k1 = GP(cfg.pre_tab)[3 * i], k2 = GP(cfg.pre_tab)[3 * i + 1];
          const float ang = __int_as_float(GP(cfg.pre_tab)[3 * i + 2]);
          if (c.lane == 0) env_project_connector_quat(c, k1, k2, ang == ang, ang);
          SYNC();
          env_connect(c, cfg, k1, k2, true);
          // config.reset_robot_after_attach: this _connect, too, ends with _initialize_robot_pos() (furniture.py:919-925) -- its draw was taken
          // between the placement's and the robot initialisation's and sits behind the latter's 101 rows of the noise table
          if (cfg.reset_robot_after_attach && io.tab_noise && io.n_noise > 101 + i) env_init_robot(c, io, 101 + i, cfg.move_speed);
          if (c.lane == 0) { E[E_CONNECTED_THIS_STEP] = 0; E[E_CONNBODY1] = 0; }
          SYNC();
        }
      if (u % 10 == 0) {
        if (c.lane == 0) for (int p = 0; p < c.D.nparts; p++) env_stop_part(c, p, 0.0f);
        SYNC();
      }
      fs_step(c);
      // _slow_objects: gravity compensation + clip |qvel| <= 0.2
      for (int p = c.lane; p < c.D.nparts; p += 64) {
        float *x = c.L + c.ly.xfrc + 6 * p;
        x[0] = 0; x[1] = 0; x[2] = -c.D.gravity[2] * GP(m.part_mass)[p]; x[3] = 0; x[4] = 0; x[5] = 0;
        int d = GP(m.part_dofadr)[p];
        for (int k = 0; k < 6; k++) { c.L[c.ly.qvel + d + k] = fminf(fmaxf(c.L[c.ly.qvel + d + k], -0.2f), 0.2f); c.L[c.ly.qfrcapp + d + k] = 0; }
      }
      SYNC();
    } else if (!init && u <= NS + 100) { // ---- the robot: initial pose + joint noise, 1 + 100 times
      if (u == NS && c.D.narm > 0) env_gravity_comp(c);
      env_init_robot(c, io, u - NS, cfg.move_speed);
      fs_step(c);
      if (u == NS) {
        for (int g = c.lane; g < c.D.ncg; g += 64)
          if (m.cg_isrobot[g]) { c.I(c.ly.contype)[g] = m.cg_contype0[g]; c.I(c.ly.conaff)[g] = m.cg_conaffinity0[g]; }
        SYNC();
        if (c.D.narm > 0) env_gravity_comp(c);
      }
    } else { // ---- the common tail
      if (u == (init ? 0 : NS + 101)) {
        for (int i = c.lane; i < c.D.nu; i += 64) L[c.ly.ctrl + i] = 0;
        for (int i = c.lane; i < c.D.nv; i += 64) { L[c.ly.qfrcapp + i] = 0; L[c.ly.qaccws + i] = 0; }
        for (int i = c.lane; i < 6 * c.D.nparts; i += 64) L[c.ly.xfrc + i] = 0;
        SYNC();
        fs_forward(c);
        if (c.D.narm > 0) env_gravity_comp(c);
      }
      fs_step(c);
    }
  }
  if (p1 == total && p0 < total) {
    if (cfg_ik) env_ik_sync(c); // furniture.py:1643-1650
    // the finger / floor touch masks describe the contact list of a forward pass run with mode bit 1; none of the reset's passes is one,
    // so what is there belongs to the state BEFORE the reset (the terminal step's last pass) -- the reset state has the fingers open and
    // away from the parts.  Cleared, so that what follows (the dense reward's _reset_reward_variables, the scheduler features) does not
    // depend on where the previous episode ended -- which also makes the reset a function of the reset table alone (look-ahead reset).
    if (c.lane == 0) { int *scal = c.I(c.ly.scal); scal[SC_TOUCHL] = 0; scal[SC_TOUCHR] = 0; scal[SC_TOUCHF] = 0; }
    SYNC();
    if (c.lane == 0) {
      env_next_subtask(c);
      if (cfg_dense) { // FurnitureSawyerDenseRewardEnv._reset: _reset_reward_variables (furniture_sawyer_dense.py:218-220)
        DenseSimP<Ctx> dp{c, cfg};
        dense_reset(env_edense(c), cfg.dense_coef, cfg.dense_sub, dp, cfg.n_pre);
      }
    }
  }
  // (sticky overflow report: this call's substeps -- the LDS scalars do not outlive a launch, the record does)
  if (c.lane == 0) E[E_OVERFLOW] |= c.I(c.ly.scal)[SC_OVERFLOW];
  SYNC();
}
template <class Ctx> DEV void env_reset(const Ctx &c, const EnvCfg *cfgp, const EnvResetIO &io) { env_reset_units(c, cfgp, io, 0, 1 << 20); }

// Smallest clearance between a robot collision geom and a furniture part's collision geom, from the body poses of the last
// forward pass: a LOWER bound on the true distance (exact point-to-solid distance from one geom's centre to the other geom's box /
// cylinder, minus the first geom's bounding radius; the larger of the two directions).  A function of the state only -- it decides
// which envs the next launch steps with the multi-wave kernel (an env about to enter robot-part contact is the expensive kind),
// so results never depend on timing.  lanes = candidate pairs (robot x part pairs only).
template <class Ctx> DEV float env_robot_clearance(const Ctx &c) {
  CModel &m = c.m;
  const float *L = c.L;
  float best = 1e9f;
  for (int p0 = 0; p0 < c.D.ncp; p0 += 64) {
    const int p = min(p0 + c.lane, c.D.ncp - 1);
    const int w = GP(m.pair_bp)[2 * p], g1 = w & 255, g2 = (w >> 8) & 255;
    const bool rp = (m.cg_fingerrole[g1] && m.cg_ispartcol[g2]) || (m.cg_fingerrole[g2] && m.cg_ispartcol[g1]); // finger geoms x part geoms
    if (!rp || p0 + c.lane >= c.D.ncp) continue;
    const int b1 = m.cg_body[g1], b2 = m.cg_body[g2];
    const M3 B1 = ldm3(L + c.ly.xmat + 9 * b1), B2 = ldm3(L + c.ly.xmat + 9 * b2);
    const V3 c1 = ldv3(L + c.ly.xpos + 3 * b1) + mulv(B1, ldv3(GP(m.cg_pos) + 3 * g1)), c2 = ldv3(L + c.ly.xpos + 3 * b2) + mulv(B2, ldv3(GP(m.cg_pos) + 3 * g2));
    const M3 R1 = mulm(B1, ldm3(GP(m.cg_mat) + 9 * g1)), R2 = mulm(B2, ldm3(GP(m.cg_mat) + 9 * g2));
    const V3 d = c2 - c1;
    float lb = norm(d) - GP(m.cg_rbound)[g1] - GP(m.cg_rbound)[g2];
#pragma unroll
    for (int side = 0; side < 2; side++) { // side 0: centre of geom 1 against the solid of geom 2
      const int gs = side ? g1 : g2, go = side ? g2 : g1, ty = m.cg_type[gs];
      const M3 &R = side ? R1 : R2;
      const V3 dw = side ? d : -d; // centre(other) - centre(solid)
      const V3 cl = multv(R, dw), sz = ldv3(GP(m.cg_size) + 3 * gs);
      const V3 e = v3(fmaxf(fabsf(cl.x) - sz.x, 0.0f), fmaxf(fabsf(cl.y) - sz.y, 0.0f), fmaxf(fabsf(cl.z) - sz.z, 0.0f));
      const float er = fmaxf(sqrtf(cl.x * cl.x + cl.y * cl.y) - sz.x, 0.0f), ez = fmaxf(fabsf(cl.z) - sz.y, 0.0f);
      if (ty == GT_BOX) lb = fmaxf(lb, norm(e) - GP(m.cg_rbound)[go]);
      else if (ty == GT_CYLINDER) lb = fmaxf(lb, sqrtf(er * er + ez * ez) - GP(m.cg_rbound)[go]);
    }
    best = fminf(best, lb);
  }
  return -wave_max(-best);
}

// ---------------------------------------------------------------------------------------------------- after a reset / a step
// What a launch leaves behind for the NEXT one, and the observation: the Newton iterations the scheduler's multi-wave rule reads
// (0 after a reset), the robot-part clearance, the sticky overflow report.  One function for the step, the in-kernel reset, the reset
// launch and the look-ahead (shadow) reset, so that all of them leave the same words.
template <class Ctx> DEV void env_post(const Ctx &c, const EnvCfg &cfg, const EnvIO &io, int niter) {
  int *E = c.I(c.ly.env);
  const int *scal = c.I(c.ly.scal);
  const float clr = c.D.narm > 0 ? env_robot_clearance(c) : 1e9f;
  if (c.lane == 0) {
    E[E_NITER] = niter; c.L[c.ly.env + E_CLEARANCE] = clr;
    E[E_TOUCH_L] = scal[SC_TOUCHL]; E[E_TOUCH_R] = scal[SC_TOUCHR]; E[E_TOUCH_FLOOR] = c.D.nr > 1 ? scal[SC_ISL + KI(r_tree, 1)] : 0; // (development: candidate features of the scheduler's rule)
    if (io.info) io.info[FSIM_INFO_OVERFLOW] = (scal[SC_OVERFLOW] & 0xff) | (E[E_OVERFLOW] << 8);
  }
  env_write_obs(c, cfg, io);
}

// ---- look-ahead reset.  The state a reset leaves is a function of the env's reset table (and the handle's configuration) alone, and
// the host uploads that table one episode ahead.  While the episode is still being stepped, waves of the step kernel that have run
// out of envs run the NEXT reset a few dozen units per launch (env_shadow_job, fsim.hip) into a SHADOW record + observation row; the
// terminal step then copies the shadow in instead of running 301 / 401 substeps on the critical path of its launch.  Same loop
// (env_reset_units), same context type, same inputs: the record is bit-identical to what the in-kernel reset leaves
// (tests/test_lookahead_gpu.py).  A shadow counts when it is complete and was computed from the table that is on the device now
// (serial numbers); everything happens on the handle's one stream, in launch order.
DEV bool env_shadow_ready(const EnvCfg &cfg, const EnvIO &io) {
  if (!io.sh_prog) return false;
  const int p = io.sh_prog0;
  return io.sh_serial == io.tab_serial && io.tab_serial > 0 && p == env_reset_total(cfg, io.init_state != nullptr);
}
template <class Ctx> DEV void env_swap_in(const Ctx &c, const EnvCfg &cfg, const EnvIO &io) {
  float *L = c.L;
  int *E = c.I(c.ly.env);
  const int *scal = c.I(c.ly.scal);
  const int episodes = E[E_EPISODE_COUNT], sticky = E[E_OVERFLOW];
  SYNC();
  for (int i = c.lane; i < c.ly.stride; i += 64) L[i] = io.sh_state[i];
  SYNC();
  if (c.D.agent == 2) env_cursor_restore_poses(c); // (the pose arrays follow the record: the launch's closing env_cursor_keep_poses then writes the reset's poses back, not this step's)
  if (c.lane == 0) {
    // (the two words of the record that belong to the env, not to the reset: how many episodes it has seen, whether it ever dropped contacts)
    E[E_EPISODE_COUNT] = episodes + 1; E[E_OVERFLOW] |= sticky;
    *io.sh_prog = env_reset_total(cfg, io.init_state != nullptr) + 1; // taken (a new table -- another serial -- starts the next one)
    if (io.stats) __hip_atomic_fetch_add(io.stats, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (io.info) {
      io.info[FSIM_INFO_SUBTASK1] = E[E_SUBTASK1]; io.info[FSIM_INFO_SUBTASK2] = E[E_SUBTASK2];
      io.info[FSIM_INFO_OVERFLOW] = (scal[SC_OVERFLOW] & 0xff) | (E[E_OVERFLOW] << 8);
    }
  }
  if (io.obs) {
    if (cfg.obs_bf16) {
      const unsigned short *src = static_cast<const unsigned short *>(io.sh_obs);
      unsigned short *dst = reinterpret_cast<unsigned short *>(io.obs);
      for (int i = c.lane; i < cfg.obs_dim; i += 64) dst[i] = src[i];
    } else {
      const float *src = static_cast<const float *>(io.sh_obs);
      for (int i = c.lane; i < cfg.obs_dim; i += 64) io.obs[i] = src[i];
    }
  }
  SYNC();
}
// The reset of ONE env inside a launch: the shadow record if one is ready, else the reset itself.  (reset launches, deferred resets
// of the multi-wave workgroups and -- through env_step -- the auto-reset of a terminal step)
template <class Ctx> DEV void env_reset_or_swap(const Ctx &c, const EnvCfg &cfg, const EnvIO &io) {
  if (env_shadow_ready(cfg, io)) { env_swap_in(c, cfg, io); return; }
  env_reset(c, io.cfg_dev, env_reset_io(io));
  if (c.lane == 0 && io.stats) __hip_atomic_fetch_add(io.stats + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  env_post(c, cfg, io, 0);
}

// ---------------------------------------------------------------------------------------------------- step
// DEFER (the multi-wave workgroups of k_env_step_x): a terminal env without a ready shadow record is NOT reset here -- env_step returns 1
// and the workgroup's wave 0 runs the reset as a one-wave job right afterwards (fsim.hip), so that a reset's bits never depend on
// whether the env happened to be stepped by four waves (a reset is one-wave arithmetic wherever a STEP launch runs it).  Returns 0 otherwise.
// The one exception is the overflow re-step (fsim.hip redo_overflowed: the generic four-wave kernel, non-DEFER, on the larger slot layout): an env
// whose terminal step or reset dropped contacts is repeated there WITH its reset inline -- four-wave arithmetic with more contact slots, i.e. a
// different (more complete) computation than the first pass by construction, not a bit-copy of it.
template <class Ctx, bool DEFER = false> DEV int env_step(const Ctx &c, const EnvCfg &cfg, const EnvIO &io) {
  const int cfg_ik = Ctx::PLAIN ? 0 : cfg.ik, cfg_controller = Ctx::PLAIN ? 0 : cfg.controller, cfg_dense = Ctx::PLAIN ? 0 : cfg.dense; // (SpecCtx::PLAIN)
  CModel &m = c.m;
  float *L = c.L;
  int *E = c.I(c.ly.env);
  int *scal = c.I(c.ly.scal);
  int dof = cfg.dof_action;
  // _before_step + action plumbing
  if (c.lane == 0) { E[E_CONNECTED_THIS_STEP] = 0; if (Ctx::NW > 1) E[E_MW_STEPS] += 1; }
  float connect = io.action[dof - 1];
  if (cfg_ik) {
    // _do_ik_step (furniture.py:2911-2958, 2999-3018): per arm d_pos = move_speed * [-a1, a0, a2]; rotation entries stay raw (x rotate_speed
    // in env_ik); the grips follow the arm commands, the last entry is connect.  Block layout per arm: EI_ACT + [dpos 3, rot 3|4, grip]
    const int nrot = cfg_ik == 1 ? 3 : 4, stride = 3 + nrot;
    for (int t = c.lane; t < c.D.narm * (stride + 1); t += 64) {
      const int arm = t / (stride + 1), k = t % (stride + 1);
      float *K = L + c.ly.eik + EI_WORDS * arm;
      const float *aa = io.action + arm * stride;
      float v;
      if (k == 0) v = -aa[1] * cfg.move_speed;
      else if (k == 1) v = aa[0] * cfg.move_speed;
      else if (k == 2) v = aa[2] * cfg.move_speed;
      else if (k < stride) v = aa[k];
      else { v = io.action[c.D.narm * stride + arm]; if (cfg.discrete_grip && cfg.agent == 0) v = v < 0 ? -1.0f : 1.0f; } // furniture_sawyer.py:72-74
      K[EI_ACT + k] = v;
    }
  } else if (cfg_controller) {
    // FurnitureSawyerEnv._step discretises the grip (furniture_sawyer.py:72-74); _do_controller_step scales the first three
    // entries by move_speed and permutes them [-a1, a0, a2] whatever the controller kind (furniture.py:3069-3071)
    float *K = L + c.ly.env + E_GROUP + c.D.nparts;
    const int cd = dof - 2;
    if (c.lane < 8) {
      float v = 0;
      if (c.lane == 0) v = -io.action[1] * cfg.move_speed;
      else if (c.lane == 1) v = io.action[0] * cfg.move_speed;
      else if (c.lane == 2) v = io.action[2] * cfg.move_speed;
      else if (c.lane < cd) v = io.action[c.lane];
      else if (c.lane == 7) { v = io.action[cd]; if (cfg.discrete_grip) v = v < 0 ? -1.0f : 1.0f; }
      K[EK_ACT + c.lane] = v;
    }
    if (c.lane == 0) reinterpret_cast<int *>(K)[EK_KIND] = cfg_controller;
  } else
  // _setup_action (impedance): clip, gripper 1 -> 2, rescale to ctrlrange, stale gravity compensation
  for (int u = c.lane; u < c.D.nu; u += 64) {
    float a;
    if (u < c.D.narmj) a = io.action[u];
    else {
      int gi = (u - c.D.narmj) >> 1;
      a = io.action[c.D.narmj + gi];
      if (cfg.discrete_grip && cfg.agent == 0) a = a < 0 ? -1.0f : 1.0f; // furniture_sawyer.py:72-74
      if (cfg.rescale_actions) a = fminf(fmaxf(a, -1.0f), 1.0f);
      if ((u - c.D.narmj) & 1) a = -a; // format_action: [g, -g]
    }
    if (u < c.D.narmj && cfg.rescale_actions) a = fminf(fmaxf(a, -1.0f), 1.0f);
    L[c.ly.ctrl + u] = cfg.rescale_actions ? GP(m.ctrl_bias)[u] + GP(m.ctrl_weight)[u] * a : a;
  }
  SYNC();
  if (c.D.agent == 2) {
    // FurnitureCursorEnv._step: _step_discrete(a) then _do_simulation(None) (furniture_cursor.py:59-70, furniture.py:2857-2897)
    connect = 0; // the arm agents' finger scan below does not apply
    // _step_discrete reads data.site_xpos / site_xmat (_try_connect -> _is_aligned, the approach target) BEFORE this step's first forward
    // pass: the poses the previous launch kept in the record (env_cursor_keep_poses).  Until round 5 a step in which neither cursor moved a
    // selected part (a move rejected at the boundary) read these arrays as whatever the LDS held (scripts/dev/r5/lds_uninit.py found word
    // xpos + 6: results depended on the CU's previous tenant).
    env_cursor_restore_poses(c);
    env_cursor_discrete(c, cfg, io.action);
    if (cfg.reset_robot_after_attach && E[E_CONNECTED_THIS_STEP]) env_init_robot(c, EnvResetIO{nullptr, nullptr, nullptr, 0}, 0, cfg.move_speed); // (furniture.py:919-925: the cursors go back to their start positions)
    if (c.lane == 0) { // parts in a selected group float (gravity compensated), the others are only stopped
      int *grp = E + E_GROUP;
      const int *eci = c.I(env_ecur(c));
      for (int i = 0; i < c.D.nparts; i++) {
        int g = env_find(grp, i);
        bool sel = false;
        for (int q = 0; q < 2; q++) if (eci[EC_SEL + q] && env_find(grp, eci[EC_SEL + q] - 1) == g) sel = true;
        env_stop_part(c, i, sel ? 1.0f : 0.0f);
      }
    }
    SYNC();
    fs_substeps(c, cfg.n_substeps, 0);
    env_stop_selected(c, 1.0f);
  } else if (cfg_ik) {
    SYNC();
    env_ik(c, cfg.rotate_speed, cfg_ik);
    const float *K0 = L + c.ly.eik;
    const float pgain = GP(m.ik_tab)[IKT_ARM * c.D.narm + IKT_GAIN];
    const int ng = 3 + (cfg_ik == 1 ? 3 : 4); // offset of the grip entry inside EI_ACT
    for (int rep = 0; rep < 3; rep++) { // action_repeat = 3 (furniture.py:172): closed loop on the commanded joint positions
      // get_control's P controller (sawyer_ik_controller.py:75-84, baxter_ik_controller.py:86-95), then _setup_action on [velocities, grips]
      for (int u = c.lane; u < c.D.nu; u += 64) {
        float a;
        if (u < c.D.narmj) {
          const float *K = K0 + EI_WORDS * (u / 7);
          a = fminf(fmaxf(-pgain * (L[c.ly.qpos + GP(m.arm_qposadr)[u]] - K[EI_QCMD + u % 7]), -1.0f), 1.0f);
        } else {
          const float *K = K0 + EI_WORDS * ((u - c.D.narmj) >> 1);
          a = K[EI_ACT + ng];
          if (cfg.rescale_actions) a = fminf(fmaxf(a, -1.0f), 1.0f);
          if ((u - c.D.narmj) & 1) a = -a;
        }
        L[c.ly.ctrl + u] = cfg.rescale_actions ? GP(m.ctrl_bias)[u] + GP(m.ctrl_weight)[u] * a : a;
      }
      SYNC();
      env_gravity_comp(c);
      fs_substeps(c, cfg.n_substeps, rep == 2 ? 2 : 0);
      if (scal[SC_BAD] & 2) break;
    }
  } else if (cfg_controller) {
    // _do_controller_step: sim.forward(), then n_substeps x (_pre_action, sim.step()); no _setup_action, so qfrc_applied keeps
    // the gravity compensation the reset left (furniture.py:1624-1632) on top of the qfrc_bias inside ctrl
    fs_substeps_t<true>(c, cfg.n_substeps, 2);
  } else {
    env_gravity_comp(c);
#if defined(FSIM_TIMELINE) && defined(FSIM_PROFILE)
    long long tl0_ = clock64();
    if (c.lane == 0) scal[48] = (int)((tl0_ - io.t0) >> 4);
#endif
    // _do_simulation: n_substeps x sim.step()
    fs_substeps(c, cfg.n_substeps, 2);
#if defined(FSIM_TIMELINE) && defined(FSIM_PROFILE)
    { long long tl1_ = clock64(); if (c.lane == 0) { scal[51] = (int)((tl1_ - tl0_) >> 4); scal[52] = (int)(tl1_ >> 4); } }
#endif
  }
  int bad = scal[SC_BAD] & 2;
  if (bad) {
    // unstable simulation: reset inside step(), flag the failure (furniture.py:2889-2897).  Under auto_reset the step is
    // terminal and the env is reset again right below (the vec-env worker's reset, subproc_vec_env.py:16-20): the first reset
    // would leave nothing behind but ONE consumed pass of the env's reset-time RNG stream, so it is not executed -- the info
    // block tells the host to drop one draw instead (FSIM_INFO_NEEDS_TABLE = 2).  (The dense reward is computed on the reset
    // state by the reference, so the dense env keeps the in-step reset and its stream runs one draw behind after a failure.)
    // Known one-episode deviation: the table already on the device is draw k+1, and it is what the reset below consumes, while
    // the reference's post-failure episode starts from draw k+2 (its in-step reset took k+1).  The host drops k+2 and uploads
    // k+3, so from the following episode on both streams agree again; only the placement of the episode right after an
    // unstable step differs (same distribution).  Keeping two tables per env on the device would remove it.
    const bool skip_reset = cfg.auto_reset && !cfg_dense;
    if (!skip_reset) {
      if (env_shadow_ready(cfg, io)) env_swap_in(c, cfg, io); // (the record only matters here: the forward pass below rebuilds the poses)
      else {
        env_reset(c, io.cfg_dev, env_reset_io(io));
        if (c.lane == 0 && io.stats) __hip_atomic_fetch_add(io.stats + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (c.lane == 0) { E[E_FAIL] = skip_reset ? 2 : 1; scal[SC_BAD] = 0; if (skip_reset) { scal[SC_TOUCHL] = 0; scal[SC_TOUCHR] = 0; scal[SC_TOUCHF] = 0; } }
    SYNC();
    if (!skip_reset) fs_substeps(c, 1, 3);
  } else if (connect > 0) {
    // finger-touch scan -> first part (in part order) pinched by both fingers of an arm -> _try_connect
    env_finger_scan(c.D.narm, scal, [&](int part) { return env_try_connect(c, cfg, part, -1); });
    SYNC();
    // config.reset_robot_after_attach (furniture.py:919-925): the last thing _connect does is _initialize_robot_pos() -- the arm's
    // joints jump to the initial pose plus ONE fresh draw of joint noise, the gripper opens, velocities stay; the IK controller is
    // re-synchronised.  The draw comes from the env's one RandomState, between the draws of two resets: the host keeps it waiting in
    // the attach table (fsim_set_attach_noise) and learns from FSIM_INFO_CONNECTED_THIS_STEP that it was taken.
    if (cfg.reset_robot_after_attach && E[E_CONNECTED_THIS_STEP]) {
      env_init_robot(c, EnvResetIO{nullptr, io.tab_attach, nullptr, 1}, 0, cfg.move_speed);
      if (cfg_ik) env_ik_sync(c, true);
    }
  }
  // post-connect re-pose of body1's (merged) group (furniture.py:426-436)
  if (E[E_CONNBODY1] > 0) {
    fs_forward(c);
    if (c.lane == 0) {
      int pA = E[E_CONNBODY1] - 1;
      V3 tp = ldv3(L + c.ly.env + E_CB1_POS);
      Q4 tq = ldq(L + c.ly.env + E_CB1_QUAT);
      env_move_group(c, pA, tp - ldv3(L + c.ly.qpos + GP(m.part_qposadr)[pA]), tq, cfg.gravity_comp ? 1.0f : 0.0f);
      E[E_CONNBODY1] = 0;
    }
    SYNC();
    fs_substeps(c, 1, 2);
  }
  SYNC();
  // reward (furniture.py:482-541): one-shot touch / pick latches, success delta, control penalty on the RAW action
  float touch_rew = 0, pick_rew = 0, succ_rew = 0, ctrl_pen = 0;
  {
    float s2 = 0;
    for (int k = c.lane; k < dof; k += 64) s2 += io.action[k] * io.action[k];
    ctrl_pen = c.D.agent == 2 ? 0.0f : -cfg.ctrl_penalty_coef * wave_sum(s2); // (wave_sum is evaluated by all lanes either way)
  }
  int success = 0, terminal = 0;
  // scheduler hint: is a robot hand within 10 cm of a furniture part's collision geom?  (an env about to enter robot-part
  // contact is the expensive kind next step even if this step was cheap)
  int near = 0;
  if (io.cost) {
    for (int g = c.lane; g < c.D.ncg; g += 64) {
      if (!m.cg_ispartcol[g]) continue;
      int b = m.cg_body[g];
      V3 ctr = ldv3(L + c.ly.xpos + 3 * b) + mulv(ldm3(L + c.ly.xmat + 9 * b), ldv3(GP(m.cg_pos) + 3 * g));
      for (int arm = 0; arm < c.D.narm; arm++) {
        V3 hp = ldv3(L + c.ly.xpos + 3 * GP(m.body_red)[GP(m.hand_body)[arm]]);
        if (norm(ctr - hp) - GP(m.cg_rbound)[g] < 0.10f) near = 1;
      }
    }
    near = wave_or(near);
  }
  float penalty = 0, dense_rew = 0;
  if (c.lane == 0) {
    for (int arm = 0; arm < c.D.narm; arm++) {
      int both = (scal[SC_TOUCHL] >> (16 * arm)) & (scal[SC_TOUCHR] >> (16 * arm)) & 0xffff;
      for (int p = 0; p < c.D.nparts; p++) {
        if (!((both >> p) & 1)) continue;
        if (!((E[E_TOUCHED] >> p) & 1)) { E[E_TOUCHED] |= 1 << p; touch_rew += cfg.touch_reward; }
        if (!((scal[SC_TOUCHF] >> p) & 1) && !((E[E_PICKED] >> p) & 1)) { E[E_PICKED] |= 1 << p; pick_rew += cfg.pick_reward; }
      }
    }
    succ_rew = cfg.success_reward * (float)(E[E_NUM_CONNECTED] - E[E_PREV_NUM_CONNECTED]);
    E[E_PREV_NUM_CONNECTED] = E[E_NUM_CONNECTED];
    if (E[E_NUM_CONNECTED] == cfg.success_num_conn && c.D.nparts > 1) { E[E_SUCCESS] = 1; success = 1; }
    terminal = success;
    int dense_phase = 0;
    if (cfg_dense) {
      // FurnitureSawyerEnv._step (furniture_sawyer.py:76-79): the dense _compute_reward replaces the reward and owns _success;
      // done = (all parts connected) or its own done
      DenseSimP<Ctx> dp{c, cfg};
      DenseOut d = dense_compute(env_edense(c), cfg.dense_coef, cfg.dense_sub, min(cfg.dense_nsub, cfg.success_num_conn), dp, io.action, dof, E[E_CONNECTED_THIS_STEP] != 0);
      success = d.success; E[E_SUCCESS] = success;
      terminal = terminal || d.done;
      dense_phase = d.phase_info;
      touch_rew = 0; pick_rew = 0; ctrl_pen = 0; succ_rew = d.phase_bonus;
      dense_rew = d.reward;
    }
    // _after_step
    E[E_EPISODE_LENGTH] += 1;
    int fail = E[E_FAIL];
    if (E[E_EPISODE_LENGTH] == cfg.max_episode_steps || fail) {
      terminal = 1;
      if (fail) { E[E_FAIL] = 0; penalty = -cfg.unstable_penalty_coef; }
    }
    float rew = cfg_dense ? dense_rew + penalty : succ_rew + touch_rew + pick_rew + ctrl_pen + penalty;
    L[c.ly.env + E_EPISODE_REWARD] += rew;
    if (io.reward) *io.reward = rew;
    if (io.done) *io.done = (uint8_t)terminal;
    if (io.info) {
      io.info[FSIM_INFO_NUM_CONNECTED] = E[E_NUM_CONNECTED]; io.info[FSIM_INFO_SUCCESS] = success; io.info[FSIM_INFO_FAIL] = fail ? 1 : 0;
      io.info[FSIM_INFO_LAST_SITE1] = E[E_SITE1]; io.info[FSIM_INFO_LAST_SITE2] = E[E_SITE2];
      io.info[FSIM_INFO_EPISODE_LENGTH] = E[E_EPISODE_LENGTH]; io.info[FSIM_INFO_CONNECTED_THIS_STEP] = E[E_CONNECTED_THIS_STEP];
      io.info[FSIM_INFO_NEEDS_TABLE] = (terminal && cfg.auto_reset) ? (fail == 2 ? 2 : 1) : 0; // 2: drop one draw first (see the unstable branch)
      if (terminal && cfg.auto_reset && io.nreset) __hip_atomic_fetch_add(io.nreset, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      io.info[FSIM_INFO_SUCCESS_REWARD_F] = __float_as_int(succ_rew); io.info[FSIM_INFO_TOUCH_REWARD_F] = __float_as_int(touch_rew);
      io.info[FSIM_INFO_PICK_REWARD_F] = __float_as_int(pick_rew); io.info[FSIM_INFO_CTRL_PENALTY_F] = __float_as_int(ctrl_pen);
      io.info[FSIM_INFO_DENSE_PHASE] = dense_phase;
      io.info[FSIM_INFO_EPISODE_REWARD_F] = __float_as_int(L[c.ly.env + E_EPISODE_REWARD]);
    }
    scal[14] = terminal;
    if (io.cost) {
      // longest-job-first key for the next launch: this step's cost predicts the next one (contact state persists);
      // an env that will hit max_episode_steps next step pays an in-kernel reset on top and goes to the front.
      long long dt = clock64() - io.t0;
      int key = (int)min(dt >> 10, (long long)(1 << 24));
      bool timeout_next = !terminal && cfg.auto_reset && E[E_EPISODE_LENGTH] + 1 >= cfg.max_episode_steps;
      *io.cost = timeout_next ? -1 : (key | (near << 30));
    }
  }
  SYNC();
  terminal = scal[14];
  const int nit_step = scal[SC_NITSUM];
  if (c.lane == 0) E[E_OVERFLOW] |= scal[SC_OVERFLOW]; // sticky: a step that dropped contacts is never lost between two host reads
  SYNC();
  if (terminal && cfg.auto_reset) { // SubprocVecEnv worker semantics (subproc_vec_env.py:15-48)
    if (DEFER && !env_shadow_ready(cfg, io)) return 1;
    env_reset_or_swap(c, cfg, io);
  } else {
    if (cfg_ik) env_ik_remember(c, cfg_ik); // (a reset stores its own poses: env_ik_sync)
    // what the multi-wave rule reads: Newton iterations per 50 substeps (an IK step runs 3 x n_substeps of them: unscaled, every IK env
    // would pass the threshold of 150 = three iterations per substep without touching anything)
    const int nsub = max(1, cfg.n_substeps * (cfg_ik ? 3 : 1));
    env_post(c, cfg, io, nsub == 50 ? nit_step : (int)((float)nit_step * 50.0f / (float)nsub));
  }
#if defined(FSIM_TIMELINE) && defined(FSIM_PROFILE)
  if (c.lane == 0) scal[52] = (int)(clock64() >> 4) - scal[52];
#endif
  return 0;
}
