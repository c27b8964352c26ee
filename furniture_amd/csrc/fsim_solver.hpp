// fsim_solver.hpp -- P4 constraint assembly + P7 primal Newton solver for one env / one wave.
//
// The reference never sets <option solver> (furniture/env/models/assets/base.xml:4), so what it runs
// is MuJoCo's default: Newton on the primal soft-constraint problem
//     min_x 1/2 (x - a_s)' M (x - a_s) + sum_i s_i(J_i x - aref_i)
// with quadratic costs for welds, one-sided quadratic for joint limits / frictionless contacts and
// the three-zone elliptic-cone cost with the impratio-regularised friction.  The convex problem has
// a unique optimum, so iterating to tolerance reproduces the reference's accelerations.
//
// GPU mapping: lanes = constraint slots (welds | joint limits | contact slots).  Jacobians are never
// stored: J*v is evaluated through per-body spatial vectors W_b = sum_{d in chain(b)} cdof_d v_d
// (one LDS 6-vector per body), J'f through per-body wrenches G_b accumulated with LDS float atomics,
// and J'WJ by a per-slot double loop over the dof chains of the two bodies.  The Hessian is a packed
// lower triangle in LDS, factored by a left-looking Cholesky with lane = row, wave shuffles for the
// pivots and a skyline start per row (block-diagonal when no constraint couples two kinematic trees).
#pragma once
#include "fsim_collide.hpp"

enum { LM_ACTIVE = 0, LM_AREF = 1, LM_D = 2, LM_JAR = 3, LM_JP = 4, LM_SIGN = 5, LM_DOF = 6 };
enum { WD_ACTIVE = 0, WD_P0 = 1, WD_C = 4, WD_AREF = 13, WD_D = 19, WD_JAR = 25, WD_JP = 31, WD_B1 = 37, WD_B2 = 38, WD_X2 = 39 };

// x^p for x in (0, 1): the MuJoCo default solimp power is 2; anything else goes through the hardware exp2/log2
DEV float fs_pow01(float x, float p) { return p == 2.0f ? x * x : __builtin_exp2f(p * __builtin_log2f(x)); }

template <class P1, class P2> DEV float fs_impedance(P1 solref, P2 solimp, float x0, float timestep, float *k, float *b) {
  float dmin = fminf(fmaxf(solimp[0], 0.0001f), 0.9999f), dmax = fminf(fmaxf(solimp[1], 0.0001f), 0.9999f);
  float width = fmaxf(solimp[2], 1e-15f), mid = fminf(fmaxf(solimp[3], 0.0001f), 0.9999f), power = fmaxf(solimp[4], 1.0f);
  float x = fabsf(x0) / width, imp;
  if (x >= 1) imp = dmax;
  else if (x <= 0) imp = dmin;
  else {
    float y;
    if (power == 1.0f) y = x;
    else if (x <= mid) y = fs_pow01(x, power) / (power == 2.0f ? mid : fs_pow01(mid, power - 1));
    else y = 1 - fs_pow01(1 - x, power) / (power == 2.0f ? 1 - mid : fs_pow01(1 - mid, power - 1));
    imp = dmin + y * (dmax - dmin);
  }
  if (solref[0] > 0) {
    float tc = fmaxf(solref[0], 2 * timestep), dr = solref[1];
    *k = 1.0f / (dmax * dmax * tc * tc * dr * dr);
    *b = 2.0f / (dmax * tc);
  } else { *k = -solref[0] / (dmax * dmax); *b = -solref[1] / dmax; }
  return imp;
}

// bt = body | tree << 8 (contact slots store it that way; fs_bt builds it for other callers)
template <class Ctx> DEV int fs_bt(const Ctx &c, int b) { return b | (KI(r_tree, b) << 8); }
template <class Ctx> DEV V3 fs_ptvel(const Ctx &c, int off, int bt, V3 p) {
  const int b = bt & 255;
  if (b == 0) return v3(0, 0, 0);
  S6 w = lds6(c.L + off + 6 * b);
  return w.l + cross(w.a, p - ldv3(c.L + c.ly.com + 3 * (bt >> 8)));
}

// returns 1 if any constraint couples two kinematic trees
template <class Ctx> DEV int fs_make_constraints(const Ctx &c) {
  CModel &m = c.m;
  float *L = c.L;
  int nslot = c.I(c.ly.scal)[SC_NSLOT];
  int coupled = 0, ncon = 0;
  int *adj = c.I(c.ly.scal) + SC_ADJ; // tree adjacency bitmasks (islands for the block Cholesky)
  if (c.lane < c.D.ntree) adj[c.lane] = 1 << c.lane;
  SYNC();
  for (int s = c.lane; s < nslot; s += 64) {
    float *r = L + c.ly.con + FSIM_CONW * s;
    int *ri = reinterpret_cast<int *>(r);
    if (ri[C_ACTIVE] != 1) continue;
    ncon++;
    const int bt1 = ri[C_B1], bt2 = ri[C_B2], b1 = bt1 & 255, b2 = bt2 & 255, g1 = ri[C_G1], g2 = ri[C_G2];
    float dist = r[C_DIST], incm = r[C_INCM];
    if (dist >= incm) { ri[C_ACTIVE] = 2; continue; }
    V3 pos = ldv3(r + C_POS);
    V3 vrel = fs_ptvel(c, c.ly.cvel, bt2, pos) - fs_ptvel(c, c.ly.cvel, bt1, pos);
    float mix = GP(m.cg_solmix)[g1] / (GP(m.cg_solmix)[g1] + GP(m.cg_solmix)[g2]);
    float sr[2], si[5];
    for (int i = 0; i < 2; i++) sr[i] = mix * GP(m.cg_solref)[2 * g1 + i] + (1 - mix) * GP(m.cg_solref)[2 * g2 + i];
    for (int i = 0; i < 5; i++) si[i] = mix * GP(m.cg_solimp)[5 * g1 + i] + (1 - mix) * GP(m.cg_solimp)[5 * g2 + i];
    float k, b;
    float imp = fs_impedance(sr, si, dist - incm, c.D.timestep, &k, &b);
    float R = fmaxf((1 - imp) / imp * (GP(m.cg_invweight)[g1] + GP(m.cg_invweight)[g2]), 1e-15f);
    r[C_DN] = 1.0f / R;
    r[C_DT] = fmaxf(c.D.impratio, 1e-15f) / R;
    V3 fx, fy, fz;
    fs_frame(r, fx, fy, fz);
    r[C_AREF] = -b * dot(fx, vrel) - k * imp * (dist - incm);
    r[C_AREF + 1] = -b * dot(fy, vrel); // (overwrites C_DIST / C_INCM, already in registers)
    r[C_AREF + 2] = -b * dot(fz, vrel);
    if (b1 != 0 && b2 != 0) {
      int t1 = bt1 >> 8, t2 = bt2 >> 8;
      if (t1 != t2) { coupled = 1; atomicOr(&adj[t1], 1 << t2); atomicOr(&adj[t2], 1 << t1); }
    }
  }
  for (int s = c.lane; s < 2 * c.D.nlim; s += 64) {
    float *r = L + c.ly.lim + FSIM_LIMW * s;
    int *ri = reinterpret_cast<int *>(r);
    int li = s >> 1, side = s & 1, d = GP(m.lim_dof)[li];
    float q = L[c.ly.qpos + GP(m.dof_qposadr)[d]];
    float dist = side ? GP(m.lim_range)[2 * li + 1] - q : q - GP(m.lim_range)[2 * li];
    float mg = GP(m.lim_margin)[li];
    int act = dist < mg;
    ri[LM_ACTIVE] = act;
    if (!act) continue;
    float sign = side ? -1.0f : 1.0f, k, b;
    float imp = fs_impedance(GP(m.lim_solref) + 2 * li, GP(m.lim_solimp) + 5 * li, dist - mg, c.D.timestep, &k, &b);
    float R = fmaxf((1 - imp) / imp * GP(m.dof_invweight0)[d], 1e-15f);
    r[LM_D] = 1.0f / R;
    r[LM_AREF] = -b * sign * L[c.ly.qvel + d] - k * imp * (dist - mg);
    r[LM_SIGN] = sign;
    ri[LM_DOF] = d;
  }
  for (int e = c.lane; e < c.D.neq; e += 64) {
    float *r = L + c.ly.weld + FSIM_WELDW * e;
    int *ri = reinterpret_cast<int *>(r);
    int act = c.I(c.ly.eqactive)[e];
    ri[WD_ACTIVE] = act;
    if (!act) continue;
    coupled = 1;
    int b1 = GP(m.eq_rbody1)[e], b2 = GP(m.eq_rbody2)[e];
    if (b1 != 0 && b2 != 0) {
      int t1 = KI(r_tree, b1), t2 = KI(r_tree, b2);
      if (t1 != t2) { atomicOr(&adj[t1], 1 << t2); atomicOr(&adj[t2], 1 << t1); }
    }
    const float *data = L + c.ly.eqdata + 7 * e;
    M3 R1 = ldm3(L + c.ly.xmat + 9 * b1);
    V3 p0 = ldv3(L + c.ly.xpos + 3 * b1) + mulv(R1, ldv3(data));
    V3 x2 = ldv3(L + c.ly.xpos + 3 * b2);
    Q4 qd = qmul(ldq(L + c.ly.xquat + 4 * b1), ldq(data + 3));
    Q4 q2c = qconj(ldq(L + c.ly.xquat + 4 * b2));
    Q4 qe = qmul(q2c, qd);
    float cpos[6];
    V3 dp = p0 - x2;
    cpos[0] = dp.x; cpos[1] = dp.y; cpos[2] = dp.z; cpos[3] = qe.x; cpos[4] = qe.y; cpos[5] = qe.z;
    // C[c][a] = 0.5 * imag_c( conj(q2) * e_a * q1*rel )
    for (int a = 0; a < 3; a++) {
      Q4 w = q4(0, a == 0, a == 1, a == 2);
      Q4 t = qmul(qmul(q2c, w), qd);
      r[WD_C + 0 * 3 + a] = 0.5f * t.x; r[WD_C + 1 * 3 + a] = 0.5f * t.y; r[WD_C + 2 * 3 + a] = 0.5f * t.z;
    }
    V3 jt = fs_ptvel(c, c.ly.cvel, fs_bt(c, b1), p0) - fs_ptvel(c, c.ly.cvel, fs_bt(c, b2), x2);
    V3 dw = lds6(L + c.ly.cvel + 6 * b1).a - lds6(L + c.ly.cvel + 6 * b2).a;
    float jv[6] = {jt.x, jt.y, jt.z, 0, 0, 0};
    for (int q = 0; q < 3; q++) jv[3 + q] = r[WD_C + 3 * q] * dw.x + r[WD_C + 3 * q + 1] * dw.y + r[WD_C + 3 * q + 2] * dw.z;
    for (int q = 0; q < 6; q++) {
      float k, b;
      float imp = fs_impedance(GP(m.eq_solref) + 2 * e, GP(m.eq_solimp) + 5 * e, cpos[q], c.D.timestep, &k, &b);
      float R = fmaxf((1 - imp) / imp * GP(m.eq_invweight)[2 * e + (q >= 3)], 1e-15f);
      r[WD_D + q] = 1.0f / R;
      r[WD_AREF + q] = -b * jv[q] - k * imp * cpos[q];
    }
    stv3(r + WD_P0, p0); stv3(r + WD_X2, x2);
    ri[WD_B1] = b1; ri[WD_B2] = b2;
  }
  int any = wave_or(coupled);
  int tot = (int)wave_sum((float)ncon);
  if (c.lane == 0) c.I(c.ly.scal)[SC_NCON] = tot;
  SYNC();
  int *scal_ = c.I(c.ly.scal);
  // the island structure rarely changes between substeps: keep last substep's closure + map when the adjacency is the same
  bool changed = c.lane < c.D.ntree && adj[c.lane] != scal_[SC_PADJ + c.lane];
  if (!__ballot(changed)) return any;
  if (c.lane < c.D.ntree) scal_[SC_PADJ + c.lane] = adj[c.lane];
  if (c.lane < c.D.ntree) { // transitive closure of the (<= 16 node) tree graph
    int reach = adj[c.lane], prev;
    do {
      prev = reach;
      for (int tt = prev; tt; tt &= tt - 1) reach |= adj[__ffs(tt) - 1];
    } while (reach != prev);
    scal_[SC_ISL + c.lane] = reach;
  }
  SYNC();
  fs_build_map(c, c.ly.hmap, scal_ + SC_ISL, SC_HWORDS);
  return any;
}

// W_b = sum_{d in chain(b)} cdof_d * vec_d
template <class Ctx> DEV void fs_body_spatial(const Ctx &c, int off_vec) {
  CModel &m = c.m;
  float *L = c.L;
  for (int b = c.lane; b < c.D.nr; b += 64) {
    S6 w = s6zero();
    {
      const int ch = KI(r_chain, b), base = (unsigned)ch >> 25;
      for (int mm = ch & 0x1ffffff; mm;) {
        FS_BITS3(mm, e0, e1, e2, h1, h2);
        const int d0 = base + e0, d1 = base + e1, d2 = base + e2;
        const S6 s0 = lds6(L + c.ly.cdof + 6 * d0), s1 = lds6(L + c.ly.cdof + 6 * d1), s2 = lds6(L + c.ly.cdof + 6 * d2);
        const float v0 = L[off_vec + d0], l1 = L[off_vec + d1], l2 = L[off_vec + d2], v1 = h1 ? l1 : 0.0f, v2 = h2 ? l2 : 0.0f;
        w = w + s0 * v0 + s1 * v1 + s2 * v2;
      }
    }
    sts6(L + c.ly.W + 6 * b, w);
  }
  SYNC();
}

// ---- the Newton solve keeps this lane's constraint records in REGISTERS: lane = contact slot (ncon_max <= 64) and lane =
// joint-limit record (2 nlim <= 64, checked by fsim_create).  The records are read from LDS once per solve (one round trip) and
// every pass of the iteration -- J v, the cone forces, the line-search evaluations, the update -- runs on registers: those passes
// used to re-read 10-20 words per slot behind `if (active)` branches, i.e. several dependent LDS round trips each, which is
// what a single wavefront per env pays for most (a wave issues ~1 instruction per 5 cycles and waits ~100 for every dependent
// LDS access).  Welds (rare: only after a connect) stay in LDS behind one wave-uniform flag.
struct SolSlot {
  bool act, dim1;          // contact slot is an active constraint / is frictionless (condim 1)
  int bt1, bt2;            // body | tree << 8 of the two geoms
  V3 r1, r2, fx, fy, fz;   // contact point relative to the CoM of body 1's / body 2's tree, and the contact frame (x = normal)
  int tb;                  // (wave-uniform) bitmask of the moving bodies that carry an active contact or weld
  float mu, dn, dt;
  float aref[3], jar[3], jp[3];
  bool lact;               // joint-limit record is active
  int ldof;
  float ld, laref, lsign, ljar, ljp;
  bool anyweld;            // (wave-uniform) some weld is active
  int pid, npc, ptot, nye; // body-pair cache (fs_pair_cache): this slot's pair block (-1 none), pairs (wave-uniform; -1 = not cached), items, column items
};
// (what the gradient pass knows about a lane's contact slot and later passes need again: fs_grad_slot)
struct SlotK { bool on; int zone; float K[6]; }; // zone: see fs_line_eval
// base: first contact slot of the set (0; 64 for the second set of a model with more than 64 contact slots -- Ctx::NS == 2: the
// solve then carries two SolSlot per lane; joint limits and welds belong to the first set)
template <class Ctx> DEV SolSlot fs_load_slots(const Ctx &c, const int base = 0) {
  float *L = c.L;
  SolSlot S;
  const int nslot = c.I(c.ly.scal)[SC_NSLOT];
  const int sc = min(base + c.lane, c.ly.ncon_max - 1), sl = min(c.lane, max(2 * c.D.nlim - 1, 0));
  const float *r = L + c.ly.con + FSIM_CONW * sc;
  const int *ri = reinterpret_cast<const int *>(r);
  S.act = base + c.lane < nslot && ri[C_ACTIVE] == 1;
  S.dim1 = ri[C_DIM] == 1;
  S.bt1 = S.act ? ri[C_B1] : 0; S.bt2 = S.act ? ri[C_B2] : 0;
  {
    const V3 pos = ldv3(r + C_POS);
    S.r1 = pos - ldv3(L + c.ly.com + 3 * (S.bt1 >> 8)); S.r2 = pos - ldv3(L + c.ly.com + 3 * (S.bt2 >> 8));
  }
  S.mu = r[C_MU]; S.dn = r[C_DN]; S.dt = r[C_DT];
  for (int a = 0; a < 3; a++) { S.aref[a] = r[C_AREF + a]; S.jar[a] = 0; S.jp[a] = 0; }
  fs_frame(r, S.fx, S.fy, S.fz);
  const float *q = L + c.ly.lim + FSIM_LIMW * sl;
  const int *qi = reinterpret_cast<const int *>(q);
  S.lact = base == 0 && c.lane < 2 * c.D.nlim && qi[LM_ACTIVE] != 0;
  S.ldof = qi[LM_DOF]; S.ld = q[LM_D]; S.laref = q[LM_AREF]; S.lsign = q[LM_SIGN]; S.ljar = 0; S.ljp = 0;
  if (!S.lact) S.ldof = 0;
  int aw = 0, tb = S.act ? ((1 << (S.bt1 & 255)) | (1 << (S.bt2 & 255))) : 0;
  for (int e = c.lane; e < c.D.neq; e += 64)
    if (c.I(c.ly.eqactive)[e]) { aw = 1; tb |= (1 << GP(c.m.eq_rbody1)[e]) | (1 << GP(c.m.eq_rbody2)[e]); }
  S.anyweld = __ballot(aw != 0) != 0;
  S.tb = wave_or(tb) & ~1;
  S.pid = -1; S.npc = 0; S.ptot = 0; S.nye = 0;
  return S;
}

// ---- body-pair cache.  A contact between two MOVING bodies (lo, hi) adds the cross block -cdof_d1' X cdof_d2 on
// chain(lo) x chain(hi) to the Newton Hessian (fs_hessian).  Which pairs exist, which slot feeds which pair block and the
// (pair, dof, dof) triple of every entry depend on the contact list only, i.e. they are fixed for the substep, while fs_hessian
// runs once per Newton iteration (5-6 times per substep when a gripper holds a part): the leader election, the chain-length
// prefix sums and the item -> (pair, e1, e2) -> (d1, d2) decoding cost five dependent LDS round trips per 64 items and
// iteration.  Built once per solve: S.pid per slot; one word per entry (pair | d1 << 8 | d2 << 16 | column << 24) in
// Layout::pitem[0 .. ptot); one word per column of Y = X * cdof(chain(hi)) (pair | d2 << 8 | column << 16) in
// pitem[FSIM_PCAP - FSIM_YCAP ..).  The projection then runs in two stages: Y (36 FMAs per column, <= 36 columns), then
// 6 FMAs per entry instead of 42.  More than FSIM_NPAIR pairs, a chain longer than FSIM_XW / 6 or more entries than fit:
// S.npc = -1 and fs_hessian takes its multi-pass path.
template <class Ctx> DEV void fs_pair_cache(const Ctx &c, SolSlot &S) {
  const int b1 = S.bt1 & 255, b2 = S.bt2 & 255, blo = min(b1, b2), bhi = max(b1, b2);
  const bool haskey = S.act && blo != 0 && bhi != blo;
  unsigned long long pending = __ballot(haskey);
  if (!pending) return; // (uniform) no moving-moving contact: nothing to do
  const int key = blo * 256 + bhi;
  int *pitem = c.I(c.ly.pitem);
  int np = 0, total = 0, ycols = 0;
  int plo[FSIM_NPAIR], phi_[FSIM_NPAIR], pbase[FSIM_NPAIR], ybase[FSIM_NPAIR]; // wave-uniform
  for (int q = 0; q < FSIM_NPAIR; q++) { plo[q] = 0; phi_[q] = 0; pbase[q] = 0x7fffffff; ybase[q] = 0x7fffffff; }
  bool ok = true;
#pragma unroll
  for (int q = 0; q < FSIM_NPAIR + 1; q++) {
    if (!pending) break;
    if (q == FSIM_NPAIR) { ok = false; break; }
    const int leader = __ffsll((long long)pending) - 1;
    const int k = __builtin_amdgcn_readlane(key, leader);
    const bool mt = haskey && key == k;
    if (mt) S.pid = q;
    const int nlo = KI(r_chainlen, k >> 8), nhi = KI(r_chainlen, k & 255);
    plo[q] = k >> 8; phi_[q] = k & 255; pbase[q] = total; ybase[q] = ycols;
    total += nlo * nhi; ycols += nhi;
    ok = ok && 6 * nhi <= FSIM_XW;
    pending &= ~__ballot(mt);
    np = q + 1;
  }
  total = __builtin_amdgcn_readfirstlane(total); ycols = __builtin_amdgcn_readfirstlane(ycols);
  if (!__builtin_amdgcn_readfirstlane(ok) || total > FSIM_PCAP - FSIM_YCAP || ycols > FSIM_YCAP) { S.pid = -1; S.npc = -1; return; }
  S.npc = np; S.ptot = total; S.nye = ycols;
  for (int it = c.lane; it < total; it += 64) {
    int q = 0;
#pragma unroll
    for (int t = 1; t < FSIM_NPAIR; t++) q += it >= pbase[t]; // (unused pairs: base = INT_MAX)
    int lo = plo[0], hi = phi_[0], base = pbase[0];
#pragma unroll
    for (int t = 1; t < FSIM_NPAIR; t++) if (q == t) { lo = plo[t]; hi = phi_[t]; base = pbase[t]; }
    const int rem = it - base, nhi = KI(r_chainlen, hi);
    const int e1 = (int)(((float)rem + 0.5f) / (float)nhi), e2 = rem - e1 * nhi;
    const int d1 = KI(chain_dofs, KI(r_chainadr, lo) + e1), d2 = KI(chain_dofs, KI(r_chainadr, hi) + e2);
    pitem[it] = q | (d1 << 8) | (d2 << 16) | (e2 << 24);
  }
  if (c.lane < ycols) {
    int q = 0;
#pragma unroll
    for (int t = 1; t < FSIM_NPAIR; t++) q += c.lane >= ybase[t];
    int hi = phi_[0], base = ybase[0];
#pragma unroll
    for (int t = 1; t < FSIM_NPAIR; t++) if (q == t) { hi = phi_[t]; base = ybase[t]; }
    const int e2 = c.lane - base;
    pitem[FSIM_PCAP - FSIM_YCAP + c.lane] = q | (KI(chain_dofs, KI(r_chainadr, hi) + e2) << 8) | (e2 << 16);
  }
  // (no barrier: fs_hessian's first barrier orders these stores before its loads)
}

// S.jar (to_jar: minus aref) or S.jp = J * vec, using W from fs_body_spatial(vec)
template <class Ctx> DEV void fs_jdot(const Ctx &c, SolSlot &S, int off_vec, bool to_jar, const bool welds = true) {
  float *L = c.L;
  {
    // (unconditional loads: inactive lanes read body 0 / valid addresses and drop the result)
    const S6 w1 = lds6(L + c.ly.W + 6 * (S.bt1 & 255)), w2 = lds6(L + c.ly.W + 6 * (S.bt2 & 255)); // (body 0's W is zero)
    const V3 rel = (w2.l + cross(w2.a, S.r2)) - (w1.l + cross(w1.a, S.r1));
    const float v0 = dot(S.fx, rel), v1 = dot(S.fy, rel), v2 = dot(S.fz, rel);
    if (to_jar) { S.jar[0] = v0 - S.aref[0]; S.jar[1] = v1 - S.aref[1]; S.jar[2] = v2 - S.aref[2]; }
    else { S.jp[0] = v0; S.jp[1] = v1; S.jp[2] = v2; }
    const float lv = S.lsign * L[off_vec + S.ldof];
    if (to_jar) S.ljar = lv - S.laref; else S.ljp = lv;
  }
  if (welds && S.anyweld) {
    for (int e = c.lane; e < c.D.neq; e += 64) {
      float *r = L + c.ly.weld + FSIM_WELDW * e;
      int *ri = reinterpret_cast<int *>(r);
      if (!ri[WD_ACTIVE]) continue;
      int b1 = ri[WD_B1], b2 = ri[WD_B2];
      V3 jt = fs_ptvel(c, c.ly.W, fs_bt(c, b1), ldv3(r + WD_P0)) - fs_ptvel(c, c.ly.W, fs_bt(c, b2), ldv3(r + WD_X2));
      V3 dw = lds6(L + c.ly.W + 6 * b1).a - lds6(L + c.ly.W + 6 * b2).a;
      int dst = to_jar ? WD_JAR : WD_JP;
      float v[6] = {jt.x, jt.y, jt.z, 0, 0, 0};
      for (int q = 0; q < 3; q++) v[3 + q] = r[WD_C + 3 * q] * dw.x + r[WD_C + 3 * q + 1] * dw.y + r[WD_C + 3 * q + 2] * dw.z;
      for (int q = 0; q < 6; q++) r[dst + q] = v[q] - (to_jar ? r[WD_AREF + q] : 0.0f);
    }
    SYNC();
  }
}

// elliptic contact block: force, cost, (optional) 3x3 Hessian w.r.t. jar.  returns state 0/1/2
DEV int fs_cone(const float *jar, float Dn, float Dt, float fri, float *f, float *cost, float *H) {
  float mu = fri * sqrtf(Dn / Dt); // friction * sqrt(R_t/R_n) = friction / sqrt(impratio)
  float U0 = jar[0] * mu, U1 = jar[1] * fri, U2 = jar[2] * fri;
  float N = U0, T = sqrtf(U1 * U1 + U2 * U2);
  if (N >= mu * T || (T <= 0 && N >= 0)) { f[0] = f[1] = f[2] = 0; *cost = 0; return 0; }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    f[0] = -Dn * jar[0]; f[1] = -Dt * jar[1]; f[2] = -Dt * jar[2];
    *cost = 0.5f * (Dn * jar[0] * jar[0] + Dt * (jar[1] * jar[1] + jar[2] * jar[2]));
    if (H) { for (int i = 0; i < 9; i++) H[i] = 0; H[0] = Dn; H[4] = Dt; H[8] = Dt; }
    return 1;
  }
  float Dm = Dn / fmaxf(mu * mu * (1 + mu * mu), 1e-15f), NT = N - mu * T;
  *cost = 0.5f * Dm * NT * NT;
  f[0] = -Dm * NT * mu;
  f[1] = -f[0] / T * U1 * fri; f[2] = -f[0] / T * U2 * fri;
  if (H) {
    float u[2] = {U1 / T, U2 / T}, sc[3] = {mu, fri, fri}, Hu[9];
    Hu[0] = Dm;
    for (int a = 0; a < 2; a++) {
      Hu[1 + a] = Hu[3 * (1 + a)] = -Dm * mu * u[a];
      for (int b = 0; b < 2; b++) Hu[3 * (1 + a) + 1 + b] = Dm * mu * mu * u[a] * u[b] - Dm * mu * NT * ((a == b ? 1.0f : 0.0f) - u[a] * u[b]) / T;
    }
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) H[3 * a + b] = Hu[3 * a + b] * sc[a] * sc[b];
  }
  return 2;
}

// Directional derivatives of the elliptic contact cost along jp at jar (what the line search needs), without forming the 3x3
// cone Hessian: with N = mu jar0, U = fri (jar1, jar2), T = |U| and primes for d/dalpha,
//   top zone:    0;   bottom zone: quadratic in jar;   middle zone: cost = Dm/2 (N - mu T)^2,
//   d1 = Dm (N - mu T)(N' - mu T'),  d2 = Dm (N' - mu T')^2 - Dm (N - mu T) mu T'',  T' = U.U'/T,  T'' = (|U'|^2 - T'^2)/T.
// Identical (up to rounding) to contracting fs_cone's force / Hessian with jp, at a third of the instructions.
DEV int fs_cone_dir(const float *jar, const float *jp, float Dn, float Dt, float fri, float *d1, float *d2) {
  float mu = fri * sqrtf(Dn / Dt);
  float U1 = jar[1] * fri, U2 = jar[2] * fri;
  float N = jar[0] * mu, T = sqrtf(U1 * U1 + U2 * U2);
  if (N >= mu * T || (T <= 0 && N >= 0)) { *d1 = 0; *d2 = 0; return 0; }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    *d1 = Dn * jar[0] * jp[0] + Dt * (jar[1] * jp[1] + jar[2] * jp[2]);
    *d2 = Dn * jp[0] * jp[0] + Dt * (jp[1] * jp[1] + jp[2] * jp[2]);
    return 1;
  }
  float Dm = Dn / fmaxf(mu * mu * (1 + mu * mu), 1e-15f), NT = N - mu * T;
  float V1 = jp[1] * fri, V2 = jp[2] * fri, rT = 1.0f / T;
  float Tp = (U1 * V1 + U2 * V2) * rT, Np = jp[0] * mu;
  float Tpp = (V1 * V1 + V2 * V2 - Tp * Tp) * rT;
  float w = Np - mu * Tp;
  *d1 = Dm * NT * w;
  *d2 = Dm * w * w - Dm * NT * mu * Tpp;
  return 2;
}

// first / second directional derivatives of the constraint cost along jp at jar + alpha*jp
// zone: per-lane record of which piece of the piecewise cost this lane's contact slot (bits 0-1: 0 top / 1 bottom / 2 middle)
// and joint-limit record (bit 2: active) were in when the gradient was taken; *nonquad: some lane's slot / limit is now in another
// piece than at `zone`, or on the cone surface (the only non-quadratic piece)
// this lane's contact slot: adds the first / second directional derivative of its cost at alpha, returns its zone there
DEV int fs_line_slot(const SolSlot &S, float alpha, float &a1, float &a2) {
  int zc = 0;
  if (S.act) {
    if (S.dim1) {
      float j = S.jar[0] + alpha * S.jp[0];
      if (j < 0) { a1 += S.dn * j * S.jp[0]; a2 += S.dn * S.jp[0] * S.jp[0]; zc = 1; }
    } else {
      float jar[3], e1, e2;
      for (int a = 0; a < 3; a++) jar[a] = S.jar[a] + alpha * S.jp[a];
      zc = fs_cone_dir(jar, S.jp, S.dn, S.dt, S.mu, &e1, &e2);
      a1 += e1; a2 += e2;
    }
  }
  return zc;
}
// (T / skT: the further slot sets of a model with more than 64 contact slots, Ctx::NS - 1 of them)
template <class Ctx> DEV void fs_line_eval(const Ctx &c, const SolSlot &S, float alpha, float *d1, float *d2, int zone, bool *nonquad, const SolSlot *T = nullptr,
                                          const SlotK *skT = nullptr) {
  float *L = c.L;
  float a1 = 0, a2 = 0;
  int zl = 0; // this lane's limit activity at alpha
  const int zc = fs_line_slot(S, alpha, a1, a2); // this lane's contact zone at alpha
  bool moved = zc == 2 || zc != (zone & 3);
  if constexpr (Ctx::NS > 1) {
#pragma unroll
    for (int k = 0; k < Ctx::NS - 1; k++) {
      const int zt = fs_line_slot(T[k], alpha, a1, a2);
      moved = moved || zt == 2 || zt != (skT[k].zone & 3);
    }
  }
  if (S.lact) {
    float j = S.ljar + alpha * S.ljp;
    if (j < 0) { a1 += S.ld * j * S.ljp; a2 += S.ld * S.ljp * S.ljp; zl = 1; }
  }
  if (S.anyweld)
    for (int e = c.lane; e < c.D.neq; e += 64) {
      float *r = L + c.ly.weld + FSIM_WELDW * e;
      if (!reinterpret_cast<int *>(r)[WD_ACTIVE]) continue;
      for (int q = 0; q < 6; q++) {
        float j = r[WD_JAR + q] + alpha * r[WD_JP + q], D = r[WD_D + q];
        a1 += D * j * r[WD_JP + q]; a2 += D * r[WD_JP + q] * r[WD_JP + q];
      }
    }
  *nonquad = __ballot(moved || zl != ((zone >> 2) & 1)) != 0;
  *d1 = wave_sum(a1); *d2 = wave_sum(a2);
}

template <class Ctx> DEV void fs_add_wrench(const Ctx &c, int bt, V3 p, V3 F, V3 T, float sign) {
  const int b = bt & 255;
  if (b == 0) return;
  float *G = c.L + c.ly.G + 6 * b;
  V3 mo = (cross(p - ldv3(c.L + c.ly.com + 3 * (bt >> 8)), F) + T) * sign;
  atomicAdd(G + 0, mo.x); atomicAdd(G + 1, mo.y); atomicAdd(G + 2, mo.z);
  atomicAdd(G + 3, sign * F.x); atomicAdd(G + 4, sign * F.y); atomicAdd(G + 5, sign * F.z);
}
// (r: application point relative to the tree's CoM)
template <class Ctx> DEV void fs_add_wrench_r(const Ctx &c, int bt, V3 r, V3 F, float sign) {
  const int b = bt & 255;
  if (b == 0) return;
  float *G = c.L + c.ly.G + 6 * b;
  const V3 mo = cross(r, F) * sign;
  atomicAdd(G + 0, mo.x); atomicAdd(G + 1, mo.y); atomicAdd(G + 2, mo.z);
  atomicAdd(G + 3, sign * F.x); atomicAdd(G + 4, sign * F.y); atomicAdd(G + 5, sign * F.z);
}

// grad = Mx - qfrc_smooth - J' f(jar)
// What the gradient pass already knows about this lane's contact slot and the Hessian pass needs again: whether the
// cone is active and its world-frame stiffness K = F' * Hcone * F (one slot per lane: ncon_max <= 64).

// this lane's contact slot: cone force -> wrenches on its two bodies; returns the cone state and world stiffness
template <class Ctx> DEV SlotK fs_grad_slot(const Ctx &c, const SolSlot &S) {
  SlotK sk;
  sk.on = false;
  sk.zone = 0;
  for (int q = 0; q < 6; q++) sk.K[q] = 0;
  if (S.act) {
    float f[3] = {0, 0, 0}, cc, Hc[9];
    bool on;
    if (S.dim1) {
      on = S.jar[0] < 0;
      if (on) f[0] = -S.dn * S.jar[0];
      for (int q = 0; q < 9; q++) Hc[q] = 0;
      Hc[0] = S.dn;
      sk.zone = on ? 1 : 0;
    } else { sk.zone = fs_cone(S.jar, S.dn, S.dt, S.mu, f, &cc, Hc); on = sk.zone != 0; }
    if (on) { // (top zone: zero force, zero Hessian)
      const V3 fx = S.fx, fy = S.fy, fz = S.fz;
      V3 w0 = fx * Hc[0] + fy * Hc[1] + fz * Hc[2], w1 = fx * Hc[3] + fy * Hc[4] + fz * Hc[5], w2 = fx * Hc[6] + fy * Hc[7] + fz * Hc[8];
      sk.on = true;
      sk.K[0] = fx.x * w0.x + fy.x * w1.x + fz.x * w2.x; sk.K[1] = fx.x * w0.y + fy.x * w1.y + fz.x * w2.y; sk.K[2] = fx.x * w0.z + fy.x * w1.z + fz.x * w2.z;
      sk.K[3] = fx.y * w0.y + fy.y * w1.y + fz.y * w2.y; sk.K[4] = fx.y * w0.z + fy.y * w1.z + fz.y * w2.z; sk.K[5] = fx.z * w0.z + fy.z * w1.z + fz.z * w2.z;
      const V3 F = fx * f[0] + fy * f[1] + fz * f[2];
      fs_add_wrench_r(c, S.bt2, S.r2, F, 1.0f);
      fs_add_wrench_r(c, S.bt1, S.r1, F, -1.0f);
    }
  }
  return sk;
}

// (T / skT: the second slot set, Ctx::NS == 2)
template <class Ctx> DEV SlotK fs_gradient(const Ctx &c, const SolSlot &S, const SolSlot *T = nullptr, SlotK *skT = nullptr) {
  float *L = c.L;
  float *tsum = L + c.ly.scal + SC_TMP; // [ntree] squared gradient norm per kinematic tree (fs_active_islands)
  for (int i = c.lane; i < 6 * c.D.nr; i += 64) L[c.ly.G + i] = 0;
  for (int d = c.lane; d < c.D.nv; d += 64) L[c.ly.grad + d] = L[c.ly.Mx + d] - L[c.ly.smooth + d];
  if (c.lane < 16) tsum[c.lane] = 0;
  SYNC();
  SlotK sk = fs_grad_slot(c, S);
  int tb = S.tb;
  if constexpr (Ctx::NS > 1) {
#pragma unroll
    for (int k = 0; k < Ctx::NS - 1; k++) { skT[k] = fs_grad_slot(c, T[k]); tb |= T[k].tb; }
  }
  if (S.lact && S.ljar < 0) {
    sk.zone |= 4;
    atomicAdd(L + c.ly.grad + S.ldof, S.lsign * S.ld * S.ljar); // -sign*f, f = -D*jar
  }
  if (S.anyweld)
    for (int e = c.lane; e < c.D.neq; e += 64) {
      float *r = L + c.ly.weld + FSIM_WELDW * e;
      int *ri = reinterpret_cast<int *>(r);
      if (!ri[WD_ACTIVE]) continue;
      float f[6];
      for (int q = 0; q < 6; q++) f[q] = -r[WD_D + q] * r[WD_JAR + q];
      V3 F = v3(f[0], f[1], f[2]);
      V3 T_ = v3(r[WD_C] * f[3] + r[WD_C + 3] * f[4] + r[WD_C + 6] * f[5], r[WD_C + 1] * f[3] + r[WD_C + 4] * f[4] + r[WD_C + 7] * f[5],
                 r[WD_C + 2] * f[3] + r[WD_C + 5] * f[4] + r[WD_C + 8] * f[5]);
      fs_add_wrench(c, fs_bt(c, ri[WD_B1]), ldv3(r + WD_P0), F, T_, 1.0f);
      fs_add_wrench(c, fs_bt(c, ri[WD_B2]), ldv3(r + WD_X2), F, T_, -1.0f);
    }
  SYNC();
  for (int d = c.lane; d < c.D.nv; d += 64) {
    const int bd = KI(dof_rbody, d);
    const S6 s_ = lds6(L + c.ly.cdof + 6 * d);
    float acc = 0;
    // only the bodies that carry a constraint hold a wrench (a free arm's dofs skip the walk); two bodies per trip, the second
    // predicated, so that a trip's loads are independent
    for (int mm = KI(r_submask, bd) & tb; mm;) {
      const int b0 = __ffs(mm) - 1;
      mm &= mm - 1;
      const bool two = mm != 0;
      const int b1 = two ? __ffs(mm) - 1 : b0;
      mm &= mm - 1;
      const S6 g0 = lds6(L + c.ly.G + 6 * b0), g1 = lds6(L + c.ly.G + 6 * b1);
      acc += dot6(s_, g0) + (two ? dot6(s_, g1) : 0.0f);
    }
    const float g = L[c.ly.grad + d] - acc;
    L[c.ly.grad + d] = g;
    atomicAdd(tsum + KI(dof_tree, d), g * g);
  }
  SYNC();
  return sk;
}

// Which islands still move.  The Newton system is block diagonal over islands (sets of kinematic trees joined by an active
// constraint), so island I's step is -H_I^-1 g_I whatever the others do, and an island whose own residual is below a quarter of the
// tolerance (scale |g_I| < tol / 4: with <= 16 islands all of them below that puts the whole gradient below tol, the stopping rule)
// takes no step worth taking: parts at rest on the floor, whose contact forces the warm start already balances, while the robot
// island iterates.  Such islands are left out of the iteration -- no contact blocks, no projection, no factorisation, p = 0 --
// which is what a gripping env spends a third of an iteration on and a median env a tenth of a substep.  An island that is below
// the threshold stays there (its x does not move), so the set only shrinks during a solve.  Returns the squared gradient norm;
// *am = bit mask of the trees in islands that still move (dof d moves iff bit KI(dof_tree, d)).
template <class Ctx> DEV float fs_active_islands(const Ctx &c, float scale, bool all, int *am) {
  float *L = c.L;
  const float *tsum = L + c.ly.scal + SC_TMP;
  const int *isl = c.I(c.ly.scal) + SC_ISL;
  const int t = min(c.lane, c.D.ntree - 1);
  const float tv = c.lane < c.D.ntree ? tsum[t] : 0.0f;
  float isum = 0;
  for (int mm = isl[t]; mm;) {
    FS_BITS3(mm, u0, u1, u2, h1, h2);
    const float x0 = tsum[u0], x1 = tsum[u1], x2 = tsum[u2];
    isum += x0 + (h1 ? x1 : 0.0f) + (h2 ? x2 : 0.0f);
  }
  const float thr = 0.25f * c.newton_tol;
  const bool moves = c.lane < c.D.ntree && (all || scale * scale * isum >= thr * thr || !(isum == isum));
  *am = (int)(unsigned)__ballot(moves);
  return wave_sum(tv);
}
#define FS_DOF_MOVES(am_, d_) ((((am_) >> KI(dof_tree, (d_))) & 1) != 0)

DEV int fs_tri(int i, int j) { return i * (i + 1) / 2 + j; }
// packed index of entry (i, j), i >= j, both in the same island, under the map at word offset mp (Layout::hmap or k_tmap)
// map word of dof i: packed row base (12 bits) | local index (7) | island size (7) (MAP_ROWB / MAP_L / MAP_NI); [nv + lane] = dof of a solver lane
template <class Ctx> DEV int fs_hidx(const Ctx &c, int mp, int i, int j) { const int *A = c.I(mp); return MAP_ROWB(A[i]) + MAP_L(A[j]); }

// column of J for chain entry: value of row-space functional on dof d.  For a contact the three rows are
// frame_a . (cdof_lin + cdof_ang x (pos - com)); sign folded in by the caller.
template <class Ctx> DEV V3 fs_col(const Ctx &c, int d, V3 pos) {
  S6 s = lds6(c.L + c.ly.cdof + 6 * d);
  return s.l + cross(s.a, pos - ldv3(c.L + c.ly.com + 3 * KI(r_tree, KI(dof_rbody, d))));
}

#ifndef FSIM_LS_TOL
#define FSIM_LS_TOL 1e-3f // line search: relative tolerance on phi'(alpha) vs phi'(0) and on the step in alpha (MuJoCo's own ls_tolerance default is 1e-2; 1e-6 costs 2 more evaluations per Newton iteration and changes no iteration count)
#endif
#define FSIM_NPAIR 4 // body-pair cross blocks assembled per pass

// H = M + J' W J assembled at BODY level, like a composite-rigid-body pass with "stiffness inertias":
//   a contact point on body b with world-frame stiffness K (3x3, = frame' * cone Hessian * frame) acts on the dofs of
//   chain(b) through P = [-[r]x, I] (r = point - tree com), so it adds the 6x6 symmetric block A_b += P' K P;
//   summing A over each subtree (A^c) gives  H[i][j] += cdof_i' A^c_body(i) cdof_j  on exactly M's sparsity pattern;
//   a contact between two moving bodies (lo, hi) additionally adds -cdof_d1' X cdof_d2, X = P_lo' K P_hi, on
//   chain(lo) x chain(hi).  The cost is independent of the number of contacts per body (20 part-floor contacts
//   collapse into 5 blocks) and every projection runs with one lane per output entry.
// ADD (second slot set of a model with more than 64 contact slots): the blocks of S are assembled the same way and ADDED to the H
// the first set left -- the projection is linear in the blocks --, without M, joint limits and welds (they came with the first set).
// am: trees of the islands that still move (fs_active_islands); contacts, entries and limits of the others are left out
template <class Ctx, bool ADD = false> DEV void fs_hessian(const Ctx &c, const SlotK &sk, const SolSlot &S, const int am = -1) {
  CModel &m = c.m;
  float *L = c.L;
  const int nH = c.I(c.ly.scal)[SC_HWORDS]; // packed island triangles
  const int hm = c.ly.hmap;
  float *A = L + c.ly.hA;                    // [nr][21]: aa(xx,xy,xz,yy,yz,zz) al(9, row = ang comp) ll(xx,xy,xz,yy,yz,zz)
  float *X = L + c.ly.hP;                    // [NPAIR][FSIM_XW] cross blocks (first 36 words), row = lo's spatial comp, col = hi's
  int *pmeta = c.I(c.ly.hP + FSIM_XW * FSIM_NPAIR); // lo[NPAIR], hi[NPAIR], base[NPAIR + 1]
#if defined(FSIM_PROFILE) && !defined(FSIM_NPPROF) && !defined(FSIM_CHOLPROF) && !defined(FSIM_TIMELINE) // (those reuse these profile slots)
  long long th_ = clock64();
#define FS_HPROF(slot) do { long long t1h_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[slot] += (int)((t1h_ - th_) >> 4); th_ = t1h_; } while (0)
#else
#define FS_HPROF(slot) do { } while (0)
#endif
  if (!ADD) for (int i = c.lane; i < nH; i += 64) L[c.ly.H + i] = 0;
  for (int i = c.lane; i < 21 * c.D.nr; i += 64) A[i] = 0;
  if (S.npc > 0) for (int i = c.lane; i < FSIM_XW * S.npc; i += 64) X[i] = 0; // cached pairs: the blocks are filled with the body blocks below
  SYNC();
  // ---- contacts: lane = slot (ncon_max <= 64)
  // cone state and world stiffness of this lane's slot, from the gradient pass of this iteration; the slot's island still moves?
  const bool on = sk.on && (((S.bt1 & 255) != 0 && ((am >> (S.bt1 >> 8)) & 1)) || ((S.bt2 & 255) != 0 && ((am >> (S.bt2 >> 8)) & 1)));
  int blo = 0, bhi = 0, tlo = 0, thi = 0;
  float K[6];
  for (int q = 0; q < 6; q++) K[q] = sk.K[q];
  V3 rlo_ = v3(0, 0, 0), rhi_ = rlo_;
  if (on) { // (this lane's slot record is in registers: SolSlot)
    const bool first_lo = (S.bt1 & 255) <= (S.bt2 & 255);
    const int wl = first_lo ? S.bt1 : S.bt2, wh = first_lo ? S.bt2 : S.bt1;
    blo = wl & 255; bhi = wh & 255; tlo = wl >> 8; thi = wh >> 8;
    rlo_ = first_lo ? S.r1 : S.r2; rhi_ = first_lo ? S.r2 : S.r1;
  }
  // rows of K, G = [r]x K (ang-lin block), and the diagonal blocks
  const V3 K0 = v3(K[0], K[1], K[2]), K1 = v3(K[1], K[3], K[4]), K2 = v3(K[2], K[4], K[5]);
  V3 Glo0 = v3(0, 0, 0), Glo1 = Glo0, Glo2 = Glo0, rhi = Glo0;
  // X = [[Glo * Rhi, Glo], [K * Rhi, K]],  v' * Rhi = v' * (-[rhi]x) = (rhi x v)'  row-wise
#define FS_ADD_X(Xp_) do {                                                                                                  \
    float *Xp = (Xp_);                                                                                                      \
    V3 x0 = cross(rhi, Glo0), x1 = cross(rhi, Glo1), x2 = cross(rhi, Glo2), y0 = cross(rhi, K0), y1 = cross(rhi, K1), y2 = cross(rhi, K2); \
    atomicAdd(Xp + 0, x0.x); atomicAdd(Xp + 1, x0.y); atomicAdd(Xp + 2, x0.z); atomicAdd(Xp + 3, Glo0.x); atomicAdd(Xp + 4, Glo0.y); atomicAdd(Xp + 5, Glo0.z); \
    atomicAdd(Xp + 6, x1.x); atomicAdd(Xp + 7, x1.y); atomicAdd(Xp + 8, x1.z); atomicAdd(Xp + 9, Glo1.x); atomicAdd(Xp + 10, Glo1.y); atomicAdd(Xp + 11, Glo1.z); \
    atomicAdd(Xp + 12, x2.x); atomicAdd(Xp + 13, x2.y); atomicAdd(Xp + 14, x2.z); atomicAdd(Xp + 15, Glo2.x); atomicAdd(Xp + 16, Glo2.y); atomicAdd(Xp + 17, Glo2.z); \
    atomicAdd(Xp + 18, y0.x); atomicAdd(Xp + 19, y0.y); atomicAdd(Xp + 20, y0.z); atomicAdd(Xp + 21, K0.x); atomicAdd(Xp + 22, K0.y); atomicAdd(Xp + 23, K0.z); \
    atomicAdd(Xp + 24, y1.x); atomicAdd(Xp + 25, y1.y); atomicAdd(Xp + 26, y1.z); atomicAdd(Xp + 27, K1.x); atomicAdd(Xp + 28, K1.y); atomicAdd(Xp + 29, K1.z); \
    atomicAdd(Xp + 30, y2.x); atomicAdd(Xp + 31, y2.y); atomicAdd(Xp + 32, y2.z); atomicAdd(Xp + 33, K2.x); atomicAdd(Xp + 34, K2.y); atomicAdd(Xp + 35, K2.z); \
  } while (0)
  if (on) {
#pragma unroll
    for (int side = 0; side < 2; side++) {
      int b = side ? bhi : blo;
      if (b == 0) continue;
      V3 rr = side ? rhi_ : rlo_;
      // G[:, c] = rr x K[:, c]  (K symmetric: column c = row c); stored by rows G_a = (G[a][0], G[a][1], G[a][2])
      V3 c0 = cross(rr, K0), c1 = cross(rr, K1), c2 = cross(rr, K2);
      V3 G0 = v3(c0.x, c1.x, c2.x), G1 = v3(c0.y, c1.y, c2.y), G2 = v3(c0.z, c1.z, c2.z);
      // A_aa rows = rr x G_a   (= [r]x K [r]x')
      V3 a0 = cross(rr, G0), a1 = cross(rr, G1), a2 = cross(rr, G2);
      float *Ab = A + 21 * b;
      atomicAdd(Ab + 0, a0.x); atomicAdd(Ab + 1, a0.y); atomicAdd(Ab + 2, a0.z); atomicAdd(Ab + 3, a1.y); atomicAdd(Ab + 4, a1.z); atomicAdd(Ab + 5, a2.z);
      atomicAdd(Ab + 6, G0.x); atomicAdd(Ab + 7, G0.y); atomicAdd(Ab + 8, G0.z); atomicAdd(Ab + 9, G1.x); atomicAdd(Ab + 10, G1.y); atomicAdd(Ab + 11, G1.z);
      atomicAdd(Ab + 12, G2.x); atomicAdd(Ab + 13, G2.y); atomicAdd(Ab + 14, G2.z);
      atomicAdd(Ab + 15, K[0]); atomicAdd(Ab + 16, K[1]); atomicAdd(Ab + 17, K[2]); atomicAdd(Ab + 18, K[3]); atomicAdd(Ab + 19, K[4]); atomicAdd(Ab + 20, K[5]);
      if (side == 0) { Glo0 = G0; Glo1 = G1; Glo2 = G2; } else rhi = rr;
    }
    if (S.pid >= 0) FS_ADD_X(X + FSIM_XW * S.pid); // cached pair block of this slot (zeroed above)
  }
  FS_HPROF(48);
  // the subtree sums are only needed when a non-root body (a robot link beyond the base) carries a contact block: dm = those bodies
  int dm = 0;
  if (on) dm = ((blo > 0 && KI(r_parent, blo) > 0) ? 1 << blo : 0) | ((bhi > 0 && KI(r_parent, bhi) > 0) ? 1 << bhi : 0);
  dm = wave_or(dm);
  SYNC();
  // cached pairs, stage 1: Y_q = X_q * cdof(chain(hi_q)), one lane per column, written over X_q
  int pairon = 0;
  if (S.npc > 0) {
#pragma unroll
    for (int q = 0; q < FSIM_NPAIR; q++) pairon |= (__ballot(on && S.pid == q) != 0) << q;
    // (a pair none of whose contacts is in an active cone zone is skipped: its dofs may lie in different islands, where
    //  fs_hidx means nothing)
    if (pairon) {
      const int w = c.I(c.ly.pitem)[FSIM_PCAP - FSIM_YCAP + min(c.lane, S.nye - 1)]; // (lanes beyond the last column redo it and store nothing)
      const int q = w & 255, d2 = (w >> 8) & 255, e2 = w >> 16;
      const float *Xq = X + FSIM_XW * q;
      const float *s2 = L + c.ly.cdof + 6 * d2;
      float y[6];
#pragma unroll
      for (int rr = 0; rr < 6; rr++) {
        float t = 0;
#pragma unroll
        for (int cc = 0; cc < 6; cc++) t += Xq[6 * rr + cc] * s2[cc];
        y[rr] = t;
      }
      SYNC(); // every column has read X before the first one overwrites it
      if (c.lane < S.nye) {
        float *Yq = X + FSIM_XW * q + 6 * e2;
#pragma unroll
        for (int rr = 0; rr < 6; rr++) Yq[rr] = y[rr];
      }
    }
  }
  // ---- composite blocks A^c_b = sum of A over the subtree of b.  Only the contact-carrying non-root bodies (dm: a finger or two)
  // have anything to pass up, so the projection below adds THEIR blocks to the block of body(i) on the fly -- instead of a chain walk
  // over every body (one dependent LDS round trip per link of the arm: 4 kcycles per Newton iteration of a gripping env, 1.6 k now).
  FS_HPROF(49);
  // ---- tree blocks on M's pattern: lane = M entry
  for (int e = c.lane; e < c.D.nM; e += 64) {
    int i = KM_I(e), j = KM_J(e);
    if (am != -1 && !FS_DOF_MOVES(am, i)) continue;
    const int bi = KI(dof_rbody, i);
    float Ab[21];
#pragma unroll
    for (int q = 0; q < 21; q++) Ab[q] = A[21 * bi + q];
    if (dm)
      for (int mm = KI(r_submask, bi) & dm & ~(1 << bi); mm; mm &= mm - 1) {
        const float *Ad = A + 21 * (__ffs(mm) - 1);
#pragma unroll
        for (int q = 0; q < 21; q++) Ab[q] += Ad[q];
      }
    S6 si = lds6(L + c.ly.cdof + 6 * i), sj = lds6(L + c.ly.cdof + 6 * j);
    // t = A * sj
    V3 ta = v3(Ab[0] * sj.a.x + Ab[1] * sj.a.y + Ab[2] * sj.a.z + Ab[6] * sj.l.x + Ab[7] * sj.l.y + Ab[8] * sj.l.z,
               Ab[1] * sj.a.x + Ab[3] * sj.a.y + Ab[4] * sj.a.z + Ab[9] * sj.l.x + Ab[10] * sj.l.y + Ab[11] * sj.l.z,
               Ab[2] * sj.a.x + Ab[4] * sj.a.y + Ab[5] * sj.a.z + Ab[12] * sj.l.x + Ab[13] * sj.l.y + Ab[14] * sj.l.z);
    V3 tl = v3(Ab[6] * sj.a.x + Ab[9] * sj.a.y + Ab[12] * sj.a.z + Ab[15] * sj.l.x + Ab[16] * sj.l.y + Ab[17] * sj.l.z,
               Ab[7] * sj.a.x + Ab[10] * sj.a.y + Ab[13] * sj.a.z + Ab[16] * sj.l.x + Ab[18] * sj.l.y + Ab[19] * sj.l.z,
               Ab[8] * sj.a.x + Ab[11] * sj.a.y + Ab[14] * sj.a.z + Ab[17] * sj.l.x + Ab[19] * sj.l.y + Ab[20] * sj.l.z);
    if (ADD) L[c.ly.H + fs_hidx(c, hm, i, j)] += dot(si.a, ta) + dot(si.l, tl);
    else L[c.ly.H + fs_hidx(c, hm, i, j)] = L[c.ly.M + KM_P(e)] + dot(si.a, ta) + dot(si.l, tl);
  }
  SYNC();
  FS_HPROF(50);
  // ---- body-pair cross blocks: -cdof_d1' X cdof_d2 on chain(lo) x chain(hi), one lane per entry
#define FS_PAIR_ITEM(q_, d1_, d2_) do {                                                   \
    const float *Xq = X + 36 * (q_);                                                      \
    const float *s1 = L + c.ly.cdof + 6 * (d1_), *s2 = L + c.ly.cdof + 6 * (d2_);        \
    float v = 0;                                                                          \
    _Pragma("unroll") for (int rr = 0; rr < 6; rr++) {                                    \
      float t = 0;                                                                        \
      _Pragma("unroll") for (int cc = 0; cc < 6; cc++) t += Xq[6 * rr + cc] * s2[cc];     \
      v += s1[rr] * t;                                                                    \
    }                                                                                     \
    if ((d1_) == (d2_)) v *= 2.0f;                                                        \
    atomicAdd(L + c.ly.H + fs_hidx(c, hm, max((d1_), (d2_)), min((d1_), (d2_))), -v);     \
  } while (0)
  if (S.npc > 0) {
    // cached pairs, stage 2: entry (d1, d2) += -cdof_d1 . Y_q[:, column of d2]   (Y was written before the last barrier)
    if (pairon) {
      const int *pitem = c.I(c.ly.pitem);
      for (int it = c.lane; it < S.ptot; it += 64) {
        const int w = pitem[it], q = w & 255, d1 = (w >> 8) & 255, d2 = (w >> 16) & 255, e2 = w >> 24;
        const float *Yq = X + FSIM_XW * q + 6 * e2;
        const float *s1 = L + c.ly.cdof + 6 * d1;
        float v = 0;
#pragma unroll
        for (int rr = 0; rr < 6; rr++) v += s1[rr] * Yq[rr];
        if (d1 == d2) v *= 2.0f;
        const int hx = fs_hidx(c, hm, max(d1, d2), min(d1, d2));
        if ((pairon >> q) & 1) atomicAdd(L + c.ly.H + hx, -v);
      }
    }
  } else if (S.npc < 0) {
  // more pairs / items than the cache holds: FSIM_NPAIR distinct pairs per pass, elected and decoded on the fly
  bool haskey = on && blo != 0 && bhi != blo;
  const int key = blo * 256 + bhi;
  unsigned long long pending = __ballot(haskey);
  while (pending) {
    int pid = -1, np = 0, total = 0;
    if (c.lane == 0) pmeta[2 * FSIM_NPAIR] = 0;
    while (pending && np < FSIM_NPAIR) {
      int leader = __ffsll((long long)pending) - 1;
      int k = __builtin_amdgcn_readlane(key, leader);
      bool mt = haskey && key == k;
      if (mt) pid = np;
      int lo = k >> 8, hi = k & 255;
      total += KI(r_chainlen, lo) * KI(r_chainlen, hi);
      if (c.lane == 0) { pmeta[np] = lo; pmeta[FSIM_NPAIR + np] = hi; pmeta[2 * FSIM_NPAIR + np + 1] = total; }
      pending &= ~__ballot(mt);
      np++;
    }
    for (int i = c.lane; i < 36 * np; i += 64) X[i] = 0;
    SYNC();
    if (pid >= 0) { FS_ADD_X(X + 36 * pid); haskey = false; }
    SYNC();
    for (int it = c.lane; it < total; it += 64) {
      int q = 0;
      while (q + 1 < np && it >= pmeta[2 * FSIM_NPAIR + q + 1]) q++;
      int rem = it - pmeta[2 * FSIM_NPAIR + q];
      int lo = pmeta[q], hi = pmeta[FSIM_NPAIR + q];
      int nhi = KI(r_chainlen, hi);
      int e1 = (int)(((float)rem + 0.5f) / (float)nhi), e2 = rem - e1 * nhi;
      int d1 = KI(chain_dofs, KI(r_chainadr, lo) + e1), d2 = KI(chain_dofs, KI(r_chainadr, hi) + e2);
      FS_PAIR_ITEM(q, d1, d2);
    }
    SYNC();
  }
  }
#undef FS_PAIR_ITEM
#undef FS_ADD_X
  FS_HPROF(53);
  if (!ADD && S.lact && S.ljar < 0 && (am == -1 || FS_DOF_MOVES(am, S.ldof))) atomicAdd(L + c.ly.H + fs_hidx(c, hm, S.ldof, S.ldof), S.ld);
  if (!ADD && S.anyweld)
  for (int e = c.lane; e < c.D.neq; e += 64) {
    float *r = L + c.ly.weld + FSIM_WELDW * e;
    int *ri = reinterpret_cast<int *>(r);
    if (!ri[WD_ACTIVE]) continue;
    int b1 = ri[WD_B1], b2 = ri[WD_B2];
    V3 p0 = ldv3(r + WD_P0), x2 = ldv3(r + WD_X2);
    int n1 = KI(r_chainlen, b1), n2 = KI(r_chainlen, b2), a1 = KI(r_chainadr, b1), a2 = KI(r_chainadr, b2);
    for (int e1 = 0; e1 < n1 + n2; e1++) {
      bool f1 = e1 < n1;
      int d1 = f1 ? KI(chain_dofs, a1 + e1) : KI(chain_dofs, a2 + e1 - n1);
      float sg1 = f1 ? 1.0f : -1.0f;
      V3 t1 = fs_col(c, d1, f1 ? p0 : x2) * sg1;
      V3 w1 = lds6(L + c.ly.cdof + 6 * d1).a * sg1;
      float j1[6] = {t1.x, t1.y, t1.z, 0, 0, 0};
      for (int q = 0; q < 3; q++) j1[3 + q] = r[WD_C + 3 * q] * w1.x + r[WD_C + 3 * q + 1] * w1.y + r[WD_C + 3 * q + 2] * w1.z;
      for (int q = 0; q < 6; q++) j1[q] *= r[WD_D + q];
      for (int e2 = 0; e2 < n1 + n2; e2++) {
        bool f2 = e2 < n1;
        int d2 = f2 ? KI(chain_dofs, a1 + e2) : KI(chain_dofs, a2 + e2 - n1);
        if (d2 > d1) continue;
        float sg2 = f2 ? 1.0f : -1.0f;
        V3 t2 = fs_col(c, d2, f2 ? p0 : x2) * sg2;
        V3 w2 = lds6(L + c.ly.cdof + 6 * d2).a * sg2;
        float v = j1[0] * t2.x + j1[1] * t2.y + j1[2] * t2.z;
        for (int q = 0; q < 3; q++) v += j1[3 + q] * (r[WD_C + 3 * q] * w2.x + r[WD_C + 3 * q + 1] * w2.y + r[WD_C + 3 * q + 2] * w2.z);
        atomicAdd(L + c.ly.H + fs_hidx(c, hm, d1, d2), v);
      }
    }
  }
  SYNC();
  FS_HPROF(54);
}

// Island Cholesky + solve: p <- -H^-1 grad.  returns false if not SPD.
//
// H is block diagonal over "islands" (sets of kinematic trees joined by an active constraint; a lone tree is its own
// island); the map at `mp` (fs_build_map) gives every dof a solver lane.  A solver lane holds its FULL symmetric row in
// registers and the factorisation is right-looking on the symmetric trailing matrix: at pivot j every lane below j needs
// (a) its own entry A[l][j] -- already in its registers -- and (b) the pivot ROW A[j][k], k > j, which by symmetry is the
// pivot column: one broadcast of lane j's register k per k.  No transposition, no LDS, no cross-lane dependency beyond
// that broadcast:
//   * row phase  -- islands of <= 16 dofs sit in the wave's four 16-lane DPP rows (several islands per row = one block-
//                   diagonal matrix) and the broadcast is a `row_newbcast:j` DPP move: four rows are factored at once,
//                   sequential depth = the fullest row (12 for a free Sawyer + five parts, 15 when it grips one);
//   * big phase  -- an island of 17..32 dofs owns a contiguous lane range and the broadcast is v_readlane from lane
//                   (first + j); big islands are factored one after the other;
//   * beyond 32 dofs (all parts welded to the arm) the factor stays in LDS: fs_chol_lds.
// Forward substitution rides along with the factorisation; the back substitution reads L[j][l] = A[l][j] / L[l][l] from
// the lane's OWN registers (lane l stopped updating its row at step l), so it needs one broadcast per step as well.
template <int J> DEV float fs_rowbc(float v) { // value of lane J of the caller's 16-lane row, in every lane of that row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + J, 0xf, 0xf, true));
}
struct RowBcast { // row phase: lane `pos` of each 16-lane row (DPP row_newbcast)
  template <int J> DEV float get(float v) const { return fs_rowbc<J>(v); }
};
struct LaneBcast { // big phase: lane first + J of the wave (v_readlane, wave-uniform source)
  int first;
  template <int J> DEV float get(float v) const { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), first + J)); }
};

// One phase: every lane owns row `l` (position `pos` in its group, which starts at position pos - l) of a symmetric matrix;
// steps = largest group fill in the wave (uniform).  A[q] = row entries by position.  Unoccupied positions carry a unit
// diagonal (set by the caller), so no per-step "is this pivot live" predicate exists; the lane-role tests (pivot / below /
// above) are single-use compares against `pos` -- kept that way on purpose: when the forward and the backward sweep shared
// them the compiler held 2 x NLOC lane masks in SGPR pairs, spilled them to VGPR lanes and a third of this routine's
// instructions were v_readlane reloads of masks.
typedef float fs_f2 __attribute__((ext_vector_type(2)));
// (the row lives in NLOC / 2 two-float vectors so that the trailing update is v_pk_fma_f32 on register pairs the allocator keeps
//  aligned -- with a plain float array the SLP-packed update paid 2.5 v_mov per packed FMA to build the pairs)
#define FS_ROW(A, k) ((A)[(k) >> 1][(k) & 1])
template <int NLOC, int JJ, class BC> struct FsCholStep {
  DEV static void fwd(fs_f2 (&A)[NLOC / 2], float &b, float &mydinv, float &dmin, const int pos, const int steps, const BC &bc) {
    if (JJ < steps) {
      const float ajj = FS_ROW(A, JJ);
      const float d = bc.template get<JJ>(ajj);
      dmin = fminf(dmin, d);
      const float rinv = rsqrtf(fmaxf(d, 1e-30f));
      // forward substitution, column form: y_j = b_j / L_jj, then b_l -= L[l][j] y_j below the pivot
      const float yj = bc.template get<JJ>(b * rinv);
      const float lij = ajj * rinv; // L[l][j] for pos > j
      int pj = pos;
      asm volatile("" : "+v"(pj) : "v"(yj)); // (ties this step's lane-role compares to this step: see the note above)
      const bool below = pj > JJ;
      b = pj == JJ ? yj : b;
      mydinv = pj == JJ ? rinv : mydinv;
      b = below ? __builtin_fmaf(-lij, yj, b) : b;
      const float w = below ? -lij * rinv : 0.0f; // A[l][k] -= L[l][j] L[k][j] = A[l][j] A[j][k] / d
      const fs_f2 w2 = {w, w};
      if ((JJ + 1) & 1) { // odd first column: scalar
        if (JJ + 1 < NLOC) FS_ROW(A, JJ + 1) = __builtin_fmaf(w, bc.template get<JJ>(FS_ROW(A, JJ + 1)), FS_ROW(A, JJ + 1));
      }
#pragma unroll
      for (int kk = (JJ + 2) >> 1; kk < NLOC / 2; kk++) {
        if (((kk - ((JJ + 2) >> 1)) & 1) == 0 && 2 * kk >= steps) break;
        const fs_f2 t = {bc.template get<JJ>(A[kk].x), bc.template get<JJ>(A[kk].y)};
        A[kk] = __builtin_elementwise_fma(w2, t, A[kk]);
      }
      FsCholStep<NLOC, JJ + 1, BC>::fwd(A, b, mydinv, dmin, pos, steps, bc);
    }
  }
  // backward: L' p = y.  L[j][l] = A[l][j] * dinv_l for l < j (lane l's row as it was when l was the pivot)
  DEV static void bwd(const fs_f2 (&A)[NLOC / 2], float &b, const float mydinv, const int pos, const int steps, const BC &bc) {
    constexpr int J = NLOC - 1 - JJ;
    if (J < steps) {
      const float pj = bc.template get<J>(b * mydinv);
      int pk = pos;
      asm volatile("" : "+v"(pk) : "v"(pj));
      b = pk == J ? pj : b;
      b = pk < J ? __builtin_fmaf(-FS_ROW(A, J) * mydinv, pj, b) : b;
    }
    FsCholStep<NLOC, JJ + 1, BC>::bwd(A, b, mydinv, pos, steps, bc);
  }
};
template <int NLOC, class BC> struct FsCholStep<NLOC, NLOC, BC> {
  DEV static void fwd(fs_f2 (&)[NLOC / 2], float &, float &, float &, const int, const int, const BC &) {}
  DEV static void bwd(const fs_f2 (&)[NLOC / 2], float &, const float, const int, const int, const BC &) {}
};

// dof: the dof this lane owns in this phase (< 0: none); pos / steps as above; writes p[dof].  returns the lane's bad flag
template <int NLOC, class BC, class Ctx> DEV int fs_chol_phase(const Ctx &c, int mp, const int dof, const int pos, const int steps, const BC &bc) {
  static_assert(NLOC % 2 == 0, "rows are stored as float pairs");
  float *L = c.L;
  const float *H = L + c.ly.H;
  const bool row = dof >= 0;
  const int B = row ? c.I(mp)[dof] : 0;
  const int l = MAP_L(B), nI = MAP_NI(B), rowb = MAP_ROWB(B), hI = rowb - l * (l + 1) / 2;
  const int p0 = pos - l; // first position of the lane's island inside its group
  fs_f2 A[NLOC / 2];
  {
    // entry (l, lq) of the island's packed triangle, lq = q - p0: tri(max) + min.  Unconditional loads on an unclamped index
    // (at worst 15 words below the island's base, inside the LDS image) and a selection afterwards: behind `ok ? load : 0`
    // every entry became an exec-masked block of ~20 instructions
    float e[NLOC];
    const int triL = l * (l + 1) / 2;
    int lq = -p0, triQ = p0 * (p0 - 1) / 2; // tri(lq) for lq = -p0 (only used once lq >= l >= 0)
#pragma unroll
    for (int q = 0; q < NLOC; q++) {
      e[q] = H[hI + (l >= lq ? triL + lq : triQ + l)];
      lq++; triQ += lq;
    }
    if (NLOC == 6)
      asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]));
    else if (NLOC == 12)
      asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(e[8]), "+v"(e[9]), "+v"(e[10]), "+v"(e[11]));
    else
#pragma unroll
      for (int q0 = 0; q0 < NLOC; q0 += 8)
        asm volatile("" : "+v"(e[q0]), "+v"(e[q0 + 1]), "+v"(e[q0 + 2]), "+v"(e[q0 + 3]), "+v"(e[q0 + 4]), "+v"(e[q0 + 5]), "+v"(e[q0 + 6]), "+v"(e[q0 + 7]));
#pragma unroll
    for (int q = 0; q < NLOC; q++) {
      const bool ok = row && q < steps && (unsigned)(q - p0) < (unsigned)nI;
      FS_ROW(A, q) = (!row && q == pos) ? 1.0f : (ok ? e[q] : 0.0f); // a lane without a dof: unit diagonal, zero row
    }
  }
  float b = row ? -L[c.ly.grad + dof] : 0.0f, mydinv = 0.0f, dmin = 1.0f;
  FsCholStep<NLOC, 0, BC>::fwd(A, b, mydinv, dmin, pos, steps, bc);
  int pos2 = pos;
  asm volatile("" : "+v"(pos2)); // (the backward sweep derives its lane roles afresh: see FsCholStep)
  FsCholStep<NLOC, 0, BC>::bwd(A, b, mydinv, pos2, steps, bc);
  if (row) L[c.ly.p + dof] = b;
  return !(dmin > 1e-30f); // a non-positive (or NaN) pivot anywhere in this lane's group
}

// ---- big phase on the matrix cores: one island of 17..31 dofs as a 32 x 32 symmetric tile in the accumulator layout of
// v_mfma_f32_32x32x2_f32 (lane = column j = lane % 32, half h = lane / 32; register v holds row 8 (v / 4) + 4 h + v % 4).
// Row 31 / column 31 carry the right-hand side, so the forward substitution IS the elimination.  At pivot p the scaled pivot
// row u (one register, lanes of one half, columns > p) is both MFMA operands: D -= u u' is ONE instruction for the whole
// trailing matrix (fp32 MFMA = an fmaf chain, bit-exact fp32), where the readlane path issues 2.5 instructions per
// (pivot, column) pair.  A wave issues one instruction per ~5 cycles whatever it is, so the 21-dof island of a gripping
// env drops from ~1100 to ~550 issued instructions per factorisation + solve.  Columns <= p are masked out of u, so
// row p and column p keep S^(p)[p][.] = L[.][p] / rinv_p for the back substitution, which is column oriented: lane i
// accumulates sum_j S[j][i] x_j from its OWN registers (one accumulator per half), x_p travels by v_readlane.
typedef float fs_f16v __attribute__((ext_vector_type(16)));
template <int P> struct FsMfmaStep {
  static constexpr int VP = 4 * (P >> 3) + (P & 3), HP = (P >> 2) & 1;
  // forward: pivots P and P + 1 (P even: both rows sit in the same half, registers VP and VP + 1) with ONE MFMA -- the
  // instruction contracts over K = 2 (lanes 0..31 carry k = 0, lanes 32..63 k = 1), so it applies two rank-1 updates at once.
  // Row P + 1 after pivot P's update is formed ahead of the MFMA on the vector ALU (one broadcast + one FMA: the same fmaf the
  // matrix core performs), pivot P + 1's scaled row follows from it, v_permlane32_swap moves it to the other half of the
  // operand register.  One MFMA latency (64 cycles + the 17-cycle read-after-MFMA gap) per TWO pivots.
  DEV static void fwd(fs_f16v &D, float &myrinv, float &dmin, const int n, const int lane) {
    static_assert((P & 1) == 0, "pivots are taken in pairs");
    if (P < n) {
      const float d0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(D[VP]), 32 * HP + P));
      dmin = fminf(dmin, d0);
      const float r0 = rsqrtf(fmaxf(d0, 1e-30f));
      // lanes of half HP with column > P: one unsigned range test
      const bool cols0 = (unsigned)(lane - (32 * HP + P + 1)) < (unsigned)(31 - P);
      const float u0 = cols0 ? D[VP] * r0 : 0.0f;
      myrinv = (lane & 31) == P ? r0 : myrinv;
      float A = u0;
      if (P + 1 < 31) {
        const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u0), 32 * HP + P + 1)); // L[P + 1][P]
        const float row1 = __builtin_fmaf(-m, u0, D[VP + 1]); // row P + 1 after pivot P (columns > P; the others are not used)
        const float d1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row1), 32 * HP + P + 1));
        dmin = fminf(dmin, P + 1 < n ? d1 : 1.0f); // (n odd: row n is empty, d1 = 0, u1 = 0)
        const float r1 = rsqrtf(fmaxf(d1, 1e-30f));
        const bool cols1 = (unsigned)(lane - (32 * HP + P + 2)) < (unsigned)(30 - P);
        const float u1 = cols1 ? row1 * r1 : 0.0f;
        myrinv = (lane & 31) == P + 1 ? r1 : myrinv;
        // u0, u1 are zero outside half HP: swapping u0's upper with u1's lower half leaves [U0 | U1] in one of the two results
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(u0), __float_as_uint(u1), false, false);
        A = __uint_as_float(sw[HP]);
      }
      D = __builtin_amdgcn_mfma_f32_32x32x2f32(A, -A, D, 0, 0, 0);
      FsMfmaStep<P + 2>::fwd(D, myrinv, dmin, n, lane);
    }
  }
  DEV static void bwd(const fs_f16v &D, const float myrinv, float &acc, float &res, const int n, const int lane) {
    constexpr int Q = 30 - P; // pivots 30 .. 0
    constexpr int VQ = 4 * (Q >> 3) + (Q & 3), HQ = (Q >> 2) & 1;
    if (Q < n) {
      const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(D[VQ]), 32 * HQ + 31));
      const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), Q));
      const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), Q + 32));
      const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myrinv), Q)); // (the caller passes 1 / d_Q = rinv_Q^2)
      const float xq = r2 * (sy - t0 - t1);
      acc = (unsigned)(lane - 32 * HQ) < (unsigned)Q ? __builtin_fmaf(D[VQ], xq, acc) : acc; // half HQ, columns < Q
      res = lane == Q ? xq : res;
    }
    FsMfmaStep<P + 1>::bwd(D, myrinv, acc, res, n, lane);
  }
};
template <> struct FsMfmaStep<31> {
  DEV static void bwd(const fs_f16v &, const float, float &, float &, const int, const int) {}
};
template <> struct FsMfmaStep<32> {
  DEV static void fwd(fs_f16v &, float &, float &, const int, const int) {}
};
// first: first big-phase lane of the island, n: its size (17..31).  returns the bad flag (uniform)
// (a real function: its 16-register accumulator tile must not weigh on the register allocation of the substep loop, and
//  only the rare env with a big island ever calls it)
// (the LDS base travels as a 32-bit LDS address: naming the dynamic-LDS symbol inside a non-kernel function costs a table lookup
//  -- s_getpc + s_load + wait, ~400 cycles -- at every use the compiler does not merge)
typedef __attribute__((address_space(3))) float fs_lds_f;
// nst_ < 0: the tile is read from the packed Hessian in LDS (what fs_hessian assembled; the integrator's M + h D).
// nst_ >= 0 (round 4): the tile is ASSEMBLED HERE, on the matrix cores.  The island's Hessian is M + sum_c Jc' Kc Jc over its nst_
// active contacts (+ D e e' per active joint limit).  Column k of Jc is the velocity the contact point picks up per unit velocity of
// dof k, v_k = [k in chain(b2)] (lin_k + ang_k x r2) - [k in chain(b1)] (lin_k + ang_k x r1) -- lane k has cdof_k in registers and
// the chain test is one bit of its body's subtree mask --, the world stiffness Kc = G G' was factored by the contact's lane
// (fs_stage_big), so the contact adds the three rank-1 terms u_a u_a', u = G' v: exactly what v_mfma_f32_32x32x2_f32 accumulates,
// two at a time (K = 2: the lanes of the lower half carry one contact, those of the upper half the next one).  1.5 MFMAs per
// contact straight into the accumulator tile the factorisation then works on, instead of 21 + 21 + 36 LDS float atomics per contact
// (which cost ~4 cycles per active lane in the CU's one LDS unit), a projection pass over M's pattern, a pass over the body-pair
// entries and the read-back of the packed triangle.  stage_: word offset of the staged records (FS_STW words each, pairs zero-padded),
// preceded by 32 words of right-hand side and 32 words of joint-limit diagonal (tile order).
#define FS_STW 16
#define FS_ST_HEAD 64
template <class Ctx> DEV int fs_mfma_tile_solve(const Ctx &c, const int mp, const int first, const int n, const int nst, const int stage) {
  float *L = c.L;
  const float *H = L + c.ly.H;
  const int nv = c.D.nv;
  float *rhs = L + (nst >= 0 ? stage : c.ly.hA); // 32 words of scratch: the Hessian body blocks (>= 21 nr + 160 words) are dead once H is assembled
  const int k = c.lane & 31, h = c.lane >> 5;
  // every load below is unconditional on a clamped index and the selection follows: behind `if` the 32 reads of the tile
  // became 32 exec-masked blocks with a wait each
  const int lw = c.I(mp)[nv + first + min(k, n - 1)];
  const int dofk = (lw >> 8) & 255;
  const float gk = L[c.ly.grad + dofk];
  const int hI = MAP_ROWB(__builtin_amdgcn_readfirstlane(c.I(mp)[__builtin_amdgcn_readfirstlane(dofk)])); // packed base of the island (its first dof -- lane 0's -- has l = 0)
  if (c.lane < 32) rhs[k] = k < n ? -gk : 0.0f;
  SYNC();
  fs_f16v D;
  float eH[16], eR[16];
  if (nst < 0) {
  // element (i, k) of the packed lower triangle: tri(max) + min.  The index is NOT clamped for rows / columns outside the
  // island: hI + 527 words is inside the LDS image whatever follows H, and the selection below drops what was read
  const int triK = k * (k + 1) / 2;
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int i0 = 8 * (v >> 2) + (v & 3), i = i0 + 4 * h;
    const int triI = h ? (i0 + 4) * (i0 + 5) / 2 : i0 * (i0 + 1) / 2;
    eH[v] = H[hI + (i >= k ? triI + k : triK + i)];
    eR[v] = rhs[i]; // column 31 (lanes 31 and 63): the right-hand side.  (row 31 is never read: zero)
  }
  } else {
    // M's block of the island, from the tree-packed triangles of M: entry (i, k) exists iff the two dofs belong to one kinematic tree.
    // The tile's columns are the island's dofs tree by tree, so the tree of column k occupies the tile indices [k - lk, k - lk + nk)
    // (lk: index of dof k inside its tree, nk: the tree's size -- both in the dof's word of the tree map), and row i = i0 + 4 h with
    // i0 a compile-time constant: no cross-lane traffic, a handful of integer instructions per entry
    const int Tk = c.I(c.ly.k_tmap)[dofk];
    const int rbk = MAP_ROWB(Tk), lk = MAP_L(Tk), nk = MAP_NI(Tk), lock = k - lk, hbk = rbk - lk * (lk + 1) / 2;
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int i0 = 8 * (v >> 2) + (v & 3);
      const int li = i0 + 4 * h - lock;
      const bool same = (unsigned)li < (unsigned)nk;
      const int lic = same ? li : 0; // (clamped: the index stays inside M whatever the row)
      int ia = hbk + lic * (lic + 1) / 2 + lk, ib = rbk + lic;
      asm volatile("" : "+v"(ia), "+v"(ib)); // (both candidates computed, then ONE select: behind `?:` each entry became an exec-masked block of ~20 instructions)
      const int idx = lic >= lk ? ia : ib;
      const float mv = L[c.ly.M + idx];
      eH[v] = same ? mv : 0.0f;
      eR[v] = rhs[i0 + 4 * h];
    }
  }
  // (pins the 32 loads where they are: the compiler otherwise sinks each of them into the branch of the selection below)
  asm volatile("" : "+v"(eH[0]), "+v"(eH[1]), "+v"(eH[2]), "+v"(eH[3]), "+v"(eH[4]), "+v"(eH[5]), "+v"(eH[6]), "+v"(eH[7]),
                    "+v"(eH[8]), "+v"(eH[9]), "+v"(eH[10]), "+v"(eH[11]), "+v"(eH[12]), "+v"(eH[13]), "+v"(eH[14]), "+v"(eH[15]));
  asm volatile("" : "+v"(eR[0]), "+v"(eR[1]), "+v"(eR[2]), "+v"(eR[3]), "+v"(eR[4]), "+v"(eR[5]), "+v"(eR[6]), "+v"(eR[7]),
                    "+v"(eR[8]), "+v"(eR[9]), "+v"(eR[10]), "+v"(eR[11]), "+v"(eR[12]), "+v"(eR[13]), "+v"(eR[14]), "+v"(eR[15]));
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int i = 8 * (v >> 2) + 4 * h + (v & 3);
    D[v] = max(i, k) < n ? eH[v] : 0.0f; // (the right-hand side column is put in after the assembly: its lanes carry u = 0 until then)
  }
  if (nst >= 0) {
    const bool valid = k < n;
    const S6 sk_ = lds6(L + c.ly.cdof + 6 * dofk);
    const int sub = valid ? KI(r_submask, KI(dof_rbody, dofk)) : 0; // bodies whose chain holds dof k
    // joint limits: D e_k e_k' per limited dof -- a DIAGONAL term (as many rank-1 terms as there are active limits, so no single MFMA):
    // entry (k, k) sits in lane k + 32 ((k >> 2) & 1), register 4 (k >> 3) + (k & 3)
    {
      const float dl = valid ? L[stage + 32 + k] : 0.0f;
#pragma unroll
      for (int v = 0; v < 16; v++) D[v] += (8 * (v >> 2) + 4 * h + (v & 3)) == k ? dl : 0.0f;
    }
    // two accumulator tiles (this routine has the registers): the three rank-1 terms of a pair of contacts alternate between them, so
    // an MFMA never waits for the one before it; the NEXT pair's record is fetched before this pair's arithmetic (one LDS round trip
    // per pair would otherwise sit in front of every iteration: a lone wave has nothing else to hide it behind)
    fs_f16v D2;
#pragma unroll
    for (int v = 0; v < 16; v++) D2[v] = 0.0f;
    const float *rec = L + stage + FS_ST_HEAD + FS_STW * h;
    float w[13];
#pragma unroll
    for (int t = 0; t < 13; t++) w[t] = rec[t];
#pragma unroll 1
    for (int q = 0; q < nst; q += 2) {
      float wn[13];
      const float *rn = rec + FS_STW * min(q + 2, nst - 1); // (the last trip re-reads a record it does not use)
#pragma unroll
      for (int t = 0; t < 13; t++) wn[t] = rn[t];
      const V3 r1 = v3(w[0], w[1], w[2]), r2 = v3(w[3], w[4], w[5]);
      const int bb = __float_as_int(w[12]);
      const bool in1 = (sub >> (bb & 255)) & 1, in2 = (sub >> ((bb >> 8) & 255)) & 1;
      const V3 v1 = sk_.l + cross(sk_.a, r1), v2 = sk_.l + cross(sk_.a, r2);
      const V3 vv = (in2 ? v2 : v3(0, 0, 0)) - (in1 ? v1 : v3(0, 0, 0));
      const float u0 = w[6] * vv.x + w[7] * vv.y + w[9] * vv.z, u1 = w[8] * vv.y + w[10] * vv.z, u2 = w[11] * vv.z; // u = G' v
      D = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, u0, D, 0, 0, 0);
      D2 = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, u1, D2, 0, 0, 0);
      D = __builtin_amdgcn_mfma_f32_32x32x2f32(u2, u2, D, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 13; t++) w[t] = wn[t];
    }
#pragma unroll
    for (int v = 0; v < 16; v++) D[v] += D2[v];
  }
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int i = 8 * (v >> 2) + 4 * h + (v & 3);
    D[v] = (k == 31 && i < n) ? eR[v] : D[v];
  }
  float myrinv = 0.0f, acc = 0.0f, res = 0.0f, dmin = 1.0f;
#ifdef FSIM_CHOLPROF
  long long tm0_ = clock64();
#endif
  FsMfmaStep<0>::fwd(D, myrinv, dmin, n, c.lane);
#ifdef FSIM_CHOLPROF
  { long long tm1_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[50] += (int)((tm1_ - tm0_) >> 4); tm0_ = tm1_; }
#endif
  FsMfmaStep<0>::bwd(D, myrinv * myrinv, acc, res, n, c.lane);
#ifdef FSIM_CHOLPROF
  { long long tm1_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[53] += (int)((tm1_ - tm0_) >> 4); }
#endif
  if (c.lane < n) L[c.ly.p + dofk] = res;
  return !(dmin > 1e-30f);
}

// cone state, world-frame stiffness K = F' Hcone F and world force of this lane's contact slot (what fs_gradient computes on its way)
DEV SlotK fs_slot_k(const SolSlot &S, V3 *Fw) {
  SlotK sk;
  sk.on = false;
  sk.zone = 0;
  for (int q = 0; q < 6; q++) sk.K[q] = 0;
  *Fw = v3(0, 0, 0);
  if (S.act) {
    float f[3] = {0, 0, 0}, cc, Hc[9];
    bool on;
    if (S.dim1) {
      on = S.jar[0] < 0;
      if (on) f[0] = -S.dn * S.jar[0];
      for (int q = 0; q < 9; q++) Hc[q] = 0;
      Hc[0] = S.dn;
      sk.zone = on ? 1 : 0;
    } else { sk.zone = fs_cone(S.jar, S.dn, S.dt, S.mu, f, &cc, Hc); on = sk.zone != 0; }
    if (on) {
      const V3 fx = S.fx, fy = S.fy, fz = S.fz;
      V3 w0 = fx * Hc[0] + fy * Hc[1] + fz * Hc[2], w1 = fx * Hc[3] + fy * Hc[4] + fz * Hc[5], w2 = fx * Hc[6] + fy * Hc[7] + fz * Hc[8];
      sk.on = true;
      sk.K[0] = fx.x * w0.x + fy.x * w1.x + fz.x * w2.x; sk.K[1] = fx.x * w0.y + fy.x * w1.y + fz.x * w2.y; sk.K[2] = fx.x * w0.z + fy.x * w1.z + fz.x * w2.z;
      sk.K[3] = fx.y * w0.y + fy.y * w1.y + fz.y * w2.y; sk.K[4] = fx.y * w0.z + fy.y * w1.z + fz.y * w2.z; sk.K[5] = fx.z * w0.z + fy.z * w1.z + fz.z * w2.z;
      *Fw = fx * f[0] + fy * f[1] + fz * f[2];
    }
  }
  if (S.lact && S.ljar < 0) sk.zone |= 4;
  return sk;
}

// What the gradient pass leaves for the assembly (fs_stage_k, called by the Newton loop once the iteration is known to go on): this
// lane's world stiffness K = F' Hcone F -- zero unless the slot is an active constraint in an active cone zone -- in the six record
// words behind C_G2 (C_AREF .. C_DT and the pad: all of them in SolSlot registers since the solve began, rewritten by the next
// substep's constraint assembly), and the limit's J a - aref in LM_JAR.  Six plain LDS stores per iteration; the cone state is NOT
// recomputed by the assembly, and nothing travels as an argument (with the staging code in the substep loop, or its inputs as VGPR
// arguments, that loop's register allocation went from 14 spill instructions to 34-70, ten to twenty reloads per substep: -7 %).
#define C_KW C_AREF
template <class Ctx> DEV void fs_stage_k(const Ctx &c, const SolSlot &S, const SlotK &sk) {
  // (the lane index goes through an opaque copy: the record addresses are functions of the lane alone, and the compiler otherwise hoists
  //  the two 64-bit pointers to the top of the substep routine and keeps them alive -- spilled -- across the whole loop)
  int ln = c.lane;
  asm volatile("" : "+v"(ln));
  float *r = c.L + c.ly.con + FSIM_CONW * min(ln, c.ly.ncon_max - 1) + C_KW;
  if (ln < c.ly.ncon_max) { r[0] = sk.K[0]; r[1] = sk.K[1]; r[2] = sk.K[2]; r[3] = sk.K[3]; r[4] = sk.K[4]; r[5] = sk.K[5]; }
  if (ln < 2 * c.D.nlim) c.L[c.ly.lim + FSIM_LIMW * ln + LM_JAR] = S.ljar;
  SYNC();
}
// Stage the active contacts and joint limits of ONE big island (kinematic trees `trees`) for the assembly.  Lane = contact slot: K is
// factored K = G G' -- positive semi-definite (the cone cost is convex), possibly singular (a frictionless contact, the cone's
// surface): a pivot below 1e-6 of the largest diagonal entry drops its column (a rank-deficient K leaves rounding noise of ~1e-7 K in
// the later pivots: not a direction).  Returns the number of staged contacts (wave-uniform).
template <class Ctx> DEV int fs_stage_big(const Ctx &c, const int trees, const int stage) {
  static_assert(C_KW + 6 <= FSIM_CONW && C_KW > C_G2, "the stiffness words lie behind the geom ids");
  float *L = c.L;
  const int nslot = c.I(c.ly.scal)[SC_NSLOT];
  const float *r = L + c.ly.con + FSIM_CONW * min(c.lane, c.ly.ncon_max - 1);
  const int *ri = reinterpret_cast<const int *>(r);
  const float K[6] = {r[C_KW], r[C_KW + 1], r[C_KW + 2], r[C_KW + 3], r[C_KW + 4], r[C_KW + 5]};
  const int bt1 = ri[C_B1], bt2 = ri[C_B2], b1 = bt1 & 255, b2 = bt2 & 255;
  const V3 pos = ldv3(r + C_POS);
  const V3 r1 = pos - ldv3(L + c.ly.com + 3 * (bt1 >> 8)), r2 = pos - ldv3(L + c.ly.com + 3 * (bt2 >> 8));
  const bool on = c.lane < nslot && ri[C_ACTIVE] == 1 && K[0] + K[3] + K[5] > 0.0f;
  const bool mine = on && ((b1 != 0 && ((trees >> (bt1 >> 8)) & 1)) || (b2 != 0 && ((trees >> (bt2 >> 8)) & 1)));
  const unsigned long long mask = __ballot(mine);
  const int nst = __popcll(mask);
  if (c.lane < 32) L[stage + 32 + c.lane] = 0.0f;
  SYNC();
  if (mine) {
    const float big = fmaxf(fmaxf(K[0], K[3]), K[5]), eps = 1e-6f * big;
    float g00 = 0, g10 = 0, g20 = 0, g11 = 0, g21 = 0, g22 = 0;
    if (K[0] > eps) { g00 = __builtin_sqrtf(K[0]); const float q = 1.0f / g00; g10 = K[1] * q; g20 = K[2] * q; }
    const float d1 = K[3] - g10 * g10;
    if (d1 > eps) { g11 = __builtin_sqrtf(d1); g21 = (K[4] - g20 * g10) / g11; }
    const float d2 = K[5] - g20 * g20 - g21 * g21;
    if (d2 > eps) g22 = __builtin_sqrtf(d2);
    float *w = L + stage + FS_ST_HEAD + FS_STW * __popcll(mask & ((1ull << c.lane) - 1ull));
    stv3(w, r1); stv3(w + 3, r2);
    w[6] = g00; w[7] = g10; w[8] = g11; w[9] = g20; w[10] = g21; w[11] = g22;
    w[12] = __int_as_float(b1 | (b2 << 8));
  }
  if ((nst & 1) && c.lane < FS_STW) L[stage + FS_ST_HEAD + FS_STW * nst + c.lane] = 0.0f; // (the odd contact's partner: G = 0, bodies 0)
  if (c.lane < 2 * c.D.nlim) {
    const float *q = L + c.ly.lim + FSIM_LIMW * c.lane;
    const int *qi = reinterpret_cast<const int *>(q);
    if (qi[LM_ACTIVE] != 0 && q[LM_JAR] < 0 && ((trees >> KI(dof_tree, qi[LM_DOF])) & 1)) atomicAdd(L + stage + 32 + MAP_L(c.I(c.ly.hmap)[qi[LM_DOF]]), q[LM_D]);
  }
  SYNC();
  return nst;
}
// The two out-of-line entry points: the tile read from the packed Hessian in LDS (fs_hessian's / the integrator's), and the tile
// assembled here from the staged contacts of the island.  (real functions: the 16-register accumulator tile must not weigh on the
// register allocation of the substep loop, and only the env with a big island ever calls them.  The LDS base travels as a 32-bit LDS
// address: naming the dynamic-LDS symbol inside a non-kernel function costs a table lookup -- s_getpc + s_load + wait, ~400 cycles --
// at every use the compiler does not merge)
template <class Ctx> FSIM_OUTLINE int fs_chol_mfma(Ctx cv, unsigned lds_addr_, int mp_, int first_, int n_) {
  float *lds_ = (float *)(fs_lds_f *)(size_t)__builtin_amdgcn_readfirstlane(lds_addr_);
  const Ctx c = fs_rebuild(cv, lds_);
  return fs_mfma_tile_solve(c, __builtin_amdgcn_readfirstlane(mp_), __builtin_amdgcn_readfirstlane(first_), __builtin_amdgcn_readfirstlane(n_), -1, 0);
}
template <class Ctx> FSIM_OUTLINE int fs_newton_mfma(Ctx cv, unsigned lds_addr_, int mp_, int first_, int n_, int trees_) {
  float *lds_ = (float *)(fs_lds_f *)(size_t)__builtin_amdgcn_readfirstlane(lds_addr_);
  const Ctx c = fs_rebuild(cv, lds_);
  const int stage = c.ly.hA;
  const int nst = fs_stage_big(c, __builtin_amdgcn_readfirstlane(trees_), stage);
  return fs_mfma_tile_solve(c, __builtin_amdgcn_readfirstlane(mp_), __builtin_amdgcn_readfirstlane(first_), __builtin_amdgcn_readfirstlane(n_), nst, stage);
}

// Which big islands of this solve have their Hessian assembled on the matrix cores (fs_chol_mfma, nst >= 0): those whose constraint-
// active contacts fit the staging area -- decided ONCE per solve on the slots that are active constraints at all (the cone zones, hence
// the contacts that actually contribute, change from iteration to iteration; the bound does not), so that the set of trees the LDS
// path leaves out is fixed for the solve.  Welds couple rows the tile assembly does not know: the solve then takes the LDS path.
// Returns the kinematic trees of those islands (bit mask).
template <class Ctx> DEV int fs_asm_trees(const Ctx &c, const SolSlot &S) {
#ifndef FSIM_MFMA_HESSIAN
  // (opt-in build: round-4 measurement, DESIGN_HISTORY.md 12 item 1 -- the eligible slow envs get ~10 % faster, the 4096-env step 4-8 % slower)
  return 0;
#elif FSIM_MFMA_HESSIAN == 2
  if (Ctx::NW == 1) return 0;
#endif
  if (Ctx::NS != 1 || S.anyweld) return 0;
  const int nv = c.D.nv;
  const int *hm = c.I(c.ly.hmap), *tail = hm + nv + 64, *isl = c.I(c.ly.scal) + SC_ISL;
  const int nbig = __builtin_amdgcn_readfirstlane(tail[MAP_NBIG]), maxbig = __builtin_amdgcn_readfirstlane(tail[MAP_MAXBIG]);
  if (nbig == 0 || maxbig > 31) return 0;
  const int cap = (21 * c.D.nr + FSIM_XW * FSIM_NPAIR + 3 * FSIM_NPAIR + 4 - FS_ST_HEAD) / FS_STW - 1; // (- 1: the zero partner of an odd contact)
  const int t1 = S.bt1 >> 8, t2 = S.bt2 >> 8;
  const bool m1 = S.act && (S.bt1 & 255) != 0, m2 = S.act && (S.bt2 & 255) != 0;
  int out = 0;
  for (int q = 0; q < nbig; q++) {
    const int first = __builtin_amdgcn_readfirstlane(tail[MAP_BIG0 + 2 * q]);
    const int d0 = __builtin_amdgcn_readfirstlane((hm[nv + first] >> 8) & 255);
    const int trees = __builtin_amdgcn_readfirstlane(isl[KI(dof_tree, d0)]);
    const int cnt = __popcll(__ballot((m1 && ((trees >> t1) & 1)) || (m2 && ((trees >> t2) & 1))));
    if (cnt <= cap) out |= trees;
  }
  return out;
}

// islands larger than 32 dofs (e.g. the fully welded table plus the robot): the factor stays in LDS, left-looking, lane = row
// of a big-phase lane range, all such islands together -- slow path, kept small on purpose
template <class Ctx> DEV int fs_chol_lds(const Ctx &c, int mp) {
  float *L = c.L;
  float *H = L + c.ly.H;
  const int nv = c.D.nv;
  const int dofb = (c.I(mp)[nv + c.lane] >> 8) & 255;
  const bool row = dofb != 255;
  const int i = row ? dofb : 0;
  const int B = row ? c.I(mp)[i] : 0;
  const int l = MAP_L(B), nI = row ? MAP_NI(B) : 0, ib = c.lane - l;
  const int rowb = MAP_ROWB(B), hI = rowb - l * (l + 1) / 2;
  const int steps = (int)wave_max((float)nI);
  int bad = 0;
  float mydinv = 0.0f;
#pragma unroll 1
  for (int jj = 0; jj < steps; jj++) {
    const bool act = jj < nI;
    const int rj = hI + jj * (jj + 1) / 2;
    float s = 0;
    if (act && l >= jj) {
      s = H[rowb + jj];
#pragma unroll 2
      for (int k = 0; k < jj; k++) s -= H[rowb + k] * H[rj + k];
    }
    float d = __shfl(s, ib + jj, 64);
    if (act && !(d > 1e-30f)) { bad = 1; d = 1e-30f; }
    const float rinv = act ? rsqrtf(d) : 0.0f;
    if (act) {
      if (l == jj) { H[rj + jj] = d * rinv; mydinv = rinv; }
      else if (l > jj) H[rowb + jj] = s * rinv;
    }
    SYNC();
  }
  float b = row ? -L[c.ly.grad + i] : 0.0f;
#pragma unroll 1
  for (int jj = 0; jj < steps; jj++) {
    float yj = __shfl(b * mydinv, ib + jj, 64);
    if (jj < nI) { if (l == jj) b = yj; else if (l > jj) b -= H[rowb + jj] * yj; }
  }
#pragma unroll 1
  for (int jj = steps - 1; jj >= 0; jj--) {
    float pj = __shfl(b * mydinv, ib + jj, 64);
    if (jj < nI) { if (l == jj) b = pj; else if (l < jj) b -= H[hI + jj * (jj + 1) / 2 + l] * pj; }
  }
  if (row) L[c.ly.p + i] = b;
  return bad;
}

// A system with an island of more than 64 dofs (MAP_HUGE; the 512-slot kernels of the re-step ladder only: furniture whose reset starts with the
// planks inside each other -- a 72-dof island for table_liden_0921, 84 for bookcase_grevback_0484 -- or eleven planks in one pile): every island, small ones included, is factored
// in LDS from the dof words alone -- lane = rows d = lane and lane + 64, left-looking, one column of every island per trip (two barriers), the
// substitutions column-oriented through a vector indexed by island position (Layout::Mp, dead until M p is formed).  ~150 kcycles per solve of such an
// island against ~9 k for the register paths: this path serves resets and re-steps of models nothing else can hold, not throughput.
template <class Ctx> DEV int fs_chol_all_lds(const Ctx &c, int mp) {
  float *L = c.L;
  float *H = L + c.ly.H, *Y = L + c.ly.Mp;
  const int nv = c.D.nv;
  const int *hm = c.I(mp);
  int l[2], nI[2], rowb[2], hI[2], co[2];
  bool row[2];
  float b[2];
  int steps = 0, bad = 0;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int d = c.lane + 64 * k;
    row[k] = d < nv;
    const int w = hm[row[k] ? d : 0];
    l[k] = MAP_L(w); nI[k] = row[k] ? MAP_NI(w) : 0; rowb[k] = MAP_ROWB(w); hI[k] = rowb[k] - l[k] * (l[k] + 1) / 2;
    co[k] = hm[nv + KI(dof_tree, row[k] ? d : 0)];
    b[k] = row[k] ? -L[c.ly.grad + d] : 0.0f;
    steps = max(steps, nI[k]);
  }
  steps = (int)wave_max((float)steps);
#pragma unroll 1
  for (int jj = 0; jj < steps; jj++) {
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (row[k] && l[k] == jj) { // the pivot of column jj of this row's island
        float dd = H[rowb[k] + jj];
        for (int q = 0; q < jj; q++) dd -= H[rowb[k] + q] * H[rowb[k] + q];
        if (!(dd > 1e-30f)) { bad = 1; dd = 1e-30f; }
        H[rowb[k] + jj] = sqrtf(dd);
      }
    SYNC();
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (row[k] && l[k] > jj && jj < nI[k]) {
        const int rj = hI[k] + jj * (jj + 1) / 2;
        float s = H[rowb[k] + jj];
        for (int q = 0; q < jj; q++) s -= H[rowb[k] + q] * H[rj + q];
        H[rowb[k] + jj] = s / H[rj + jj];
      }
    SYNC();
  }
#pragma unroll 1
  for (int jj = 0; jj < steps; jj++) { // L y = b
#pragma unroll
    for (int k = 0; k < 2; k++) if (row[k] && l[k] == jj) { b[k] = b[k] / H[rowb[k] + jj]; Y[co[k] + jj] = b[k]; }
    SYNC();
#pragma unroll
    for (int k = 0; k < 2; k++) if (row[k] && l[k] > jj && jj < nI[k]) b[k] -= H[rowb[k] + jj] * Y[co[k] + jj];
  }
  SYNC();
#pragma unroll 1
  for (int jj = steps - 1; jj >= 0; jj--) { // L' p = y
#pragma unroll
    for (int k = 0; k < 2; k++) if (row[k] && l[k] == jj && jj < nI[k]) { b[k] = b[k] / H[rowb[k] + jj]; Y[co[k] + jj] = b[k]; }
    SYNC();
#pragma unroll
    for (int k = 0; k < 2; k++) if (row[k] && l[k] < jj && jj < nI[k]) b[k] -= H[hI[k] + jj * (jj + 1) / 2 + l[k]] * Y[co[k] + jj];
  }
#pragma unroll
  for (int k = 0; k < 2; k++) if (row[k]) L[c.ly.p + c.lane + 64 * k] = b[k];
  return bad;
}

// (inlined at its two call sites -- the Newton step and the damped integrator -- both inside fs_substeps)
// am (-1: every island): trees of the islands that take a step (fs_active_islands).  Lanes of the others act as empty lanes (unit
// diagonal) and set p = 0, the row phase only runs as many pivots as the last moving lane needs, a big island that does not move is skipped.
// asm_ok: trees of the big islands whose Hessian is assembled on the matrix cores (fs_asm_trees; their contacts' stiffness staged by
// fs_stage_k) (Newton solve only; 0 / null: every tile is read from the packed Hessian in LDS)
template <class Ctx> DEV bool fs_chol_solve(const Ctx &c, int mp, const int am = -1, const int asm_ok = 0) {
  const int nv = c.D.nv;
  const int *tail = c.I(mp) + nv + 64;
  const int lw = c.I(mp)[nv + c.lane];
  const int rs2 = __builtin_amdgcn_readfirstlane(tail[MAP_RSTEPS]), nbig = __builtin_amdgcn_readfirstlane(tail[MAP_NBIG]);
#ifdef FSIM_PROFILE
  if (c.lane == 0 && mp == c.ly.hmap) { int *ps_ = c.I(c.ly.scal); ps_[51] += nbig > 0 ? tail[MAP_MAXBIG] : 0; ps_[52] += (nbig > 0) + ((nbig > 0 && tail[MAP_MAXBIG] > 31) << 10); } // (bits 10..15: solves with an island beyond the MFMA tile)
#endif
  int bad = 0;
  if constexpr (Ctx::NS >= 4) {
    if (__builtin_amdgcn_readfirstlane(tail[MAP_MAXBIG]) == MAP_HUGE) { bad = fs_chol_all_lds(c, mp); SYNC(); return !wave_or(bad); }
  }
#ifdef FSIM_CHOLPROF
  long long tc_ = clock64();
#define FS_CHPROF(slot) do { long long t1c_ = clock64(); if (c.lane == 0 && mp == c.ly.hmap) c.I(c.ly.scal)[slot] += (int)((t1c_ - tc_) >> 4); tc_ = t1c_; } while (0)
#else
#define FS_CHPROF(slot) do { } while (0)
#endif
  // (second row pass: only models with more than 64 dofs have one -- a compile-time fact for the specialised kernels)
  const int npass = c.D.nv > 64 ? 2 : 1;
#pragma unroll 1
  for (int pass = 0; pass < npass; pass++) {
    int rsteps = (rs2 >> (8 * pass)) & 255;
    if (rsteps > 0) {
      const int dofr = (lw >> (16 * pass)) & 255;
      int dof = dofr == 255 ? -1 : dofr;
      if (am != -1) {
        if (dof >= 0 && !FS_DOF_MOVES(am, dof)) { c.L[c.ly.p + dof] = 0.0f; dof = -1; }
        // the islands of a row sit at consecutive positions and are factored independently: pivots beyond the last moving lane are
        // not needed (and an island is all moving or all still, so no island is cut)
        rsteps = min(rsteps, (int)wave_max(dof >= 0 ? (float)((c.lane & 15) + 1) : 0.0f));
        if (rsteps == 0) continue;
      }
      // (<= 6: what is left for the rows when the robot and the part it holds form a big island -- single parts, one per row)
      if (rsteps <= 6) bad |= fs_chol_phase<6>(c, mp, dof, c.lane & 15, rsteps, RowBcast());
      else if (rsteps <= 12) bad |= fs_chol_phase<12>(c, mp, dof, c.lane & 15, rsteps, RowBcast());
      else bad |= fs_chol_phase<16>(c, mp, dof, c.lane & 15, rsteps, RowBcast());
    }
  }
  FS_CHPROF(48);
  if (nbig > 0) {
    const int maxbig = __builtin_amdgcn_readfirstlane(tail[MAP_MAXBIG]);
    if (maxbig > 31) bad |= fs_chol_lds(c, mp);
    else {
      const int dofb = (lw >> 8) & 255;
#pragma unroll 1
      for (int q = 0; q < nbig; q++) {
        const int first = __builtin_amdgcn_readfirstlane(tail[MAP_BIG0 + 2 * q]), n = __builtin_amdgcn_readfirstlane(tail[MAP_BIG0 + 2 * q + 1]);
        if (am != -1 && !FS_DOF_MOVES(am, __builtin_amdgcn_readfirstlane((c.I(mp)[nv + first] >> 8) & 255))) { // this island does not move
          if (c.lane >= first && c.lane < first + n && dofb != 255) c.L[c.ly.p + dofb] = 0.0f;
          continue;
        }
        if (asm_ok) {
          const int trees = __builtin_amdgcn_readfirstlane(c.I(c.ly.scal)[SC_ISL + KI(dof_tree, __builtin_amdgcn_readfirstlane((c.I(mp)[nv + first] >> 8) & 255))]);
          if (trees & asm_ok) {
            bad |= fs_newton_mfma(c, (unsigned)(size_t)(fs_lds_f *)c.wg_lds(), mp, first, n, trees);
            continue;
          }
        }
        bad |= fs_chol_mfma(c, (unsigned)(size_t)(fs_lds_f *)c.wg_lds(), mp, first, n);
      }
    }
  }
  FS_CHPROF(49);
  SYNC();
  return !wave_or(bad);
}

// ------------------------------------------------------------------------------------------ multi-wave Newton iteration
// The step of a launch lasts as long as its slowest env, and that env is a robot holding a part: 5-8 Newton iterations per
// substep on a coupled island, one instruction stream.  The multi-wave kernels (Ctx::NW = 4 waves per env, k_env_step_mw)
// shorten that stream: wave 0 ("main") runs the env exactly like the one-wave kernel and, at fork points, hands whole passes to
// the three helper waves, which otherwise sleep at a workgroup barrier:
//   * fs_collide on helper 1 beside composite inertia / M / RNE bias / actuation on main (fs_forward_body);
//   * per Newton iteration (mw_iterate): the gradient on main beside the Hessian on the helpers -- contact blocks of the lower
//     bodies / the higher bodies / the body-pair cross blocks on one helper each, composite sums with one lane per
//     (body, component), the projection on M's pattern over all 256 lanes --, the DPP-row factorisation on helper 1 beside the
//     MFMA island on main, M p on helper 1 beside J p on main.
// Protocol: main posts a command word and everybody meets at a workgroup barrier (mw_post); the helpers run the command and go
// back to the barrier.  Command slots alternate (a slow helper may still be reading the previous one).  Every accumulator is
// filled by exactly ONE wave in lane order and sums across accumulators are taken in a fixed order, so the result is a function
// of the state alone, as in the one-wave kernel (run-to-run bit-identical) -- though not bit-identical TO the one-wave kernel.
enum { MW_IDLE = 0, MW_EXIT = 1, MW_COLLIDE = 2, MW_ITER = 3, MW_CHOL = 4, MW_MULM = 5 };
enum { MWC_CMD0 = 0, MWC_CMD1 = 1, MWC_SEQ = 2, MWC_BAD = 3, MWC_CONT = 4, MWC_NPC = 5, MWC_PTOT = 6, MWC_NYE = 7, MWC_A0 = 8, MWC_A1 = 9, MWC_SOLVE = 10, MWC_AM = 11 /* trees of the islands that still move (fs_active_islands) and are assembled in LDS */, MWC_ASM = 12 /* fs_asm_trees of the solve */ };

template <class Ctx> DEV void mw_post(const Ctx &c, int cmd) { // main wave only
  int *w = c.I(c.ly.mwc);
  const int k = w[MWC_SEQ];
  if (c.lane == 0) { w[k & 1] = cmd; w[MWC_SEQ] = k + 1; }
  c.xbar();
}

// zero the Hessian work array THIS helper wave fills next (no barrier between the zeroing and the wave's own atomics): helper 1
// the lower bodies' blocks, helper 2 the higher bodies', helper 3 the cached pair blocks and H itself (complete at barrier [3])
template <class Ctx> DEV void mw_zero(const Ctx &c, int npc) {
  float *L = c.L;
  if (c.wave == 1) for (int i = c.lane; i < 21 * c.D.nr; i += 64) L[c.ly.hA + i] = 0;
  else if (c.wave == 2) for (int i = c.lane; i < 21 * c.D.nr; i += 64) L[c.ly.hAhi + i] = 0;
  else {
    if (npc > 0) for (int i = c.lane; i < FSIM_XW * npc; i += 64) L[c.ly.hP + i] = 0;
    const int nH = c.I(c.ly.scal)[SC_HWORDS];
    for (int i = c.lane; i < nH; i += 64) L[c.ly.H + i] = 0;
  }
  SYNC();
}

// contact blocks: role 1 = the lower body of every contact -> hA, role 2 = the higher body -> hAhi, role 3 = the cached pair blocks
template <class Ctx> DEV void mw_blocks(const Ctx &c, const SolSlot &S, const SlotK &sk, int role) {
  float *L = c.L;
  if (!sk.on) return; // (sk.on: in an active cone zone AND in an island that still moves -- mw_iter_helper)
  const bool first_lo = (S.bt1 & 255) <= (S.bt2 & 255);
  const int blo = (first_lo ? S.bt1 : S.bt2) & 255, bhi = (first_lo ? S.bt2 : S.bt1) & 255;
  const V3 rlo = first_lo ? S.r1 : S.r2, rhi = first_lo ? S.r2 : S.r1;
  const float *K = sk.K;
  const V3 K0 = v3(K[0], K[1], K[2]), K1 = v3(K[1], K[3], K[4]), K2 = v3(K[2], K[4], K[5]);
  if (role <= 2) {
    // (the linear-linear part K of the HIGHER body's block is added by helper 1, into its own array -- the composite pass sums the two
    //  arrays anyway: most contacts have the table (body 0, no block) as their lower body, so helper 1 has lanes to spare while helper 2
    //  carries every contact; an LDS atomic costs ~4 cycles per active lane, scripts/dev/micro/atom.hip)
    if (role == 1 && bhi != 0) {
      float *Ah = L + c.ly.hA + 21 * bhi;
      atomicAdd(Ah + 15, K[0]); atomicAdd(Ah + 16, K[1]); atomicAdd(Ah + 17, K[2]); atomicAdd(Ah + 18, K[3]); atomicAdd(Ah + 19, K[4]); atomicAdd(Ah + 20, K[5]);
    }
    const int b = role == 1 ? blo : bhi;
    if (b == 0) return;
    const V3 rr = role == 1 ? rlo : rhi;
    V3 c0 = cross(rr, K0), c1 = cross(rr, K1), c2 = cross(rr, K2);
    V3 G0 = v3(c0.x, c1.x, c2.x), G1 = v3(c0.y, c1.y, c2.y), G2 = v3(c0.z, c1.z, c2.z);
    V3 a0 = cross(rr, G0), a1 = cross(rr, G1), a2 = cross(rr, G2);
    float *Ab = L + (role == 1 ? c.ly.hA : c.ly.hAhi) + 21 * b;
    atomicAdd(Ab + 0, a0.x); atomicAdd(Ab + 1, a0.y); atomicAdd(Ab + 2, a0.z); atomicAdd(Ab + 3, a1.y); atomicAdd(Ab + 4, a1.z); atomicAdd(Ab + 5, a2.z);
    atomicAdd(Ab + 6, G0.x); atomicAdd(Ab + 7, G0.y); atomicAdd(Ab + 8, G0.z); atomicAdd(Ab + 9, G1.x); atomicAdd(Ab + 10, G1.y); atomicAdd(Ab + 11, G1.z);
    atomicAdd(Ab + 12, G2.x); atomicAdd(Ab + 13, G2.y); atomicAdd(Ab + 14, G2.z);
    if (role == 1) { atomicAdd(Ab + 15, K[0]); atomicAdd(Ab + 16, K[1]); atomicAdd(Ab + 17, K[2]); atomicAdd(Ab + 18, K[3]); atomicAdd(Ab + 19, K[4]); atomicAdd(Ab + 20, K[5]); }
  } else if (S.pid >= 0) {
    // X = [[Glo * Rhi, Glo], [K * Rhi, K]],  v' * Rhi = (rhi x v)'  row-wise  (fs_hessian, FS_ADD_X)
    V3 c0 = cross(rlo, K0), c1 = cross(rlo, K1), c2 = cross(rlo, K2);
    V3 Glo0 = v3(c0.x, c1.x, c2.x), Glo1 = v3(c0.y, c1.y, c2.y), Glo2 = v3(c0.z, c1.z, c2.z);
    float *Xp = L + c.ly.hP + FSIM_XW * S.pid;
    V3 x0 = cross(rhi, Glo0), x1 = cross(rhi, Glo1), x2 = cross(rhi, Glo2), y0 = cross(rhi, K0), y1 = cross(rhi, K1), y2 = cross(rhi, K2);
    atomicAdd(Xp + 0, x0.x); atomicAdd(Xp + 1, x0.y); atomicAdd(Xp + 2, x0.z); atomicAdd(Xp + 3, Glo0.x); atomicAdd(Xp + 4, Glo0.y); atomicAdd(Xp + 5, Glo0.z);
    atomicAdd(Xp + 6, x1.x); atomicAdd(Xp + 7, x1.y); atomicAdd(Xp + 8, x1.z); atomicAdd(Xp + 9, Glo1.x); atomicAdd(Xp + 10, Glo1.y); atomicAdd(Xp + 11, Glo1.z);
    atomicAdd(Xp + 12, x2.x); atomicAdd(Xp + 13, x2.y); atomicAdd(Xp + 14, x2.z); atomicAdd(Xp + 15, Glo2.x); atomicAdd(Xp + 16, Glo2.y); atomicAdd(Xp + 17, Glo2.z);
    atomicAdd(Xp + 18, y0.x); atomicAdd(Xp + 19, y0.y); atomicAdd(Xp + 20, y0.z); atomicAdd(Xp + 21, K0.x); atomicAdd(Xp + 22, K0.y); atomicAdd(Xp + 23, K0.z);
    atomicAdd(Xp + 24, y1.x); atomicAdd(Xp + 25, y1.y); atomicAdd(Xp + 26, y1.z); atomicAdd(Xp + 27, K1.x); atomicAdd(Xp + 28, K1.y); atomicAdd(Xp + 29, K1.z);
    atomicAdd(Xp + 30, y2.x); atomicAdd(Xp + 31, y2.y); atomicAdd(Xp + 32, y2.z); atomicAdd(Xp + 33, K2.x); atomicAdd(Xp + 34, K2.y); atomicAdd(Xp + 35, K2.z);
  }
}

// which cached pairs have a contact in an active cone zone (a pair with none is skipped: its dofs may lie in different islands)
DEV int mw_pairon(const SolSlot &S, const SlotK &sk) {
  int pairon = 0;
#pragma unroll
  for (int q = 0; q < FSIM_NPAIR; q++) pairon |= (__ballot(sk.on && S.pid == q) != 0) << q;
  return pairon;
}

// composite blocks: hAc[b] = sum over the subtree of b of (hA + hAhi), one lane per (body, component), helper waves 1..3
template <class Ctx> DEV void mw_composite(const Ctx &c) {
  float *L = c.L;
  const float *Alo = L + c.ly.hA, *Ahi = L + c.ly.hAhi;
  for (int i = 64 * (c.wave - 1) + c.lane; i < 21 * c.D.nr; i += 64 * (Ctx::NW - 1)) {
    const int b = (int)(((float)i + 0.5f) * (1.0f / 21.0f)), k = i - 21 * b;
    float acc = 0;
    for (int mm = KI(r_submask, b); mm;) {
      FS_BITS3(mm, b0, b1, b2, h1, h2);
      const float x0 = Alo[21 * b0 + k] + Ahi[21 * b0 + k], x1 = Alo[21 * b1 + k] + Ahi[21 * b1 + k], x2 = Alo[21 * b2 + k] + Ahi[21 * b2 + k];
      acc += x0 + (h1 ? x1 : 0.0f) + (h2 ? x2 : 0.0f);
    }
    L[c.ly.hAc + i] = acc;
  }
}

// cached pairs, stage 1 (one wave): Y_q = X_q * cdof(chain(hi_q)), one lane per column, written over X_q
template <class Ctx> DEV void mw_pair_y(const Ctx &c, int nye, int pairon) {
  float *L = c.L;
  if (!pairon) return;
  float *X = L + c.ly.hP;
  const int w = c.I(c.ly.pitem)[FSIM_PCAP - FSIM_YCAP + min(c.lane, nye - 1)];
  const int q = w & 255, d2 = (w >> 8) & 255, e2 = w >> 16;
  const float *Xq = X + FSIM_XW * q;
  const float *s2 = L + c.ly.cdof + 6 * d2;
  float y[6];
#pragma unroll
  for (int rr = 0; rr < 6; rr++) {
    float t = 0;
#pragma unroll
    for (int cc = 0; cc < 6; cc++) t += Xq[6 * rr + cc] * s2[cc];
    y[rr] = t;
  }
  SYNC(); // every column has read X before the first one overwrites it
  if (c.lane < nye) {
    float *Yq = X + FSIM_XW * q + 6 * e2;
#pragma unroll
    for (int rr = 0; rr < 6; rr++) Yq[rr] = y[rr];
  }
}

// tree blocks on M's pattern, waves 0 .. NW - 2: H[i][j] = M[i][j] + cdof_i' hAc[body(i)] cdof_j
template <class Ctx> DEV void mw_project(const Ctx &c, const int am) {
  float *L = c.L;
  const int hm = c.ly.hmap;
  for (int e = 64 * c.wave + c.lane; e < c.D.nM; e += 64 * (Ctx::NW - 1)) { // (waves 0 .. NW - 2; the last one adds the pair entries meanwhile)
    int i = KM_I(e), j = KM_J(e);
    if (!FS_DOF_MOVES(am, i)) continue; // (an island that takes no step: fs_active_islands)
    const float *Ab = L + c.ly.hAc + 21 * KI(dof_rbody, i);
    S6 si = lds6(L + c.ly.cdof + 6 * i), sj = lds6(L + c.ly.cdof + 6 * j);
    V3 ta = v3(Ab[0] * sj.a.x + Ab[1] * sj.a.y + Ab[2] * sj.a.z + Ab[6] * sj.l.x + Ab[7] * sj.l.y + Ab[8] * sj.l.z,
               Ab[1] * sj.a.x + Ab[3] * sj.a.y + Ab[4] * sj.a.z + Ab[9] * sj.l.x + Ab[10] * sj.l.y + Ab[11] * sj.l.z,
               Ab[2] * sj.a.x + Ab[4] * sj.a.y + Ab[5] * sj.a.z + Ab[12] * sj.l.x + Ab[13] * sj.l.y + Ab[14] * sj.l.z);
    V3 tl = v3(Ab[6] * sj.a.x + Ab[9] * sj.a.y + Ab[12] * sj.a.z + Ab[15] * sj.l.x + Ab[16] * sj.l.y + Ab[17] * sj.l.z,
               Ab[7] * sj.a.x + Ab[10] * sj.a.y + Ab[13] * sj.a.z + Ab[16] * sj.l.x + Ab[18] * sj.l.y + Ab[19] * sj.l.z,
               Ab[8] * sj.a.x + Ab[11] * sj.a.y + Ab[14] * sj.a.z + Ab[17] * sj.l.x + Ab[19] * sj.l.y + Ab[20] * sj.l.z);
    L[c.ly.H + fs_hidx(c, hm, i, j)] = L[c.ly.M + KM_P(e)] + dot(si.a, ta) + dot(si.l, tl);
  }
}

// cached pairs, stage 2.  An entry (d1, d2) whose dofs belong to DIFFERENT kinematic trees (a finger against a part: the usual
// kind) lies outside M's pattern, i.e. outside what mw_project stores, so one wave adds those beside the projection
// (cross = true); entries inside a tree (robot self-contact) and the joint limits wait for the projection's stores
// (cross = false, helper 1).  One wave per kind: entries of different pairs may share a word of H.
template <class Ctx> DEV void mw_pair_items(const Ctx &c, int ptot, int pairon, bool cross) {
  float *L = c.L;
  const int hm = c.ly.hmap;
  if (!pairon) return;
  const float *X = L + c.ly.hP;
  const int *pitem = c.I(c.ly.pitem);
  for (int it = c.lane; it < ptot; it += 64) {
    const int w = pitem[it], q = w & 255, d1 = (w >> 8) & 255, d2 = (w >> 16) & 255, e2 = w >> 24;
    const bool same = KI(r_tree, KI(dof_rbody, d1)) == KI(r_tree, KI(dof_rbody, d2));
    const float *Yq = X + FSIM_XW * q + 6 * e2;
    const float *s1 = L + c.ly.cdof + 6 * d1;
    float v = 0;
#pragma unroll
    for (int rr = 0; rr < 6; rr++) v += s1[rr] * Yq[rr];
    if (d1 == d2) v *= 2.0f;
    const int hx = fs_hidx(c, hm, max(d1, d2), min(d1, d2));
    if (((pairon >> q) & 1) && same != cross) atomicAdd(L + c.ly.H + hx, -v);
  }
}

// the helper waves' side of one Newton iteration (command MW_ITER); barriers pair up with mw_iterate_main's
// (S: the helper's copy of the slot records, loaded once per solve -- sid tells which solve it belongs to.  The cone state of the
//  iteration -- which slots are in an active zone, their world stiffness K -- is computed ONCE, by main, and staged in LDS: three
//  helpers re-deriving it from J a - aref cost more than the block atomics they then issue)
template <class Ctx> DEV void mw_iter_helper(const Ctx &c, SolSlot &S, int &sid, int &am_last) {
  const int *w = c.I(c.ly.mwc);
  const int npc = __builtin_amdgcn_readfirstlane(w[MWC_NPC]), ptot = __builtin_amdgcn_readfirstlane(w[MWC_PTOT]), nye = __builtin_amdgcn_readfirstlane(w[MWC_NYE]);
  const int solve = __builtin_amdgcn_readfirstlane(w[MWC_SOLVE]);
#if defined(FSIM_PROFILE) && !defined(FSIM_NPPROF) && !defined(FSIM_CHOLPROF) && !defined(FSIM_TIMELINE)
  // development: where helper 1 / helper 3 spend the time between barriers [1] and [3] (slots 58..60 / 61..63: zero | state + blocks | wait)
  long long thl_ = clock64();
#define FS_HLPROF(k_) do { long long t1h_ = clock64(); if (c.lane == 0 && (c.wave & 1)) c.I(c.ly.scal)[58 + 3 * (c.wave >> 1) + (k_)] += (int)((t1h_ - thl_) >> 4); thl_ = t1h_; } while (0)
#else
#define FS_HLPROF(k_) do { } while (0)
#endif
  mw_zero(c, npc);
  FS_HLPROF(0);
  const float *j = c.L + c.ly.jst + FSIM_JSTW * c.lane;
  if (solve != sid) {
    S = fs_load_slots(c);
    S.pid = reinterpret_cast<const int *>(j)[7];
    sid = solve;
    am_last = ~__builtin_amdgcn_readfirstlane(w[MWC_ASM]); // (first iteration of a solve: every island the LDS path assembles)
  }
  SlotK sk;
#pragma unroll
  for (int q = 0; q < 6; q++) sk.K[q] = j[q];
  const int fl = reinterpret_cast<const int *>(j)[6];
  // the islands that were still moving after the LAST iteration's gradient (this iteration's is being computed by main right now; the
  // set only shrinks, so last iteration's is a superset): the contacts of the others get no blocks
  const int am0 = am_last;
  sk.on = (fl & 1) != 0 && (((S.bt1 & 255) != 0 && ((am0 >> (S.bt1 >> 8)) & 1)) || ((S.bt2 & 255) != 0 && ((am0 >> (S.bt2 >> 8)) & 1)));
  sk.zone = 0;
  const bool limit_on = (fl & 2) != 0;
  const int pairon = npc > 0 ? mw_pairon(S, sk) : 0;
  mw_blocks(c, S, sk, c.wave);
  FS_HLPROF(1);
  c.xbar(); // [3] blocks complete, H zeroed; main has decided whether the iteration goes on
  FS_HLPROF(2);
  const bool cont = __builtin_amdgcn_readfirstlane(w[MWC_CONT]) != 0;
  const int am = __builtin_amdgcn_readfirstlane(w[MWC_AM]); // (this iteration's: main has stored it before barrier [3])
  am_last = am;
  if (cont) {
    if (c.wave == Ctx::NW - 1 && npc > 0) mw_pair_y(c, nye, pairon);
    mw_composite(c);
  }
  c.xbar(); // [4] composite blocks, Y
  if (!cont) return;
  if (c.wave == Ctx::NW - 1) { if (npc > 0) mw_pair_items(c, ptot, pairon, true); } // (the projection's 3 x 64 lanes cover M's entries of every in-scope model in one or two passes)
  else mw_project(c, am);
  c.xbar(); // [5] tree blocks stored
  if (c.wave == 1) {
    if (npc > 0) mw_pair_items(c, ptot, pairon, false);
    if (limit_on && FS_DOF_MOVES(am, S.ldof)) atomicAdd(c.L + c.ly.H + fs_hidx(c, c.ly.hmap, S.ldof, S.ldof), S.ld);
  }
}

// main's side: gradient (the code of fs_gradient, split at its barriers) beside the helpers' Hessian.  Returns the gradient
// norm; *go = false: converged, the helpers have left the iteration and H is not complete.
template <class Ctx> DEV float mw_iterate_main(const Ctx &c, const SolSlot &S, SlotK &sk, float scale, bool *go, int *am, const int asm_ok) {
  float *L = c.L;
  int *w = c.I(c.ly.mwc);
  float *tsum = L + c.ly.scal + SC_TMP; // (fs_active_islands)
#if defined(FSIM_PROFILE) && !defined(FSIM_NPPROF) && !defined(FSIM_CHOLPROF) && !defined(FSIM_TIMELINE)
  long long tm_ = clock64();
#define FS_MWPROF(slot) do { long long t1m_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[slot] += (int)((t1m_ - tm_) >> 4); tm_ = t1m_; } while (0)
#else
#define FS_MWPROF(slot) do { } while (0)
#endif
  V3 Fw;
  sk = fs_slot_k(S, &Fw);
  { // stage this iteration's cone state for the helpers
    float *j = L + c.ly.jst + FSIM_JSTW * c.lane;
#pragma unroll
    for (int q = 0; q < 6; q++) j[q] = sk.K[q];
    reinterpret_cast<int *>(j)[6] = (sk.on ? 1 : 0) | ((S.lact && S.ljar < 0) ? 2 : 0);
  }
  mw_post(c, MW_ITER); // [1]
  if (c.lane < 16) tsum[c.lane] = 0;
  for (int i = c.lane; i < 6 * c.D.nr; i += 64) L[c.ly.G + i] = 0;
  for (int d = c.lane; d < c.D.nv; d += 64) L[c.ly.grad + d] = L[c.ly.Mx + d] - L[c.ly.smooth + d];
  SYNC();
  FS_MWPROF(48);
  if (sk.on) {
    fs_add_wrench_r(c, S.bt2, S.r2, Fw, 1.0f);
    fs_add_wrench_r(c, S.bt1, S.r1, Fw, -1.0f);
  }
  if (S.lact && S.ljar < 0) atomicAdd(L + c.ly.grad + S.ldof, S.lsign * S.ld * S.ljar);
  FS_MWPROF(50);
  SYNC();
  // J'f, the island norms and the decision whether the iteration goes on: main's own data only (the wrenches above are this wave's
  // atomics), so all of it runs BEFORE barrier [3], while the helpers are still filling the Hessian blocks
  for (int d = c.lane; d < c.D.nv; d += 64) {
    const int bd = KI(dof_rbody, d);
    const S6 s_ = lds6(L + c.ly.cdof + 6 * d);
    float acc = 0;
    for (int mm = KI(r_submask, bd) & S.tb; mm;) {
      const int b0 = __ffs(mm) - 1;
      mm &= mm - 1;
      const bool two = mm != 0;
      const int b1 = two ? __ffs(mm) - 1 : b0;
      mm &= mm - 1;
      const S6 g0 = lds6(L + c.ly.G + 6 * b0), g1 = lds6(L + c.ly.G + 6 * b1);
      acc += dot6(s_, g0) + (two ? dot6(s_, g1) : 0.0f);
    }
    const float g = L[c.ly.grad + d] - acc;
    L[c.ly.grad + d] = g;
    atomicAdd(tsum + KI(dof_tree, d), g * g);
  }
  SYNC();
  const int *tailh = c.I(c.ly.hmap) + c.D.nv + 64; // (the LDS-resident factorisation of islands beyond the MFMA tile takes them all)
  const bool every = __builtin_amdgcn_readfirstlane(tailh[MAP_NBIG]) > 0 && __builtin_amdgcn_readfirstlane(tailh[MAP_MAXBIG]) > 31;
  const float gn = sqrtf(fs_active_islands(c, scale, every, am));
  *go = !(scale * gn < c.newton_tol);
  // (the helpers gate THIS iteration's blocks with the set they kept from the last one -- a register of theirs, not this word)
  // (the islands assembled on the matrix cores -- asm_ok -- are none of the helpers' business: no blocks, no projection)
  if (c.lane == 0) { w[MWC_CONT] = *go ? 1 : 0; w[MWC_AM] = *am & ~asm_ok; }
  FS_MWPROF(54);
  c.xbar(); // [3]
  FS_MWPROF(53);
  c.xbar(); // [4]
  FS_MWPROF(55);
  if (!*go) return gn;
  mw_project(c, *am & ~asm_ok);
  FS_MWPROF(56);
  c.xbar(); // [5]
  FS_MWPROF(57);
  return gn;
}

// the factorisation of fs_chol_solve in two halves: the DPP rows (helper 1) and the big islands (main)
template <class Ctx> DEV void mw_chol_rows(const Ctx &c, int mp, const int am) {
  const int nv = c.D.nv;
  const int *tail = c.I(mp) + nv + 64;
  const int lw = c.I(mp)[nv + c.lane];
  int rsteps = __builtin_amdgcn_readfirstlane(tail[MAP_RSTEPS]) & 255;
  int bad = 0;
  const int dofr = lw & 255;
  int dof = dofr == 255 ? -1 : dofr;
  if (dof >= 0 && !FS_DOF_MOVES(am, dof)) { c.L[c.ly.p + dof] = 0.0f; dof = -1; } // (as in fs_chol_solve)
  rsteps = min(rsteps, (int)wave_max(dof >= 0 ? (float)((c.lane & 15) + 1) : 0.0f));
  if (rsteps == 0) return;
  if (rsteps <= 6) bad |= fs_chol_phase<6>(c, mp, dof, c.lane & 15, rsteps, RowBcast());
  else if (rsteps <= 12) bad |= fs_chol_phase<12>(c, mp, dof, c.lane & 15, rsteps, RowBcast());
  else bad |= fs_chol_phase<16>(c, mp, dof, c.lane & 15, rsteps, RowBcast());
  if (__ballot(bad != 0) && c.lane == 0) c.I(c.ly.mwc)[MWC_BAD] = 1;
}
template <class Ctx> DEV int mw_chol_big(const Ctx &c, int mp, const int am, const int asm_ok) {
  const int nv = c.D.nv;
  const int *tail = c.I(mp) + nv + 64;
  const int nbig = __builtin_amdgcn_readfirstlane(tail[MAP_NBIG]), maxbig = __builtin_amdgcn_readfirstlane(tail[MAP_MAXBIG]);
  int bad = 0;
  if (maxbig > 31) bad |= fs_chol_lds(c, mp);
  else {
#pragma unroll 1
    for (int q = 0; q < nbig; q++) {
      const int first = __builtin_amdgcn_readfirstlane(tail[MAP_BIG0 + 2 * q]), n = __builtin_amdgcn_readfirstlane(tail[MAP_BIG0 + 2 * q + 1]);
      const int lwb = c.I(mp)[nv + c.lane];
      if (!FS_DOF_MOVES(am, __builtin_amdgcn_readfirstlane((c.I(mp)[nv + first] >> 8) & 255))) { // this island does not move
        if (c.lane >= first && c.lane < first + n && ((lwb >> 8) & 255) != 255) c.L[c.ly.p + ((lwb >> 8) & 255)] = 0.0f;
        continue;
      }
      if (asm_ok) {
        const int trees = __builtin_amdgcn_readfirstlane(c.I(c.ly.scal)[SC_ISL + KI(dof_tree, __builtin_amdgcn_readfirstlane((c.I(mp)[nv + first] >> 8) & 255))]);
        if (trees & asm_ok) {
          bad |= fs_newton_mfma(c, (unsigned)(size_t)(fs_lds_f *)c.wg_lds(), mp, first, n, trees);
          continue;
        }
      }
      bad |= fs_chol_mfma(c, (unsigned)(size_t)(fs_lds_f *)c.wg_lds(), mp, first, n);
    }
  }
  return wave_or(bad);
}

// helper waves: sleep at the barrier, run what main posts (fs_collide is instantiated here only: main never runs it)
template <class Ctx> DEV void mw_helper_loop(const Ctx &c) {
  const int *w = c.I(c.ly.mwc);
  SolSlot S = {};
  int sid = -1, am_last = -1;
  for (int k = 0;; k++) {
    c.xbar();
    const int cmd = __builtin_amdgcn_readfirstlane(w[k & 1]);
    if (cmd == MW_EXIT) break;
    if (cmd == MW_COLLIDE) { sid = -1; if (c.wave == 1) fs_collide(c); } // (a new substep: the slot records will change)
    else if (cmd == MW_ITER) mw_iter_helper(c, S, sid, am_last);
    else if (cmd == MW_CHOL) { if (c.wave == 1) mw_chol_rows(c, c.ly.hmap, __builtin_amdgcn_readfirstlane(w[MWC_AM])); }
    else if (cmd == MW_MULM) { if (c.wave == 1) fs_mulM(c, __builtin_amdgcn_readfirstlane(w[MWC_A0]), __builtin_amdgcn_readfirstlane(w[MWC_A1])); }
  }
}

// The helper waves' loop as a REAL function: inlined into k_env_step_x it shared one register allocation with the kernel's two env
// loops and the look-ahead job (63 k instructions, 3 092 scratch instructions in the kernel body, against 1 008 in round 3), and what the
// helpers run -- the collision pipeline, the Hessian blocks, the row factorisation -- is per-substep code.
template <class Ctx> static FSIM_OUTLINE void mw_helper_fn(Ctx cv) {
  FS_REBUILD_CTX(cv);
  mw_helper_loop(c);
}

template <class Ctx> DEV float fs_dotv(const Ctx &c, int a, int b) {
  float s = 0;
  for (int d = c.lane; d < c.D.nv; d += 64) s += c.L[a + d] * c.L[b + d];
  return wave_sum(s);
}

// Solve for qacc (c.ly.x) and M*qacc (c.ly.Mx).
#ifdef FSIM_PROFILE
#define FS_SPROF(slot) do { long long t1s_ = clock64(); if (c.lane == 0) c.I(c.ly.scal)[16 + slot] += (int)((t1s_ - t0s_) >> 4); t0s_ = t1s_; } while (0)
#else
#define FS_SPROF(slot) do { } while (0)
#endif
template <class Ctx> DEV void fs_solve(const Ctx &c, int coupled) {
#ifdef FSIM_PROFILE
  long long t0s_ = clock64();
#endif
  CModel &m = c.m;
  float *L = c.L;
  int *scal = c.I(c.ly.scal);
  int nslot = scal[SC_NSLOT];
  // Start from the previous step's acceleration (qacc_warmstart).  Unconstrained envs take the same path: with no
  // active slot the first Newton step (H = M, alpha = 1) is exactly M^-1 qfrc_smooth, so M is never factored on its own.
  for (int d = c.lane; d < c.D.nv; d += 64) L[c.ly.x + d] = L[c.ly.qaccws + d];
  SYNC();
  if constexpr (Ctx::NW > 1) { // M x on helper 1 beside the slot records and J x here
    if (c.lane == 0) { int *w = c.I(c.ly.mwc); w[MWC_A0] = c.ly.Mx; w[MWC_A1] = c.ly.x; }
    mw_post(c, MW_MULM);
  }
  SolSlot S = fs_load_slots(c);
  SolSlot T[Ctx::NS > 1 ? Ctx::NS - 1 : 1] = {};  // (further slot sets: models with more than 64 contact slots -- Ctx::NS == 2: 128, == 8: 512; dead code otherwise)
  if constexpr (Ctx::NS > 1) {
    // several sets: the body-pair cache is not used (its election runs over one set of lanes); every set takes fs_hessian's multi-pass path
    {
      const int b1 = S.bt1 & 255, b2 = S.bt2 & 255;
      S.npc = __ballot(S.act && min(b1, b2) != 0 && b1 != b2) ? -1 : 0;
    }
#pragma unroll
    for (int k = 0; k < Ctx::NS - 1; k++) {
      T[k] = fs_load_slots(c, 64 * (k + 1));
      const int u1 = T[k].bt1 & 255, u2 = T[k].bt2 & 255;
      T[k].npc = __ballot(T[k].act && min(u1, u2) != 0 && u1 != u2) ? -1 : 0;
    }
  } else
  fs_pair_cache(c, S);
#ifdef FSIM_PROFILE
  if (c.lane == 0) scal[52] += ((S.npc < 0) << 16) + ((S.npc > 0) << 24); // solves whose body pairs did not fit the cache / did
#endif
  if constexpr (Ctx::NW == 1) fs_mulM(c, c.ly.Mx, c.ly.x);
  fs_body_spatial(c, c.ly.x);
  fs_jdot(c, S, c.ly.x, true);
  if constexpr (Ctx::NS > 1) {
#pragma unroll
    for (int k = 0; k < Ctx::NS - 1; k++) fs_jdot(c, T[k], c.ly.x, true, false);
  }
  if constexpr (Ctx::NW > 1) mw_post(c, MW_IDLE);
  float scale = c.D.meaninertia_scale;
  int it = 0;
  // multi-wave kernels: the helper waves take part in the iteration unless the solve is one of the rare kinds they do not
  // know (welds, body pairs beyond the cache, a second row pass) -- then main iterates alone, as in the one-wave kernel
  bool mw = false;
  if constexpr (Ctx::NW > 1) {
    mw = !S.anyweld && S.npc >= 0 && c.D.nv <= 64 && Ctx::NS == 1;
    if (mw) {
      c.I(c.ly.jst)[FSIM_JSTW * c.lane + 7] = S.pid;
      if (c.lane == 0) { int *w = c.I(c.ly.mwc); w[MWC_NPC] = S.npc; w[MWC_PTOT] = S.ptot; w[MWC_NYE] = S.nye; w[MWC_SOLVE] += 1; w[MWC_AM] = -1; }
    }
  }
  // big islands whose Hessian is assembled on the matrix cores, straight into the tile the factorisation works on (fs_chol_mfma)
  // (kept in an LDS scalar and read back where it is used: one more value alive across the whole Newton loop cost the substep loop
  //  ten spill reloads per substep)
#ifdef FSIM_MFMA_HESSIAN
  {
    const int asm_set = fs_asm_trees(c, S);
    if (c.lane == 0) { scal[SC_ASM] = asm_set; if constexpr (Ctx::NW > 1) { if (mw) c.I(c.ly.mwc)[MWC_ASM] = asm_set; } }
    SYNC();
  }
#define FS_ASM_OK() __builtin_amdgcn_readfirstlane(scal[SC_ASM])
#else
  // (default build: no island is assembled on the matrix cores -- a compile-time zero, so that none of that path's code, nor the LDS
  //  read per iteration, is in the substep loop)
#define FS_ASM_OK() 0
#endif
  for (; it < c.newton_maxit; it++) {
    SlotK sk, skT[Ctx::NS > 1 ? Ctx::NS - 1 : 1] = {};
    bool ok;
    bool iterated = false;
    if constexpr (Ctx::NW > 1) if (mw) {
      iterated = true;
      bool go;
      int am;
      const int asm_ok = FS_ASM_OK();
      mw_iterate_main(c, S, sk, scale, &go, &am, asm_ok);
      FS_SPROF(24);
      if (!go) break;
      if (asm_ok) fs_stage_k(c, S, sk);
      int *w = c.I(c.ly.mwc);
      const int *tail = c.I(c.ly.hmap) + c.D.nv + 64;
      const int rsteps = __builtin_amdgcn_readfirstlane(tail[MAP_RSTEPS]) & 255, nbig = __builtin_amdgcn_readfirstlane(tail[MAP_NBIG]);
      if (nbig > 0 && rsteps > 0) { // the DPP rows on helper 1 beside the big island(s) here
        if (c.lane == 0) w[MWC_BAD] = 0;
        mw_post(c, MW_CHOL);
        const int bad = mw_chol_big(c, c.ly.hmap, am, asm_ok);
        mw_post(c, MW_IDLE);
        ok = !(bad | __builtin_amdgcn_readfirstlane(w[MWC_BAD]));
      } else {
        mw_post(c, MW_IDLE); // (helper 1 is still adding the body-pair entries)
        ok = fs_chol_solve(c, c.ly.hmap, am, asm_ok);
      }
      FS_SPROF(25);
      if (!ok) { if (c.lane == 0) scal[SC_BAD] |= 1; break; }
      if (c.lane == 0) { w[MWC_A0] = c.ly.Mp; w[MWC_A1] = c.ly.p; }
      mw_post(c, MW_MULM); // M p on helper 1 beside J p here
      fs_body_spatial(c, c.ly.p);
      fs_jdot(c, S, c.ly.p, false);
      mw_post(c, MW_IDLE);
    }
    if (!iterated) {
      sk = fs_gradient(c, S, T, skT);
      int am;
      // (welds and the LDS-resident factorisation of islands beyond the MFMA tile keep every island in the iteration)
      const int *tailh = c.I(c.ly.hmap) + c.D.nv + 64;
      const bool every = S.anyweld || (__builtin_amdgcn_readfirstlane(tailh[MAP_NBIG]) > 0 && __builtin_amdgcn_readfirstlane(tailh[MAP_MAXBIG]) > 31);
      const float gn = sqrtf(fs_active_islands(c, scale, every, &am));
      FS_SPROF(23);
      if (scale * gn < c.newton_tol) break;
      // (the LDS assembly serves the islands the matrix cores do not take: none at all when a robot island is all that still moves)
      const int asm_ok = FS_ASM_OK();
      if (asm_ok) fs_stage_k(c, S, sk);
      if (!asm_ok || (am & ~asm_ok)) fs_hessian(c, sk, S, am & ~asm_ok);
      if constexpr (Ctx::NS > 1) {
#pragma unroll
        for (int k = 0; k < Ctx::NS - 1; k++) fs_hessian<Ctx, true>(c, skT[k], T[k], am);
      }
      FS_SPROF(24);
      ok = fs_chol_solve(c, c.ly.hmap, am, asm_ok);
      FS_SPROF(25);
      if (!ok) { if (c.lane == 0) scal[SC_BAD] |= 1; break; }
      fs_mulM(c, c.ly.Mp, c.ly.p);
      fs_body_spatial(c, c.ly.p);
      fs_jdot(c, S, c.ly.p, false);
      if constexpr (Ctx::NS > 1) {
#pragma unroll
        for (int k = 0; k < Ctx::NS - 1; k++) fs_jdot(c, T[k], c.ly.p, false, false);
      }
    }
    // phi'(0) along the Newton direction (= -g' H^-1 g < 0), p'Mp and p'(Mx - smooth) in one pass over the dofs
    float dphi0 = 0, pMp = 0, pg0 = 0;
    for (int d = c.lane; d < c.D.nv; d += 64) {
      const float pd = L[c.ly.p + d];
      dphi0 += pd * L[c.ly.grad + d]; pMp += pd * L[c.ly.Mp + d]; pg0 += pd * (L[c.ly.Mx + d] - L[c.ly.smooth + d]);
    }
    dphi0 = wave_sum(dphi0); pMp = wave_sum(pMp); pg0 = wave_sum(pg0);
    FS_SPROF(26);
    // exact line search: safeguarded Newton on phi'(alpha)
    float lo = 0, hi = -1, alpha = 1, best = 0;
    bool nonquad = true; // the full step alpha = 1 stayed on one quadratic piece of the cost? (set by the first evaluation)
    int nls = 0;
    for (int ls = 0; ls < 20; ls++) {
      float d1, d2;
      bool nq;
      fs_line_eval(c, S, alpha, &d1, &d2, sk.zone, &nq, T, skT);
      if (ls == 0) nonquad = nq;
#ifdef FSIM_PROFILE
      if (c.lane == 0) { scal[16 + 13] += 1; }
#endif
      d1 += pg0 + alpha * pMp;
      d2 += pMp;
      best = alpha;
      nls = ls;
      if (fabsf(d1) <= FSIM_LS_TOL * fabsf(dphi0) + 1e-30f) break;
      if (d1 < 0) lo = alpha; else hi = alpha;
      float na = alpha - d1 / fmaxf(d2, 1e-30f);
      if (hi > 0) { if (na <= lo || na >= hi) na = 0.5f * (lo + hi); }
      else if (na <= lo) na = 2 * alpha;
      if (fabsf(na - alpha) < FSIM_LS_TOL * (1 + alpha)) break;
      alpha = na;
    }
    alpha = best;
#ifdef FSIM_PROFILE
    if (c.lane == 0) { scal[16 + 14] += 1; }
#endif
    FS_SPROF(27);
    for (int d = c.lane; d < c.D.nv; d += 64) { L[c.ly.x + d] += alpha * L[c.ly.p + d]; L[c.ly.Mx + d] += alpha * L[c.ly.Mp + d]; }
    for (int a = 0; a < 3; a++) S.jar[a] += alpha * S.jp[a];
    if constexpr (Ctx::NS > 1) {
#pragma unroll
      for (int k = 0; k < Ctx::NS - 1; k++) for (int a = 0; a < 3; a++) T[k].jar[a] += alpha * T[k].jp[a];
    }
    S.ljar += alpha * S.ljp;
    if (S.anyweld)
      for (int e = c.lane; e < c.D.neq; e += 64) {
        float *r = L + c.ly.weld + FSIM_WELDW * e;
        if (reinterpret_cast<int *>(r)[WD_ACTIVE]) for (int q = 0; q < 6; q++) r[WD_JAR + q] += alpha * r[WD_JP + q];
      }
    SYNC();
    // MuJoCo's second stopping rule: scaled cost improvement of the step below tolerance.  The improvement is taken
    // from the line-search model, phi(0) - phi(alpha) = -1/2 alpha phi'(0) at an exact minimiser of a (piecewise)
    // quadratic, instead of differencing two O(1e3) costs: in fp32 that difference is noise at the 1e-6 level and
    // kept robot-contact envs iterating (5.5 iterations/substep where the fp64 oracle needs 2.5).
    FS_SPROF(28);
    float improvement = scale * 0.5f * alpha * fmaxf(-dphi0, 0.0f);
    if (improvement < c.newton_tol) { it++; break; }
    // The full Newton step was accepted at the first trial and every constraint stayed in the quadratic piece it was in when
    // H was assembled: x is the exact minimiser of that piece, the new gradient is zero up to rounding.  Evaluating it (a
    // full J'f pass, a quarter of an uncoupled env's solve) could only confirm that -- the iteration ends here.
    if (!nonquad && nls == 0 && alpha == 1.0f) { it++; break; }
  }
  if (c.lane == 0) { scal[SC_NITER] = it; scal[SC_NITSUM] += it; }
  SYNC();
}

// ------------------------------------------------------------------------------------------ P8
// qacc in c.ly.x, M*qacc in c.ly.Mx (valid).  Semi-implicit Euler with implicit joint damping.
template <class Ctx> DEV void fs_integrate_body(const Ctx &c) {
  CModel &m = c.m;
  float *L = c.L;
  float h = c.D.timestep;
  // (M + h D) a' = M a : the same lane-per-row block Cholesky as the Newton step, on H = M + h diag(damping)
  int nH = c.I(c.ly.scal)[SC_TWORDS];
  for (int d = c.lane; d < c.D.nv; d += 64) { L[c.ly.qaccws + d] = L[c.ly.x + d]; L[c.ly.grad + d] = -L[c.ly.Mx + d]; }
  for (int k = c.lane; k < nH; k += 64) L[c.ly.H + k] = L[c.ly.M + k]; // same tree-packed layout
  SYNC();
  for (int d = c.lane; d < c.D.nv; d += 64) L[c.ly.H + fs_hidx(c, c.ly.k_tmap, d, d)] += h * KF(dof_damping, d);
  SYNC();
  fs_chol_solve(c, c.ly.k_tmap);
  for (int d = c.lane; d < c.D.nv; d += 64) L[c.ly.qvel + d] += h * L[c.ly.p + d];
  SYNC();
  for (int b = c.lane; b < c.D.nr; b += 64) {
    if (b == 0) continue;
    int jt = KI(r_jtype, b), qa = KI(r_qposadr, b), d = KI(r_dofadr, b);
    if (jt == JT_FREE) {
      for (int k = 0; k < 3; k++) L[c.ly.qpos + qa + k] += h * L[c.ly.qvel + d + k];
      V3 w = ldv3(L + c.ly.qvel + d + 3);
      float wn;
      V3 ax = normalized(w, &wn);
      float ang = wn * h;
      if (ang > 0) {
        Q4 q = qnormalized(qmul(ldq(L + c.ly.qpos + qa + 3), axisangle(ax, ang)));
        stq(L + c.ly.qpos + qa + 3, q);
      }
    } else
      L[c.ly.qpos + qa] += h * L[c.ly.qvel + d];
  }
  SYNC();
}
