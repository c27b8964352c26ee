// fsim_physics.hpp -- one physics substep (mj_step analogue) for ONE env executed by ONE wavefront.
//
// Mapping to CDNA4: a 64-lane wave owns an env; its state record and all intermediates live in
// LDS; lanes parallelise over bodies / dofs / mass-matrix entries / geoms / candidate pairs /
// constraint slots, with __syncthreads() (single-wave workgroup => nearly free) between stages.
// Stage list follows SURVEY.md section 8(a) P1-P9 (the work the reference reaches through
// sim.forward()/sim.step(), furniture/env/furniture.py:2877-2879).
#pragma once
#include "fsim_math.hpp"
#include "fsim_model.hpp"

// Ordering point between the lanes working on one env.
//   one-wave kernels (NW == 1): workgroup = wave = env, SYNC() is __syncthreads() (a compiler-only barrier -- a wave's LDS
//     instructions execute in issue order -- passes every test but measures the same, so the real barrier stays);
//   multi-wave kernels (NW > 1): a workgroup of NW waves steps ONE env.  Wave 0 ("main") runs the same code as the one-wave
//     kernel and hands whole passes to the helper waves at fork points (fsim_solver.hpp, "multi-wave"); inside a pass a single
//     wave works, so SYNC() is a compiler barrier only (the hardware keeps one wave's LDS operations in order) and the waves meet
//     at explicit workgroup barriers, c.xbar().
#define SYNC() (c.sync())

// The per-wave context.  Two flavours with the same member names, so the physics / solver / env code is written once as
// templates over the context type:
//   GenCtx  -- layout offsets and model dimensions are run-time values (scalar loads from constant memory): any model;
//   SpecCtx -- they are compile-time constants of the specialisation S (fsim_spec.hpp): offsets become instruction
//              immediates, `for (i = lane; i < n; i += 64)` loops become a single test, no SGPRs are spent on the layout.
// Both take the number of waves per env as a template parameter (NW; lane = lane inside the wave, wave = which wave).
// BUNDLE: four one-wave envs share a workgroup (k_env_step_x: one launch holds the multi-wave workgroups and, behind them in
// dispatch order, bundles of one-wave envs, so both kinds are 4-wave workgroups and the long jobs are placed first).  A bundled
// wave is on its own: SYNC() must not be a workgroup barrier, and its LDS image starts at wave * FSIM_BUNDLE_STRIDE(lds_words).
#define FSIM_BUNDLE_STRIDE(words) (((words) + 3) / 4 * 4)
template <int NW_, bool BUNDLE_> struct FsSync {
  DEV static void sync() { if (NW_ == 1 && !BUNDLE_) __syncthreads(); else __asm__ volatile("" ::: "memory"); }
  DEV static void xbar() { __syncthreads(); }
};
template <int NW_, bool BUNDLE_ = false, int NS_ = 1> struct GenCtxT {
  static constexpr bool PLAIN = false; // (the generic kernels serve every configuration: controllers, IK, dense reward)
  static constexpr int NW = NW_;
  static constexpr bool BUNDLE = BUNDLE_;
  static constexpr int NS = NS_; // contact-slot sets of 64 the Newton solve carries per lane (2: models with more than 64 contact slots)
  float *L;          // LDS base (state image followed by work arrays)
  CModel &m;         // model tables (pointers)
  CLayout &ly;       // LDS / record layout
  CModel &D;         // dimensions + scalar options (the Dims base of the same struct)
  int lane, wave;
  int newton_maxit;
  float newton_tol;
  // lds_: the workgroup's dynamic LDS; tid: threadIdx.x
  __device__ GenCtxT(float *lds_, CModel &m_, CLayout &ly_, int tid, int it, float tol)
      : L(lds_), m(m_), ly(ly_), D(m_), lane(tid & 63), wave((NW_ > 1 || BUNDLE_) ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0), newton_maxit(it), newton_tol(tol) {
    if (BUNDLE_) L = lds_ + wave * FSIM_BUNDLE_STRIDE(ly_.lds_words);
  }
  DEV int *I(int off) const { return reinterpret_cast<int *>(L + off); }
  DEV float *wg_lds() const { return BUNDLE_ ? L - wave * FSIM_BUNDLE_STRIDE(ly.lds_words) : L; } // the workgroup's LDS base (what fs_rebuild takes)
  DEV void sync() const { FsSync<NW_, BUNDLE_>::sync(); }
  DEV void xbar() const { FsSync<NW_, BUNDLE_>::xbar(); }
};
typedef GenCtxT<1> GenCtx;
template <class S, int NW_> struct FsSpecLayout { static constexpr Layout ly = make_layout(S::in, NW_); };
template <class S, int NW_ = 1, bool BUNDLE_ = false> struct SpecCtx {
  // the env record of a specialised kernel has no controller / IK / dense-reward block (its LayoutIn says so, and pick_kernels only
  // takes the kernel for a configuration with the same LayoutIn): EnvCfg::controller / ik / dense are compile-time zeros here, and
  // the code behind them -- env_ik, the controller instantiation of fs_substeps_t, the dense reward -- is not in these kernels
  static constexpr bool PLAIN = S::plain;
  static constexpr int NW = NW_;
  static constexpr bool BUNDLE = BUNDLE_;
  float *L;
  CModel &m;
  static constexpr Layout ly = FsSpecLayout<S, NW_>::ly;
  static constexpr Dims D = S::D;
  static constexpr int NS = FsSpecLayout<S, NW_>::ly.ncon_max > 64 ? 2 : 1;
  int lane, wave;
  int newton_maxit;
  float newton_tol;
  __device__ SpecCtx(float *lds_, CModel &m_, CLayout &, int tid, int it, float tol)
      : L(lds_), m(m_), lane(tid & 63), wave((NW_ > 1 || BUNDLE_) ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0), newton_maxit(it), newton_tol(tol) {
    if (BUNDLE_) L = lds_ + wave * FSIM_BUNDLE_STRIDE(ly.lds_words);
  }
  __device__ SpecCtx(float *lds_, CModel &m_, int tid, int it, float tol)
      : L(lds_), m(m_), lane(tid & 63), wave((NW_ > 1 || BUNDLE_) ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0), newton_maxit(it), newton_tol(tol) {
    if (BUNDLE_) L = lds_ + wave * FSIM_BUNDLE_STRIDE(ly.lds_words);
  }
  DEV int *I(int off) const { return reinterpret_cast<int *>(L + off); }
  DEV float *wg_lds() const { return BUNDLE_ ? L - wave * FSIM_BUNDLE_STRIDE(ly.lds_words) : L; } // the workgroup's LDS base (what fs_rebuild takes)
  DEV void sync() const { FsSync<NW_, BUNDLE_>::sync(); }
  DEV void xbar() const { FsSync<NW_, BUNDLE_>::xbar(); }
};

// Re-derive wave-uniform values after a real function call so the callee's model-table loads stay scalar.
template <class T> DEV T *fs_uniform_ptr(T *p) {
  unsigned long long v = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
// (lds: the workgroup's dynamic LDS base -- a bundled wave's image offset is applied by the constructor)
template <int NW_, bool B_, int NS_> DEV GenCtxT<NW_, B_, NS_> fs_rebuild(const GenCtxT<NW_, B_, NS_> &cv, float *lds) {
  return GenCtxT<NW_, B_, NS_>(lds, *fs_uniform_ptr(&cv.m), *fs_uniform_ptr(&cv.ly), (int)threadIdx.x, __builtin_amdgcn_readfirstlane(cv.newton_maxit),
                          __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cv.newton_tol))));
}
template <class S, int NW_, bool B_> DEV SpecCtx<S, NW_, B_> fs_rebuild(const SpecCtx<S, NW_, B_> &cv, float *lds) {
  CModel &mu = *fs_uniform_ptr(&cv.m);
  return SpecCtx<S, NW_, B_>(lds, mu, (int)threadIdx.x, __builtin_amdgcn_readfirstlane(cv.newton_maxit),
                             __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cv.newton_tol))));
}
#define FS_REBUILD_CTX(cv)            \
  extern __shared__ float fs_lds_[];  \
  const Ctx c = fs_rebuild(cv, fs_lds_)
// The same context under another TYPE, for the code under env_reset: the out-of-line physics routine that code calls is then an
// instantiation of its own, and the one the step calls is called from KERNELS only.  A function whose callers are all kernels
// keeps no callee-saved registers (nothing above it has live registers); called from env_reset as well, fs_substeps_t saved 113
// VGPRs per lane to scratch on every call -- 29 KB per wave, three calls per env-step: the 72 KB of HBM writes per env-step the
// counters showed (profiles/r03_e_pmc_counters.txt).
template <class B> struct FsIn : B {
  DEV explicit FsIn(const B &b) : B(b) {}
};
template <class B> DEV FsIn<B> fs_rebuild(const FsIn<B> &cv, float *lds) { return FsIn<B>(fs_rebuild(static_cast<const B &>(cv), lds)); }


// Walk the set bits of a mask three per trip: i0 is valid, i1 / i2 fall back to i0 with their flag cleared.  A loop with a
// run-time trip count pays one dependent LDS round trip per iteration (a wave has no other work to hide it); three independent
// element loads per trip cut a 9-element walk from 9 round trips to 3.
#define FS_BITS3(mm, i0, i1, i2, h1, h2)             \
  const int i0 = __ffs(mm) - 1; mm &= mm - 1;        \
  const bool h1 = mm != 0;                           \
  const int i1 = h1 ? __ffs(mm) - 1 : i0; mm &= mm - 1; \
  const bool h2 = mm != 0;                           \
  const int i2 = h2 ? __ffs(mm) - 1 : i0; mm &= mm - 1

#define KI(field, idx) (c.I(c.ly.k_##field)[idx])
#define KM_I(e) ((c.I(c.ly.k_M_ij)[e] >> 8) & 255)
#define KM_J(e) (c.I(c.ly.k_M_ij)[e] & 255)
#define KM_P(e) (c.I(c.ly.k_M_ij)[e] >> 16)
#define KF(field, idx) (c.L[c.ly.k_##field + (idx)])
#define KFP(field) (c.L + c.ly.k_##field)

// ---- map of a block-diagonal SPD system over "islands" (sets of kinematic trees): Layout::hmap (Newton Hessian, islands
// = trees joined by an active constraint, rebuilt when the adjacency changes) and Layout::k_tmap (M + h D of the
// integrator, islands = trees, built once per launch).  Words, relative to `mp`:
//   [0, nv)        per dof:  row base in the packed triangles (13 bits) | local index l in its island (7) << 13 |
//                            island size nI (7) << 20            (MAP_ROWB / MAP_L / MAP_NI; round 6: l had six bits -- an island of
//                            more than 64 dofs, which only the 512-slot re-step kernels accept, needs seven -- and the row base twelve)
//   [nv, nv + 64)  per lane: byte 0 = dof this lane owns in the ROW phase (0xff none), byte 1 = dof it owns in the BIG phase,
//                            byte 2 = dof it owns in the SECOND row pass (models with more than 64 dofs: the four 16-lane rows are
//                            filled twice)
//   [nv + 64, +16) tail:     [0] row-phase steps (largest row fill) of pass 0 | pass 1 << 8, [1] number of big islands,
//                            [2 + 2b], [3 + 2b] = first lane and size of big island b (b < 6), [14] largest big island
// Islands of <= 16 dofs are packed into the four 16-lane DPP rows of the wave (several islands may share a row: they are
// factored as one block-diagonal matrix); the factorisation then needs no LDS traffic at all -- the pivot row travels by
// `row_newbcast` DPP moves.  Larger islands (a robot holding two parts, Baxter's 19-dof tree) get a contiguous lane range
// and are factored one after the other with v_readlane broadcasts.  (fs_chol_solve, fsim_solver.hpp)
enum { MAP_RSTEPS = 0, MAP_NBIG = 1, MAP_BIG0 = 2, MAP_MAXBIG = 14, MAP_BIGCAP = 6 };
#define MAP_ROWB(w) ((w) & 0x1fff) // (13 bits: the packed triangle of 93 dofs -- Sawyer + bookcase_grevback_0484 -- has 4371 words)
#define MAP_L(w) (((w) >> 13) & 127)
#define MAP_NI(w) (((w) >> 20) & 127)
#define MAP_HUGE 200 // tail[MAP_MAXBIG] of a system with an island of more than 64 dofs (or big islands beyond the 64 big-phase lanes): no lane tables --
                     // fs_chol_all_lds factors every island from the dof words alone (contexts with Ctx::NS >= 4 only: the last rung of the re-step ladder)
// (FSIM_BIG_MIN: the island of the robot's tree counts as "large" from this many dofs on.  17 = only when it does not fit a DPP row;
//  12 with the matrix-core Hessian assembly, -DFSIM_MFMA_HESSIAN, whose tile then also takes the robot + one part islands)
//  -DFSIM_MFMA_HESSIAN=2: only in the multi-wave kernels, whose main wave is the critical path of the slowest envs)
#ifdef FSIM_BIG_MIN
#define FSIM_BIG_MIN_OF(Ctx) FSIM_BIG_MIN
#elif !defined(FSIM_MFMA_HESSIAN)
#define FSIM_BIG_MIN_OF(Ctx) 17
#elif FSIM_MFMA_HESSIAN == 2
#define FSIM_BIG_MIN_OF(Ctx) (Ctx::NW > 1 ? 12 : 17)
#else
#define FSIM_BIG_MIN_OF(Ctx) 12
#endif
template <class Ctx> DEV void fs_build_map(const Ctx &c, int mp, const int *isl, int hwords_slot) {
  const int nv = c.D.nv, ntree = c.D.ntree;
  int *scal_ = c.I(c.ly.scal);
  int *tmp = scal_ + SC_TMP; // [t] = local offset of tree t in its island | island size << 8 | (rep only: first lane << 16 | big << 24); [16 + t] = island H base
  int *hm = c.I(mp);
  int *tail = hm + nv + 64;
  if (c.lane < ntree) {
    int t = c.lane, my = isl[t], loc = 0, nI = 0;
    for (int u = 0; u < ntree; u++) {
      int nu = KI(tree_dofnum, u);
      if ((my >> u) & 1) { nI += nu; if (u < t) loc += nu; }
    }
    tmp[t] = loc | (nI << 8);
  }
  SYNC();
  if (c.lane == 0) {
    // islands in the order of their lowest tree: H base, then a lane range.  Small islands go to the least-filled row that
    // still has room (keeps the row phase short: 9 | 6+6 | 6+6 | 6 for a free Sawyer + table_lack), the others (and any small
    // island that no row can take) become "big".
    // (a model with more than 64 dofs fills the four rows a second time: second row pass of fs_chol_solve)
    const int nrow = nv > 64 ? 8 : 4;
    int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hb = 0, sb = 0, nbig = 0, maxbig = 0;
    bool huge = false;
    for (int u = 0; u < ntree; u++) {
      if (__ffs(isl[u]) - 1 != u) continue;
      const int n = (tmp[u] >> 8) & 255;
      tmp[16 + u] = hb;
      hb += n * (n + 1) / 2;
      int row = -1;
      // (round 4: an island of 12..16 dofs around the robot -- tree 0: the arm and the part it touches -- is a BIG island too: its
      //  Hessian is then assembled on the matrix cores, straight into the tile that is factored (fs_newton_mfma), instead of through
      //  the body blocks in LDS and the row phase -- opt-in build, see FSIM_BIG_MIN_OF)
      if (n <= 16 && !(u == 0 && n >= FSIM_BIG_MIN_OF(Ctx))) {
        for (int r = 0; r < 4; r++) if (fill[r] + n <= 16 && (row < 0 || fill[r] < fill[row])) row = r;
        if (row < 0) for (int r = 4; r < nrow; r++) if (fill[r] + n <= 16 && (row < 0 || fill[r] < fill[row])) row = r;
      }
      int lane0;
      if (row >= 0) { lane0 = 16 * (row & 3) + fill[row]; fill[row] += n; }
      else { lane0 = sb; if (nbig < MAP_BIGCAP) { tail[MAP_BIG0 + 2 * nbig] = sb; tail[MAP_BIG0 + 2 * nbig + 1] = n; } nbig++; sb += n; maxbig = max(maxbig, n); }
      tmp[u] |= (lane0 << 16) | ((row < 0 ? 1 : 0) << 24) | ((row >= 4 ? 1 : 0) << 25);
      // an island of more than 64 dofs (or big islands that overflow the 64 big-phase lanes) cannot be mapped onto lanes: the solve is
      // flagged bad, which the env treats like an unstable simulation -- a robot holding ten mutually coupled parts, eleven planks in one pile.
      // The 512-slot kernels of the re-step ladder, Ctx::NS >= 4, take such a system through the LDS-resident factorisation instead (`huge`), and
      // every other kernel ALSO raises the capacity report (SC_OVERFLOW bit 1: "this kernel's layout does not hold this step"), which lists the
      // env for the ladder: its step or reset is repeated from the kept record, rung by rung, and the last rung solves it.
      if (n > 64 || sb > 64) { if (Ctx::NS >= 4 && n <= 127) huge = true; else { scal_[SC_BAD] |= 3; if (Ctx::NS < 4) scal_[SC_OVERFLOW] |= 2; } }
    }
    tail[MAP_RSTEPS] = max(max(fill[0], fill[1]), max(fill[2], fill[3])) | (max(max(fill[4], fill[5]), max(fill[6], fill[7])) << 8);
    tail[MAP_NBIG] = nbig;
    tail[MAP_MAXBIG] = huge ? MAP_HUGE : (nbig > MAP_BIGCAP ? 99 : maxbig); // (more big islands than the table holds: LDS fallback handles them all)
    scal_[hwords_slot] = hb;
  }
  SYNC();
  hm[nv + c.lane] = 0xffffff;
  SYNC();
  for (int i = c.lane; i < nv; i += 64) {
    const int t = KI(r_tree, KI(dof_rbody, i)), rep = __ffs(isl[t]) - 1;
    const int nI = (tmp[rep] >> 8) & 255, l = (tmp[t] & 255) + i - KI(tree_dofadr, t), lane = ((tmp[rep] >> 16) & 255) + l;
    hm[i] = (tmp[16 + rep] + l * (l + 1) / 2) | (l << 13) | (nI << 20);
    const bool big = (tmp[rep] >> 24) & 1, second = (tmp[rep] >> 25) & 1;
    if (lane < 64) reinterpret_cast<unsigned char *>(hm + nv)[4 * lane + (big ? 1 : (second ? 2 : 0))] = (unsigned char)i;
  }
  SYNC();
  if constexpr (Ctx::NS >= 4) {
    // huge system (fs_chol_all_lds): the lane tables mean nothing; [nv + t] = COMPACT offset of tree t's island -- the islands' dofs numbered
    // island by island in the order of their lowest tree, so that (offset + local index) addresses a vector by island position
    if (tail[MAP_MAXBIG] == MAP_HUGE) {
      SYNC();
      if (c.lane < ntree) {
        const int rep = __ffs(isl[c.lane]) - 1;
        int off = 0;
        for (int u = 0; u < rep; u++) if (__ffs(isl[u]) - 1 == u) off += (tmp[u] >> 8) & 255;
        hm[nv + c.lane] = off;
      }
      SYNC();
    }
  }
}

// copy the hot model tables HBM -> LDS (once per kernel launch; the 50 substeps then never leave the CU for them)
template <class Ctx> DEV void fs_load_cache(const Ctx &c) {
  CModel &m = c.m;
  int nb = c.D.nr, nv = c.D.nv, nchain = 0;
  for (int b = 0; b < nb; b++) nchain = max(nchain, GP(m.r_chainadr)[b] + GP(m.r_chainlen)[b]);
#define CPI(field, n) for (int i_ = c.lane; i_ < (n); i_ += 64) c.I(c.ly.k_##field)[i_] = m.field[i_]
#define CPF(field, n) for (int i_ = c.lane; i_ < (n); i_ += 64) c.L[c.ly.k_##field + i_] = m.field[i_]
  CPI(dof_tree, nv); CPI(dof_rbody, nv);
  CPI(r_parent, nb); CPI(r_jtype, nb); CPI(r_qposadr, nb); CPI(r_dofadr, nb); CPI(r_tree, nb);
  CPI(r_chainadr, nb); CPI(r_chainlen, nb); CPI(chain_dofs, nchain);
  CPI(tree_dofadr, c.D.ntree); CPI(tree_dofnum, c.D.ntree); CPI(tree_bodyadr, c.D.ntree); CPI(tree_bodynum, c.D.ntree);
  // (body-frame constants r_pos/r_quat/r_jpos/r_jaxis/r_ipos/r_inertia are read once per substep by lane-per-body
  //  passes: they stay in HBM/L2 and their 330 words of LDS buy an extra workgroup per CU instead)
  CPF(r_mass, nb); CPF(dof_damping, nv); CPF(dof_armature, nv);
#undef CPI
#undef CPF
  // bit tables that replace index-list walks (an index load feeding a data load is two dependent LDS round trips per
  // element): r_submask[b] = bodies in b's subtree; r_chain[b] = (first dof of b's tree) << 25 | bitmask of the dofs on
  // the path root -> b, relative to that first dof (<= 25 dofs per tree and <= 128 dofs in all, checked by fsim_create.  Round 4:
  // the shift was 26, which left six bits for the first dof -- in a model with more than 64 dofs the parts whose dofs start at 64
  // or later took their J v from other bodies' dofs: bookcase_grevback_0484's part 10, tests/test_all_furniture_gpu.py)
  for (int b = c.lane; b < nb; b += 64) {
    int sub = 0, ch = 0, base = b > 0 ? GP(m.tree_dofadr)[GP(m.r_tree)[b]] : 0;
    for (int d = b; d < nb; d++) if (b > 0 && ((GP(m.r_ancmask)[d] >> b) & 1)) sub |= 1 << d;
    if (b > 0) for (int k = 0; k < GP(m.r_chainlen)[b]; k++) ch |= 1 << (GP(m.chain_dofs)[GP(m.r_chainadr)[b] + k] - base);
    c.I(c.ly.k_r_submask)[b] = sub;
    c.I(c.ly.k_r_chain)[b] = (int)(((unsigned)base << 25) | (unsigned)ch);
  }
  SYNC();
  // static "tree map" of the block-diagonal-by-tree system M + h*D of fs_integrate (islands = trees): same format as the
  // per-substep island map of the Newton system; also fixes the tree-packed triangle layout of M itself
  if (c.lane < c.D.ntree) c.I(c.ly.scal)[SC_ISL + c.lane] = 1 << c.lane;
  SYNC();
  fs_build_map(c, c.ly.k_tmap, c.I(c.ly.scal) + SC_ISL, SC_TWORDS);
  // M entry e -> (i, j, packed index in the tree-packed triangle); the entries of M that are structurally zero
  // (two branches of one tree) are zeroed once here and never written again
  for (int e = c.lane; e < c.D.nM; e += 64) {
    int i = GP(m.M_i)[e], j = GP(m.M_j)[e];
    int pidx = MAP_ROWB(c.I(c.ly.k_tmap)[i]) + MAP_L(c.I(c.ly.k_tmap)[j]);
    c.I(c.ly.k_M_ij)[e] = (pidx << 16) | (i << 8) | j;
  }
  {
    int w = 0;
    for (int u = 0; u < c.D.ntree; u++) w += GP(m.tree_dofnum)[u] * (GP(m.tree_dofnum)[u] + 1) / 2;
    for (int k = c.lane; k < w; k += 64) c.L[c.ly.M + k] = 0.0f;
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------ P1
template <class Ctx> DEV void fs_kinematics(const Ctx &c) {
  // lane = body from start to end, poses in registers.  (A) joint transform relative to the parent frame; (B) world poses by
  // pointer doubling up the kinematic tree: every body holds its pose relative to an ancestor and in each round composes it with
  // that ancestor's own record and jumps to the ancestor's ancestor -- ceil(log2(depth)) rounds of (one LDS write, one LDS read)
  // instead of a serial walk down the 8-deep Sawyer chain (which was 9 x three dependent LDS round trips on one lane while 63
  // idled: half of this stage); (C) rotation matrix, joint anchor / axis and inertial-frame origin in world coordinates.
  CModel &m = c.m;
  float *L = c.L;
  int *ancs = c.I(c.ly.surv); // (the broadphase survivor list: dead until fs_collide)
  const int b = c.lane;
  const bool on = b < c.D.nr && b > 0;
  const int bb = on ? b : 0;
  // every load this stage needs from global memory, issued first (one round trip)
  const Q4 q0 = ldq(GP(m.r_quat) + 4 * bb);
  const V3 p0 = ldv3(GP(m.r_pos) + 3 * bb), jpos = ldv3(GP(m.r_jpos) + 3 * bb), jax = ldv3(GP(m.r_jaxis) + 3 * bb), ipos = ldv3(GP(m.r_ipos) + 3 * bb);
  const int jt = KI(r_jtype, bb), qa = KI(r_qposadr, bb), parent = KI(r_parent, bb);
  V3 P = v3(0, 0, 0), al = v3(0, 0, 0), axl = v3(0, 0, 1);
  Q4 Q = q4(1, 0, 0, 0);
  if (on) {
    if (jt == JT_FREE) {
      const float *q = L + c.ly.qpos + qa;
      P = ldv3(q);
      Q = qnormalized(ldq(q + 3));
      stq(L + c.ly.qpos + qa + 3, Q); // MuJoCo normalises the stored quaternion in place
      al = P;
    } else {
      al = p0 + qrot(q0, jpos);
      axl = qrot(q0, jax);
      const float q = L[c.ly.qpos + qa]; // joint reference positions are zero in every in-scope model (checked by the compiler)
      if (jt == JT_SLIDE) { Q = q0; P = p0 + axl * q; }
      else { Q = qmul(q0, axisangle(jax, q)); P = al - qrot(Q, jpos); }
    }
  }
  int anc = on ? parent : 0;
  for (int span = 1; span < c.D.maxdepth; span <<= 1) {
    if (b < c.D.nr) { stv3(L + c.ly.xpos + 3 * b, P); stq(L + c.ly.xquat + 4 * b, Q); ancs[b] = anc; }
    SYNC();
    if (anc > 0) {
      const V3 pa = ldv3(L + c.ly.xpos + 3 * anc);
      const Q4 qa_ = ldq(L + c.ly.xquat + 4 * anc);
      const int na = ancs[anc];
      P = pa + qrot(qa_, P);
      Q = qnormalized(qmul(qa_, Q));
      anc = na;
    }
    SYNC();
  }
  if (b < c.D.nr) { stv3(L + c.ly.xpos + 3 * b, P); stq(L + c.ly.xquat + 4 * b, Q); }
  SYNC();
  if (b < c.D.nr) {
    const M3 R = q2m(Q);
    stm3(L + c.ly.xmat + 9 * b, R);
    stv3(L + c.ly.xipos + 3 * b, P + mulv(R, ipos));
    if (b > 0) { // joint anchor / axis are in the parent frame
      const V3 pp = ldv3(L + c.ly.xpos + 3 * parent);
      const Q4 pq = ldq(L + c.ly.xquat + 4 * parent);
      stv3(L + c.ly.xanchor + 3 * b, pp + qrot(pq, al));
      stv3(L + c.ly.xaxis + 3 * b, qrot(pq, axl));
    }
  }
  SYNC();
}

// per-tree centre of mass, body inertias about it, motion axes (mj_comPos)
template <class Ctx> DEV void fs_com_inertia(const Ctx &c) {
  CModel &m = c.m;
  float *L = c.L;
  for (int t = c.lane; t < c.D.ntree; t += 64) {
    V3 s = v3(0, 0, 0);
    float tot = 0;
    for (int b = KI(tree_bodyadr, t); b < KI(tree_bodyadr, t) + KI(tree_bodynum, t); b++) {
      float ms = KF(r_mass, b);
      s = s + ms * ldv3(L + c.ly.xipos + 3 * b);
      tot += ms;
    }
    stv3(L + c.ly.com + 3 * t, tot > 0 ? s * (1.0f / tot) : ldv3(L + c.ly.xpos + 3 * KI(tree_bodyadr, t)));
  }
  SYNC();
  for (int b = c.lane; b < c.D.nr; b += 64) {
    float *I = L + c.ly.cinert + 10 * b;
    if (b == 0) { for (int k = 0; k < 10; k++) I[k] = 0; continue; }
    M3 R = ldm3(L + c.ly.xmat + 9 * b);
    auto ib = GP(m.r_inertia) + 6 * b; // xx yy zz xy xz yz in body frame
    M3 Ib;
    Ib.m[0] = ib[0]; Ib.m[4] = ib[1]; Ib.m[8] = ib[2]; Ib.m[1] = Ib.m[3] = ib[3]; Ib.m[2] = Ib.m[6] = ib[4]; Ib.m[5] = Ib.m[7] = ib[5];
    M3 T = mulm(R, Ib);
    // Iw = T * R^T
    float w[6];
    const int ij[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
    for (int e = 0; e < 6; e++) {
      int i = ij[e][0], j = ij[e][1];
      w[e] = T.m[3 * i] * R.m[3 * j] + T.m[3 * i + 1] * R.m[3 * j + 1] + T.m[3 * i + 2] * R.m[3 * j + 2];
    }
    float ms = KF(r_mass, b);
    V3 d = ldv3(L + c.ly.xipos + 3 * b) - ldv3(L + c.ly.com + 3 * KI(r_tree, b));
    float dd = dot(d, d);
    I[0] = w[0] + ms * (dd - d.x * d.x); I[1] = w[1] + ms * (dd - d.y * d.y); I[2] = w[2] + ms * (dd - d.z * d.z);
    I[3] = w[3] - ms * d.x * d.y; I[4] = w[4] - ms * d.x * d.z; I[5] = w[5] - ms * d.y * d.z;
    I[6] = ms * d.x; I[7] = ms * d.y; I[8] = ms * d.z; I[9] = ms;
  }
  for (int d = c.lane; d < c.D.nv; d += 64) {
    int b = KI(dof_rbody, d), jt = KI(r_jtype, b), k = d - KI(r_dofadr, b);
    V3 com = ldv3(L + c.ly.com + 3 * KI(r_tree, b));
    S6 s;
    if (jt == JT_FREE) {
      if (k < 3) { s.a = v3(0, 0, 0); s.l = v3(k == 0, k == 1, k == 2); }
      else {
        M3 R = ldm3(L + c.ly.xmat + 9 * b);
        s.a = colv(R, k - 3);
        s.l = cross(s.a, com - ldv3(L + c.ly.xpos + 3 * b));
      }
    } else if (jt == JT_SLIDE) { s.a = v3(0, 0, 0); s.l = ldv3(L + c.ly.xaxis + 3 * b); }
    else { s.a = ldv3(L + c.ly.xaxis + 3 * b); s.l = cross(s.a, com - ldv3(L + c.ly.xanchor + 3 * b)); }
    sts6(L + c.ly.cdof + 6 * d, s);
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------ P2
template <class Ctx> DEV void fs_crb_factor(const Ctx &c) {
  CModel &m = c.m;
  float *L = c.L;
  // composite inertia: sum over descendants (no serial tree pass: each lane scans the body list)
  for (int b = c.lane; b < c.D.nr; b += 64) {
    float acc[10];
    for (int k = 0; k < 10; k++) acc[k] = 0;
    for (int mm = KI(r_submask, b); mm;) {
      FS_BITS3(mm, b0, b1, b2, h1, h2);
      const float *I0 = L + c.ly.cinert + 10 * b0, *I1 = L + c.ly.cinert + 10 * b1, *I2 = L + c.ly.cinert + 10 * b2;
#pragma unroll
      for (int k = 0; k < 10; k++) { const float x0 = I0[k], x1 = I1[k], x2 = I2[k]; acc[k] += x0 + (h1 ? x1 : 0.0f) + (h2 ? x2 : 0.0f); }
    }
    for (int k = 0; k < 10; k++) L[c.ly.crb + 10 * b + k] = acc[k];
  }
  SYNC();
  for (int e = c.lane; e < c.D.nM; e += 64) {
    int i = KM_I(e), j = KM_J(e);
    S6 f = inert_mul(L + c.ly.crb + 10 * KI(dof_rbody, i), lds6(L + c.ly.cdof + 6 * i));
    float v = dot6(lds6(L + c.ly.cdof + 6 * j), f);
    if (i == j) v += KF(dof_armature, i);
    L[c.ly.M + KM_P(e)] = v;
  }
  SYNC();
}

// y = M v.  M is a dense packed lower triangle per kinematic tree (layout of k_tmap), so lane = dof gathers its row
// with computed addresses: no index loads, no atomics, one barrier.
template <class Ctx> DEV void fs_mulM(const Ctx &c, int off_y, int off_v) {
  CModel &m = c.m;
  float *L = c.L;
  for (int i = c.lane; i < c.D.nv; i += 64) {
    const int w = c.I(c.ly.k_tmap)[i];
    const int li = MAP_L(w), n = MAP_NI(w), a = i - li; // local index, tree size, first dof
    const int rowi = MAP_ROWB(w), tb = rowi - li * (li + 1) / 2;
    const float *Mt = L + c.ly.M;
    float acc = 0;
    // one walk over the row (left of the diagonal: the packed row itself, right of it: column li of the later rows), three
    // independent entries per trip
    for (int lj0 = 0; lj0 < n; lj0 += 3) {
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int lj = lj0 + u;
        const bool ok = lj < n;
        const int ljc = ok ? lj : li;
        const int adr = ljc <= li ? rowi + ljc : tb + ljc * (ljc + 1) / 2 + li;
        const float mv = Mt[adr] * L[off_v + a + ljc];
        acc += ok ? mv : 0.0f;
      }
    }
    L[off_y + i] = acc;
  }
  SYNC();
}

// ------------------------------------------------------------------------------------------ P5/P6
template <class Ctx> DEV void fs_velocity_bias(const Ctx &c) {
  // Body velocities and RNE bias accelerations without walking the chains: all spatial vectors of a tree share one reference
  // point (its CoM), so  v_b = sum_{i in chain(b)} u_i  with u_i = (joint i's motion axes) * qvel, and the velocity-product
  // acceleration  a_b + g = sum_i cdof_dot_i qvel_i = sum_{j before i} u_j x u_i  (x = spatial motion cross product, bilinear).
  // Segments of a chain therefore combine as (U, C) o (U', C') = (U + U', C + C' + U x U'), an associative rule: lane = body
  // and ceil(log2(depth)) pointer-doubling rounds replace the per-dof pointer chase up the chain plus the per-body walk down it.
  CModel &m = c.m;
  float *L = c.L;
  int *ancs = c.I(Ctx::NW > 1 ? c.ly.anc2 : c.ly.surv); // (broadphase survivor list: dead here -- unless a helper wave runs fs_collide beside this pass)
  float *SU = L + c.ly.cacc, *SC = L + c.ly.cfrc; // round scratch: cacc is free (anchors / axes were consumed by fs_com_inertia)
  const int b = c.lane;
  const bool on = b < c.D.nr && b > 0;
  const int bb = on ? b : 0;
  const int jt = KI(r_jtype, bb), da = KI(r_dofadr, bb), parent = KI(r_parent, bb);
  S6 u = s6zero(), cc = s6zero();
  if (on) {
    if (jt == JT_FREE) {
      S6 vt = s6zero();
#pragma unroll
      for (int t = 0; t < 3; t++) vt = vt + lds6(L + c.ly.cdof + 6 * (da + t)) * L[c.ly.qvel + da + t];
      u = vt;
#pragma unroll
      for (int k = 3; k < 6; k++) {
        const S6 sk = lds6(L + c.ly.cdof + 6 * (da + k));
        const float qd = L[c.ly.qvel + da + k];
        u = u + sk * qd;
        cc = cc + cross_motion(vt, sk) * qd;
      }
    } else
      u = lds6(L + c.ly.cdof + 6 * da) * L[c.ly.qvel + da];
  }
  int anc = on ? parent : 0;
  for (int span = 1; span < c.D.maxdepth; span <<= 1) {
    if (b < c.D.nr) { sts6(SU + 6 * b, u); sts6(SC + 6 * b, cc); ancs[b] = anc; }
    SYNC();
    if (anc > 0) {
      const S6 ua = lds6(SU + 6 * anc), ca = lds6(SC + 6 * anc);
      const int na = ancs[anc];
      cc = ca + cc + cross_motion(ua, u);
      u = ua + u;
      anc = na;
    }
    SYNC();
  }
  if (b < c.D.nr) {
    S6 a = cc;
    a.l = a.l + v3(-c.D.gravity[0], -c.D.gravity[1], -c.D.gravity[2]);
    sts6(L + c.ly.cvel + 6 * b, u);
    S6 f = s6zero();
    if (b > 0) {
      const float *I = L + c.ly.cinert + 10 * b;
      f = inert_mul(I, a) + cross_force(u, inert_mul(I, u));
    }
    sts6(L + c.ly.cfrc + 6 * b, f);
  }
  SYNC();
  for (int d = c.lane; d < c.D.nv; d += 64) {
    const int bd = KI(dof_rbody, d);
    const S6 s_ = lds6(L + c.ly.cdof + 6 * d);
    float acc = 0;
    // two subtree bodies per trip, the second one predicated (its address falls back to the first): the loads of a trip are
    // independent, so a 9-body subtree costs 5 LDS round trips instead of 9
    for (int mm = KI(r_submask, bd); mm;) {
      const int b0 = __ffs(mm) - 1;
      mm &= mm - 1;
      const bool two = mm != 0;
      const int b1 = two ? __ffs(mm) - 1 : b0;
      mm &= mm - 1;
      const S6 f0 = lds6(L + c.ly.cfrc + 6 * b0), f1 = lds6(L + c.ly.cfrc + 6 * b1);
      acc += dot6(s_, f0) + (two ? dot6(s_, f1) : 0.0f);
    }
    L[c.ly.qfrcbias + d] = acc;
  }
  SYNC();
}

template <class Ctx> DEV void fs_smooth(const Ctx &c) {
  CModel &m = c.m;
  float *L = c.L;
  for (int d = c.lane; d < c.D.nv; d += 64)
    L[c.ly.smooth + d] = -KF(dof_damping, d) * L[c.ly.qvel + d] - L[c.ly.qfrcbias + d] + L[c.ly.qfrcapp + d];
  SYNC();
  for (int u = c.lane; u < c.D.nu; u += 64) {
    float ct = L[c.ly.ctrl + u];
    if (GP(m.act_ctrllimited)[u]) ct = fminf(fmaxf(ct, GP(m.act_ctrlrange)[2 * u]), GP(m.act_ctrlrange)[2 * u + 1]);
    float g = GP(m.act_gear)[u];
    float len = L[c.ly.qpos + GP(m.act_qpos)[u]] * g, vel = L[c.ly.qvel + GP(m.act_dof)[u]] * g;
    float f = GP(m.act_gain)[u] * ct + GP(m.act_bias)[3 * u] + GP(m.act_bias)[3 * u + 1] * len + GP(m.act_bias)[3 * u + 2] * vel;
    if (GP(m.act_forcelimited)[u]) f = fminf(fmaxf(f, GP(m.act_forcerange)[2 * u]), GP(m.act_forcerange)[2 * u + 1]);
    atomicAdd(L + c.ly.smooth + GP(m.act_dof)[u], f * g);
  }
  for (int p = c.lane; p < c.D.nparts; p += 64) {
    const float *F = L + c.ly.xfrc + 6 * p;
    V3 f = ldv3(F), t = ldv3(F + 3);
    if (f.x != 0 || f.y != 0 || f.z != 0 || t.x != 0 || t.y != 0 || t.z != 0) {
      int b = GP(m.part_rbody)[p], d = GP(m.part_dofadr)[p];
      M3 R = ldm3(L + c.ly.xmat + 9 * b);
      V3 tq = multv(R, t + cross(ldv3(L + c.ly.xipos + 3 * b) - ldv3(L + c.ly.xpos + 3 * b), f));
      atomicAdd(L + c.ly.smooth + d, f.x); atomicAdd(L + c.ly.smooth + d + 1, f.y); atomicAdd(L + c.ly.smooth + d + 2, f.z);
      atomicAdd(L + c.ly.smooth + d + 3, tq.x); atomicAdd(L + c.ly.smooth + d + 4, tq.y); atomicAdd(L + c.ly.smooth + d + 5, tq.z);
    }
  }
  SYNC();
}

