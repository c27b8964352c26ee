// fsim_model.hpp -- device-side model tables and per-env memory layout.
//
// Model constants are shared by all envs: they live in one HBM buffer, stay L2-resident
// (tens of KB) and are read through wave-uniform (scalar) loads wherever the index is
// uniform.  Per-env state is one contiguous record ("AoS per env"): with one wavefront per
// env the wave streams its own record in and out with unit-stride 64-lane accesses, so the
// HBM traffic of a fused 50-substep step is one read + one write of the record.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FSIM_MAXANG 8
#define FSIM_CONW 19     // words per contact slot: 18 used + 1 pad -- an ODD stride spreads lane = slot accesses over all 64 LDS banks (18 would hit 32)
#define FSIM_PCAP 192    // words of the body-pair cache (fs_pair_cache): FSIM_PCAP - FSIM_YCAP projection items (a gripper holding a part that touches another: 132) + FSIM_YCAP column items
#define FSIM_YCAP 36     // (pair, dof of the higher body's chain) items: FSIM_NPAIR pairs x chains of <= FSIM_XW / 6 dofs
#define FSIM_XW 54       // words per body-pair block: the 6 x 6 cross block X, overwritten by Y = X * cdof (6 x chain length <= 9)
#define FSIM_WELDW 44    // words per weld record
#define FSIM_LIMW 7      // words per joint-limit record (odd stride, see FSIM_CONW)
#define FSIM_JSTW 9      // words per staged slot (multi-wave kernel): world stiffness K[6], cone-active flag, pair-block id, limit flag; odd stride
#define FSIM_MWCW 16     // command words of the multi-wave protocol (MWC_*)
#define FSIM_MAXSURV 48  // broadphase survivors per substep of models with <= 8 parts (22 is the most seen on Sawyer + table_lack); LayoutIn::maxsurv

enum { JT_FREE = 0, JT_BALL = 1, JT_SLIDE = 2, JT_HINGE = 3 };
enum { GT_PLANE = 0, GT_SPHERE = 2, GT_CAPSULE = 3, GT_CYLINDER = 5, GT_BOX = 6, GT_MESH = 7 };
enum { PT_PLANE_SPHERE = 0, PT_PLANE_BOX, PT_PLANE_CYL, PT_SPHERE_SPHERE, PT_SPHERE_BOX, PT_SPHERE_CYL, PT_BOX_BOX, PT_CYL_BOX, PT_CYL_CYL,
       PT_PLANE_CAP, PT_CONVEX /* every other pair with a capsule or a convex mesh in it, through the portal routine */,
       PT_PLANE_MESH /* a convex mesh (its hull vertices: DModel::mesh_vert) against a plane */ };

// Dimensions and scalar options of a model.  The generic kernels read them at run time (scalar loads from the DModel
// in constant memory); the kernels specialised for one (agent, furniture, config) carry them as compile-time constants
// (fsim_spec.hpp), which turns every layout offset into an instruction immediate and every `lane < n` loop into a test.
struct Dims {
  int nq, nv, nu, nr, ntree, ncg, ncp, nsite, neq, nparts, nM, maxdepth, nconn, narm, nlim, nbody, ngeom, agent;
  int narmj, ngripj;
  float timestep, gravity[3], impratio, meaninertia_scale;
};

struct DModel : Dims {
  // reduced bodies
  const int *r_parent, *r_jtype, *r_qposadr, *r_dofadr, *r_dofnum, *r_depth, *r_tree, *r_chainadr, *r_chainlen, *r_ancmask;
  const float *r_pos, *r_quat, *r_jaxis, *r_jpos, *r_mass, *r_ipos, *r_inertia;
  const int *chain_dofs;
  const int *tree_dofadr, *tree_dofnum, *tree_bodyadr, *tree_bodynum;
  // dofs
  const int *dof_parent, *dof_Madr, *dof_rbody, *dof_tree, *dof_qposadr, *M_i, *M_j;
  const float *dof_armature, *dof_damping, *dof_invweight0;
  // joint limits (one entry per limited hinge/slide dof)
  const int *lim_dof;
  const float *lim_range, *lim_margin, *lim_solref, *lim_solimp;
  // colliding geoms
  const int *cg_body, *cg_type, *cg_condim, *cg_partid, *cg_fingerrole, *cg_isfloor, *cg_isrobot, *cg_ispartcol, *cg_orig,
      *cg_contype0, *cg_conaffinity0, *cg_cursor, *cg_namepart; // cg_cursor: bit K = geom of cursor K; cg_namepart: parts named in the geom's name
  const float *cursor_pos0; // [2][3] model body_pos of the cursor bodies (Cursor agent)
  const float *cg_pos, *cg_mat, *cg_size, *cg_rbound, *cg_friction, *cg_solref, *cg_solimp, *cg_margin, *cg_gap, *cg_solmix, *cg_invweight;
  const int *cp; // [ncp][3] = cg1, cg2, pair type
  // [ncp][16] one 64-byte record per candidate pair with everything the broad/narrow phase needs from the two geoms:
  // g1 g2 pairtype (type1 | type2 << 8) | margin gap rbound1 rbound2 | size1 xyz mesh1 | size2 xyz mesh2   (built by fsim_create;
  // mesh = first hull vertex | vertex count << 16 for a convex-mesh geom, as int bits)
  const float *pair_rec;
  const float *mesh_vert; // [nvert][3] convex-hull vertices of the colliding mesh geoms, geom frame (three furniture: SURVEY C.1)
  // [ncp][2] broadphase stage-1 record: g1 | g2 << 8 | (geom 1 is a plane) << 16, then the float bound the centre distance (plane:
  // the signed distance of geom 2's centre) is tested against: r1 + r2 + margin (plane: r2 + margin)
  const int *pair_bp;
  // sites
  const int *s_body;
  const float *s_pos, *s_quat;
  // actuators
  const int *act_dof, *act_qpos, *act_ctrllimited, *act_forcelimited;
  const float *act_gain, *act_bias, *act_ctrlrange, *act_forcerange, *act_gear, *ctrl_bias, *ctrl_weight;
  // welds
  const int *eq_rbody1, *eq_rbody2, *eq_part1, *eq_part2;
  const float *eq_solref, *eq_solimp, *eq_invweight, *eq_data0;
  const int *eq_active0;
  // env tables
  const int *part_rbody, *part_qposadr, *part_dofadr, *body_red;
  const float *body_relpos, *body_relquat, *part_mass;
  const int *arm_qposadr, *arm_dofadr, *grip_qposadr, *grip_dofadr, *eef_siteid, *hand_body;
  const float *arm_initqpos, *grip_initqpos, *qpos0;
  const float *ik_tab; // control_type "ik": URDF chain of the reference's IK controller, rest pose, base pose (fsim_ik.hpp IKT_*)
  const int *conn_siteid, *conn_partid, *conn_keya, *conn_keyb, *conn_nangle;
  const float *conn_angles;
  const int *part_site_adr, *part_site_num, *part_sites; // all sites living on each part body (bounding boxes)
};

// word offsets inside one env record (HBM) == start of the LDS image
struct Layout {
  int qpos, qvel, qaccws, qfrcbias, ctrl, qfrcapp, xfrc, eqdata, eqactive, contype, conaff, env, eik, stride;  // eik: first word of the IK blocks inside the env record (after the dense block, if any)
  // LDS-only work arrays (word offsets from LDS base)
  int xpos, xquat, xmat, xanchor, xaxis, xipos, com;
  int cinert, crb, cdofdot, cvel, cacc, cfrc, H; // H aliases cinert..cfrc
  int cdof, M, LD, Dinv, LDh, Dhinv;
  int smooth, asmooth, x, Mx, grad, p, Mp;
  int gpos, gmat, surv, con, weld, lim, W, G, scal;
  int hmap;   // per-substep island map of the Newton system (FSIM_MAPW(nv) words, format at fs_build_map in fsim_solver.hpp)
  int hA, hP; // Hessian body blocks / pair blocks: alias gpos+gmat (geom poses are dead once the contacts exist)
  int pitem;  // [FSIM_PCAP] body-pair projection items of this substep (fs_pair_cache)
  // multi-wave layout only (make_layout(in, nw > 1); zero otherwise): see fsim_solver.hpp "multi-wave Newton iteration"
  int hAhi, hAc; // second set of contact blocks (the higher body of each contact) and the composite blocks; hA = the lower bodies' set
  int jst;       // [64][FSIM_JSTW] this iteration's cone state of every contact slot (world stiffness, active flag), staged by the main wave for the helpers
  int mwc;       // [FSIM_MWCW] command words of the workgroup's main wave / helper waves protocol
  int anc2;      // [nr] pointer-doubling scratch of fs_velocity_bias (the survivor list is live: fs_collide runs beside it)
  int lds_words, ncon_max, maxsurv;
  // LDS cache of the small model tables that sit inside serial / dependent loops (loaded once per launch)
  int k_dof_parent, k_r_submask, k_dof_rbody, k_dof_tree, k_r_parent, k_r_jtype, k_r_qposadr, k_r_dofadr, k_r_chain, k_r_tree, k_r_chainadr, k_r_chainlen, k_r_ancmask, k_chain_dofs, k_tree_dofadr, k_tree_dofnum, k_tree_bodyadr, k_tree_bodynum, k_M_ij, k_r_pos, k_r_quat, k_r_jpos, k_r_jaxis, k_r_ipos, k_r_mass, k_r_inertia, k_dof_damping, k_dof_armature, k_tmap;
  int k_begin, k_end;
};

// per-env scalars in LDS (Layout::scal).  0..8 named, 9..14 scratch of the env logic (fsim_env.hpp), then -- in
// -DFSIM_PROFILE builds only -- 48 words of profile counters (16..63), then the island bookkeeping of the Newton system.
#ifdef FSIM_PROFILE
#define FSIM_SC_BASE 64
#else
#define FSIM_SC_BASE 16
#endif
enum { SC_NSURV = 0, SC_NSLOT = 1, SC_OVERFLOW = 2, SC_NITER = 3, SC_TOUCHL = 4, SC_TOUCHR = 5, SC_TOUCHF = 6, SC_NCON = 7, SC_BAD = 8,
       SC_NITSUM = 15, // Newton iterations of all solves since the launch began (env_step stores it in E_NITER)
       SC_ADJ = FSIM_SC_BASE, SC_ISL = FSIM_SC_BASE + 16, SC_TMP = FSIM_SC_BASE + 32, SC_PADJ = FSIM_SC_BASE + 64,
       SC_HWORDS = FSIM_SC_BASE + 80, SC_TWORDS = FSIM_SC_BASE + 81,
       SC_ASM = FSIM_SC_BASE + 82, // trees of the big islands assembled on the matrix cores in this solve (fs_asm_trees; kept here, not in a register)
       SC_WORDS = FSIM_SC_BASE + 84 };

// env-logic block (word offsets relative to Layout::env)
enum {
  E_NUM_CONNECTED = 0, E_PREV_NUM_CONNECTED, E_CONNECT_STEP, E_EPISODE_LENGTH, E_SUCCESS, E_FAIL,
  E_OVERFLOW /* sticky SC_OVERFLOW bits: some launch of this env (a step, a reset, a look-ahead reset swapped in) dropped contacts; survives resets */,
  E_CONNECTED_THIS_STEP, E_SITE1, E_SITE2, E_SUBTASK1, E_SUBTASK2, E_TOUCHED, E_PICKED, E_CONNSITES0, E_CONNSITES1,
  E_CONNSITES2, E_CONNSITES3, E_TOUCH_L, E_TOUCH_R, E_TOUCH_FLOOR, E_CONNBODY1, E_CB1_POS, E_CB1_QUAT = E_CB1_POS + 3,
  E_TARGET_QUAT = E_CB1_QUAT + 4, E_EPISODE_REWARD = E_TARGET_QUAT + 4, E_CLEARANCE /* float: env_robot_clearance at the end of the last step */, E_NITER /* Newton iterations of the last step */,
  E_MW_STEPS /* steps of this episode that four waves took (multi-wave workgroups): lets a test see which kernel path an env ran */, E_EPISODE_COUNT,
  E_GROUP, // nparts ints follow
  E_FIXED_WORDS = E_GROUP
};
// Cursor agent block, placed after the group table (env_extra_words): word offsets relative to Layout::env + E_GROUP + nparts
enum { EC_SEL = 0 /* 2 ints: selected part + 1, 0 = none */, EC_POS = 2 /* model.body_pos of the two cursors */,
       EC_XPOS = 8 /* data.xpos of the cursors = EC_POS as of the last forward pass */,
       EC_P2Q0 = 14 /* gradual connect (furniture.py:993-1034): part pose at connect step 0 */, EC_BODY_POS = 21, EC_BODY_ROT = 24,
       EC_TOUCH = 28 /* 2 ints: parts in contact with cursor K at the last forward pass */, EC_WORDS = 30 };

#define FSIM_NPAIR 4 // body-pair cross blocks assembled per Hessian pass (fsim_solver.hpp)
// island / tree map of a block-diagonal SPD system: [nv] dof words, [64] lane words, [16] tail (fs_build_map, fsim_solver.hpp)
#define FSIM_MAPW(nv) ((nv) + 64 + 16)

// Everything the LDS layout depends on (host: from the model blob + config; specialised kernels: compile-time constants)
struct LayoutIn {
  int nq, nv, nu, nr, ntree, ncg, nparts, neq, nlim, nM;
  int Mwords;    // sum over kinematic trees of n (n + 1) / 2: the tree-packed lower triangles of M
  int nchain;    // entries of chain_dofs
  int env_words; // env-logic block of the record: E_FIXED_WORDS + nparts + agent / reward / controller extras
  int eik_rel;   // offset of the IK blocks behind the group table (the dense block, if any, comes first)
  int ncon_max;
  int maxsurv;   // broadphase survivor list (FSIM_MAXSURV; more for furniture with many long parts lying next to each other)
};

// One function for host and device: fsim_create calls it at run time, fsim_spec.hpp at compile time.
constexpr Layout make_layout(const LayoutIn &in, int nw = 1) {
  Layout ly{};
  int o = 0;
#define TAKE(field, n) do { ly.field = o; o += (n); } while (0)
  TAKE(qpos, in.nq); TAKE(qvel, in.nv); TAKE(qaccws, in.nv); TAKE(qfrcbias, in.nv); TAKE(ctrl, in.nu);
  TAKE(qfrcapp, in.nv); TAKE(xfrc, 6 * in.nparts); TAKE(eqdata, 7 * in.neq); TAKE(eqactive, in.neq);
  TAKE(contype, in.ncg); TAKE(conaff, in.ncg); TAKE(env, in.env_words);
  ly.eik = ly.env + E_GROUP + in.nparts + in.eik_rel;
  o = (o + 3) / 4 * 4;
  ly.stride = o;
  // LDS-only
  TAKE(xpos, 3 * in.nr); TAKE(xquat, 4 * in.nr); TAKE(xmat, 9 * in.nr); TAKE(xipos, 3 * in.nr); TAKE(com, 3 * in.ntree);
  TAKE(cvel, 6 * in.nr); // read by constraint assembly AND by the observation (site velocities): never aliased
  const int hstart = o;
  TAKE(cinert, 10 * in.nr); TAKE(crb, 10 * in.nr); TAKE(cdofdot, 6 * in.nv); TAKE(cacc, 6 * in.nr); TAKE(cfrc, 6 * in.nr);
  // joint anchors / axes are only live between kinematics and the motion-axis computation: they sit in the unused cacc slot
  ly.xanchor = ly.cacc; ly.xaxis = ly.cacc + 3 * in.nr;
  // H (Newton Hessian, packed lower triangle) is only live inside fs_solve, after the rigid-body temporaries above are
  // dead, so it aliases them.
  const int nH = in.nv * (in.nv + 1) / 2;
  ly.H = hstart;
  if (hstart + nH > o) o = hstart + nH;
  TAKE(cdof, 6 * in.nv);
  TAKE(M, in.Mwords); // tree-packed triangle layout of k_tmap (dense lower triangle per kinematic tree)
  ly.LD = ly.M; ly.Dinv = ly.M; ly.LDh = ly.M; ly.Dhinv = ly.M;
  TAKE(smooth, in.nv); ly.asmooth = ly.smooth; TAKE(x, in.nv); TAKE(Mx, in.nv); TAKE(grad, in.nv); TAKE(p, in.nv); TAKE(Mp, in.nv);
  TAKE(gpos, 3 * in.ncg); TAKE(gmat, 9 * in.ncg);
  {
    // (multi-wave: three sets of body blocks -- lower body, higher body, composite -- because different waves fill them)
    const int nsets = nw > 1 ? 3 : 1;
    int need = 21 * in.nr * nsets + FSIM_XW * FSIM_NPAIR + 3 * FSIM_NPAIR + 4;
    ly.hA = ly.gpos; ly.hP = ly.gpos + 21 * in.nr * nsets;
    if (nw > 1) { ly.hAhi = ly.gpos + 21 * in.nr; ly.hAc = ly.gpos + 42 * in.nr; }
    if (need < 12 * in.nr) need = 12 * in.nr;
    if (need > 12 * in.ncg) o += need - 12 * in.ncg;
    // per-body spatial vectors W (J*v) and wrenches G (J'f) are dead while the Hessian blocks are live and vice versa
    ly.W = ly.gpos; ly.G = ly.gpos + 6 * in.nr;
  }
  if (nw > 1) { // the gradient's wrenches are accumulated while the helper waves fill the Hessian blocks: no aliasing
    TAKE(G, 6 * in.nr); TAKE(jst, FSIM_JSTW * 64); TAKE(mwc, FSIM_MWCW); TAKE(anc2, in.nr);
  }
  TAKE(surv, in.maxsurv); TAKE(pitem, FSIM_PCAP);
  TAKE(con, FSIM_CONW * in.ncon_max); TAKE(weld, FSIM_WELDW * in.neq); TAKE(lim, FSIM_LIMW * 2 * in.nlim);
  TAKE(scal, SC_WORDS); TAKE(hmap, FSIM_MAPW(in.nv));
  // LDS model cache
  ly.k_begin = o;
  TAKE(k_dof_tree, in.nv); ly.k_dof_parent = 0; TAKE(k_r_submask, in.nr); TAKE(k_dof_rbody, in.nv); // (dof -> kinematic tree: the island bookkeeping of the Newton solve; the dof-parent table it replaced was never read)
  TAKE(k_r_parent, in.nr); TAKE(k_r_jtype, in.nr); TAKE(k_r_qposadr, in.nr); TAKE(k_r_dofadr, in.nr); TAKE(k_r_chain, in.nr);
  TAKE(k_r_tree, in.nr); TAKE(k_r_chainadr, in.nr); TAKE(k_r_chainlen, in.nr); ly.k_r_ancmask = 0; TAKE(k_chain_dofs, in.nchain);
  TAKE(k_tree_dofadr, in.ntree); TAKE(k_tree_dofnum, in.ntree); TAKE(k_tree_bodyadr, in.ntree); TAKE(k_tree_bodynum, in.ntree);
  TAKE(k_M_ij, in.nM);
  ly.k_r_pos = 0; ly.k_r_quat = 0; ly.k_r_jpos = 0; ly.k_r_jaxis = 0; ly.k_r_ipos = 0; ly.k_r_inertia = 0; // not cached
  TAKE(k_r_mass, in.nr); TAKE(k_dof_damping, in.nv); TAKE(k_dof_armature, in.nv); TAKE(k_tmap, FSIM_MAPW(in.nv));
  ly.k_end = o;
#undef TAKE
  ly.lds_words = o;
  ly.ncon_max = in.ncon_max;
  ly.maxsurv = in.maxsurv;
  return ly;
}

// Device code reaches the (read-only, launch-constant) model and layout structs through the CONSTANT address space:
// their fields then come in by scalar loads (s_load_dword through the scalar cache, hoistable and CSE-able across
// stores) instead of flat vector loads that tick both memory counters and serialise behind every LDS wait.
typedef const __attribute__((address_space(4))) DModel CModel;
typedef const __attribute__((address_space(4))) Layout CLayout;

