/* fsim_oracle.c -- see fsim_oracle.h.  TEST INFRASTRUCTURE: the product never links this.
 *
 * One environment, real precision, plain C.  Follows the stage list of
 * SURVEY.md section 8(a) P1-P9; each stage cites the reference call site that
 * triggers it (all physics is reached via sim.forward()/sim.step(),
 * furniture/env/furniture.py:2877-2879).
 *
 * Spatial algebra here is referenced at the WORLD ORIGIN (the device code uses
 * per-tree centre-of-mass references), M comes from a composite-rigid-body pass
 * stored dense per kinematic tree with a dense Cholesky, and Jacobian rows are
 * kept as sparse (dof, value) lists -- deliberately a different formulation from
 * the HIP kernels so that agreement between the two is evidence, not tautology.
 */
#include "fsim_oracle.h"
#include <tgmath.h> /* (type-generic: sqrt / sin / fabs ... of a float stay float in the fp32 control build) */
#undef I /* (<complex.h>, which <tgmath.h> brings in) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
/* a constant of the fp64 checker | of the fp32 control build (tolerances below fp32 resolution would never be met) */
#define RSEL(d, f) (sizeof(real) == 8 ? (real)(d) : (real)(f))
#define MAXCON 1024 /* contacts of one substep; more are DROPPED and counted in `ndropped` (never silently: the callers raise on it) */
#define MAXEFC 4096 /* constraint rows, likewise */
#define MAXNNZ 128
#define MAXCONV 64

enum { G_PLANE = 0, G_HFIELD, G_SPHERE, G_CAPSULE, G_ELLIPSOID, G_CYLINDER, G_BOX, G_MESH };
enum { J_FREE = 0, J_BALL, J_SLIDE, J_HINGE };
enum { C_EQ = 0, C_LIMIT = 1, C_CONTACT = 2 };

static char g_err[256];
const char *osim_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ blob */
typedef struct { char name[48]; int32_t code; int32_t pad; int64_t count; int64_t off; } BlobEnt;

typedef struct {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, neq, npair, nM;
  real timestep, gravity[3], impratio;
  const real *qpos0;
  const int32_t *body_parentid, *body_rootid, *body_weldid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  real *body_pos;  /* mutable (cursor bodies, furniture.py:3139) */
  const real *body_quat, *body_ipos, *body_iquat, *body_mass, *body_inertia, *body_invweight0;
  const int32_t *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const real *jnt_pos, *jnt_axis, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp;
  const int32_t *dof_bodyid, *dof_jntid, *dof_parentid;
  const real *dof_armature, *dof_damping, *dof_invweight0;
  const int32_t *geom_type, *geom_bodyid, *geom_condim;
  int32_t *geom_contype, *geom_conaffinity; /* mutable (furniture.py:874-878) */
  const real *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solref, *geom_solimp, *geom_margin,
      *geom_gap, *geom_rbound, *geom_solmix;
  const int32_t *site_bodyid;
  const real *site_pos, *site_quat;
  const int32_t *actuator_jntid, *actuator_ctrllimited, *actuator_forcelimited;
  const real *actuator_gain, *actuator_bias, *actuator_ctrlrange, *actuator_forcerange, *actuator_gear;
  const int32_t *eq_obj1id, *eq_obj2id;
  int32_t *eq_active; /* mutable */
  real *eq_data;    /* mutable (furniture.py:2772) */
  const real *eq_solref, *eq_solimp;
  const int32_t *pair_geom;
  const int32_t *geom_meshadr, *geom_meshnum; /* convex-mesh colliders: hull vertices in mesh_vert (geom frame); NULL in tables compiled before round 5 */
  const real *mesh_vert;
  int ntree;
  int *tree_dofadr, *tree_dofnum, *dof_treeid;
} Model;

typedef struct {
  real dist, pos[3], frame[9], includemargin, mu, solref[2], solimp[5];
  int geom1, geom2, dim, efc_address;
} Contact;

typedef struct {
  int type, nnz, dim; /* dim: rows in this block (contact normal row carries 3, others 1; friction rows 0) */
  int idx[MAXNNZ];
  real J[MAXNNZ], B[MAXNNZ];
  real pos, margin, R, D, aref, force, diagA;
  real mu; /* contacts */
} Row;

struct osim {
  void *blob;
  Model m;
  /* state */
  real *qpos, *qvel, *ctrl, *qfrc_applied, *xfrc_applied, *qacc, *qacc_warmstart, time_;
  /* derived */
  real *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
  real *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  real *M, *L, *Lh; /* dense nv*nv; only per-tree blocks used */
  real *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *actuator_force;
  Contact *contact;
  int32_t ncon, *contact_geom1, *contact_geom2;
  Row *row;
  int nefc;
  int32_t ndropped; /* contacts / rows that did not fit MAXCON / MAXEFC since the last osim_reset (sticky) */
  real *conv[MAXCONV]; /* blob entries converted to `real` (fp32 control build only) */
  int nconv;
  int solver_iters, solver_kind; /* kind: 0 PGS, 1 Newton */
  real solver_tol;
  int last_iters;
};

static const void *blob_find(const void *blob, size_t nbytes, const char *name, int code, int64_t *count) {
  const char *p = (const char *)blob;
  int32_t n;
  memcpy(&n, p + 12, 4);
  const BlobEnt *e = (const BlobEnt *)(p + 16);
  for (int i = 0; i < n; i++) {
    if (strncmp(e[i].name, name, 48) == 0) {
      if (e[i].code != code) { snprintf(g_err, sizeof g_err, "blob entry %s has wrong dtype", name); return NULL; }
      if ((size_t)e[i].off + (size_t)e[i].count * (code == 0 ? 8 : 4) > nbytes) { snprintf(g_err, sizeof g_err, "blob entry %s out of range", name); return NULL; }
      if (count) *count = e[i].count;
      return p + e[i].off;
    }
  }
  snprintf(g_err, sizeof g_err, "blob entry %s missing", name);
  return NULL;
}
/* a float64 entry of the blob as `real`: in place in the (private copy of the) blob when real is double, converted into an array the
 * simulator owns otherwise (the control build in fp32) */
static real *blob_reals(osim_t *s, size_t nbytes, const char *name) {
  int64_t cnt = 0;
  const void *p = blob_find(s->blob, nbytes, name, 0, &cnt);
  if (!p) return NULL;
  if (sizeof(real) == 8) return (real *)p;
  if (s->nconv >= MAXCONV) { snprintf(g_err, sizeof g_err, "too many converted blob entries"); return NULL; }
  real *r = (real *)malloc(sizeof(real) * (size_t)(cnt + 1));
  for (int64_t i = 0; i < cnt; i++) { double v; memcpy(&v, (const char *)p + 8 * i, 8); r[i] = (real)v; }
  s->conv[s->nconv++] = r;
  return r;
}
#define GETD(field) do { m->field = blob_reals(s, nbytes, #field); if (!m->field) return -1; } while (0)
#define GETI(field) do { m->field = (const int32_t *)blob_find(s->blob, nbytes, #field, 1, NULL); if (!m->field) return -1; } while (0)
#define GETDM(field) GETD(field)
#define GETIM(field) do { m->field = (int32_t *)blob_find(s->blob, nbytes, #field, 1, NULL); if (!m->field) return -1; } while (0)

static int load_model(osim_t *s, size_t nbytes) {
  Model *m = &s->m;
  if (memcmp(s->blob, "FSIMBLOB", 8) != 0) { snprintf(g_err, sizeof g_err, "bad blob magic"); return -1; }
  const int32_t *dims = (const int32_t *)blob_find(s->blob, nbytes, "dims", 1, NULL);
  if (!dims) return -1;
  m->nq = dims[0]; m->nv = dims[1]; m->nu = dims[2]; m->nbody = dims[3]; m->njnt = dims[4]; m->ngeom = dims[5];
  m->nsite = dims[6]; m->neq = dims[7]; m->npair = dims[8]; m->nM = dims[9];
  const real *opt = blob_reals(s, nbytes, "opt");
  if (!opt) return -1;
  m->timestep = opt[0]; m->gravity[0] = opt[1]; m->gravity[1] = opt[2]; m->gravity[2] = opt[3]; m->impratio = opt[4];
  GETD(qpos0);
  GETI(body_parentid); GETI(body_rootid); GETI(body_weldid); GETI(body_jntadr); GETI(body_jntnum); GETI(body_dofadr); GETI(body_dofnum);
  GETDM(body_pos); GETD(body_quat); GETD(body_ipos); GETD(body_iquat); GETD(body_mass); GETD(body_inertia); GETD(body_invweight0);
  GETI(jnt_type); GETI(jnt_qposadr); GETI(jnt_dofadr); GETI(jnt_bodyid); GETI(jnt_limited);
  GETD(jnt_pos); GETD(jnt_axis); GETD(jnt_range); GETD(jnt_margin); GETD(jnt_solref); GETD(jnt_solimp);
  GETI(dof_bodyid); GETI(dof_jntid); GETI(dof_parentid); GETD(dof_armature); GETD(dof_damping); GETD(dof_invweight0);
  GETI(geom_type); GETI(geom_bodyid); GETI(geom_condim); GETIM(geom_contype); GETIM(geom_conaffinity);
  GETD(geom_size); GETD(geom_pos); GETD(geom_quat); GETD(geom_friction); GETD(geom_solref); GETD(geom_solimp);
  GETD(geom_margin); GETD(geom_gap); GETD(geom_rbound); GETD(geom_solmix);
  GETI(site_bodyid); GETD(site_pos); GETD(site_quat);
  GETI(actuator_jntid); GETI(actuator_ctrllimited); GETI(actuator_forcelimited);
  GETD(actuator_gain); GETD(actuator_bias); GETD(actuator_ctrlrange); GETD(actuator_forcerange); GETD(actuator_gear);
  GETI(eq_obj1id); GETI(eq_obj2id); GETD(eq_solref); GETD(eq_solimp);
  GETI(pair_geom);
  m->geom_meshadr = (const int32_t *)blob_find(s->blob, nbytes, "geom_meshadr", 1, NULL); /* optional */
  m->geom_meshnum = (const int32_t *)blob_find(s->blob, nbytes, "geom_meshnum", 1, NULL);
  m->mesh_vert = blob_reals(s, nbytes, "mesh_vert");
  g_err[0] = 0;
  /* mutable copies seeded from the *0 arrays */
  {
    int64_t c;
    const int32_t *a0 = (const int32_t *)blob_find(s->blob, nbytes, "eq_active0", 1, &c);
    const real *d0 = blob_reals(s, nbytes, "eq_data0");
    if (!a0 || !d0) return -1;
    m->eq_active = (int32_t *)calloc(m->neq + 1, sizeof(int32_t));
    m->eq_data = (real *)calloc(7 * m->neq + 1, sizeof(real));
    memcpy(m->eq_active, a0, sizeof(int32_t) * m->neq);
    memcpy(m->eq_data, d0, sizeof(real) * 7 * m->neq);
  }
  /* kinematic trees = maximal runs of dofs sharing a root body (dofs of a tree are contiguous in DFS order) */
  m->dof_treeid = (int *)calloc(m->nv + 1, sizeof(int));
  m->tree_dofadr = (int *)calloc(m->nv + 1, sizeof(int));
  m->tree_dofnum = (int *)calloc(m->nv + 1, sizeof(int));
  m->ntree = 0;
  for (int d = 0; d < m->nv; d++) {
    int root = m->body_rootid[m->dof_bodyid[d]];
    if (d == 0 || root != m->body_rootid[m->dof_bodyid[d - 1]]) {
      m->tree_dofadr[m->ntree] = d; m->tree_dofnum[m->ntree] = 0; m->ntree++;
    }
    m->dof_treeid[d] = m->ntree - 1;
    m->tree_dofnum[m->ntree - 1]++;
  }
  return 0;
}

/* ------------------------------------------------------------------ math */
static inline real dot3(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(real *r, const real *a, const real *b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void copy3(real *r, const real *a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void sub3(real *r, const real *a, const real *b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void add3(real *r, const real *a, const real *b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void addscl3(real *r, const real *a, const real *b, real s) { r[0] = a[0] + s * b[0]; r[1] = a[1] + s * b[1]; r[2] = a[2] + s * b[2]; }
static inline void scl3(real *r, const real *a, real s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
static inline real norm3(const real *a) { return sqrt(dot3(a, a)); }
static inline real normalize3(real *a) {
  real n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n; return n;
}
static void quat_mul(real *r, const real *a, const real *b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_norm(real *q) {
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat2mat(real *R, const real *q) { /* row-major 3x3 */
  real w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
static inline void mulMV(real *r, const real *R, const real *v) {
  real x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulMTV(real *r, const real *R, const real *v) {
  real x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2], z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axisangle_quat(real *q, const real *axis, real ang) {
  real s = sin(0.5 * ang);
  q[0] = cos(0.5 * ang); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* spatial vectors are [ang(3), lin(3)] about the world origin */
static void cross_motion(real *r, const real *v, const real *m) {
  real a[3], b[3], c[3];
  cross3(a, v, m); cross3(b, v, m + 3); cross3(c, v + 3, m);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(real *r, const real *v, const real *f) {
  real a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
/* inertia record: [Ixx Iyy Izz Ixy Ixz Iyz, hx hy hz, m]  (I about origin, h = m*c) */
static void inert_mul(real *f, const real *I, const real *v) {
  const real *w = v, *l = v + 3;
  real hxl[3], hxw[3];
  cross3(hxl, I + 6, l); cross3(hxw, I + 6, w);
  f[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2] + hxl[0];
  f[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2] + hxl[1];
  f[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2] + hxl[2];
  f[3] = I[9] * l[0] - hxw[0]; f[4] = I[9] * l[1] - hxw[1]; f[5] = I[9] * l[2] - hxw[2];
}

/* ------------------------------------------------------------------ P1 kinematics */
static void kinematics(osim_t *s) {
  Model *m = &s->m;
  real *xpos = s->xpos, *xquat = s->xquat;
  xpos[0] = xpos[1] = xpos[2] = 0; xquat[0] = 1; xquat[1] = xquat[2] = xquat[3] = 0;
  quat2mat(s->xmat, xquat);
  copy3(s->xipos, xpos); memcpy(s->ximat, s->xmat, 9 * sizeof(real));
  for (int i = 1; i < m->nbody; i++) {
    int p = m->body_parentid[i], jn = m->body_jntnum[i], ja = m->body_jntadr[i];
    real pos[3], quat[4];
    if (jn == 1 && m->jnt_type[ja] == J_FREE) {
      real *q = s->qpos + m->jnt_qposadr[ja];
      quat_norm(q + 3); /* MuJoCo normalises the stored quaternion in place */
      copy3(pos, q); memcpy(quat, q + 3, 4 * sizeof(real));
      copy3(s->xanchor + 3 * ja, pos);
      s->xaxis[3 * ja] = 0; s->xaxis[3 * ja + 1] = 0; s->xaxis[3 * ja + 2] = 1;
    } else {
      real t[3];
      mulMV(t, s->xmat + 9 * p, m->body_pos + 3 * i);
      add3(pos, xpos + 3 * p, t);
      quat_mul(quat, xquat + 4 * p, m->body_quat + 4 * i);
      for (int j = ja; j < ja + jn; j++) {
        real R[9], ql[4];
        quat2mat(R, quat);
        mulMV(t, R, m->jnt_pos + 3 * j); add3(s->xanchor + 3 * j, pos, t);
        mulMV(s->xaxis + 3 * j, R, m->jnt_axis + 3 * j);
        real q = s->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
        if (m->jnt_type[j] == J_SLIDE) {
          addscl3(pos, pos, s->xaxis + 3 * j, q);
        } else if (m->jnt_type[j] == J_HINGE) {
          axisangle_quat(ql, m->jnt_axis + 3 * j, q);
          quat_mul(quat, quat, ql);
          quat2mat(R, quat);
          mulMV(t, R, m->jnt_pos + 3 * j); sub3(pos, s->xanchor + 3 * j, t);
        }
      }
    }
    quat_norm(quat);
    copy3(xpos + 3 * i, pos); memcpy(xquat + 4 * i, quat, 4 * sizeof(real));
    quat2mat(s->xmat + 9 * i, quat);
    real t[3], qi[4];
    mulMV(t, s->xmat + 9 * i, m->body_ipos + 3 * i); add3(s->xipos + 3 * i, pos, t);
    quat_mul(qi, quat, m->body_iquat + 4 * i); quat2mat(s->ximat + 9 * i, qi);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    real t[3], q[4];
    mulMV(t, s->xmat + 9 * b, m->geom_pos + 3 * g); add3(s->geom_xpos + 3 * g, xpos + 3 * b, t);
    quat_mul(q, xquat + 4 * b, m->geom_quat + 4 * g); quat2mat(s->geom_xmat + 9 * g, q);
  }
  for (int k = 0; k < m->nsite; k++) {
    int b = m->site_bodyid[k];
    real t[3], q[4];
    mulMV(t, s->xmat + 9 * b, m->site_pos + 3 * k); add3(s->site_xpos + 3 * k, xpos + 3 * b, t);
    quat_mul(q, xquat + 4 * b, m->site_quat + 4 * k); quat2mat(s->site_xmat + 9 * k, q);
  }
}

/* ------------------------------------------------------------------ P2 inertia, CRB, factor */
static void motion_axes(osim_t *s) {
  Model *m = &s->m;
  for (int j = 0; j < m->njnt; j++) {
    int d = m->jnt_dofadr[j], b = m->jnt_bodyid[j];
    real *S = s->cdof + 6 * d;
    if (m->jnt_type[j] == J_FREE) {
      for (int k = 0; k < 3; k++) {
        real *St = S + 6 * k, *Sr = S + 6 * (3 + k);
        memset(St, 0, 6 * sizeof(real)); St[3 + k] = 1;
        real a[3] = {s->xmat[9 * b + k], s->xmat[9 * b + 3 + k], s->xmat[9 * b + 6 + k]};
        copy3(Sr, a); cross3(Sr + 3, s->xpos + 3 * b, a);
      }
    } else if (m->jnt_type[j] == J_SLIDE) {
      memset(S, 0, 6 * sizeof(real)); copy3(S + 3, s->xaxis + 3 * j);
    } else {
      copy3(S, s->xaxis + 3 * j); cross3(S + 3, s->xanchor + 3 * j, s->xaxis + 3 * j);
    }
  }
}

static void body_inertias(osim_t *s) {
  Model *m = &s->m;
  memset(s->cinert, 0, sizeof(real) * 10 * m->nbody);
  for (int b = 1; b < m->nbody; b++) {
    real mass = m->body_mass[b];
    const real *R = s->ximat + 9 * b, *c = s->xipos + 3 * b, *di = m->body_inertia + 3 * b;
    real *I = s->cinert + 10 * b;
    /* R diag(di) R^T */
    real Ic[6];
    int ij[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
    for (int e = 0; e < 6; e++) {
      int i = ij[e][0], j = ij[e][1];
      Ic[e] = R[3 * i] * di[0] * R[3 * j] + R[3 * i + 1] * di[1] * R[3 * j + 1] + R[3 * i + 2] * di[2] * R[3 * j + 2];
    }
    real cc = dot3(c, c);
    I[0] = Ic[0] + mass * (cc - c[0] * c[0]); I[1] = Ic[1] + mass * (cc - c[1] * c[1]); I[2] = Ic[2] + mass * (cc - c[2] * c[2]);
    I[3] = Ic[3] - mass * c[0] * c[1]; I[4] = Ic[4] - mass * c[0] * c[2]; I[5] = Ic[5] - mass * c[1] * c[2];
    I[6] = mass * c[0]; I[7] = mass * c[1]; I[8] = mass * c[2]; I[9] = mass;
  }
}

static int chol_block(real *L, const real *A, int nv, int a0, int n) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      real sum = A[(a0 + i) * nv + a0 + j];
      for (int k = 0; k < j; k++) sum -= L[(a0 + i) * nv + a0 + k] * L[(a0 + j) * nv + a0 + k];
      if (i == j) {
        if (sum < MINVAL) return -1;
        L[(a0 + i) * nv + a0 + i] = sqrt(sum);
      } else
        L[(a0 + i) * nv + a0 + j] = sum / L[(a0 + j) * nv + a0 + j];
    }
  return 0;
}
static void chol_solve_block(const real *L, real *x, int nv, int a0, int n) {
  for (int i = 0; i < n; i++) {
    real sum = x[a0 + i];
    for (int k = 0; k < i; k++) sum -= L[(a0 + i) * nv + a0 + k] * x[a0 + k];
    x[a0 + i] = sum / L[(a0 + i) * nv + a0 + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real sum = x[a0 + i];
    for (int k = i + 1; k < n; k++) sum -= L[(a0 + k) * nv + a0 + i] * x[a0 + k];
    x[a0 + i] = sum / L[(a0 + i) * nv + a0 + i];
  }
}

static int crb_and_factor(osim_t *s) {
  Model *m = &s->m;
  int nv = m->nv;
  memcpy(s->crb, s->cinert, sizeof(real) * 10 * m->nbody);
  for (int b = m->nbody - 1; b >= 1; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int k = 0; k < 10; k++) s->crb[10 * p + k] += s->crb[10 * b + k];
  }
  memset(s->M, 0, sizeof(real) * nv * nv);
  for (int i = 0; i < nv; i++) {
    real f[6];
    inert_mul(f, s->crb + 10 * m->dof_bodyid[i], s->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      const real *S = s->cdof + 6 * j;
      real v = S[0] * f[0] + S[1] * f[1] + S[2] * f[2] + S[3] * f[3] + S[4] * f[4] + S[5] * f[5];
      s->M[i * nv + j] = v; s->M[j * nv + i] = v;
    }
    s->M[i * nv + i] += m->dof_armature[i];
  }
  memset(s->L, 0, sizeof(real) * nv * nv);
  for (int t = 0; t < m->ntree; t++)
    if (chol_block(s->L, s->M, nv, m->tree_dofadr[t], m->tree_dofnum[t])) return -1;
  return 0;
}

static void solve_M(osim_t *s, real *x) {
  Model *m = &s->m;
  for (int t = 0; t < m->ntree; t++) chol_solve_block(s->L, x, m->nv, m->tree_dofadr[t], m->tree_dofnum[t]);
}

/* ------------------------------------------------------------------ P5 velocity, RNE bias, passive */
static void com_vel_and_bias(osim_t *s) {
  Model *m = &s->m;
  int nb = m->nbody;
  memset(s->cvel, 0, 6 * sizeof(real)); memset(s->cacc, 0, 6 * sizeof(real));
  s->cacc[3] = -m->gravity[0]; s->cacc[4] = -m->gravity[1]; s->cacc[5] = -m->gravity[2];
  for (int b = 1; b < nb; b++) {
    int p = m->body_parentid[b];
    real v[6], a[6];
    memcpy(v, s->cvel + 6 * p, 6 * sizeof(real)); memcpy(a, s->cacc + 6 * p, 6 * sizeof(real));
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, d = m->jnt_dofadr[j];
      int nd = m->jnt_type[j] == J_FREE ? 6 : 1;
      if (m->jnt_type[j] == J_FREE) {
        /* translations: constant axes */
        for (int k = 0; k < 3; k++) { memset(s->cdof_dot + 6 * (d + k), 0, 6 * sizeof(real)); for (int c = 0; c < 6; c++) v[c] += s->cdof[6 * (d + k) + c] * s->qvel[d + k]; }
        /* rotations: all three axes advance with the pre-rotation velocity (their own rotation
           contributes S x S = 0 in the sum) */
        for (int k = 3; k < 6; k++) cross_motion(s->cdof_dot + 6 * (d + k), v, s->cdof + 6 * (d + k));
        for (int k = 3; k < 6; k++) for (int c = 0; c < 6; c++) v[c] += s->cdof[6 * (d + k) + c] * s->qvel[d + k];
      } else {
        cross_motion(s->cdof_dot + 6 * d, v, s->cdof + 6 * d);
        for (int c = 0; c < 6; c++) v[c] += s->cdof[6 * d + c] * s->qvel[d];
      }
      for (int k = 0; k < nd; k++) for (int c = 0; c < 6; c++) a[c] += s->cdof_dot[6 * (d + k) + c] * s->qvel[d + k];
    }
    memcpy(s->cvel + 6 * b, v, 6 * sizeof(real)); memcpy(s->cacc + 6 * b, a, 6 * sizeof(real));
    real Ia[6], Iv[6], vxIv[6];
    inert_mul(Ia, s->cinert + 10 * b, a); inert_mul(Iv, s->cinert + 10 * b, v); cross_force(vxIv, v, Iv);
    for (int c = 0; c < 6; c++) s->cfrc[6 * b + c] = Ia[c] + vxIv[c];
  }
  for (int b = nb - 1; b >= 1; b--) {
    int p = m->body_parentid[b];
    if (p > 0) for (int c = 0; c < 6; c++) s->cfrc[6 * p + c] += s->cfrc[6 * b + c];
  }
  for (int d = 0; d < m->nv; d++) {
    const real *S = s->cdof + 6 * d, *f = s->cfrc + 6 * m->dof_bodyid[d];
    s->qfrc_bias[d] = S[0] * f[0] + S[1] * f[1] + S[2] * f[2] + S[3] * f[3] + S[4] * f[4] + S[5] * f[5];
    s->qfrc_passive[d] = -m->dof_damping[d] * s->qvel[d];
  }
}

/* velocity Jacobian column helpers: point p (world) on body b, for every ancestor dof */
static int jac_point_sparse(osim_t *s, int body, const real *p, const real *dir, real sign, int *idx, real *val, int n) {
  /* appends sign * dir . (S_lin + S_ang x p) for dofs in body's chain (merging duplicates) */
  Model *m = &s->m;
  int b = body;
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b <= 0) return n;
  for (int d = m->body_dofadr[b] + m->body_dofnum[b] - 1; d >= 0; d = m->dof_parentid[d]) {
    const real *S = s->cdof + 6 * d;
    real t[3];
    cross3(t, S, p);
    real v = sign * (dir[0] * (S[3] + t[0]) + dir[1] * (S[4] + t[1]) + dir[2] * (S[5] + t[2]));
    int k;
    for (k = 0; k < n; k++) if (idx[k] == d) { val[k] += v; break; }
    if (k == n) { idx[n] = d; val[n] = v; n++; }
  }
  return n;
}
static int jac_rot_sparse(osim_t *s, int body, const real *dir, real sign, int *idx, real *val, int n) {
  Model *m = &s->m;
  int b = body;
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b <= 0) return n;
  for (int d = m->body_dofadr[b] + m->body_dofnum[b] - 1; d >= 0; d = m->dof_parentid[d]) {
    const real *S = s->cdof + 6 * d;
    real v = sign * dot3(dir, S);
    int k;
    for (k = 0; k < n; k++) if (idx[k] == d) { val[k] += v; break; }
    if (k == n) { idx[n] = d; val[n] = v; n++; }
  }
  return n;
}

/* ------------------------------------------------------------------ P6 actuation */
static void actuation(osim_t *s) {
  Model *m = &s->m;
  memset(s->qfrc_actuator, 0, sizeof(real) * m->nv);
  for (int u = 0; u < m->nu; u++) {
    int j = m->actuator_jntid[u];
    real c = s->ctrl[u];
    if (m->actuator_ctrllimited[u]) { real lo = m->actuator_ctrlrange[2 * u], hi = m->actuator_ctrlrange[2 * u + 1]; c = c < lo ? lo : (c > hi ? hi : c); }
    real len = s->qpos[m->jnt_qposadr[j]] * m->actuator_gear[u], vel = s->qvel[m->jnt_dofadr[j]] * m->actuator_gear[u];
    const real *bp = m->actuator_bias + 3 * u;
    real f = m->actuator_gain[u] * c + bp[0] + bp[1] * len + bp[2] * vel;
    if (m->actuator_forcelimited[u]) { real lo = m->actuator_forcerange[2 * u], hi = m->actuator_forcerange[2 * u + 1]; f = f < lo ? lo : (f > hi ? hi : f); }
    s->actuator_force[u] = f;
    s->qfrc_actuator[m->jnt_dofadr[j]] += f * m->actuator_gear[u];
  }
}

#include "fsim_oracle_collide.inc"
#include "fsim_oracle_solve.inc"

/* ------------------------------------------------------------------ forward / step */
static int forward_impl(osim_t *s) {
  Model *m = &s->m;
  int nv = m->nv;
  kinematics(s);
  motion_axes(s);
  body_inertias(s);
  if (crb_and_factor(s)) return 1;
  collide(s);
  com_vel_and_bias(s);
  actuation(s);
  /* qfrc_smooth = passive - bias + applied + actuator + J^T xfrc (applied at body CoM) */
  for (int d = 0; d < nv; d++) s->qfrc_smooth[d] = s->qfrc_passive[d] - s->qfrc_bias[d] + s->qfrc_applied[d] + s->qfrc_actuator[d];
  for (int b = 1; b < m->nbody; b++) {
    const real *F = s->xfrc_applied + 6 * b;
    if (F[0] == 0 && F[1] == 0 && F[2] == 0 && F[3] == 0 && F[4] == 0 && F[5] == 0) continue;
    int idx[MAXNNZ]; real val[MAXNNZ]; int n;
    for (int c = 0; c < 3; c++) {
      real e[3] = {0, 0, 0}; e[c] = 1;
      n = jac_point_sparse(s, b, s->xipos + 3 * b, e, 1.0, idx, val, 0);
      for (int k = 0; k < n; k++) s->qfrc_smooth[idx[k]] += val[k] * F[c];
      n = jac_rot_sparse(s, b, e, 1.0, idx, val, 0);
      for (int k = 0; k < n; k++) s->qfrc_smooth[idx[k]] += val[k] * F[3 + c];
    }
  }
  memcpy(s->qacc_smooth, s->qfrc_smooth, sizeof(real) * nv);
  solve_M(s, s->qacc_smooth);
  make_constraints(s);
  solve_constraints(s);
  for (int d = 0; d < nv; d++) if (!isfinite(s->qacc[d]) || fabs(s->qacc[d]) > 1e10) return 1;
  return 0;
}

void osim_forward(osim_t *s) { forward_impl(s); }

int osim_step(osim_t *s) {
  Model *m = &s->m;
  int nv = m->nv;
  real h = m->timestep;
  for (int i = 0; i < m->nq; i++) if (!isfinite(s->qpos[i]) || fabs(s->qpos[i]) > 1e10) return 1;
  for (int i = 0; i < nv; i++) if (!isfinite(s->qvel[i]) || fabs(s->qvel[i]) > 1e10) return 2;
  if (forward_impl(s)) return 3;
  memcpy(s->qacc_warmstart, s->qacc, sizeof(real) * nv);
  /* semi-implicit Euler, joint damping treated implicitly: (M + h D) a' = M a */
  real *rhs = s->qfrc_smooth; /* reuse as scratch after the solve */
  for (int i = 0; i < nv; i++) {
    real sum = 0;
    int t = m->dof_treeid[i], a0 = m->tree_dofadr[t], n = m->tree_dofnum[t];
    for (int j = a0; j < a0 + n; j++) sum += s->M[i * nv + j] * s->qacc[j];
    rhs[i] = sum;
  }
  memcpy(s->Lh, s->M, sizeof(real) * nv * nv);
  for (int i = 0; i < nv; i++) s->Lh[i * nv + i] += h * m->dof_damping[i];
  {
    real *Lt = (real *)calloc((size_t)nv * nv, sizeof(real));
    for (int t = 0; t < m->ntree; t++) {
      if (chol_block(Lt, s->Lh, nv, m->tree_dofadr[t], m->tree_dofnum[t])) { free(Lt); return 4; }
      chol_solve_block(Lt, rhs, nv, m->tree_dofadr[t], m->tree_dofnum[t]);
    }
    free(Lt);
  }
  for (int i = 0; i < nv; i++) s->qvel[i] += h * rhs[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], d = m->jnt_dofadr[j];
    if (m->jnt_type[j] == J_FREE) {
      for (int k = 0; k < 3; k++) s->qpos[qa + k] += h * s->qvel[d + k];
      real w[3] = {s->qvel[d + 3], s->qvel[d + 4], s->qvel[d + 5]};
      real ang = norm3(w) * h;
      if (ang > 0) {
        real ax[3] = {w[0], w[1], w[2]}, ql[4], qn[4];
        normalize3(ax);
        axisangle_quat(ql, ax, ang);
        quat_mul(qn, s->qpos + qa + 3, ql);
        quat_norm(qn);
        memcpy(s->qpos + qa + 3, qn, 4 * sizeof(real));
      }
    } else
      s->qpos[qa] += h * s->qvel[d];
  }
  s->time_ += h;
  return 0;
}

/* ------------------------------------------------------------------ API */
#define ALLOCD(f, n) s->f = (real *)calloc((size_t)(n) + 1, sizeof(real))
osim_t *osim_create(const void *blob, size_t nbytes) {
  osim_t *s = (osim_t *)calloc(1, sizeof(osim_t));
  s->blob = malloc(nbytes);
  memcpy(s->blob, blob, nbytes);
  if (load_model(s, nbytes)) { for (int i = 0; i < s->nconv; i++) free(s->conv[i]); free(s->blob); free(s); return NULL; }
  Model *m = &s->m;
  int nb = m->nbody, nv = m->nv;
  ALLOCD(qpos, m->nq); ALLOCD(qvel, nv); ALLOCD(ctrl, m->nu); ALLOCD(qfrc_applied, nv); ALLOCD(xfrc_applied, 6 * nb);
  ALLOCD(qacc, nv); ALLOCD(qacc_warmstart, nv);
  ALLOCD(xpos, 3 * nb); ALLOCD(xquat, 4 * nb); ALLOCD(xmat, 9 * nb); ALLOCD(xipos, 3 * nb); ALLOCD(ximat, 9 * nb);
  ALLOCD(xanchor, 3 * m->njnt); ALLOCD(xaxis, 3 * m->njnt);
  ALLOCD(geom_xpos, 3 * m->ngeom); ALLOCD(geom_xmat, 9 * m->ngeom); ALLOCD(site_xpos, 3 * m->nsite); ALLOCD(site_xmat, 9 * m->nsite);
  ALLOCD(cinert, 10 * nb); ALLOCD(crb, 10 * nb); ALLOCD(cdof, 6 * nv); ALLOCD(cdof_dot, 6 * nv);
  ALLOCD(cvel, 6 * nb); ALLOCD(cacc, 6 * nb); ALLOCD(cfrc, 6 * nb);
  ALLOCD(M, nv * nv); ALLOCD(L, nv * nv); ALLOCD(Lh, nv * nv);
  ALLOCD(qfrc_bias, nv); ALLOCD(qfrc_passive, nv); ALLOCD(qfrc_actuator, nv); ALLOCD(qfrc_smooth, nv);
  ALLOCD(qacc_smooth, nv); ALLOCD(qfrc_constraint, nv); ALLOCD(actuator_force, m->nu);
  s->contact = (Contact *)calloc(MAXCON, sizeof(Contact));
  s->contact_geom1 = (int32_t *)calloc(MAXCON, sizeof(int32_t));
  s->contact_geom2 = (int32_t *)calloc(MAXCON, sizeof(int32_t));
  s->row = (Row *)calloc(MAXEFC, sizeof(Row));
  s->solver_iters = 100;
  s->solver_tol = 1e-8;
  osim_reset_data(s);
  return s;
}

void osim_destroy(osim_t *s) {
  if (!s) return;
  real **ds[] = {&s->qpos, &s->qvel, &s->ctrl, &s->qfrc_applied, &s->xfrc_applied, &s->qacc, &s->qacc_warmstart, &s->xpos, &s->xquat,
                   &s->xmat, &s->xipos, &s->ximat, &s->xanchor, &s->xaxis, &s->geom_xpos, &s->geom_xmat, &s->site_xpos, &s->site_xmat,
                   &s->cinert, &s->crb, &s->cdof, &s->cdof_dot, &s->cvel, &s->cacc, &s->cfrc, &s->M, &s->L, &s->Lh, &s->qfrc_bias,
                   &s->qfrc_passive, &s->qfrc_actuator, &s->qfrc_smooth, &s->qacc_smooth, &s->qfrc_constraint, &s->actuator_force};
  for (size_t i = 0; i < sizeof(ds) / sizeof(ds[0]); i++) free(*ds[i]);
  free(s->contact); free(s->contact_geom1); free(s->contact_geom2); free(s->row);
  free(s->m.eq_active); free(s->m.eq_data); free(s->m.dof_treeid); free(s->m.tree_dofadr); free(s->m.tree_dofnum);
  for (int i = 0; i < s->nconv; i++) free(s->conv[i]);
  free(s->blob); free(s);
}

void osim_reset_data(osim_t *s) {
  Model *m = &s->m;
  memcpy(s->qpos, m->qpos0, sizeof(real) * m->nq);
  memset(s->qvel, 0, sizeof(real) * m->nv); memset(s->ctrl, 0, sizeof(real) * m->nu);
  memset(s->qfrc_applied, 0, sizeof(real) * m->nv); memset(s->xfrc_applied, 0, sizeof(real) * 6 * m->nbody);
  memset(s->qacc, 0, sizeof(real) * m->nv); memset(s->qacc_warmstart, 0, sizeof(real) * m->nv);
  memset(s->qfrc_bias, 0, sizeof(real) * m->nv);
  s->time_ = 0; s->ncon = 0; s->nefc = 0; s->ndropped = 0;
}

real *osim_dptr(osim_t *s, const char *name, int *count) {
  Model *m = &s->m;
  struct { const char *n; real *p; int c; } tab[] = {
      {"qpos", s->qpos, m->nq}, {"qvel", s->qvel, m->nv}, {"ctrl", s->ctrl, m->nu}, {"qfrc_applied", s->qfrc_applied, m->nv},
      {"xfrc_applied", s->xfrc_applied, 6 * m->nbody}, {"qacc", s->qacc, m->nv}, {"qacc_warmstart", s->qacc_warmstart, m->nv},
      {"qfrc_bias", s->qfrc_bias, m->nv}, {"qfrc_constraint", s->qfrc_constraint, m->nv}, {"qfrc_actuator", s->qfrc_actuator, m->nv},
      {"qfrc_passive", s->qfrc_passive, m->nv}, {"qacc_smooth", s->qacc_smooth, m->nv}, {"actuator_force", s->actuator_force, m->nu},
      {"xpos", s->xpos, 3 * m->nbody}, {"xquat", s->xquat, 4 * m->nbody}, {"xmat", s->xmat, 9 * m->nbody}, {"xipos", s->xipos, 3 * m->nbody},
      {"geom_xpos", s->geom_xpos, 3 * m->ngeom}, {"geom_xmat", s->geom_xmat, 9 * m->ngeom}, {"site_xpos", s->site_xpos, 3 * m->nsite},
      {"site_xmat", s->site_xmat, 9 * m->nsite}, {"eq_data", m->eq_data, 7 * m->neq}, {"body_pos", m->body_pos, 3 * m->nbody},
      {"time", &s->time_, 1}, {"cvel", s->cvel, 6 * m->nbody}};
  for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); i++)
    if (strcmp(tab[i].n, name) == 0) { if (count) *count = tab[i].c; return tab[i].p; }
  snprintf(g_err, sizeof g_err, "no real field %s", name);
  return NULL;
}
int32_t *osim_iptr(osim_t *s, const char *name, int *count) {
  Model *m = &s->m;
  struct { const char *n; int32_t *p; int c; } tab[] = {
      {"geom_contype", m->geom_contype, m->ngeom}, {"geom_conaffinity", m->geom_conaffinity, m->ngeom}, {"eq_active", m->eq_active, m->neq},
      {"contact_geom1", s->contact_geom1, MAXCON}, {"contact_geom2", s->contact_geom2, MAXCON}, {"ncon", &s->ncon, 1}, {"nefc", &s->nefc, 1}, {"ndropped", &s->ndropped, 1}};
  for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); i++)
    if (strcmp(tab[i].n, name) == 0) { if (count) *count = tab[i].c; return tab[i].p; }
  snprintf(g_err, sizeof g_err, "no int field %s", name);
  return NULL;
}

void osim_site_vel(osim_t *s, int site, real *velp, real *velr) {
  /* mj_objectVelocity(site, flg_local=0) as read by furniture_sawyer.py:144-149: uses the body
     velocity left by the last forward pass */
  int b = s->m.site_bodyid[site];
  const real *v = s->cvel + 6 * b, *p = s->site_xpos + 3 * site;
  real t[3];
  cross3(t, v, p);
  velr[0] = v[0]; velr[1] = v[1]; velr[2] = v[2];
  velp[0] = v[3] + t[0]; velp[1] = v[4] + t[1]; velp[2] = v[5] + t[2];
}

void osim_body_jac(osim_t *s, int body, const real *point, real *jacp, real *jacr) {
  int nv = s->m.nv;
  memset(jacp, 0, sizeof(real) * 3 * nv); memset(jacr, 0, sizeof(real) * 3 * nv);
  for (int c = 0; c < 3; c++) {
    int idx[MAXNNZ]; real val[MAXNNZ];
    real e[3] = {0, 0, 0}; e[c] = 1;
    int n = jac_point_sparse(s, body, point, e, 1.0, idx, val, 0);
    for (int k = 0; k < n; k++) jacp[c * nv + idx[k]] = val[k];
    n = jac_rot_sparse(s, body, e, 1.0, idx, val, 0);
    for (int k = 0; k < n; k++) jacr[c * nv + idx[k]] = val[k];
  }
}
void osim_full_M(osim_t *s, real *M) { memcpy(M, s->M, sizeof(real) * s->m.nv * s->m.nv); }
void osim_set_solver(osim_t *s, int it, real tol) { s->solver_iters = it; s->solver_tol = tol; }
void osim_set_solver_kind(osim_t *s, int kind) { s->solver_kind = kind; }
int osim_last_solver_iters(osim_t *s) { return s->last_iters; }
/* development / diagnostics (scripts/dev/release_diag.py): the constraint rows of the last forward pass and the solver's objective
 * at a given acceleration -- which of two answers to one substep is the minimiser */
int osim_contact_row(osim_t *s, int i) { return (i >= 0 && i < s->ncon) ? s->contact[i].efc_address : -1; }
int osim_row_info(osim_t *s, int i, real *out8) {
  if (i < 0 || i >= s->nefc) return -1;
  const Row *r = &s->row[i];
  out8[0] = r->type; out8[1] = r->dim; out8[2] = r->aref; out8[3] = r->R; out8[4] = r->D; out8[5] = r->mu; out8[6] = r->pos - r->margin; out8[7] = r->force;
  return 0;
}
real osim_row_dot(osim_t *s, int i, const real *a) { return (i >= 0 && i < s->nefc) ? row_dot_a(&s->row[i], a) : 0.0; }
real osim_cost_at(osim_t *s, const real *qacc) {
  int nv = s->m.nv, ne = s->nefc;
  real *Mx = (real *)malloc(sizeof(real) * nv), *jar = (real *)malloc(sizeof(real) * (ne > 0 ? ne : 1));
  RowState *rs = (RowState *)malloc(sizeof(RowState) * (ne > 0 ? ne : 1));
  mul_M(s, Mx, qacc);
  for (int i = 0; i < ne; i++) jar[i] = row_dot_a(&s->row[i], qacc) - s->row[i].aref;
  real c = total_cost(s, qacc, Mx, jar, rs);
  free(Mx); free(jar); free(rs);
  return c;
}
/* signed distance of listed contact i (< 0: penetration), as data.contact[i].dist */
real osim_contact_dist(osim_t *s, int i) { return (i >= 0 && i < s->ncon) ? s->contact[i].dist : 0.0; }
