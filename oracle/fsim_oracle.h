/* fsim_oracle.h -- CPU restatement (double precision, one env) of the physics the
 * reference reaches through mujoco-py (sim.forward()/sim.step(), furniture/env/furniture.py:2857-2879).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in furniture_amd/ may link or call this;
 * it is the checker for the HIP path (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).
 *
 * PARITY UNPINNED: MuJoCo 2.0 (closed binary, un-vendored: README.md:44,
 * requirements.txt:12) is absent from /root/reference and from this image and the
 * reference ships no numeric tests for this path.  The pipeline
 * below restates MuJoCo's *published* computation (kinematics, CRB, RNE, soft
 * constraint model with solref/solimp impedance, elliptic cones, PGS dual solver,
 * semi-implicit Euler with implicit joint damping) and is pinned only by analytic
 * invariants (tests/test_oracle_physics.py), by the reference's MJCF geometry, and by ONE
 * trajectory recorded from MuJoCo itself: the first 62 frames of the reference's
 * demos/Cursor_7.pkl are replayed within 1.5 mm (tests/test_demo_replay.py).  That is
 * evidence, not a pin of the solver: the status stays PARITY UNPINNED.
 */
#ifndef FSIM_ORACLE_H
#define FSIM_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* `real` is the arithmetic type of the whole checker: double in libfsim_oracle.so / libfsim_cpu.so (THE checker), float in the
 * control build libfsim_cpu32.so (-DOSIM_REAL=float -fsingle-precision-constant, <tgmath.h>: the same source in fp32 throughout --
 * what `fp32 alone` does to a trajectory, scripts/divergence_control.py).  The Python binding (oracle/oracle_sim.py) is fp64 only. */
#ifndef OSIM_REAL
#define OSIM_REAL double
#endif
typedef OSIM_REAL real;

typedef struct osim osim_t;

osim_t *osim_create(const void *model_blob, size_t nbytes);
void osim_destroy(osim_t *);
const char *osim_last_error(void);

/* named views into the simulator's own memory (mujoco_py-style in-place access) */
real *osim_dptr(osim_t *, const char *name, int *count);
int32_t *osim_iptr(osim_t *, const char *name, int *count);

void osim_reset_data(osim_t *);          /* sim.reset(): qpos=qpos0, everything else 0 */
void osim_forward(osim_t *);             /* sim.forward() */
int osim_step(osim_t *);                 /* sim.step(); !=0 -> unstable (MujocoException analogue) */
void osim_site_vel(osim_t *, int site, real *velp3, real *velr3); /* data.site_xvelp/xvelr */
void osim_body_jac(osim_t *, int body, const real *point3, real *jacp_3xnv, real *jacr_3xnv);
void osim_full_M(osim_t *, real *M_nvxnv);

/* solver knobs: iterations, tolerance (<=0: fixed iterations), order (0 canonical) */
void osim_set_solver(osim_t *, int iterations, real tolerance);
int osim_last_solver_iters(osim_t *);
real osim_contact_dist(osim_t *, int i);
/* diagnostics: efc row of contact i; (type, dim, aref, R, D, mu, pos - margin, force) of row i; J_i . a; the Newton objective at qacc */
int osim_contact_row(osim_t *, int i);
int osim_row_info(osim_t *, int i, real *out8);
real osim_row_dot(osim_t *, int i, const real *a);
real osim_cost_at(osim_t *, const real *qacc);
void osim_set_solver_kind(osim_t *, int kind); /* 0 = PGS (dual), 1 = Newton (primal, MuJoCo default) */

#ifdef __cplusplus
}
#endif
#endif
