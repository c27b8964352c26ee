/* fsim.h -- C-ABI of the MI355X batched furniture-assembly simulator (libfsim.so).
 *
 * This is the drop-in boundary "B2" of SURVEY.md section 8(b): the operator surface the
 * reference env code sits on is mujoco_py.MjSim driven from Python
 * (furniture/env/furniture.py:1838 MjSim(...), :2877-2879 forward()/step() hot loop,
 * :3332-3379 action->ctrl, :1344-1387 + furniture_sawyer.py:103-155 observation,
 * :482-541 reward, :926-1042/:847-924 connector state machine, :1406-1663 reset).
 * The reference has no FFI of its own (it is pure Python over a closed binary), so the
 * entry points below are what a ctypes binding for that path binds; INTEGRATION.md shows
 * the stub.  Plain pointers and sizes only -- no torch types.  Device pointers are raw
 * HIP device addresses (e.g. torch.Tensor.data_ptr() of a tensor on the handle's device).
 *
 * Conventions: every call returns 0 on success or a negative code and leaves a message in
 * fsim_last_error(); the caller owns every buffer it passes; the library owns the per-env
 * state; all GPU work is enqueued on the handle's stream and is asynchronous unless stated.
 */
#ifndef FSIM_H
#define FSIM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fsim fsim_t;

enum {
  FSIM_OK = 0,
  FSIM_EINVAL = -1,   /* bad argument / malformed model blob */
  FSIM_ENOMEM = -2,   /* model does not fit the per-env LDS budget or hipMalloc failed */
  FSIM_EHIP = -3,     /* a HIP runtime call failed */
  FSIM_ENODEV = -4    /* no usable gfx950 device */
};

/* env configuration that the reference keeps in its argparse Namespace (furniture/config/furniture.py) */
typedef struct fsim_config {
  int32_t control_type;       /* 0 impedance -- `_do_simulation` on whatever actuators the compiled model has: the velocity-actuated robot for the
                                 reference's "impedance", the motor-actuated one (robot_torque.xml) for its "torque" (furniture.py:1268); 1 is not
                                 used --, and the torque-level arm controllers of
                                 furniture/env/controllers/arm_controller.py run per physics substep (furniture.py:41-47, 3065-3093;
                                 Sawyer, motor-actuated model): 2 position_orientation, 3 position, 4 joint_impedance,
                                 5 joint_velocity, 6 joint_torque.  Action = [arm command (6|3|7|7|7), grip, connect].
                                 7 ik (furniture.py:2899-2991): action = [dpos 3, rotation 3 (deg / rotate_speed), grip, connect]; a batched
                                 damped-least-squares solver stands in for pybullet.calculateInverseKinematics (parity unpinned),
                                 3 closed-loop repeats of 50 substeps per step (action_repeat, furniture.py:172).
                                 8 ik_quaternion (furniture.py:2994-3063): action = [dpos 3, quaternion wxyz 4, grip, connect], the quaternion
                                 is composed with the current hand orientation. */
  int32_t n_substeps;         /* int(control_timestep/model_timestep) = 50 (furniture.py:2878) */
  int32_t max_episode_steps;  /* config/furniture.py:163-168 */
  int32_t discrete_grip;      /* furniture_sawyer.py:72-74 */
  int32_t rescale_actions;    /* furniture.py:3333,3359 */
  int32_t auto_align;         /* furniture.py:881 */
  int32_t num_connect_steps;  /* 0 for arms, 10 for Cursor (furniture_cursor.py:28-32) */
  int32_t auto_reset;         /* SubprocVecEnv semantics: reset in place when done (subproc_vec_env.py:15-48) */
  int32_t solver_iterations;  /* Newton iteration cap (MuJoCo default 100) */
  int32_t reset_robot_after_attach; /* config/furniture.py:298-303 */
  float solver_tolerance;     /* scaled-gradient tolerance (MuJoCo default 1e-8 in fp64; fp32 floor ~1e-6) */
  float alignment_pos_dist, alignment_rot_dist_up, alignment_rot_dist_forward, alignment_project_dist;
  float ctrl_penalty_coef, unstable_penalty_coef, success_reward, touch_reward, pick_reward;
  float furn_xyz_rand, furn_rot_rand, agent_xyz_rand;
  float move_speed, rotate_speed, cursor_boundary; /* Cursor agent: config/furniture.py move_speed 0.05? see furniture_cursor.py; degrees per step; workspace half-extent */
  int32_t dense_reward;       /* 1: FurnitureSawyerDenseRewardEnv (furniture_sawyer_dense.py): the 8-phase reward replaces the sparse
                                 one; the tables must be uploaded with fsim_set_dense_reward before the first reset.  Sawyer only; control_type 0
                                 (impedance, the reference's dense config) or 7 / 8 (ik / ik_quaternion). */
  int32_t obs_bf16;           /* 1: the observation slab handed to fsim_step / fsim_reset is bfloat16 [n, obs_dim] (round to nearest even) --
                                 half the bytes of the per-step all-gather to the learner (BASELINE config 2).  State, reward and the whole
                                 computation stay float32; only the store of the finished observation is narrowed. */
  int32_t multi_wave;         /* which step kernel the handle runs -- decided once, here, because an env's arithmetic (summation order) depends on it:
                                 0 auto: 2 if the model has the kernel and n_envs <= 8 x the device's CU count (a launch with more envs than wave
                                   slots is throughput-bound and four waves per env only cost slots), else 1.  A function of n_envs and the device;
                                 1 off:  one wave per env in every launch (k_env_step);
                                 2 rule: ONE launch of persistent 4-wave workgroups (k_env_step_x): an env whose previous step took >= 150 Newton
                                   iterations -- its OWN state, nothing else -- is stepped by four cooperating waves, the others by one wave each;
                                 3 all:  four waves for every env (development / tests).
                                 Within one mode env i's results depend on env i's state and actions alone: not on the batch it is in, the slab
                                 layout, other handles or timing.  Across modes they agree like two fp32 implementations.  fsim_step_kernel() reports it. */
  int32_t lookahead_reset;    /* 1 (default): the reset of every env's NEXT episode is computed ahead of time from the reset table the host has already
                                 uploaded (fsim_set_reset_tables), a few dozen substeps per step launch, by the waves of that launch that have no env
                                 left to step, into a shadow record; the step / reset that ends the episode copies it in instead of running the
                                 301 / 401 reset substeps inside its launch.  Same loop on the same inputs: bit-identical to the in-launch reset,
                                 which remains the fallback whenever the shadow is not complete.  Not used with the arm controllers, ik /
                                 ik_quaternion, reset_robot_after_attach and multi_wave = all.  0: off. */
  int32_t overflow_restep;    /* 1 (default): a step or reset that needs more contact slots than the kernel's LDS image holds (48 / 64 / 128) is repeated
                                 up a ladder of layouts -- 64, 128, 512 slots -- before fsim_sync returns (fsim_overflow_resteps).  0: off -- the sticky
                                 report of FSIM_INFO_OVERFLOW is then all there is. */
} fsim_config_t;

void fsim_default_config(fsim_config_t *cfg);

/* model_blob: FSIMBLOB produced by furniture_amd.mjcf.model.CompiledModel.to_blob()  (replaces
 * load_model_from_xml + MjSim(model), furniture/env/models/base.py:113-115, furniture.py:1838) */
int fsim_create(const void *model_blob, size_t nbytes, int n_envs, int device, const fsim_config_t *cfg, fsim_t **out);
void fsim_destroy(fsim_t *);
const char *fsim_last_error(void);

/* dimensions the caller needs to size its buffers */
int fsim_dims(const fsim_t *, int32_t *nq, int32_t *nv, int32_t *nu, int32_t *dof_action, int32_t *obs_dim,
              int32_t *info_dim, int32_t *state_stride_words);

/* the handle's HIP stream (hipStream_t), so a collective can be enqueued behind a step */
int fsim_stream(fsim_t *, void **hip_stream);
int fsim_sync(fsim_t *);

/* ---- raw physics (sim.forward()/sim.step() on every env) -------------------------------- */
/* Run n physics substeps on the current state (data.ctrl/qfrc_applied/xfrc_applied as stored). */
int fsim_physics_step(fsim_t *, int n_substeps);
/* Derived quantities of the current state without integrating (sim.forward()). */
int fsim_physics_forward(fsim_t *);

/* ---- state access (sim.data.* / sim.model.* mutable fields), device pointers, [n_envs, dim] row-major.
 * Any pointer may be NULL to skip that field.  (get_env_state/set_env_state furniture.py:1781-1803, plus the
 * weld/mask/group state the reference's snapshot omits) */
typedef struct fsim_state_ptrs {
  float *qpos, *qvel, *qacc_warmstart, *qfrc_bias, *ctrl, *qfrc_applied, *xfrc_applied /* [n, nparts*6] */;
  float *eq_data /* [n, neq*7] */;
  int32_t *eq_active /* [n, neq] */, *geom_contype /* [n, ngeom] */, *geom_conaffinity /* [n, ngeom] */;
  int32_t *group /* [n, nparts] */;
  /* The `out only` fields below are results of the last fsim_physics_step / fsim_physics_forward launch: fsim_step and
   * fsim_reset do NOT refresh them (their forward passes live and die inside the fused kernel). */
  float *qacc /* out only */, *xpos /* out only: [n, nbody*3] */, *xquat /* out only: [n, nbody*4] */;
  int32_t *ncon /* out only: [n] */, *contact_geoms /* out only: [n, max_contacts*2], -1 padded */;
  int32_t *solver_iters /* out only: [n] Newton iterations of the last substep */;
  float *cursor /* Cursor agent only: [n, 8] = model.body_pos of cursor0, cursor1 (furniture.py:3139), then the selected part
                   index + 1 of each cursor (0 = none) as floats (furniture_cursor.py _cursor_selected) */;
  float *dense /* dense_reward handles only: [n, FSIM_DENSE_STATEW] the reward state machine's variables
                  (furniture_sawyer_dense.py:128-216: subtask, phase, flags, fine-aligned count, init table-site / leg / lift / eef
                  positions, the eleven _prev_* distances), so that a snapshot restores the reward too */;
  int32_t *env_block /* [n, fsim_env_block_words()] the env-logic variables of the record as opaque words (episode length and reward,
                        _num_connected, connected-site set, touched/picked latches, pending connect, groups, Cursor / dense
                        blocks): with qpos, qvel, qacc_warmstart, eq_*, geom_* this makes get/set_state a complete snapshot */;
} fsim_state_ptrs_t;
int fsim_get_state(fsim_t *, const fsim_state_ptrs_t *dst);
int fsim_set_state(fsim_t *, const fsim_state_ptrs_t *src);
int fsim_max_contacts(const fsim_t *);
/* Which step kernel the handle runs: "generic" (run-time layout, any model) or the name of a kernel specialised at build
 * time for this (agent, furniture, config) -- same arithmetic, layout offsets as instruction immediates (csrc/fsim_spec.hpp). */
const char *fsim_kernel_variant(const fsim_t *);
/* Which launch structure fsim_step uses (fsim_config_t::multi_wave as resolved at fsim_create): "k_env_step_x (...)" / "k_env_step (...)". */
const char *fsim_step_kernel(const fsim_t *);
int fsim_env_block_words(const fsim_t *);

/* ---- the env hot path ------------------------------------------------------------------- */
/* Initial placements for the next reset of each env: part poses [n, nparts*7] (pos, quat wxyz) as drawn by
 * the reference's UniformRandomSampler (tasks/placement_sampler.py:138-190) and robot joint noise
 * [n, n_noise, narmjoints] as drawn by _initialize_robot_pos (furniture.py:1761-1779), n_noise = 101.
 * Host pointers; copied on a transfer stream of the handle and complete on return.  mask: host uint8 [n] or NULL = all.  No launch of
 * THIS handle may be in flight: besides the terminal envs' resets, the look-ahead jobs of a step launch read the tables of any env
 * whose episode is old enough -- the call waits for a step still in flight itself (as fsim_sync would).  Other handles (the other
 * slabs of an asynchronously stepped batch) keep running. */
int fsim_set_reset_tables(fsim_t *, const uint8_t *mask, const float *part_qpos, const float *robot_noise, int n_noise);

/* config.reset_robot_after_attach (config/furniture.py:298-303; furniture.py:919-925: `_connect` ends with `_initialize_robot_pos()`):
 * the joint noise [n, narmjoints] the NEXT attach of each env adds to the arm's initial pose -- one `_init_random(.., "agent")` draw of
 * the env's RandomState, taken BETWEEN the draws of two resets, which is why the host keeps it waiting here and advances the env's
 * stream when FSIM_INFO_CONNECTED_THIS_STEP reports that the kernel used it (furniture_amd/envs.py).  Host pointers, copied on the
 * handle's transfer stream, complete on return; mask as above.  Only for handles created with reset_robot_after_attach = 1. */
int fsim_set_attach_noise(fsim_t *, const uint8_t *mask, const float *noise);

/* FurnitureEnv.reset() on the masked envs (device uint8 mask or NULL = all); writes obs if non-NULL. */
/* FurnitureEnv.set_init_qpos (furniture.py:315-316; applied inside _reset, :1505-1519, 1568, 1617): the resets of the masked envs
 * (host uint8 [n], NULL = all) start from the given state -- qpos [n][nq], qvel [n][nv], host float32 (the format of
 * get_env_state) -- instead of sampling a placement, settling the parts and initialising the robot; no reset table is consumed.
 * qpos = NULL: back to sampled resets (set_init_qpos(None)). */
int fsim_set_init_state(fsim_t *, const uint8_t *mask, const float *qpos, const float *qvel);

int fsim_reset(fsim_t *, const uint8_t *mask_dev, void *obs_dev /* float32 | bfloat16, as fsim_step */);

/* FurnitureEnv.step(action) on every env.  action [n, dof_action] float32, obs [n, obs_dim] float32,
 * reward [n] float32, done [n] uint8, info [n, info_dim] int32/float bits -- all device pointers.
 * info columns: see FSIM_INFO_* below. */
int fsim_step(fsim_t *, const float *action_dev, void *obs_dev /* float32, or bfloat16 with cfg.obs_bf16 */, float *reward_dev, uint8_t *done_dev,
              int32_t *info_dev);

/* Device -> host copy of a caller buffer (e.g. the info block of the last step) on the handle's transfer stream; complete on return.
 * With a pinned destination this is a plain DMA transfer: no kernel, no allocation.  The handle's in-flight step is waited for first
 * (as by fsim_sync, overflow re-step included), so the rows read are the step's final ones. */
int fsim_read(fsim_t *, void *host_dst, const void *dev_src, size_t nbytes);

/* Number of envs whose FSIM_INFO_NEEDS_TABLE is set by the last fsim_step, valid once that step has completed (fsim_sync): lets the
 * host skip the scan of the info block on the (many) steps in which no episode ended. */
int fsim_tables_needed(const fsim_t *);

/* Look-ahead reset bookkeeping (fsim_config_t::lookahead_reset): out[6] = { enabled, reset units (= physics substeps of a reset) run by
 * look-ahead jobs so far, resets taken from a shadow record, resets executed inside a step / reset launch, look-ahead jobs per launch,
 * reset units per job }.  The device-side counters are valid once the launches that bumped them have completed (fsim_sync).  Every
 * reset still costs its 301 / 401 substeps; the counters say where they ran. */
int fsim_lookahead_stats(fsim_t *, int64_t *out);

/* Contact-overflow re-step (no reference counterpart: MuJoCo's contact arrays are sized nconmax = 5000, `base.xml`).  Models that run on
 * the default 48 contact slots (what lets eight envs share a CU's LDS) occasionally need more: Sawyer + table_lack_0825 about 1.6 times
 * per million env-steps.  Every step launch keeps each env's pre-step record and lists the envs that dropped contacts; fsim_sync() steps
 * those again from the kept record with a 64-slot layout (one four-wave workgroup per env) and overwrites their record and output rows
 * before it returns.  Reset launches are covered the same way, and the re-step is a LADDER: an env that overflows 64 slots too is repeated with
 * 128 (one-wave kernel, two slots per lane), and one that overflows 128 -- the furniture with eleven and more planks, which the reference's
 * sampler places inside each other: 240 - 370 simultaneous contacts and every part in one island of up to 84 dofs while the reset throws them
 * apart -- with 512 (eight slots per lane; islands of more than 64 dofs factored in LDS).  Models start on the rung their size gives them (48,
 * 64 or 128 slots).  Returns how many env-steps were repeated so far (one count per rung taken).  FSIM_NO_OVERFLOW_REDO=1 switches it off (the sticky report of
 * FSIM_INFO_OVERFLOW is then all there is).  Outputs read in stream order without fsim_sync carry the first pass's rows.
 * Buffer lifetime: the action and output buffers of a step must stay valid (and the action unchanged) until the next fsim_sync or
 * launch of the handle -- the re-step reads and writes them again; every setter of the handle waits for it first. */
int64_t fsim_overflow_resteps(const fsim_t *);

/* (Round 5's shared work pool -- fsim_pool_*: a resident kernel that several handles posted their steps to -- measured 20 % slower
 * than a launch per slab-step (663 / 634 k against 828 k env-steps/s, profiles/r05_b_bench_pool_*) and left the library in round 6;
 * the code is in the history at commit 71a76b7.) */

/* FurnitureEnv.set_max_episode_steps (furniture.py:312-313, forwarded by FurnitureGym :46-48): takes effect from the next step. */
int fsim_set_max_episode_steps(fsim_t *, int max_episode_steps);

enum {
  FSIM_INFO_NUM_CONNECTED = 0, FSIM_INFO_SUCCESS = 1, FSIM_INFO_FAIL = 2, FSIM_INFO_LAST_SITE1 = 3,
  FSIM_INFO_LAST_SITE2 = 4, FSIM_INFO_EPISODE_LENGTH = 5, FSIM_INFO_CONNECTED_THIS_STEP = 6,
  FSIM_INFO_NEEDS_TABLE = 7, /* env consumed its reset table this step: upload the next one (fsim_set_reset_tables with a mask).
                              * 2 = after DROPPING one pass of the env's reset-time RNG stream: an unstable simulation resets inside
                              * step() (furniture.py:2889-2897) and the vec-env worker resets again; only the second reset is run */
  FSIM_INFO_SUCCESS_REWARD_F = 8, FSIM_INFO_TOUCH_REWARD_F = 9, FSIM_INFO_PICK_REWARD_F = 10,
  FSIM_INFO_CTRL_PENALTY_F = 11, /* float bits */
  FSIM_INFO_OVERFLOW = 12, /* bits 0-1: this launch -- bit 0 broadphase survivor list truncated, bit 1 contact slots exhausted (contacts dropped) or an island of more than 64 dofs met on a kernel without the LDS-resident factorisation;
                              bits 8-9: the same two flags, STICKY: raised by any launch of this env so far (a step, a reset, the look-ahead
                              reset that was copied in) and kept in its record across resets, so a host that reads the block every k-th step
                              misses nothing */
  FSIM_INFO_DENSE_PHASE = 13, /* dense-reward env: info["phase_i"] = phase + 8 * subtask (furniture_sawyer_dense.py:347); then
                                 FSIM_INFO_SUCCESS_REWARD_F carries info["phase_bonus"] and the other *_F columns are 0 */
  FSIM_INFO_EPISODE_REWARD_F = 14, /* float bits: the episode's reward so far incl. this step (step_log["episode_reward"] at done,
                                      furniture.py:468-470) */
  FSIM_INFO_SUBTASK1 = 15, FSIM_INFO_SUBTASK2 = 16, /* _subtask_part1 / _subtask_part2 of the state the returned observation describes
                                                       (part indices, -1 = none; furniture.py:2723-2736) -- what object_ob_all=False
                                                       and subtask_ob=True select / report (furniture.py:1360-1385) */
  FSIM_INFO_DIM = 17
};

/* ---- dense-reward env (FurnitureSawyerDenseRewardEnv) -------------------------------------- */
/* coef: FSIM_DENSE_NCOEF floats = the config/furniture_sawyer_dense.py coefficients + z_finedist + the griptip/grip site ids + phase_ob, in
 * the order of furniture_amd/dense.py DENSE_COEF_DEFAULTS; subtasks: [nsub][FSIM_DENSE_SUBW] floats, one row per recipe step
 * (furniture_sawyer_dense.py:149-216: leg/table part, leg/table connector site, grasp-target sites, angle (NaN = None), ...).
 * Host pointers, copied before return. */
enum { FSIM_DENSE_NCOEF = 34, FSIM_DENSE_SUBW = 16, FSIM_DENSE_OBSW = 40, FSIM_DENSE_STATEW = 27 };
int fsim_set_dense_reward(fsim_t *, const float *coef, int ncoef, const float *subtasks, int nsub);

/* ---- pre-assembled starts: FurnitureEnv.set_subtask / config.preassembled / config.num_connects ------------------------
 * (furniture.py:163, 204-207, 1476-1503, 1542-1566).  Applies to every reset that follows, for all envs of the handle.
 *   ids[n_pre]           the reference's `preassembled` list: weld equality ids for a furniture WITHOUT a recipe file (the welds are
 *                        switched on and the part groups merged before the parts are placed), recipe step indices for one WITH a
 *                        recipe (models/assets/recipes/<name>.yaml), in which case the caller also passes, per listed step,
 *   conn_pairs[n_pre][2] the connector indices (rows of the model's connector table) of the recipe's site2 and site1 -- the
 *                        arguments of the reset's _connect(site2_id, site1_id) -- and
 *   angles[n_pre]        the recipe's angle in degrees, NaN for none (_project_connector_quat);
 *   num_connects         config.num_connects: success when num_connected == num_connects + n_pre; < 0 = None (all parts).
 * conn_pairs = NULL: ids are weld ids whatever the furniture (config.assembled, furniture.py:1502-1503: all welds on, the host leaves
 * the parts at the XML's assembled poses).
 * n_pre = 0 restores the default.  Not combined with fsim_set_init_state (either call then returns FSIM_EINVAL).  Host pointers. */
int fsim_set_preassembled(fsim_t *, int n_pre, const int32_t *ids, const int32_t *conn_pairs, const float *angles, int num_connects);

/* Parity hook: run the device implementation of the reward state machine alone, on recorded sensor values (the layout of
 * oracle/dense_reward.py O_*: obs0 [nsub][40] at reset, obs [T][nsub][40], ac [T][dof], connected [T]); out_reward [T],
 * out_flags [T][4] = done, success, phase, subtask after each step.  Host pointers; synchronous; needs no handle. */
int fsim_dense_replay(int device, const float *coef, int ncoef, const float *subtasks, int nsub, int n_pre, const float *obs0,
                      const float *obs, const float *ac, int dof, const uint8_t *connected, int T, float *out_reward,
                      int32_t *out_flags);

/* ---- env-logic replay (parity hooks, like fsim_dense_replay): the device functions of the connector state machine that run
 * inside fsim_step, fed with recorded inputs so that the reference's golden vectors can be checked against them directly.
 * Host pointers; synchronous.
 *   fsim_replay_is_aligned   FurnitureEnv._is_aligned (furniture.py:1057-1153) on n site-pose pairs: p [n][3], R [n][9] row-major
 *                            world rotation of the site, nang [n] allowed forward angles of site 1 (0 = any), angles [n][4] degrees;
 *                            out_ok [n], out_tq [n][4] = _target_connector_xquat (wxyz; NaN where the reference leaves it unset).
 *   fsim_replay_try_connect  the search + _connect_step bookkeeping of _try_connect (furniture.py:926-1042) on the handle's connector /
 *                            weld tables: part12 [n][2] (part2 = -1: any), group [n][nparts] union-find parent table, used [n][nconn]
 *                            connected-site flags, aligned [n][nconn][nconn] recorded outcome of _is_aligned per connector pair,
 *                            step_in [n]; out [n][5] = connector index 1, connector index 2 (-1: none), return value, _connect_step
 *                            after the call, part moved by the approach phase (-1: none).
 *   fsim_replay_touch_scan   finger-touch masks (furniture.py:500-513, 1298-1322) + the connect scan of _step_continuous
 *                            (furniture.py:1290-1330) on n contact lists: ncon [n], geoms [n][maxc][2] COLLIDING-geom indices,
 *                            script [n][4] outcome of the k-th _try_connect; out_masks [n][3] = left / right finger touch bits
 *                            (arm * 16 + part), floor touch bits; out_tried [n][4] parts tried in order, -1 padded. */
int fsim_replay_is_aligned(int device, float pos_dist, float rot_up, float rot_fwd, float proj_dist, int n, const float *p1, const float *R1,
                           const float *p2, const float *R2, const int32_t *nang, const float *angles, int32_t *out_ok, float *out_tq);
int fsim_replay_try_connect(fsim_t *, int n, int num_connect_steps, const int32_t *part12, const int32_t *group, const int32_t *used,
                            const uint8_t *aligned, const int32_t *step_in, int32_t *out);
int fsim_replay_touch_scan(fsim_t *, int n, int maxc, const int32_t *ncon, const int32_t *geoms, const uint8_t *script, int32_t *out_masks,
                           int32_t *out_tried);

/* timing helper for bench.py: average device time (ms) of the fsim_step kernel launches since the previous call, measured with
 * HIP events on the handle's stream; resets the accumulator.  Timing is OFF until the first call (which returns n = 0): while
 * it is on, each fsim_step / fsim_reset first waits for the end event of the handle's previous launch, i.e. a launch is then
 * no longer fully asynchronous with respect to the one before it. */
int fsim_kernel_time_ms(fsim_t *, double *avg_ms, int32_t *n_launches);

#ifdef __cplusplus
}
#endif
#endif
