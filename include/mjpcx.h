/* mjpcx.h -- C ABI of the MI355X rollout-and-evaluate library (libmjpcx.so).
 *
 * This is the drop-in boundary for the batched rollout hot path of
 * google-deepmind/mujoco_mpc: everything the reference does between
 * `SamplingPlanner::Rollouts` (mjpc/planners/sampling/planner.cc:355-393) and
 * the `partial_sort` on total_return (planner.cc:184-188), i.e. N x
 * `Trajectory::Rollout` (mjpc/trajectory.cc:92-210) + `UpdateReturn`
 * (trajectory.cc:312-326), runs behind these entry points on one gfx950 GPU.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success, a negative MJPCX_E* code otherwise;
 *     no exceptions cross the boundary; `mjpcx_error_string` maps codes.
 *   - the caller owns all host buffers; the library owns device buffers inside
 *     the opaque context and deep-copies model/task at create time.
 *   - calls on ONE context are serialised by the caller (the planning thread,
 *     cf. mjpc/agent.cc:283-357); different contexts are independent.
 *   - all host-side reals are fp64 (mjtNum=double in the reference), whatever
 *     precision the device kernels compute in.
 */
#ifndef MJPCX_H_
#define MJPCX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJPCX_VERSION 1

/* ---- error codes ------------------------------------------------------- */
enum {
  MJPCX_OK = 0,
  MJPCX_EINVAL = -1,       /* bad argument / size mismatch                   */
  MJPCX_EUNSUPPORTED = -2, /* model feature / size has no device kernel      */
  MJPCX_EDEVICE = -3,      /* HIP runtime error (see mjpcx_last_hip_error)   */
  MJPCX_ENOMEM = -4,       /* host or device allocation failed               */
  MJPCX_ESTATE = -5,       /* call out of order (e.g. fetch before rollout)  */
};

/* ---- enums shared with the reference ------------------------------------ */
/* mjtJoint (MuJoCo) */
enum { MJPCX_JNT_FREE = 0, MJPCX_JNT_BALL = 1, MJPCX_JNT_SLIDE = 2, MJPCX_JNT_HINGE = 3 };
/* mjtIntegrator (MuJoCo). Euler (with implicit joint damping) and RK4 are integrated as such. IMPLICITFAST is accepted for models whose
 * only velocity-dependent smooth force is joint damping (no actuator with a velocity term in its bias, which is all this ABI carries
 * besides damping): there MuJoCo's M - h dqfrc_smooth/dqvel is M + h diag(damping), i.e. mj_implicit's update IS mj_Euler's -- the
 * context integrates with the Euler path. IMPLICIT (Coriolis derivatives) and other IMPLICITFAST models are refused by mjpcx_create. */
enum { MJPCX_INT_EULER = 0, MJPCX_INT_RK4 = 1, MJPCX_INT_IMPLICIT = 2, MJPCX_INT_IMPLICITFAST = 3 };
/* mjpc::spline::SplineInterpolation, mjpc/spline/spline.h:29-33 */
enum { MJPCX_SPLINE_ZERO = 0, MJPCX_SPLINE_LINEAR = 1, MJPCX_SPLINE_CUBIC = 2 };
/* mjpc::NormType, mjpc/norm.h:24-35 */
enum {
  MJPCX_NORM_NULL = -1, MJPCX_NORM_QUADRATIC = 0, MJPCX_NORM_L22 = 1, MJPCX_NORM_L2 = 2,
  MJPCX_NORM_COSH = 3, MJPCX_NORM_POWER_LOSS = 5, MJPCX_NORM_SMOOTH_ABS = 6,
  MJPCX_NORM_SMOOTH_ABS2 = 7, MJPCX_NORM_RECTIFY = 8,
};
/* actuator gain / bias types (mjtGain / mjtBias subset) */
enum { MJPCX_GAIN_FIXED = 0 };
enum { MJPCX_BIAS_NONE = 0, MJPCX_BIAS_AFFINE = 1 };
/* opt.disableflags bits used by the hot path (MuJoCo mjtDisableBit subset) */
enum {
  MJPCX_DSBL_CONSTRAINT = 1 << 0, MJPCX_DSBL_FRICTIONLOSS = 1 << 2, MJPCX_DSBL_LIMIT = 1 << 3, MJPCX_DSBL_CONTACT = 1 << 4,
  MJPCX_DSBL_PASSIVE = 1 << 5, MJPCX_DSBL_GRAVITY = 1 << 6, MJPCX_DSBL_CLAMPCTRL = 1 << 7,
  MJPCX_DSBL_ACTUATION = 1 << 10, MJPCX_DSBL_REFSAFE = 1 << 11, MJPCX_DSBL_EULERDAMP = 1 << 14,
};

/* Device residual functions = the `ResidualFn::Residual` overrides of the
 * reference tasks, resolved to an id at task-bake time. */
enum {
  MJPCX_RESIDUAL_PARTICLE = 1,      /* mjpc/test/testdata/particle_residual.h:33-43   */
  MJPCX_RESIDUAL_PARTICLE_COPY = 2, /* mjpc/test/agent/rollout_test.cc:37-42          */
  MJPCX_RESIDUAL_CARTPOLE = 3,      /* mjpc/tasks/cartpole/cartpole.cc:36-49          */
  MJPCX_RESIDUAL_QUADRUPED_FLAT = 4,/* mjpc/tasks/quadruped/quadruped.cc:33-226       */
  MJPCX_RESIDUAL_HUMANOID_TRACK = 5,/* mjpc/tasks/humanoid/tracking/tracking.cc:94-216 */
};

/* ---- flat model ("mjModel" subset, compiled) ----
 * Array sizes are given next to each pointer. All arrays are host memory,
 * row-major, and are deep-copied by mjpcx_create. */
typedef struct mjpcx_model {
  /* sizes */
  int32_t nq, nv, nu, na, nbody, njnt, nsite, nmocap, nuserdata;
  /* options (mjOption) */
  double timestep;
  double gravity[3];
  int32_t integrator;   /* MJPCX_INT_*                                        */
  int32_t disableflags; /* MJPCX_DSBL_* bits                                  */
  int32_t solver_iterations; /* opt.iterations (default 100)                  */
  double solver_tolerance;   /* opt.tolerance (default 1e-8)                  */
  double meaninertia;        /* stat.meaninertia                              */
  /* bodies (nbody, body 0 = world) */
  const int32_t* body_parentid; /* nbody */
  const int32_t* body_rootid;   /* nbody */
  const int32_t* body_jntnum;   /* nbody */
  const int32_t* body_jntadr;   /* nbody (-1: none) */
  const int32_t* body_dofnum;   /* nbody */
  const int32_t* body_dofadr;   /* nbody (-1: none) */
  const int32_t* body_mocapid;  /* nbody (-1: not mocap) */
  const double* body_pos;       /* nbody x 3 */
  const double* body_quat;      /* nbody x 4 */
  const double* body_ipos;      /* nbody x 3 */
  const double* body_iquat;     /* nbody x 4 */
  const double* body_mass;      /* nbody */
  const double* body_inertia;   /* nbody x 3 */
  /* joints (njnt) */
  const int32_t* jnt_type;    /* njnt */
  const int32_t* jnt_qposadr; /* njnt */
  const int32_t* jnt_dofadr;  /* njnt */
  const int32_t* jnt_bodyid;  /* njnt */
  const int32_t* jnt_limited; /* njnt */
  const double* jnt_pos;      /* njnt x 3 */
  const double* jnt_axis;     /* njnt x 3 */
  const double* jnt_stiffness;/* njnt */
  const double* jnt_range;    /* njnt x 2 */
  const double* jnt_margin;   /* njnt */
  const double* jnt_solref;   /* njnt x 2 */
  const double* jnt_solimp;   /* njnt x 5 */
  /* dofs (nv) */
  const int32_t* dof_bodyid;   /* nv */
  const int32_t* dof_jntid;    /* nv */
  const int32_t* dof_parentid; /* nv (-1: none) */
  const double* dof_armature;  /* nv */
  const double* dof_damping;   /* nv */
  const double* dof_frictionloss; /* nv */
  const double* dof_invweight0;   /* nv */
  /* reference configuration */
  const double* qpos0;       /* nq */
  const double* qpos_spring; /* nq */
  /* sites (nsite) */
  const int32_t* site_bodyid; /* nsite */
  const double* site_pos;     /* nsite x 3 */
  const double* site_quat;    /* nsite x 4 */
  /* actuators (nu): joint transmission only */
  const int32_t* actuator_trnid;      /* nu: joint id */
  const int32_t* actuator_gaintype;   /* nu */
  const int32_t* actuator_biastype;   /* nu */
  const int32_t* actuator_ctrllimited;  /* nu */
  const int32_t* actuator_forcelimited; /* nu */
  const double* actuator_gear;       /* nu (gear[0]) */
  const double* actuator_gainprm;    /* nu x 3 */
  const double* actuator_biasprm;    /* nu x 3 */
  const double* actuator_ctrlrange;  /* nu x 2 */
  const double* actuator_forcerange; /* nu x 2 */
  /* ---- contacts and constraint options (all zero / NULL for contact-free models) ---- */
  int32_t ngeom;
  int32_t nkey;
  int32_t cone;            /* opt.cone: 0 pyramidal, 1 elliptic                */
  double impratio;         /* opt.impratio                                     */
  const int32_t* geom_type;        /* ngeom, MJPCX_GEOM_* (= mjtGeom)          */
  const int32_t* geom_bodyid;      /* ngeom */
  const int32_t* geom_contype;     /* ngeom */
  const int32_t* geom_conaffinity; /* ngeom */
  const int32_t* geom_condim;      /* ngeom */
  const int32_t* geom_priority;    /* ngeom */
  const int32_t* geom_group;       /* ngeom */
  const double* geom_size;     /* ngeom x 3 */
  const double* geom_pos;      /* ngeom x 3 (body frame) */
  const double* geom_quat;     /* ngeom x 4 */
  const double* geom_friction; /* ngeom x 3 (sliding, torsional, rolling) */
  const double* geom_solref;   /* ngeom x 2 */
  const double* geom_solimp;   /* ngeom x 5 */
  const double* geom_margin;   /* ngeom */
  const double* geom_gap;      /* ngeom */
  const double* geom_solmix;   /* ngeom */
  const double* body_invweight0;  /* nbody x 2 (translational, rotational) */
  const double* body_subtreemass; /* nbody */
  const double* dof_solref;       /* nv x 2 (friction loss) */
  const double* dof_solimp;       /* nv x 5 */
  const double* key_qpos;         /* nkey x nq (residuals that read keyframes) */
  /* ---- fixed tendons (limits only), body-pair collision filter, mocap keyframes: the Humanoid class of models.
   * All zero / NULL for models without them. Spatial tendons are not supported (mjpcx_create: MJPCX_EUNSUPPORTED). */
  int32_t ntendon;
  int32_t nwrap;
  int32_t nexclude;
  const int32_t* tendon_adr;        /* ntendon: first wrap entry                                   */
  const int32_t* tendon_num;        /* ntendon: number of wrap entries                             */
  const int32_t* tendon_limited;    /* ntendon                                                     */
  const int32_t* wrap_objid;        /* nwrap: joint id (mjWRAP_JOINT entries only)                 */
  const double* wrap_prm;           /* nwrap: coefficient                                          */
  const double* tendon_range;       /* ntendon x 2                                                 */
  const double* tendon_margin;      /* ntendon                                                     */
  const double* tendon_solref_lim;  /* ntendon x 2                                                 */
  const double* tendon_solimp_lim;  /* ntendon x 5                                                 */
  const double* tendon_invweight0;  /* ntendon                                                     */
  const int32_t* exclude_signature; /* nexclude: (body1 << 16) + body2, body1 < body2 (mjModel)    */
  const int32_t* body_weldid;       /* nbody: the body this one is welded to (mjModel.body_weldid) */
  const double* key_mpos;           /* nkey x nmocap x 3 (Humanoid tracking residual)              */
} mjpcx_model;

/* moving-geom pairs (self-collision) are built for sphere and capsule geoms; other pairs are skipped */
enum { MJPCX_GEOM_PLANE = 0, MJPCX_GEOM_SPHERE = 2, MJPCX_GEOM_CAPSULE = 3, MJPCX_GEOM_CYLINDER = 5, MJPCX_GEOM_BOX = 6 };

/* ---- task / cost specification (mjpc::Task after Task::Reset,
 * mjpc/task.cc:147-248, plus the frozen ResidualFn copy of agent.cc:319) ---- */
#define MJPCX_MAX_COST_TERMS 128 /* kMaxCostTerms, mjpc/task.h:31 */
typedef struct mjpcx_task {
  int32_t residual_id;  /* MJPCX_RESIDUAL_*                                   */
  int32_t num_residual; /* Task::num_residual                                 */
  int32_t num_term;     /* Task::num_term                                     */
  int32_t num_trace;    /* Task::num_trace                                    */
  int32_t num_parameter;/* Task::parameters.size()                            */
  const int32_t* dim_norm_residual;  /* num_term */
  const int32_t* norm;               /* num_term, MJPCX_NORM_* */
  const int32_t* num_norm_parameter; /* num_term */
  const double* weight;              /* num_term */
  const double* norm_parameter;      /* sum(num_norm_parameter) */
  const double* parameters;          /* num_parameter (residual_* numerics) */
  const int32_t* trace_site;         /* num_trace: site id of sensor "trace%i", or -1 - body id for a body frame */
  double risk;                       /* Task::risk */
  /* task-specific frozen ResidualFn state (the members Transition manages and Reset resolves, e.g.
   * QuadrupedFlat::ResidualFn, quadruped.h:186-246); layout documented per residual in csrc/residuals.h */
  int32_t num_residual_int;
  int32_t num_residual_real;
  const int32_t* residual_int;
  const double* residual_real;
} mjpcx_task;

/* ---- outputs ---------------------------------------------------------------
 * One candidate trajectory in the reference's own layout (mjpc::Trajectory,
 * mjpc/trajectory.h:74-86): row-major by time. Any pointer may be NULL to skip
 * that buffer. */
typedef struct mjpcx_traj_view {
  int32_t horizon;   /* in: capacity in steps; out: trajectory length         */
  double* states;    /* horizon x dim_state                                   */
  double* actions;   /* horizon x nu                                          */
  double* times;     /* horizon                                               */
  double* residual;  /* horizon x num_residual                                */
  double* costs;     /* horizon                                               */
  double* trace;     /* horizon x 3*num_trace                                 */
  double total_return; /* out */
  int32_t failure;     /* out */
} mjpcx_traj_view;

/* Noise specification for device-side candidate generation: the counter-based
 * replacement of the function-local absl::BitGen in
 * SamplingPlanner::AddNoiseToPolicy (sampling/planner.cc:326-352) and
 * CrossEntropyPlanner::AddNoiseToPolicy (cross_entropy/planner.cc:351-385).
 * Philox4x32-10, key = (seed_lo, seed_hi), counter = (i, c, iteration, stream)
 * for global candidate i. stream 0: the four words of counter c give two 53-bit
 * uniforms u1,u2 in (0,1) and Box-Muller gives z0 = sqrt(-2 ln u1) cos(2 pi u2),
 * z1 = ... sin(...); parameter j = node*nu + actuator uses pair c = j/2, z(j%2).
 * stream 1, c = 0: u1 < 0.2 is the per-candidate Bernoulli choosing std1. */
enum {
  /* sigma(k) = 0.5*(ctrlrange_hi-lo)(k) * std, std = std0, or std1 w.p. 0.2 per
   * candidate when std1 > 0 (sampling/planner.cc:331-345) */
  MJPCX_NOISE_SAMPLING = 0,
  /* sigma(j) = max(sqrt(param_variance[j]), floor), floor = std0 (std_initial)
   * for global candidates < explore_count, else std1 (std_min); not scaled by
   * ctrlrange (cross_entropy/planner.cc:351-385, 399-405) */
  MJPCX_NOISE_CROSS_ENTROPY = 1,
};
typedef struct mjpcx_noise_spec {
  uint64_t seed;
  uint32_t iteration;        /* plan-iteration counter                          */
  int32_t mode;              /* MJPCX_NOISE_*                                   */
  int32_t candidate_offset;  /* global index of local candidate 0 (rank sharding) */
  int32_t nominal_candidate; /* global candidate left un-noised (PS: 0, planner.cc:374); -1: none (CE) */
  int32_t explore_count;     /* CE only                                         */
  double std0;
  double std1;
  const double* param_variance; /* CE only: P*nu                                */
} mjpcx_noise_spec;

typedef struct mjpcx_ctx mjpcx_ctx;

/* ---- lifecycle --------------------------------------------------------------
 * precision: 64 (fp64, the reference's mjtNum) or 32 (every rollout kernel family has a float instantiation; Trajectory
 * buffers come back as doubles either way; the iLQG entry points of the contact-model family are fp64 only). Replaces
 * Planner::Initialize/Allocate/ResizeMjData (planners/planner.cc:23-33). */
int mjpcx_create(const mjpcx_model* model, const mjpcx_task* task, int device,
                 int precision, mjpcx_ctx** out);
void mjpcx_destroy(mjpcx_ctx* ctx);
const char* mjpcx_create_error(void); /* detail of the calling thread's last failed mjpcx_create; after a successful one: "" or a
                                         non-fatal warning naming what of the model the device does not reproduce (e.g. collidable
                                         pairs of non-sphere/capsule geoms between two moving bodies, which are left out) */
const char* mjpcx_error_string(int code);
const char* mjpcx_last_error(const mjpcx_ctx* ctx); /* detail of last failure */
const char* mjpcx_kernel_name(const mjpcx_ctx* ctx); /* rollout kernel variant */

/* Planner::SetState (sampling/planner.cc:150-153, State::CopyTo
 * states/state.cc:128-135): state = [qpos,qvel,act], mocap = 7*nmocap. */
int mjpcx_set_state(mjpcx_ctx* ctx, const double* state, double time,
                    const double* mocap, const double* userdata);

/* Per-plan frozen copy of weights/parameters (BaseResidualFn::Update,
 * task.cc:112-123). Any pointer may be NULL to keep the current value. */
int mjpcx_set_task_params(mjpcx_ctx* ctx, const double* weight,
                          const double* norm_parameter, const double* parameters,
                          double risk);

/* ---- the hot path -----------------------------------------------------------
 * Roll out N candidate spline policies for `horizon` steps from the state set
 * by mjpcx_set_state and evaluate their returns. Equivalent to N x
 * Trajectory::Rollout + UpdateReturn with SamplingPolicy::Action
 * (sampling/policy.cc:52-59) as the policy. Asynchronous on the context's
 * stream; results are read with the getters below (which synchronise).
 *   node_times : P                (shared by all candidates)
 *   node_values: N x P x nu       (candidate-major, each block = the reference's
 *                                  TimeSpline values, already noised+clamped) */
int mjpcx_rollout_splines(mjpcx_ctx* ctx, int num_candidates, int horizon,
                          int num_nodes, int interpolation, const double* node_times,
                          const double* node_values);

/* Trajectory::NoisyRollout (mjpc/trajectory.cc:100-210; RobustPlanner::OptimizePolicy, planners/robust/robust_planner.cc:
 * 120-140): as mjpcx_rollout_splines, with Ornstein-Uhlenbeck force / torque noise on every body's xfrc_applied,
 * xfrc[i] <- exp(-dt / xfrc_rate) xfrc[i] + N(0, xfrc_std sqrt(1 - exp(-2 dt / xfrc_rate))) before every step (starting
 * from zero). The normals are counter-based, keyed on (seed, candidate_offset + candidate, step, entry) -- the reference's
 * absl::BitGen is unseeded, so there is no stream to reproduce. Both kernel families. */
int mjpcx_rollout_splines_noisy(mjpcx_ctx* ctx, int num_candidates, int horizon, int num_nodes, int interpolation,
                                const double* node_times, const double* node_values, double xfrc_std, double xfrc_rate,
                                uint64_t seed, int candidate_offset);

/* Same, but candidates are generated on the device: nominal spline (P x nu)
 * + clamped Gaussian noise per mjpcx_noise_spec. */
int mjpcx_rollout_noise(mjpcx_ctx* ctx, int num_candidates, int horizon,
                        int num_nodes, int interpolation, const double* node_times,
                        const double* nominal_values, const mjpcx_noise_spec* noise);

/* The task-specific frozen ResidualFn state (mjpcx_task::residual_int / residual_real) for the next rollouts: what
 * Task::Transition changed since the context was created (mode, gait phase origin, Walk/Flip state of the Quadruped).
 * Either pointer may be NULL (keep). */
int mjpcx_set_residual_state(mjpcx_ctx* ctx, const int32_t* residual_int, const double* residual_real);

/* Block until everything queued on the context's stream has finished. */
int mjpcx_sync(mjpcx_ctx* ctx);

/* total_return[N], failure[N] (Trajectory::total_return / failure). failure: 0 = the rollout completed, non-zero = it
 * stopped at a warning (trajectory.cc:169-173). The wavefront-per-candidate kernels put diagnostics above the low
 * byte: (warning bits << 8) | (failing step << 16); bits: 16 = Hessian not positive definite, 32 = contact list full. */
int mjpcx_get_returns(mjpcx_ctx* ctx, double* total_return, int32_t* failure);

/* total_return / failure of one candidate (e.g. trajectory[0], the nominal,
 * for `improvement`, sampling/planner.cc:207-208). */
int mjpcx_get_return_at(mjpcx_ctx* ctx, int candidate, double* total_return, int32_t* failure);

/* Device-side selection replacing std::partial_sort (sampling/planner.cc:184):
 * indices and returns of the k best candidates, ascending, ties by index. */
int mjpcx_topk(mjpcx_ctx* ctx, int k, int32_t* index, double* total_return);

/* The reads one Predictive-Sampling policy update needs, fused into one launch and one sync:
 * argmin over total_return (ties by index) -> *index, *best_return; the winner's spline values
 * (P x nu, may be NULL); and the return of `ref_candidate` (the nominal, candidate 0; -1: skip)
 * for `improvement` (sampling/planner.cc:197-212, 534-543). */
int mjpcx_best(mjpcx_ctx* ctx, int ref_candidate, int32_t* index, double* best_return, double* ref_return,
               double* spline_values);

/* Cross-Entropy elite statistics (cross_entropy/planner.cc:231-270) over the spline parameters
 * of `n` local candidates: out[j] = sum_i p_ij when mean == NULL, else sum_i (p_ij - mean[j])^2,
 * j < P*nu; *sum_return = sum_i total_return_i. The caller divides (n_elite, n_elite - 1) -- and, when
 * the candidates are sharded, all-reduces the partial sums first. */
int mjpcx_elite_moments(mjpcx_ctx* ctx, int n, const int32_t* candidates, const double* mean,
                        double* out, double* sum_return);

/* Gather one candidate into the reference's Trajectory layout. */
int mjpcx_fetch_trajectory(mjpcx_ctx* ctx, int candidate, mjpcx_traj_view* out);

/* The (noised, clamped) spline values of one candidate: P x nu. This is
 * candidate_policy[i].plan of the reference (sampling/planner.cc:534-543). */
int mjpcx_fetch_spline(mjpcx_ctx* ctx, int candidate, double* node_values);

/* ---- iLQG (mjpc/planners/ilqg, model_derivatives.cc, cost_derivatives.cc) -------------------------
 * All arrays are host fp64, row-major, time-major as in the reference's std::vectors; dim_state = nq+nv,
 * ndx = 2*nv (state-derivative dimension), nr = num_residual. */

/* N candidate rollouts under a feedback policy built on a shared nominal trajectory of Tn steps.
 *   mode 0: Trajectory::RolloutDiscrete with the index policy of iLQGPlanner::ActionRollouts
 *           (ilqg/planner.cc:630-692): u = clamp(actions[t] + alpha_i * improvement[t] + gains[t] (x - states[t]))
 *   mode 1: Trajectory::Rollout with iLQGPolicy::Action (ilqg/policy.cc:82-161), representation 0 / 1 / 2 (zero-order, linear, cubic)
 *           (zero-order / linear interpolation over `times`), feedback scaled by alpha_i, applied iff use_state
 *           (FeedbackRollouts, ilqg/planner.cc:695-724)
 * gains: Tn x nu x ndx (feedback_gain), improvement: Tn x nu (action_improvement), alpha: N. */
int mjpcx_rollout_feedback(mjpcx_ctx* ctx, int num_candidates, int horizon, int mode, int representation,
                           int use_state, int nominal_horizon, const double* times, const double* states,
                           const double* actions, const double* gains, const double* improvement,
                           const double* alpha);

/* ModelDerivatives::Compute (model_derivatives.cc:45-106) = Tn x mjd_transitionFD(eps, centered): A (Tn x ndx x ndx),
 * B (Tn x ndx x nu) of the next state, C (Tn x nr x ndx), D (Tn x nr x nu) of the residual sensors. Any output may be NULL. */
int mjpcx_transition_fd(mjpcx_ctx* ctx, int nominal_horizon, const double* times, const double* states,
                        const double* actions, double eps, int centered, double* A, double* B, double* C, double* D);

/* CostDerivatives::Compute (cost_derivatives.cc:112-230) with the context's current cost specification:
 * residual (T x nr), C, D as above -> cx (T x ndx), cu (T x nu), cxx, cxu (T x ndx x nu), cuu. */
int mjpcx_cost_derivatives(mjpcx_ctx* ctx, int T, const double* residual, const double* C, const double* D,
                           double* cx, double* cu, double* cxx, double* cxu, double* cuu);

/* One Riccati sweep at regularisation mu (iLQGBackwardPass::RiccatiStep for t = T-2..0, backward_pass.cc:65-250;
 * reg_type 0 control / 1 state-control / 2 value; use_limits: box-QP on ctrlrange `limits` (nu x 2)).
 * Outputs Vx (T x n), Vxx, K = feedback_gain (T x m x n), du = action_improvement (T x m), dV[2];
 * *status = 1 on success, 0 if some Quu was not positive definite (the caller scales mu and retries,
 * ilqg/planner.cc:429-520). n <= 48, m <= 16. kernel_ms (optional): HIP-event time of the kernel. */
int mjpcx_backward_pass(mjpcx_ctx* ctx, int n, int m, int T, double mu, int reg_type, int use_limits,
                        const double* A, const double* B, const double* cx, const double* cu, const double* cxx,
                        const double* cxu, const double* cuu, const double* actions, const double* limits,
                        double* Vx, double* Vxx, double* K, double* du, double* dV, int32_t* status,
                        double* kernel_ms);

/* ---- measurement --------------------------------------------------------------
 * HIP-event timing of the rollout kernel on the context's own stream.
 * mjpcx_timing_reset zeroes the accumulators; mjpcx_timing_read synchronises and
 * returns the summed kernel time [ms] and launch count since the reset. */
int mjpcx_timing_reset(mjpcx_ctx* ctx);
int mjpcx_timing_read(mjpcx_ctx* ctx, double* kernel_ms, int64_t* launches);
/* The same accumulation for the rollout's FIRST kernel alone -- the one that rolls the batch out (rollout_quad_kernel, the first pass
 * of rollout_tree_kernel, the time loop of the lane family); what follows it (the pass over the candidates it handed on, the sensor
 * stage of the lane family) is in mjpcx_timing_read's total only. Call before mjpcx_timing_read (which ends the timing window). */
int mjpcx_timing_read_main(mjpcx_ctx* ctx, double* main_kernel_ms, int64_t* launches);
/* rollout_quad_kernel only (zeros otherwise): of the last rollout, [0] candidates handed to the wavefront-per-candidate kernel, then by
 * reason: [1] contact list full [2] contact between two legs [3] indefinite Hessian [4] non-finite value [5] both limits of a joint
 * [6] contact between the trunk and a leg [7] a joint beyond the range over which the geom pairs the kernel leaves out are proven
 * apart (csrc/pair_cull.h). Synchronises. */
int mjpcx_quad_stats(mjpcx_ctx* ctx, int32_t* handed_on /* 8 */);

/* Algorithmic bytes of one candidate rollout:
 * w*[H*(dim_state+nu+1+nr+3*ntrace+1) + P*nu + P + 2]. */
int64_t mjpcx_algorithmic_bytes(const mjpcx_ctx* ctx, int horizon, int num_nodes);

/* mjData kinematics of the state last given to mjpcx_set_state (mj_kinematics, mj_comPos, mj_comVel, mj_subtreeVel), for
 * Task::Transition implementations that read them (QuadrupedFlat::TransitionLocked: torso pose, subtree centre of mass and
 * linear velocity, head site; quadruped.cc:229-391) when the host has no physics of its own. Any output may be NULL; sizes
 * follow the model (nbody x 3 / 4 / 9, nsite x 3). Contact-model kernel family only. Synchronous. */
int mjpcx_kinematics(mjpcx_ctx* ctx, double* xpos, double* xquat, double* xmat, double* xipos, double* site_xpos,
                     double* subtree_com, double* subtree_linvel);

/* Raw device pointers (for zero-copy wrapping, e.g. torch tensors feeding an
 * RCCL collective). which: 0 = total_return (N x fp64), 1 = failure (N x i32). */
int mjpcx_device_buffer(mjpcx_ctx* ctx, int which, void** ptr, size_t* bytes);

/* ---- multi-GPU: one process per GPU, candidates partitioned by rank (candidate_offset in the noise spec), ONE small
 * exchange per plan iteration over RCCL (xGMI). SURVEY.md 8b / 8e. The library resolves librccl.so.1 at the first call
 * (MJPCX_EUNSUPPORTED if it is absent); every rank must make the same sequence of collective calls.
 *   mjpcx_comm_unique_id : rank 0 only; the caller ships the 128 bytes to the other ranks (any side channel)
 *   mjpcx_comm_init      : every rank; ncclCommInitRank on the context's device
 *   mjpcx_exchange_best  : Predictive Sampling (replaces the partial_sort over one process's candidates,
 *                          sampling/planner.cc:184-188): all-gather of (best return, global index, nominal return), winner =
 *                          lowest return, ties to the lowest global index, NaN ranked last; then the winner's owner broadcasts
 *                          its n spline values. In/out: this rank's record -> the global winner's. nominal_return is rank 0's
 *                          (global candidate 0 lives there).
 *   mjpcx_merge_topk     : Cross-Entropy (cross_entropy/planner.cc:240-250): in = this rank's k best (global index, return),
 *                          unused slots index -1; out = the global k best, identical on every rank (ties by global index)
 *   mjpcx_elite_allreduce: in-place sum over the ranks of a small fp64 vector (the elite moments)
 *   mjpcx_comm_barrier   : all ranks reach it (a 1-element all-reduce + stream sync)
 */
#define MJPCX_COMM_ID_BYTES 128
int mjpcx_comm_unique_id(void* id_out);
int mjpcx_comm_init(mjpcx_ctx* ctx, const void* unique_id, int rank, int world);
int mjpcx_comm_info(const mjpcx_ctx* ctx, int* rank, int* world); /* world = 1, rank = 0 before mjpcx_comm_init */
int mjpcx_exchange_best(mjpcx_ctx* ctx, int32_t* index, double* best_return, double* nominal_return, double* spline_values, int n);
int mjpcx_merge_topk(mjpcx_ctx* ctx, int k, int64_t* index, double* total_return);
int mjpcx_elite_allreduce(mjpcx_ctx* ctx, double* values, int n);
int mjpcx_comm_barrier(mjpcx_ctx* ctx);
int mjpcx_comm_destroy(mjpcx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MJPCX_H_ */
