/* oracle.h -- CPU fp64 restatement of the MJPC rollout-and-evaluate hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (mujoco_mpc_amd/, the C-ABI
 * library, bench.py's timed GPU leg) may import, link or execute this code; it
 * is the checker for tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg.
 *
 * Parity status
 *   - mjpc-owned arithmetic (TimeSpline, Norm, CostTerms/CostValue, Trajectory
 *     bookkeeping, sampling-planner update, iLQG Riccati step) is PINNED against
 *     the reference's own known-answer tests (tests/test_oracle_*.py cite them).
 *   - the rigid-body physics (`mj_step`/`mj_forward`, MuJoCo @ 088079ef, a
 *     third-party dependency that is absent from /root/reference and from this
 *     image) is restated from MuJoCo's documented pipeline: PARITY UNPINNED
 *     against MuJoCo itself. It is validated by analytic checks (closed-form
 *     cart-pole dynamics, energy conservation, free fall, pendulum period) and
 *     the reference's behavioural tests (rollout_test.cc).
 *   - capacity: OMAXEFC = 64 constraint rows and OMAXCON = 16 contacts per step mirror the device kernel's lane = row
 *     layout; where MuJoCo would grow its arena, the oracle raises a warning and the rollout fails (as the device does).
 *     DESIGN.md section 2 states how often that happens on the BASELINE workloads.
 */
#ifndef MJPC_ORACLE_H_
#define MJPC_ORACLE_H_

#include <stdint.h>
#include "../include/mjpcx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- TimeSpline (mjpc/spline/spline.{h,cc}) ------------------ */
typedef struct OSpline {
  int dim, interp, size, cap;
  double* times;  /* size, sorted            */
  double* values; /* size x dim, time order  */
} OSpline;
void ospline_init(OSpline* s, int dim, int interp);
void ospline_free(OSpline* s);
void ospline_clear(OSpline* s);
void ospline_copy(OSpline* dst, const OSpline* src);
void ospline_set_interpolation(OSpline* s, int interp);
/* returns node index, or -1 when `time` falls strictly inside the node range
 * (spline.cc:217-219 CHECKs that) */
int ospline_add_node(OSpline* s, double time, const double* values);
void ospline_sample(const OSpline* s, double time, double* out);
int ospline_discard_before(OSpline* s, double time);
void ospline_shift_time(OSpline* s, double start_time);

/* ---------------- Norm (mjpc/norm.cc) ------------------------------------- */
int onorm_parameter_dimension(int type);
double onorm(double* g, double* H, const double* x, const double* params, int n, int type);

/* ---------------- cost (mjpc/task.cc:71-110) ------------------------------ */
void ocost_terms(const mjpcx_task* task, const double* residual, double* terms, int weighted);
double ocost_value(const mjpcx_task* task, const double* residual);

/* ---------------- counter-based noise (replaces absl::BitGen) --- */
void ophilox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void ogaussian_pair(uint64_t seed, uint32_t candidate, uint32_t pair, uint32_t iteration, double z[2]);
/* candidate spline values = clamp(nominal + noise): sampling/planner.cc:326-352 */
void onoise_candidate(const mjpcx_model* m, const mjpcx_noise_spec* ns, int num_nodes,
                      const double* nominal, int global_candidate, double* out_values);

/* ---------------- physics (MuJoCo pipeline restatement) ------------------- */
typedef struct OData OData;
OData* odata_new(const mjpcx_model* m);
void odata_free(OData* d);
void odata_set_state(OData* d, const double* state, double time, const double* mocap,
                     const double* userdata);
void odata_set_ctrl(OData* d, const double* ctrl);
void o_forward(OData* d);           /* mj_forward */
void o_step(OData* d);              /* mj_step    */
/* same, with the task residual evaluated at the sensor-callback point */
void o_forward_task(OData* d, const mjpcx_task* task, double* residual);
void o_step_task(OData* d, const mjpcx_task* task, double* residual);
const double* odata_site_xpos(const OData* d);
double* odata_xfrc_applied(OData* d); /* 6 nbody: force, torque per body (mjData.xfrc_applied) */
const double* odata_trace_point(const OData* d, int id);
int odata_warning(const OData* d);  /* !=0: BADQPOS/BADQVEL/BADQACC/BADCTRL seen */
/* introspection for tests: name in {"qpos","qvel","qacc","qacc_smooth","M",
 * "xpos","xquat","xmat","xipos","site_xpos","subtree_com","qfrc_bias",
 * "qfrc_passive","qfrc_actuator","qfrc_constraint","actuator_force","time",
 * "efc_force","nefc","energy"}; returns count written (<= cap), -1 unknown. */
int odata_get(const OData* d, const char* name, double* out, int cap);
/* tests only: the dual of the constraint problem of the last o_forward by projected Gauss-Seidel (contact.inc); qacc_out[nv],
 * force_out[nefc] (may be NULL); returns the sweeps taken */
int o_solve_pgs(OData* d, int max_sweeps, double tol, double* qacc_out, double* force_out);

/* tests only: the narrow phase of a thin geom (sphere | capsule) against a solid (box | cylinder), in the solid's frame (contact.inc) */
double thin_vs_solid(int is_cylinder, const double* size, const double* p, const double* a, double h, double r, double* n, double* c);

/* residual dispatch (the ResidualFn::Residual overrides) */
void oresidual(const mjpcx_task* task, const OData* d, double* residual);

/* ---------------- Trajectory::Rollout (mjpc/trajectory.cc:92-210) --------- */
/* policy = SamplingPolicy::Action over `spline` (sampling/policy.cc:52-59). */
int orollout_spline(const mjpcx_model* m, const mjpcx_task* task, OData* d,
                    const double* state, double time, const double* mocap,
                    const double* userdata, int horizon, const OSpline* spline,
                    mjpcx_traj_view* out);
/* policy = PD feedback of rollout_test.cc:84-103 (particle), for the ported test */
int orollout_pd(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state,
                double time, const double* mocap, int horizon, const double* pos_goal,
                const double* vel_goal, double P, double D, mjpcx_traj_view* out);

/* ---------------- batched rollouts over a thread pool --------------------- */
/* The reference's fan-out (sampling/planner.cc:355-393): one task per candidate,
 * one physics arena per worker. node_values: N x P x nu. Outputs: returns[N],
 * failure[N]; optional full trajectories in reference layout, candidate-major
 * (NULL to skip). Returns 0. */
typedef struct OBatchOut {
  double* total_return; /* N */
  int32_t* failure;     /* N */
  double* states;   /* N x H x dim_state or NULL */
  double* actions;  /* N x H x nu or NULL */
  double* times;    /* N x H or NULL */
  double* residual; /* N x H x nr or NULL */
  double* costs;    /* N x H or NULL */
  double* trace;    /* N x H x 3*ntrace or NULL */
} OBatchOut;
int orollout_batch(const mjpcx_model* m, const mjpcx_task* task, const double* state,
                   double time, const double* mocap, const double* userdata, int N, int H,
                   int P, int interp, const double* node_times, const double* node_values,
                   int num_threads, OBatchOut* out);
/* Trajectory::NoisyRollout (trajectory.cc:100-210) for N candidate splines: Ornstein-Uhlenbeck xfrc_applied noise
 * (rate = exp(-dt / xfrc_rate), scale = xfrc_std sqrt(1 - rate^2)), counter-based normals keyed on
 * (seed, candidate_offset + i, step, entry) instead of the reference's unseeded absl::BitGen */
int orollout_batch_noisy(const mjpcx_model* m, const mjpcx_task* task, const double* state,
                         double time, const double* mocap, const double* userdata, int N, int H,
                         int P, int interp, const double* node_times, const double* node_values,
                         double xfrc_std, double xfrc_rate, uint64_t seed, int candidate_offset,
                         int num_threads, OBatchOut* out);

/* ---------------- iLQG derivatives and feedback rollouts (oracle/ilqg.c) ---------------- */
/* StateDiff (mjpc/utilities.cc:543-553): (s2 - s1) / h in the tangent space, 2 nv entries */
void ostate_diff(const mjpcx_model* m, double* dx, const double* s1, const double* s2, double h);
int otransition_fd(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state, double time,
                   const double* ctrl, double eps, int centered, double* A, double* B, double* C, double* D);
void ocost_derivatives(const mjpcx_task* task, int T, int ndx, int nu, const double* r, const double* rx, const double* ru,
                       double* cx, double* cu, double* cxx, double* cxu, double* cuu);
int orollout_feedback(const mjpcx_model* m, const mjpcx_task* task, const double* state, double time, const double* mocap,
                      int N, int H, int mode, int representation, int use_state, int Tn, const double* times,
                      const double* states, const double* actions, const double* gains, const double* improvement,
                      const double* alpha, OBatchOut* out);
/* the same fanned over worker threads (one physics arena each), as the reference schedules them on its ThreadPool: the candidates of
 * the line search (ilqg/planner.cc:630-692) and the time steps of ModelDerivatives::Compute (model_derivatives.cc:45-106) */
int orollout_feedback_mt(const mjpcx_model* m, const mjpcx_task* task, const double* state, double time, const double* mocap,
                         int N, int H, int mode, int representation, int use_state, int Tn, const double* times,
                         const double* states, const double* actions, const double* gains, const double* improvement,
                         const double* alpha, int num_threads, OBatchOut* out);
int otransition_fd_batch(const mjpcx_model* m, const mjpcx_task* task, const double* mocap, int T, const double* states,
                         const double* times, const double* actions, double eps, int centered, double* A, double* B, double* C,
                         double* D, int num_threads);

/* ---------------- iLQG backward pass (mjpc/planners/ilqg/backward_pass.cc) ---------------- */
/* box-constrained QP (MuJoCo mju_boxQP): returns the number of free dimensions, -1 if not PD */
int oboxqp(double* res, double* R, int* index, const double* H, const double* g, int n, const double* lower,
           const double* upper);
/* One backward sweep over T steps at regularisation mu. Row-major layouts as in the reference:
 * A[T*n*n] B[T*n*m] cx[T*n] cu[T*m] cxx[T*n*n] cxu[T*n*m] cuu[T*m*m]; out Vx[T*n] Vxx[T*n*n]
 * K[T*m*n] (feedback_gain) du[T*m] (action_improvement) dV[2]. reg_type 0 control, 1 state-control,
 * 2 value. Returns 1 on success, 0 if a step's Quu was not positive definite. */
int oriccati(int n, int m, int T, double mu, int reg_type, int use_limits, const double* A, const double* B,
             const double* cx, const double* cu, const double* cxx, const double* cxu, const double* cuu,
             const double* actions, const double* action_limits, double* Vx, double* Vxx, double* K, double* du,
             double* dV);

#ifdef __cplusplus
}
#endif
#endif
