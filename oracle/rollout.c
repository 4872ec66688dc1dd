/* rollout.c -- oracle restatement of Trajectory::NoisyRollout / UpdateReturn
 * (mjpc/trajectory.cc:100-210, 312-326) with SamplingPolicy::Action
 * (mjpc/planners/sampling/policy.cc:52-59) as the policy, and of the ThreadPool
 * fan-out of SamplingPlanner::Rollouts (mjpc/planners/sampling/planner.cc:355-393).
 * TEST INFRASTRUCTURE ONLY (see oracle.h). */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define KMAX_RETURN 1.0e6 /* kMaxReturnValue, trajectory.cc:29 */

typedef void (*policy_fn)(void* user, double* action, const double* state, double time);

/* trajectory.cc:100-210 */
static int rollout_core(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state,
                        double time, const double* mocap, const double* userdata, int horizon,
                        void (*policy)(void*, double*, const double*, double), void* user, mjpcx_traj_view* out);
#define OXFRC_STREAM 0x58465243u /* "XFRC": Philox stream of the force noise */
typedef struct { double std, rate; uint64_t seed; int candidate; } XfrcNoise;

static int rollout_core_noisy(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state,
                              double time, const double* mocap, const double* userdata, int horizon,
                              policy_fn policy, void* user, mjpcx_traj_view* out, const XfrcNoise* noise) {
  int nq = m->nq, nv = m->nv, nu = m->nu, nr = task->num_residual, ntr = task->num_trace;
  int ds = nq + nv + m->na;
  double* r = (double*)malloc(sizeof(double) * (nr + nu + ds + 8));
  double* act = r + nr;
  double* st = act + nu;
  double total = 0;
  out->horizon = horizon;

  odata_set_state(d, state, time, mocap, userdata);
  memcpy(st, state, sizeof(double) * ds);
  if (out->states) memcpy(out->states, state, sizeof(double) * ds);
  if (out->times) out->times[0] = time;

  double cur_time = time;
  for (int t = 0; t < horizon - 1; t++) {
    policy(user, act, st, cur_time);
    if (out->actions) memcpy(out->actions + t * nu, act, sizeof(double) * nu);
    odata_set_ctrl(d, act);
    if (noise && noise->std > 0) { /* Ornstein-Uhlenbeck force / torque noise on every body (trajectory.cc:147-155) */
      double rate = exp(-m->timestep / noise->rate), scale = noise->std * sqrt(1 - rate * rate);
      double* x = odata_xfrc_applied(d);
      for (int j = 0; j < 6 * m->nbody; j += 2) {
        double z[2];
        ogaussian_pair(noise->seed, (uint32_t)noise->candidate, (uint32_t)(t * 3 * m->nbody + j / 2), OXFRC_STREAM, z);
        x[j] = rate * x[j] + scale * z[0];
        x[j + 1] = rate * x[j + 1] + scale * z[1];
      }
    }
    /* mj_step; the residual is evaluated inside its forward pass by the
     * mjSTAGE_ACC sensor callback (app.cc:110-126): it sees the pre-integration
     * state, which pairs residual[t] with states[t] (rollout_test.cc:140-145) */
    o_step_task(d, task, r);
    if (out->residual) memcpy(out->residual + t * nr, r, sizeof(double) * nr);
    for (int k = 0; k < ntr && out->trace; k++) /* GetTraces, utilities.cc:268-286 */
      memcpy(out->trace + (t * ntr + k) * 3, odata_trace_point(d, task->trace_site[k]), 3 * sizeof(double));
    if (odata_warning(d)) { /* CheckWarnings, trajectory.cc:169-173 */
      if (getenv("ODEBUG_FAIL")) fprintf(stderr, "rollout failed at step %d: warning bits %d\n", t, odata_warning(d));
      out->total_return = KMAX_RETURN;
      out->failure = 1;
      free(r);
      return 0;
    }
    double c = ocost_value(task, r); /* UpdateReturn, hoisted: same order of summation */
    if (out->costs) out->costs[t] = c;
    total += c;
    odata_get(d, "qpos", st, nq);
    odata_get(d, "qvel", st + nq, nv);
    odata_get(d, "time", &cur_time, 1);
    if (out->states) memcpy(out->states + (t + 1) * ds, st, sizeof(double) * ds);
    if (out->times) out->times[t + 1] = cur_time;
  }
  /* final action: copy of the previous one (trajectory.cc:190-196) */
  if (out->actions) {
    if (horizon > 1) memcpy(out->actions + (horizon - 1) * nu, out->actions + (horizon - 2) * nu, sizeof(double) * nu);
    else memset(out->actions, 0, sizeof(double) * nu);
  }
  /* final mj_forward with the last control still in data->ctrl */
  o_forward_task(d, task, r);
  if (out->residual) memcpy(out->residual + (horizon - 1) * nr, r, sizeof(double) * nr);
  for (int k = 0; k < ntr && out->trace; k++)
    memcpy(out->trace + ((horizon - 1) * ntr + k) * 3, odata_trace_point(d, task->trace_site[k]), 3 * sizeof(double));
  double c = ocost_value(task, r);
  if (out->costs) out->costs[horizon - 1] = c;
  total += c;
  /* UpdateReturn, trajectory.cc:312-326 */
  out->total_return = total / (horizon > 1 ? horizon : 1);
  out->failure = 0;
  free(r);
  return 0;
}

/* SamplingPolicy::Action: spline sample then Clamp (sampling/policy.cc:52-59) */
typedef struct { const OSpline* spline; const mjpcx_model* m; } SplinePolicy;
static void spline_policy(void* user, double* action, const double* state, double time) {
  (void)state;
  SplinePolicy* p = (SplinePolicy*)user;
  ospline_sample(p->spline, time, action);
  for (int k = 0; k < p->m->nu; k++) {
    double lo = p->m->actuator_ctrlrange[2 * k], hi = p->m->actuator_ctrlrange[2 * k + 1];
    action[k] = action[k] < lo ? lo : (action[k] > hi ? hi : action[k]);
  }
}
int orollout_spline(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state,
                    double time, const double* mocap, const double* userdata, int horizon,
                    const OSpline* spline, mjpcx_traj_view* out) {
  SplinePolicy p = {spline, m};
  return rollout_core(m, task, d, state, time, mocap, userdata, horizon, spline_policy, &p, out);
}

static int rollout_core(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state,
                        double time, const double* mocap, const double* userdata, int horizon,
                        policy_fn policy, void* user, mjpcx_traj_view* out) {
  return rollout_core_noisy(m, task, d, state, time, mocap, userdata, horizon, policy, user, out, NULL);
}

/* PD feedback policy of mjpc/test/agent/rollout_test.cc:84-103 */
typedef struct { const double *pg, *vg; double P, D; } PdPolicy;
static void pd_policy(void* user, double* action, const double* state, double time) {
  (void)time;
  PdPolicy* p = (PdPolicy*)user;
  for (int i = 0; i < 2; i++)
    action[i] = -p->P * (state[i] - p->pg[i]) - p->D * (state[2 + i] - p->vg[i]);
}
int orollout_pd(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state,
                double time, const double* mocap, int horizon, const double* pos_goal,
                const double* vel_goal, double P, double D, mjpcx_traj_view* out) {
  PdPolicy p = {pos_goal, vel_goal, P, D};
  return rollout_core(m, task, d, state, time, mocap, NULL, horizon, pd_policy, &p, out);
}

/* ---------------- batched fan-out over worker threads ---------------------- */
typedef struct {
  const mjpcx_model* m; const mjpcx_task* task;
  const double *state, *mocap, *userdata, *node_times, *node_values;
  double time;
  int N, H, P, interp;
  OBatchOut* out;
  int next; /* shared work counter = the FIFO queue of one task per candidate */
  double xfrc_std, xfrc_rate; uint64_t seed; int candidate_offset; /* NoisyRollout (0: plain Rollout) */
} Batch;

static void* batch_worker(void* arg) {
  Batch* b = (Batch*)arg;
  const mjpcx_model* m = b->m;
  int nu = m->nu, ds = m->nq + m->nv + m->na, nr = b->task->num_residual, ntr = b->task->num_trace;
  OData* d = odata_new(m); /* one physics arena per worker (planner.cc:23-33) */
  OSpline sp;
  ospline_init(&sp, nu, b->interp);
  for (;;) {
    int i = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED);
    if (i >= b->N) break;
    ospline_clear(&sp);
    for (int p = 0; p < b->P; p++)
      ospline_add_node(&sp, b->node_times[p], b->node_values + ((size_t)i * b->P + p) * nu);
    mjpcx_traj_view v;
    memset(&v, 0, sizeof v);
    OBatchOut* o = b->out;
    size_t H = b->H;
    if (o->states) v.states = o->states + (size_t)i * H * ds;
    if (o->actions) v.actions = o->actions + (size_t)i * H * nu;
    if (o->times) v.times = o->times + (size_t)i * H;
    if (o->residual) v.residual = o->residual + (size_t)i * H * nr;
    if (o->costs) v.costs = o->costs + (size_t)i * H;
    if (o->trace) v.trace = o->trace + (size_t)i * H * 3 * ntr;
    if (b->xfrc_std > 0) {
      XfrcNoise nz = {b->xfrc_std, b->xfrc_rate, b->seed, b->candidate_offset + i};
      SplinePolicy pol = {&sp, m};
      rollout_core_noisy(m, b->task, d, b->state, b->time, b->mocap, b->userdata, b->H, spline_policy, &pol, &v, &nz);
    } else {
      orollout_spline(m, b->task, d, b->state, b->time, b->mocap, b->userdata, b->H, &sp, &v);
    }
    o->total_return[i] = v.total_return;
    o->failure[i] = v.failure;
  }
  ospline_free(&sp);
  odata_free(d);
  return NULL;
}

int orollout_batch(const mjpcx_model* m, const mjpcx_task* task, const double* state, double time,
                   const double* mocap, const double* userdata, int N, int H, int P, int interp,
                   const double* node_times, const double* node_values, int num_threads,
                   OBatchOut* out) {
  return orollout_batch_noisy(m, task, state, time, mocap, userdata, N, H, P, interp, node_times, node_values, 0.0, 1.0, 0, 0,
                              num_threads, out);
}

int orollout_batch_noisy(const mjpcx_model* m, const mjpcx_task* task, const double* state, double time,
                         const double* mocap, const double* userdata, int N, int H, int P, int interp,
                         const double* node_times, const double* node_values, double xfrc_std, double xfrc_rate,
                         uint64_t seed, int candidate_offset, int num_threads, OBatchOut* out) {
  Batch b = {m, task, state, mocap, userdata, node_times, node_values, time, N, H, P, interp, out, 0,
             xfrc_std, xfrc_rate, seed, candidate_offset};
  if (num_threads < 1) num_threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * num_threads);
  for (int t = 0; t < num_threads; t++) pthread_create(&th[t], NULL, batch_worker, &b);
  for (int t = 0; t < num_threads; t++) pthread_join(th[t], NULL);
  free(th);
  return 0;
}
