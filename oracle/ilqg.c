/* ilqg.c -- oracle restatement of the iLQG planner's derivative and rollout pieces:
 *   - mjd_transitionFD (MuJoCo @088079ef engine_derivative_fd.c; third-party, restated from its documented
 *     behaviour: forward/centred differences of the next state and of sensordata w.r.t. state and control,
 *     control nudges kept inside ctrlrange), as called by ModelDerivatives::Compute
 *     (mjpc/planners/model_derivatives.cc:75-106)
 *   - CostDerivatives::DerivativeStep / Compute (mjpc/planners/cost_derivatives.cc:77-230)
 *   - iLQGPolicy::Action (mjpc/planners/ilqg/policy.cc:82-161) and the index feedback policy of
 *     iLQGPlanner::ActionRollouts (ilqg/planner.cc:640-668), as policies of Trajectory::Rollout /
 *     RolloutDiscrete (mjpc/trajectory.cc:92-309)
 * TEST INFRASTRUCTURE ONLY (see oracle.h). */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ---- quaternion helpers for the tangent-space state difference (mj_differentiatePos / mj_integratePos) ---- */
static void iq_mul(double* r, const double* a, const double* b) {
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
/* res(3) = rotation vector taking qb to qa, expressed in qb's frame (mju_subQuat) */
static void iq_sub(double* res, const double* qa, const double* qb) {
  const double qn[4] = {qb[0], -qb[1], -qb[2], -qb[3]};
  double qd[4];
  iq_mul(qd, qn, qa);
  double ax[3] = {qd[1], qd[2], qd[3]};
  const double s = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (s > 1e-15) for (int k = 0; k < 3; k++) ax[k] /= s;
  double speed = 2 * atan2(s, qd[0]);
  if (speed > 3.14159265358979323846) speed -= 2 * 3.14159265358979323846;
  for (int k = 0; k < 3; k++) res[k] = ax[k] * speed;
}
/* q <- q * exp(v h / 2) (mju_quatIntegrate) */
static void iq_integrate(double* q, const double* v, double h) {
  double ax[3] = {v[0], v[1], v[2]};
  const double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (n < 1e-15) { ax[0] = 1; ax[1] = ax[2] = 0; } else for (int k = 0; k < 3; k++) ax[k] /= n;
  const double a = 0.5 * h * n;
  const double qr[4] = {cos(a), ax[0] * sin(a), ax[1] * sin(a), ax[2] * sin(a)};
  double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (nq < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else for (int k = 0; k < 4; k++) q[k] /= nq;
  iq_mul(q, q, qr);
}
/* StateDiff (mjpc/utilities.cc:543-553): dx = (s2 - s1) / h in the tangent space: mj_differentiatePos for the
 * positions, plain differences for velocities. dx has 2 nv entries. */
void ostate_diff(const mjpcx_model* m, double* dx, const double* s1, const double* s2, double h) {
  const int nq = m->nq, nv = m->nv;
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    switch (m->jnt_type[j]) {
      case MJPCX_JNT_FREE:
        for (int k = 0; k < 3; k++) dx[da + k] = (s2[qa + k] - s1[qa + k]) / h;
        qa += 3; da += 3;
        /* fallthrough */
      case MJPCX_JNT_BALL: {
        double r[3];
        iq_sub(r, s2 + qa, s1 + qa);
        for (int k = 0; k < 3; k++) dx[da + k] = r[k] / h;
        break;
      }
      default: dx[da] = (s2[qa] - s1[qa]) / h;
    }
  }
  for (int i = 0; i < nv; i++) dx[nv + i] = (s2[nq + i] - s1[nq + i]) / h;
}
/* x_out = x with tangent coordinate j (< 2 nv) moved by eps (mj_integratePos for positions) */
static void state_perturb(const mjpcx_model* m, double* xo, const double* x, int j, double eps) {
  const int nq = m->nq, nv = m->nv;
  memcpy(xo, x, sizeof(double) * (nq + nv));
  if (j >= nv) { xo[nq + j - nv] += eps; return; }
  const int jn = m->dof_jntid[j], qa = m->jnt_qposadr[jn], da = m->jnt_dofadr[jn], t = m->jnt_type[jn];
  if (t == MJPCX_JNT_FREE && j - da < 3) xo[qa + (j - da)] += eps;
  else if (t == MJPCX_JNT_FREE || t == MJPCX_JNT_BALL) {
    double v[3] = {0, 0, 0};
    const int qq = t == MJPCX_JNT_FREE ? qa + 3 : qa, k = t == MJPCX_JNT_FREE ? j - da - 3 : j - da;
    v[k] = 1;
    iq_integrate(xo + qq, v, eps);
  } else xo[qa] += eps;
}

/* ---- one perturbed mj_step: next state + residual ---- */
static void fd_step(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state, double time,
                    const double* ctrl, double* next, double* sensor) {
  odata_set_state(d, state, time, NULL, NULL);
  odata_set_ctrl(d, ctrl);
  o_step_task(d, task, sensor);
  odata_get(d, "qpos", next, m->nq);
  odata_get(d, "qvel", next + m->nq, m->nv);
}

static int in_range(double a, double b, const double* range) {
  return a >= range[0] && a <= range[1] && b >= range[0] && b <= range[1];
}

/* A: ndx x ndx, B: ndx x nu, C: nr x ndx, D: nr x nu (row-major, ndx = 2 nv); any may be NULL. Positions are perturbed and
 * differenced in the tangent space (mj_integratePos / mj_differentiatePos), so quaternion joints are covered.
 * Mocap pose must already be set on `d`. */
/* tangent coordinates (2 nv) of a state y for differencing: scalar joints, free-joint translations and velocities keep
 * their raw values (so differences of them are plain subtractions); a quaternion is represented by its rotation
 * vector relative to the same joint's quaternion in `y0` (mj_differentiatePos of that joint against y0) */
static void state_tangent(const mjpcx_model* m, double* z, const double* y, const double* y0) {
  const int nq = m->nq, nv = m->nv;
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    switch (m->jnt_type[j]) {
      case MJPCX_JNT_FREE:
        for (int k = 0; k < 3; k++) z[da + k] = y[qa + k];
        qa += 3; da += 3;
        /* fallthrough */
      case MJPCX_JNT_BALL: iq_sub(z + da, y + qa, y0 + qa); break;
      default: z[da] = y[qa];
    }
  }
  for (int i = 0; i < nv; i++) z[nv + i] = y[nq + i];
}

int otransition_fd(const mjpcx_model* m, const mjpcx_task* task, OData* d, const double* state, double time,
                   const double* ctrl, double eps, int centered, double* A, double* B, double* C, double* D) {
  const int nq = m->nq, nv = m->nv, nu = m->nu, ndx = 2 * nv, ds = nq + nv, nr = task->num_residual;
  double* w = (double*)malloc(sizeof(double) * (size_t)(3 * ds + 3 * nr + nu + 3 * ndx));
  double *y0 = w, *yq = y0 + ds, *x = yq + ds, *s0 = x + ds, *sp = s0 + nr, *sm = sp + nr, *u = sm + nr, *z0 = u + nu, *zp = z0 + ndx, *zm = zp + ndx;
  fd_step(m, task, d, state, time, ctrl, y0, s0);
  state_tangent(m, z0, y0, y0);
  for (int j = 0; j < ndx + nu; j++) {
    memcpy(u, ctrl, sizeof(double) * nu);
    int fwd = 1, back = centered != 0;
    if (j >= ndx) {
      const int k = j - ndx;
      if (m->actuator_ctrllimited[k]) {
        const double* range = m->actuator_ctrlrange + 2 * k;
        fwd = in_range(ctrl[k], ctrl[k] + eps, range);
        back = (centered || !fwd) && in_range(ctrl[k] - eps, ctrl[k], range);
      }
    }
    if (fwd) {
      if (j < ndx) state_perturb(m, x, state, j, eps); else { memcpy(x, state, sizeof(double) * ds); u[j - ndx] = ctrl[j - ndx] + eps; }
      fd_step(m, task, d, x, time, u, yq, sp);
      state_tangent(m, zp, yq, y0);
    }
    if (back) {
      memcpy(u, ctrl, sizeof(double) * nu);
      if (j < ndx) state_perturb(m, x, state, j, -eps); else { memcpy(x, state, sizeof(double) * ds); u[j - ndx] = ctrl[j - ndx] - eps; }
      fd_step(m, task, d, x, time, u, yq, sm);
      state_tangent(m, zm, yq, y0);
    }
    for (int i = 0; i < ndx + nr; i++) {
      const double v0 = i < ndx ? z0[i] : s0[i - ndx], vp = i < ndx ? zp[i] : sp[i - ndx], vm = i < ndx ? zm[i] : sm[i - ndx];
      double dv = 0;
      if (fwd && back) dv = (vp - vm) / (2 * eps);
      else if (fwd) dv = (vp - v0) / eps;
      else if (back) dv = (v0 - vm) / eps;
      if (i < ndx) {
        if (j < ndx) { if (A) A[i * ndx + j] = dv; } else if (B) B[i * nu + (j - ndx)] = dv;
      } else {
        if (j < ndx) { if (C) C[(i - ndx) * ndx + j] = dv; } else if (D) D[(i - ndx) * nu + (j - ndx)] = dv;
      }
    }
  }
  free(w);
  return 0;
}

/* CostDerivatives::Compute for one timestep t of T (the 1/T weight scaling is the caller's T) */
void ocost_derivatives(const mjpcx_task* task, int T, int ndx, int nu, const double* r, const double* rx, const double* ru,
                       double* cx, double* cu, double* cxx, double* cxu, double* cuu) {
  memset(cx, 0, sizeof(double) * ndx); memset(cu, 0, sizeof(double) * nu);
  memset(cxx, 0, sizeof(double) * ndx * ndx); memset(cxu, 0, sizeof(double) * ndx * nu); memset(cuu, 0, sizeof(double) * nu * nu);
  double g[64], H[64 * 64];
  double* Hrx = (double*)malloc(sizeof(double) * 64 * (size_t)(ndx + nu));
  double* Hru = Hrx + 64 * ndx;
  int shift = 0, pshift = 0;
  double c = 0;
  for (int k = 0; k < task->num_term; k++) {
    const int nk = task->dim_norm_residual[k];
    const double w = task->weight[k] / T;
    const double* rxk = rx + (size_t)shift * ndx;
    const double* ruk = ru + (size_t)shift * nu;
    c += w * onorm(g, H, r + shift, task->norm_parameter + pshift, nk, task->norm[k]);
    for (int a = 0; a < nk; a++) {
      for (int j = 0; j < ndx; j++) { double s = 0; for (int b = 0; b < nk; b++) s += H[a * nk + b] * rxk[b * ndx + j]; Hrx[a * ndx + j] = s; }
      for (int j = 0; j < nu; j++) { double s = 0; for (int b = 0; b < nk; b++) s += H[a * nk + b] * ruk[b * nu + j]; Hru[a * nu + j] = s; }
    }
    for (int i = 0; i < ndx; i++) { double s = 0; for (int a = 0; a < nk; a++) s += rxk[a * ndx + i] * g[a]; cx[i] += w * s; }
    for (int i = 0; i < nu; i++) { double s = 0; for (int a = 0; a < nk; a++) s += ruk[a * nu + i] * g[a]; cu[i] += w * s; }
    for (int i = 0; i < ndx; i++)
      for (int j = 0; j < ndx; j++) { double s = 0; for (int a = 0; a < nk; a++) s += Hrx[a * ndx + i] * rxk[a * ndx + j]; cxx[i * ndx + j] += w * s; }
    for (int i = 0; i < ndx; i++)
      for (int j = 0; j < nu; j++) { double s = 0; for (int a = 0; a < nk; a++) s += Hrx[a * ndx + i] * ruk[a * nu + j]; cxu[i * nu + j] += w * s; }
    for (int i = 0; i < nu; i++)
      for (int j = 0; j < nu; j++) { double s = 0; for (int a = 0; a < nk; a++) s += Hru[a * nu + i] * ruk[a * nu + j]; cuu[i * nu + j] += w * s; }
    shift += nk;
    pshift += task->num_norm_parameter[k];
  }
  free(Hrx);
  if (fabs(task->risk) < 1.0e-6) return;
  /* cost_derivatives.cc:156-226, including its use of the already scaled cx / cu in the rank-one terms */
  const double s = exp(task->risk * c);
  for (int i = 0; i < ndx; i++) cx[i] *= s;
  for (int i = 0; i < nu; i++) cu[i] *= s;
  for (int i = 0; i < ndx; i++)
    for (int j = 0; j < ndx; j++) cxx[i * ndx + j] = cxx[i * ndx + j] * s + task->risk * s * cx[i] * cx[j];
  for (int i = 0; i < ndx; i++)
    for (int j = 0; j < nu; j++) cxu[i * nu + j] = cxu[i * nu + j] * s + task->risk * s * cx[i] * cu[j];
  for (int i = 0; i < nu; i++)
    for (int j = 0; j < nu; j++) cuu[i * nu + j] = cuu[i * nu + j] * s + task->risk * s * cu[i] * cu[j];
}

/* ---- feedback policies ---- */
typedef struct {
  const mjpcx_model* m;
  int Tn, mode, representation, use_state;
  const double *times, *states, *actions, *gains, *improvement;
  double alpha;
} FbPolicy;

static void find_interval(const double* xs, double v, int length, int* b) { /* utilities.h:124-144 */
  int up = 0;
  while (up < length && xs[up] <= v) up++;
  int lo = up - 1;
  if (lo < 0) { b[0] = b[1] = 0; }
  else if (lo > length - 1) { b[0] = b[1] = length - 1; }
  else { b[0] = lo; b[1] = up < length - 1 ? up : length - 1; }
}
/* FiniteDifferenceSlope (utilities.cc:362-395): slope of component i at grid value x */
static double fd_slope(double x, const double* xs, const double* ys, int dim, int length, int i) {
  int b[2];
  find_interval(xs, x, length, b);
  if (b[0] == 0 && b[1] == 0) return length > 2 ? (ys[dim * 1 + i] - ys[i]) / (xs[1] - xs[0]) : 0.0;
  if (b[0] == length - 1 && b[1] == length - 1)
    return length > 2 ? (ys[dim * b[0] + i] - ys[dim * (b[0] - 1) + i]) / (xs[b[0]] - xs[b[0] - 1]) : 0.0;
  if (b[0] == 0) return (ys[dim * b[1] + i] - ys[dim * b[0] + i]) / (xs[b[1]] - xs[b[0]]);
  return 0.5 * (ys[dim * b[1] + i] - ys[dim * b[0] + i]) / (xs[b[1]] - xs[b[0]]) +
         0.5 * (ys[dim * b[0] + i] - ys[dim * (b[0] - 1) + i]) / (xs[b[0]] - xs[b[0] - 1]);
}
/* Zero / Linear / CubicInterpolation (utilities.cc:304-422); representation: 0 zero-order, 1 linear, 2 cubic */
static void interp(double* out, double x, const double* xs, const double* ys, int dim, int length, int zero, int representation) {
  int b[2];
  find_interval(xs, x, length, b);
  if (zero || b[0] == b[1]) { memcpy(out, ys + dim * b[0], sizeof(double) * dim); return; }
  const double span = xs[b[1]] - xs[b[0]], t = (x - xs[b[0]]) / span;
  if (representation != 2) {
    for (int i = 0; i < dim; i++) out[i] = ys[dim * b[0] + i] * (1.0 - t) + ys[dim * b[1] + i] * t;
    return;
  }
  const double c0 = 2.0 * t * t * t - 3.0 * t * t + 1.0, c1 = (t * t * t - 2.0 * t * t + t) * span;
  const double c2 = -2.0 * t * t * t + 3 * t * t, c3 = (t * t * t - t * t) * span;
  for (int i = 0; i < dim; i++)
    out[i] = c0 * ys[dim * b[0] + i] + c1 * fd_slope(xs[b[0]], xs, ys, dim, length, i) + c2 * ys[dim * b[1] + i] +
             c3 * fd_slope(xs[b[1]], xs, ys, dim, length, i);
}
static void clamp_ctrl(const mjpcx_model* m, double* u) {
  for (int k = 0; k < m->nu; k++) {
    const double lo = m->actuator_ctrlrange[2 * k], hi = m->actuator_ctrlrange[2 * k + 1];
    u[k] = u[k] < lo ? lo : (u[k] > hi ? hi : u[k]);
  }
}
/* action for step index t (mode 0) or time (mode 1) */
static void fb_action(const FbPolicy* p, double* action, const double* state, double time, int t) {
  const mjpcx_model* m = p->m;
  const int nu = m->nu, ds = m->nq + m->nv, ndx = 2 * m->nv;
  double dx[64], xi[64], K[64 * 16];
  if (p->mode == 0) {
    for (int k = 0; k < nu; k++) action[k] = p->actions[t * nu + k] + p->alpha * p->improvement[t * nu + k];
    ostate_diff(m, dx, p->states + (size_t)t * ds, state, 1.0); /* StateDiff(model, dx, states[t], state, 1.0) */
    for (int k = 0; k < nu; k++) { double s = 0; for (int j = 0; j < ndx; j++) s += p->gains[(t * nu + k) * ndx + j] * dx[j]; action[k] += s; }
  } else {
    int b[2];
    find_interval(p->times, time, p->Tn, b);
    const int zero = b[0] == b[1] || p->representation == 0;
    interp(action, time, p->times, p->actions, nu, p->Tn - 1, zero, p->representation);
    if (p->use_state) {
      interp(xi, time, p->times, p->states, ds, p->Tn, zero, p->representation);
      interp(K, time, p->times, p->gains, nu * ndx, p->Tn - 1, zero, p->representation);
      for (int j = 0; j < m->njnt; j++) { /* policy.cc:118-125: renormalise interpolated quaternions */
        const int tj = m->jnt_type[j];
        if (tj == MJPCX_JNT_FREE || tj == MJPCX_JNT_BALL) {
          double* q = xi + m->jnt_qposadr[j] + (tj == MJPCX_JNT_FREE ? 3 : 0);
          double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
          if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else for (int k = 0; k < 4; k++) q[k] /= n;
        }
      }
      ostate_diff(m, dx, xi, state, 1.0);
      for (int k = 0; k < nu; k++) { double s = 0; for (int j = 0; j < ndx; j++) s += K[k * ndx + j] * dx[j]; action[k] += p->alpha * s; }
    }
  }
  clamp_ctrl(m, action);
}

/* N rollouts sharing the nominal trajectory, differing by alpha[i]; outputs candidate-major like orollout_batch. The candidates are
 * fanned over num_threads workers, one physics arena each, as iLQGPlanner::ActionRollouts schedules them on the ThreadPool
 * (ilqg/planner.cc:630-692); a candidate's numbers do not depend on the worker that rolls it out. */
typedef struct {
  const mjpcx_model* m; const mjpcx_task* task;
  const double *state, *mocap, *times, *states, *actions, *gains, *improvement, *alpha;
  double time;
  int N, H, mode, representation, use_state, Tn;
  OBatchOut* out;
  int next, failed;
} FbBatch;

static void* fb_worker(void* arg) {
  FbBatch* b = (FbBatch*)arg;
  const mjpcx_model* m = b->m;
  const mjpcx_task* task = b->task;
  OBatchOut* out = b->out;
  const int nu = m->nu, ds = m->nq + m->nv, nr = task->num_residual, ntr = task->num_trace, H = b->H;
  OData* d = odata_new(m);
  double* r = (double*)malloc(sizeof(double) * (size_t)(nr + nu + ds));
  if (!d || !r) { __atomic_store_n(&b->failed, 1, __ATOMIC_RELAXED); free(r); if (d) odata_free(d); return NULL; }
  double *act = r + nr, *st = act + nu;
  for (;;) {
    const int i = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED);
    if (i >= b->N) break;
    FbPolicy p = {m, b->Tn, b->mode, b->representation, b->use_state, b->times, b->states, b->actions, b->gains, b->improvement, b->alpha[i]};
    odata_set_state(d, b->state, b->time, b->mocap, NULL);
    memcpy(st, b->state, sizeof(double) * ds);
    memset(act, 0, sizeof(double) * nu);
    double cur = b->time, total = 0;
    int failure = 0;
    for (int t = 0; t < H; t++) {
      const int last = t == H - 1;
      if (!last) { fb_action(&p, act, st, cur, t); odata_set_ctrl(d, act); }
      if (out->states) memcpy(out->states + ((size_t)i * H + t) * ds, st, sizeof(double) * ds);
      if (out->actions) memcpy(out->actions + ((size_t)i * H + t) * nu, act, sizeof(double) * nu);
      if (out->times) out->times[(size_t)i * H + t] = cur;
      if (last) o_forward_task(d, task, r); else o_step_task(d, task, r);
      if (out->residual) memcpy(out->residual + ((size_t)i * H + t) * nr, r, sizeof(double) * nr);
      for (int k = 0; k < ntr && out->trace; k++)
        memcpy(out->trace + (((size_t)i * H + t) * ntr + k) * 3, odata_trace_point(d, task->trace_site[k]), 3 * sizeof(double));
      if (!last && odata_warning(d)) { failure = 1; break; }
      const double c = ocost_value(task, r);
      if (out->costs) out->costs[(size_t)i * H + t] = c;
      total += c;
      if (!last) { odata_get(d, "qpos", st, m->nq); odata_get(d, "qvel", st + m->nq, m->nv); odata_get(d, "time", &cur, 1); }
    }
    out->total_return[i] = failure ? 1.0e6 : total / (H > 1 ? H : 1);
    out->failure[i] = failure;
  }
  free(r);
  odata_free(d);
  return NULL;
}

int orollout_feedback_mt(const mjpcx_model* m, const mjpcx_task* task, const double* state, double time, const double* mocap,
                         int N, int H, int mode, int representation, int use_state, int Tn, const double* times,
                         const double* states, const double* actions, const double* gains, const double* improvement,
                         const double* alpha, int num_threads, OBatchOut* out) {
  FbBatch b = {m, task, state, mocap, times, states, actions, gains, improvement, alpha, time, N, H, mode, representation, use_state, Tn, out, 0, 0};
  if (num_threads < 1) num_threads = 1;
  if (num_threads > N) num_threads = N > 0 ? N : 1;
  if (num_threads == 1) { fb_worker(&b); return b.failed ? -1 : 0; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * num_threads);
  for (int t = 0; t < num_threads; t++) pthread_create(&th[t], NULL, fb_worker, &b);
  for (int t = 0; t < num_threads; t++) pthread_join(th[t], NULL);
  free(th);
  return b.failed ? -1 : 0;
}

int orollout_feedback(const mjpcx_model* m, const mjpcx_task* task, const double* state, double time, const double* mocap,
                      int N, int H, int mode, int representation, int use_state, int Tn, const double* times,
                      const double* states, const double* actions, const double* gains, const double* improvement,
                      const double* alpha, OBatchOut* out) {
  return orollout_feedback_mt(m, task, state, time, mocap, N, H, mode, representation, use_state, Tn, times, states, actions, gains,
                              improvement, alpha, 1, out);
}

/* ModelDerivatives::Compute (model_derivatives.cc:45-106): the T calls of mjd_transitionFD fanned over num_threads workers, one
 * physics arena each (the reference schedules one task per time step on the ThreadPool). A, B, C, D time-major. */
typedef struct {
  const mjpcx_model* m; const mjpcx_task* task;
  const double *mocap, *states, *times, *actions;
  double eps;
  int T, centered;
  double *A, *B, *C, *D;
  int next, failed;
} FdBatch;

static void* fd_worker(void* arg) {
  FdBatch* b = (FdBatch*)arg;
  const mjpcx_model* m = b->m;
  const size_t ds = (size_t)(m->nq + m->nv), ndx = 2 * (size_t)m->nv, nu = (size_t)m->nu, nr = (size_t)b->task->num_residual;
  OData* d = odata_new(m);
  if (!d) { __atomic_store_n(&b->failed, 1, __ATOMIC_RELAXED); return NULL; }
  for (;;) {
    const int t = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED);
    if (t >= b->T) break;
    odata_set_state(d, b->states + t * ds, b->times[t], b->mocap, NULL);  /* (the mocap poses: otransition_fd sets the rest itself) */
    if (otransition_fd(m, b->task, d, b->states + t * ds, b->times[t], b->actions + t * nu, b->eps, b->centered, b->A + t * ndx * ndx,
                       b->B + t * ndx * nu, b->C + t * nr * ndx, b->D + t * nr * nu) != 0)
      __atomic_store_n(&b->failed, 1, __ATOMIC_RELAXED);
  }
  odata_free(d);
  return NULL;
}

int otransition_fd_batch(const mjpcx_model* m, const mjpcx_task* task, const double* mocap, int T, const double* states,
                         const double* times, const double* actions, double eps, int centered, double* A, double* B, double* C,
                         double* D, int num_threads) {
  FdBatch b = {m, task, mocap, states, times, actions, eps, T, centered, A, B, C, D, 0, 0};
  if (num_threads < 1) num_threads = 1;
  if (num_threads > T) num_threads = T > 0 ? T : 1;
  if (num_threads == 1) { fd_worker(&b); return b.failed ? -1 : 0; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * num_threads);
  for (int t = 0; t < num_threads; t++) pthread_create(&th[t], NULL, fd_worker, &b);
  for (int t = 0; t < num_threads; t++) pthread_join(th[t], NULL);
  free(th);
  return b.failed ? -1 : 0;
}
