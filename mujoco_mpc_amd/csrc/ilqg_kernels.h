// ilqg_kernels.h -- device side of the iLQG planner's data-parallel pieces for the small-model family:
//   * rollout_feedback_kernel : Trajectory::RolloutDiscrete with the time-indexed feedback policy of
//       iLQGPlanner::ActionRollouts (mjpc/planners/ilqg/planner.cc:630-692), and Trajectory::Rollout with
//       iLQGPolicy::Action (mjpc/planners/ilqg/policy.cc:82-161) of FeedbackRollouts (planner.cc:695-724)
//   * transition_fd_kernel + fd_assemble_kernel : ModelDerivatives::Compute = T x mjd_transitionFD
//       (mjpc/planners/model_derivatives.cc:45-106); one LANE per (timestep, perturbation column)
// Both reuse the per-candidate step engine of rollout_lane.h.
#pragma once
#include "rollout_lane.h"

namespace mjpcx {

template <typename T>
struct FeedbackArgs {
  const T* times;        // [Tn]            nominal trajectory (policy.trajectory)
  const T* states;       // [Tn][DS]
  const T* actions;      // [Tn][NU]
  const T* gains;        // [Tn][NU][NDX]   feedback_gain
  const T* improvement;  // [Tn][NU]        action_improvement
  const T* alpha;        // [N]             line-search step (mode 0) / feedback_scaling (mode 1)
  int Tn;
  int mode;            // 0: index policy (RolloutDiscrete), 1: continuous-time iLQGPolicy::Action
  int representation;  // mode 1: 0 zero-order, 1 linear, 2 cubic (ilqg_representation)
  int use_state;       // mode 1: settings.nominal_feedback_scaling
};

// FindInterval (mjpc/utilities.h:124-144) on a wave-uniform query
template <typename T>
__device__ __forceinline__ void find_interval(const T* xs, T value, int length, int& b0, int& b1) {
  int up = 0;
  while (up < length && xs[up] <= value) up++;  // std::upper_bound
  const int lo = up - 1;
  if (lo < 0) { b0 = b1 = 0; }
  else if (lo > length - 1) { b0 = b1 = length - 1; }
  else { b0 = lo; b1 = up < length - 1 ? up : length - 1; }
}

// Weights of Zero / Linear / CubicInterpolation (mjpc/utilities.cc:304-422) over the grid xs[0 .. length) at `value`: the
// interpolant of any series y on that grid is  w[0] y[g - 1] + w[1] y[g] + w[2] y[g + 1] + w[3] y[g + 2]  with g = bounds[0]
// (indices clamped to the grid where the weight is zero). Cubic = Hermite with finite-difference slopes (mean of the two
// neighbouring secants, one-sided at the ends of the grid, zero on a two-point grid: FiniteDifferenceSlope).
template <typename T>
struct InterpWeights { int g; int i[4]; T w[4]; };
template <typename T>
__device__ __forceinline__ InterpWeights<T> interp_weights(const T* xs, T value, int length, int representation) {
  InterpWeights<T> r;
  int b0, b1;
  find_interval(xs, value, length, b0, b1);
  r.g = b0;
  const int last = length - 1;
  r.i[0] = b0 > 0 ? b0 - 1 : 0; r.i[1] = b0; r.i[2] = b0 < last ? b0 + 1 : last; r.i[3] = b0 + 2 <= last ? b0 + 2 : last;
  r.w[0] = r.w[2] = r.w[3] = 0; r.w[1] = 1;
  if (b0 == b1 || representation == 0) return r;
  const T span = xs[b1] - xs[b0], t = (value - xs[b0]) / span;
  if (representation != 2) { r.w[1] = T(1) - t; r.w[2] = t; return r; }
  const T t2 = t * t, t3 = t2 * t;
  const T c0 = T(2) * t3 - T(3) * t2 + T(1), c1 = (t3 - T(2) * t2 + t) * span, c2 = -T(2) * t3 + T(3) * t2, c3 = (t3 - t2) * span;
  // slope at grid point b0 (never the last point here): secant(b1, b0) if b0 == 0, else the mean of the secants on both sides
  // slope at grid point b1: one-sided secant(b1, b1 - 1) if b1 is the last point (0 on a two-point grid), else the mean
  T m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};  // slopes as weights on y[g-1], y[g], y[g+1], y[g+2]
  const T is1 = T(1) / span;
  if (b0 == 0) { m0[1] = -is1; m0[2] = is1; }
  else { const T isl = T(1) / (xs[b0] - xs[b0 - 1]); m0[0] = -T(0.5) * isl; m0[1] = T(0.5) * isl - T(0.5) * is1; m0[2] = T(0.5) * is1; }
  if (b1 == last) { if (length > 2) { m1[1] = -is1; m1[2] = is1; } }
  else { const T isr = T(1) / (xs[b1 + 1] - xs[b1]); m1[1] = -T(0.5) * is1; m1[2] = T(0.5) * is1 - T(0.5) * isr; m1[3] = T(0.5) * isr; }
  r.w[0] = c1 * m0[0];
  r.w[1] = c0 + c1 * m0[1] + c3 * m1[1];
  r.w[2] = c2 + c1 * m0[2] + c3 * m1[2];
  r.w[3] = c3 * m1[3];
  return r;
}

template <class TP, class TK, typename T, class MC>
__global__ __launch_bounds__(64) void rollout_feedback_kernel(const LaneModel<T> m_karg, const LaneTask<T> tk,
                                                               const RolloutArgs<T> a, const FeedbackArgs<T> fb) {
  decltype(auto) m = MC::template get<T>(m_karg);
  constexpr int NV = TP::NV, NU = TP::NU, NS = TP::NSITE, NR = TK::NR, DS = 2 * NV, NDX = 2 * NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // the nominal trajectory is shared by all candidates: staged once into LDS, broadcast reads afterwards
  T* l_times = reinterpret_cast<T*>(smem_raw);
  T* l_states = l_times + fb.Tn;
  T* l_actions = l_states + (size_t)fb.Tn * DS;
  T* l_gains = l_actions + (size_t)fb.Tn * NU;
  T* l_impr = l_gains + (size_t)fb.Tn * NU * NDX;
  for (int i = threadIdx.x; i < fb.Tn; i += 64) l_times[i] = fb.times[i];
  for (int i = threadIdx.x; i < fb.Tn * DS; i += 64) l_states[i] = fb.states[i];
  for (int i = threadIdx.x; i < fb.Tn * NU; i += 64) { l_actions[i] = fb.actions[i]; l_impr[i] = fb.improvement[i]; }
  for (int i = threadIdx.x; i < fb.Tn * NU * NDX; i += 64) l_gains[i] = fb.gains[i];
  __syncthreads();

  const int lane = threadIdx.x;
  const int cand = blockIdx.x * 64 + lane;
  const bool live = cand < a.N;
  const int ci = live ? cand : a.N - 1;
  const T alpha = fb.alpha[ci];
  const int H = a.H;

  T qpos[NV], qvel[NV], ctrl[NU];
#pragma unroll
  for (int i = 0; i < NV; i++) { qpos[i] = tk.qpos[i]; qvel[i] = tk.qvel[i]; }
#pragma unroll
  for (int k = 0; k < NU; k++) ctrl[k] = 0;
  T time = tk.time;
  const T h = m.timestep;
  double total = 0;
  bool failed = false;

  for (int t = 0; t < H; t++) {
    const bool last = (t == H - 1);
    bool bad = false;
    if (!last) {
      T x[DS];
#pragma unroll
      for (int i = 0; i < NV; i++) { x[i] = qpos[i]; x[NV + i] = qvel[i]; }
      if (fb.mode == 0) {
        // u = (ubar_t + alpha du_t) + K_t (x - xbar_t), planner.cc:640-668
        const int tt = t < fb.Tn ? t : fb.Tn - 1;
#pragma unroll
        for (int k = 0; k < NU; k++) {
          T u = l_actions[tt * NU + k] + alpha * l_impr[tt * NU + k];
#pragma unroll
          for (int j = 0; j < NDX; j++) u += l_gains[(tt * NU + k) * NDX + j] * (x[j] - l_states[tt * DS + j]);
          ctrl[k] = u;
        }
      } else {
        // iLQGPolicy::Action, policy.cc:82-161
        int b0, b1;
        find_interval(l_times, time, fb.Tn, b0, b1);
        const int rep = (b0 == b1) ? 0 : fb.representation;
        const InterpWeights<T> wa = interp_weights(l_times, time, fb.Tn - 1, rep);  // actions / gains: horizon - 1 entries
        const InterpWeights<T> ws = interp_weights(l_times, time, fb.Tn, rep);      // states: horizon entries
        T dx[NDX];
        if (fb.use_state) {
#pragma unroll
          for (int j = 0; j < NDX; j++) {
            T xi = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) xi += ws.w[q] * l_states[ws.i[q] * DS + j];
            dx[j] = x[j] - xi;
          }
        }
#pragma unroll
        for (int k = 0; k < NU; k++) {
          T u = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) u += wa.w[q] * l_actions[wa.i[q] * NU + k];
          if (fb.use_state) {
            T fbk = 0;
#pragma unroll
            for (int j = 0; j < NDX; j++) {
              T g = 0;
#pragma unroll
              for (int q = 0; q < 4; q++) g += wa.w[q] * l_gains[(wa.i[q] * NU + k) * NDX + j];
              fbk += g * dx[j];
            }
            u += alpha * fbk;  // feedback_scaling
          }
          ctrl[k] = u;
        }
      }
#pragma unroll
      for (int k = 0; k < NU; k++) {
        bad |= is_bad(ctrl[k]);
        ctrl[k] = clampv(ctrl[k], m.act_ctrlrange[k][0], m.act_ctrlrange[k][1]);
      }
#pragma unroll
      for (int i = 0; i < NV; i++) bad |= is_bad(qpos[i]) || is_bad(qvel[i]);
    }
    T qacc[NV], qfrc[NV], qfrc_c[NV], M[NV][NV];
    T site_xpos[NS > 0 ? NS : 1][3];
    lane_forward<TP, T>(m, tk, qpos, qvel, ctrl, qacc, qfrc, qfrc_c, M, site_xpos);
    if (!last) {
#pragma unroll
      for (int i = 0; i < NV; i++) bad |= is_bad(qacc[i]);
    }
    T r[NR];
    lane_residual<TP, TK, T>(tk, qpos, qvel, ctrl, r);
    const T cost = lane_cost<TK, T>(tk, r);
    if (live && !failed) lane_record<TP, TK, T>(a, t, cand, qpos, qvel, ctrl, time, r, site_xpos, cost, bad);
    if (bad) failed = true;
    total += (double)cost;
    if (last) break;
    lane_integrate<TP, T>(m, tk, qpos, qvel, ctrl, qacc, qfrc, qfrc_c, M, nullptr, site_xpos);
    if (m.integrator == 1 && live && !failed) lane_record_trace<TP, TK, T>(a, t, cand, site_xpos);
    time += h;
  }
  if (live) {
    a.total_return[cand] = failed ? kMaxReturn : total / (double)(H > 1 ? H : 1);
    a.failure[cand] = failed ? 1 : 0;
  }
}

// ------------------------------------------------------------------ finite-difference transition derivatives
template <typename T>
struct FdArgs {
  const T* times;    // [Tn]
  const T* states;   // [Tn][DS]
  const T* actions;  // [Tn][NU]
  int Tn;
  T eps;
  T* next;      // [Tn][NC][NDX]  next state of each perturbed step
  T* sensor;    // [Tn][NC][NR]   residual (the leading user sensors) of each perturbed step
};

// columns: 0 nominal | 1..NDX: +eps on x_j | NDX+1..2NDX: -eps | then +eps on u_k | then -eps on u_k
template <class TP> constexpr int fd_columns() { return 1 + 2 * (2 * TP::NV + TP::NU); }

template <class TP, class TK, typename T, class MC>
__global__ __launch_bounds__(64) void transition_fd_kernel(const LaneModel<T> m_karg, const LaneTask<T> tk,
                                                            const FdArgs<T> f) {
  decltype(auto) m = MC::template get<T>(m_karg);
  constexpr int NV = TP::NV, NU = TP::NU, NS = TP::NSITE, NR = TK::NR, DS = 2 * NV, NDX = 2 * NV;
  constexpr int NC = fd_columns<TP>();
  const int item = blockIdx.x * 64 + threadIdx.x;
  if (item >= f.Tn * NC) return;
  const int t = item / NC, c = item % NC;
  T qpos[NV], qvel[NV], ctrl[NU];
#pragma unroll
  for (int i = 0; i < NV; i++) { qpos[i] = f.states[t * DS + i]; qvel[i] = f.states[t * DS + NV + i]; }
#pragma unroll
  for (int k = 0; k < NU; k++) ctrl[k] = f.actions[t * NU + k];
  // perturb (slide/hinge: mj_integratePos of a unit tangent is qpos[j] += eps)
  if (c >= 1) {
    const int cc = c - 1;
    const T sgn = (cc < NDX || (cc >= 2 * NDX && cc < 2 * NDX + NU)) ? T(1) : T(-1);
    const int j = cc < 2 * NDX ? cc % NDX : (cc - 2 * NDX) % NU;
    if (cc < 2 * NDX) {
#pragma unroll
      for (int i = 0; i < NV; i++) {
        if (j == i) qpos[i] += sgn * f.eps;
        if (j == NV + i) qvel[i] += sgn * f.eps;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NU; k++)
        if (j == k) ctrl[k] += sgn * f.eps;
    }
  }
  T qacc[NV], qfrc[NV], qfrc_c[NV], M[NV][NV];
  T site_xpos[NS > 0 ? NS : 1][3];
  lane_forward<TP, T>(m, tk, qpos, qvel, ctrl, qacc, qfrc, qfrc_c, M, site_xpos);
  T r[NR];
  // sensors see the clamped control, as data->ctrl is clamped only inside mj_fwdActuation: the residual reads
  // data->ctrl (unclamped) in the reference, so pass the raw perturbed value
  lane_residual<TP, TK, T>(tk, qpos, qvel, ctrl, r);
  lane_integrate<TP, T>(m, tk, qpos, qvel, ctrl, qacc, qfrc, qfrc_c, M);
  T* y = f.next + ((size_t)t * NC + c) * NDX;
#pragma unroll
  for (int i = 0; i < NV; i++) { y[i] = qpos[i]; y[NV + i] = qvel[i]; }
  T* s = f.sensor + ((size_t)t * NC + c) * NR;
#pragma unroll
  for (int i = 0; i < NR; i++) s[i] = r[i];
}

// A[t] = dy/dx (NDX x NDX), B[t] = dy/du (NDX x NU), C[t] = ds/dx (NR x NDX), D[t] = ds/du (NR x NU), row-major,
// from the perturbed steps (mjd_transitionFD: forward or centred differences; control nudges respect ctrlrange)
template <typename T>
__global__ void fd_assemble_kernel(const T* __restrict__ next, const T* __restrict__ sensor, const T* __restrict__ actions,
                                   const T* __restrict__ ctrlrange, const int* __restrict__ ctrllimited, int Tn, int ndx,
                                   int nu, int nr, T eps, int centered, double* A, double* B, double* C, double* D) {
  const int nc = 1 + 2 * (ndx + nu);
  const int rows = ndx + nr, cols = ndx + nu;
  const int total = Tn * rows * cols;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int t = idx / (rows * cols), rem = idx % (rows * cols);
    const int i = rem / cols, j = rem % cols;  // output row (state or sensor), input column (state or control)
    bool fwd = true, back = centered != 0;
    int cp, cm;
    if (j < ndx) { cp = 1 + j; cm = 1 + ndx + j; }
    else {
      const int k = j - ndx;
      cp = 1 + 2 * ndx + k; cm = 1 + 2 * ndx + nu + k;
      if (ctrllimited[k]) {  // inRange nudges of mjd_transitionFD
        const T u = actions[t * nu + k], lo = ctrlrange[2 * k], hi = ctrlrange[2 * k + 1];
        fwd = (u >= lo && u <= hi && u + eps >= lo && u + eps <= hi);
        const bool can_back = (u - eps >= lo && u - eps <= hi && u >= lo && u <= hi);
        back = (centered || !fwd) && can_back;
      }
    }
    auto val = [&](int c) -> double {
      return i < ndx ? (double)next[((size_t)t * nc + c) * ndx + i] : (double)sensor[((size_t)t * nc + c) * nr + (i - ndx)];
    };
    double d = 0;
    if (fwd && back) d = (val(cp) - val(cm)) / (2 * (double)eps);
    else if (fwd) d = (val(cp) - val(0)) / (double)eps;
    else if (back) d = (val(0) - val(cm)) / (double)eps;
    if (i < ndx) { if (j < ndx) A[((size_t)t * ndx + i) * ndx + j] = d; else B[((size_t)t * ndx + i) * nu + (j - ndx)] = d; }
    else { if (j < ndx) C[((size_t)t * nr + (i - ndx)) * ndx + j] = d; else D[((size_t)t * nr + (i - ndx)) * nu + (j - ndx)] = d; }
  }
}

}  // namespace mjpcx
