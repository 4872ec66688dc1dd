#include "planner.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>

#include "../../utilities.h"

namespace mjpc {

// ilqg/planner.cc:41-75
void GpuILQGPlanner::Initialize(mjModel* m, const Task& t) {
  model = m;
  task = &t;
  dim_state = m->nq + m->nv + m->na;
  dim_state_derivative = 2 * m->nv + m->na;
  dim_action = m->nu;
  dim_sensor = t.num_residual;
  num_rollouts_gui_ = GetNumberOrDefault(10, m, "ilqg_num_rollouts");
  settings.regularization_type = GetNumberOrDefault(settings.regularization_type, m, "ilqg_regularization_type");
  num_trajectory_ = num_rollouts_gui_;
}

// ilqg/planner.cc:78-113
void GpuILQGPlanner::Allocate() {
  state.resize(dim_state);
  mocap.resize(7 * (size_t)model->nmocap);
  userdata.resize(model->nuserdata);
  for (iLQGPolicy* p : {&policy, &previous_policy, &candidate_policy0, &winner_policy_})
    p->Allocate(model, *task, kMaxTrajectoryHorizon);
  // gradient-based planners plan on the differentiable model copy unless agent_differentiable says otherwise (agent.cc:156-164)
  const bool differentiable = GetNumberOrDefault(1, model, "agent_differentiable") != 0;
  ctx_ = std::make_unique<gpu::Context>(model, *task, device_, precision_, differentiable);
}

// ilqg/planner.cc:116-153 + iLQGBackwardPass::Reset (backward_pass.cc:50-62)
void GpuILQGPlanner::Reset(int horizon, const double* initial_repeated_action) {
  std::fill(state.begin(), state.end(), 0.0);
  std::fill(mocap.begin(), mocap.end(), 0.0);
  std::fill(userdata.begin(), userdata.end(), 0.0);
  time = 0.0;
  {
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    policy.Reset(horizon, initial_repeated_action);
    previous_policy.Reset(horizon, initial_repeated_action);
  }
  candidate_policy0.Reset(horizon, initial_repeated_action);
  winner_policy_.Reset(horizon, initial_repeated_action);
  regularization = 1.0;
  regularization_rate = 1.0;
  regularization_factor = 2.0;
  dV[0] = dV[1] = 0;
  action_step = feedback_scaling = improvement = expected = surprise = 0.0;
  derivative_skip_ = GetNumberOrDefault(0, model, "derivative_skip");
  winner = 0;
}

void GpuILQGPlanner::SetState(const State& s) { s.CopyTo(state.data(), mocap.data(), userdata.data(), &time); }

void GpuILQGPlanner::ScaleRegularization(double factor, double reg_min, double reg_max) {
  if (factor > 1) regularization_rate = mju_max(regularization_rate * factor, factor);
  else regularization_rate = mju_min(regularization_rate * factor, factor);
  regularization = mju_min(mju_max(regularization * regularization_rate, reg_min), reg_max);
}

void GpuILQGPlanner::UpdateRegularization(double reg_min, double reg_max, double z, double s) {
  auto bad = [](double v) { return !std::isfinite(v) || std::fabs(v) > 1e10; };
  const double f = regularization_factor;
  if (bad(z) || bad(s)) ScaleRegularization(f * f, reg_min, reg_max);
  else if (z > 0.5 || s > 0.3) ScaleRegularization(1.0 / f, reg_min, reg_max);
  else if (z < 0.1 || s < 0.06) ScaleRegularization(f, reg_min, reg_max);
}

// log-spaced steps in [min_linesearch_step, 1] plus a zero step (planner.cc:179-182, 535-538)
void GpuILQGPlanner::LineSearchSteps() {
  const int n = num_trajectory_;
  linesearch_steps.assign(n, 0.0);
  if (n > 1) LogScale(linesearch_steps.data(), 1.0, settings.min_linesearch_step, n - 1);
  linesearch_steps[n - 1] = 0.0;
}

int GpuILQGPlanner::BestRollout(const std::vector<double>& ret, const std::vector<std::int32_t>& fail) {
  int best = -1;
  double best_return = 0;
  for (int j = (int)ret.size() - 1; j >= 0; j--) {
    if (fail[j]) continue;
    if (best == -1 || ret[j] < best_return) { best = j; best_return = ret[j]; }
  }
  return best;
}

// ilqg/planner.cc:156-164
void GpuILQGPlanner::OptimizePolicy(int horizon, ThreadPool& pool) {
  num_trajectory_ = num_rollouts_gui_;  // the reference clamps to kMaxTrajectory = 128 here (lifted)
  NominalTrajectory(horizon, pool);
  Iteration(horizon, pool);
}

void GpuILQGPlanner::TakeTrajectory(iLQGPolicy* p, int index) { ctx_->FetchTrajectory(index, &p->trajectory); }

// ilqg/planner.cc:167-223 with FeedbackRollouts (:695-724)
void GpuILQGPlanner::NominalTrajectory(int horizon, ThreadPool& pool) {
  if (num_trajectory_ == 0) return;
  const auto start = std::chrono::steady_clock::now();
  policy.trajectory.horizon = horizon;
  LineSearchSteps();
  const Trajectory& tr = policy.trajectory;
  ctx_->SyncTask(*task);  // the per-plan frozen ResidualFn copy (agent.cc:319)
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state.data(), time, mocap.data(), userdata.data()));
  ctx_->Check(mjpcx_rollout_feedback(ctx_->handle(), num_trajectory_, horizon, /*mode=*/1, policy.representation,
                                     settings.nominal_feedback_scaling, horizon, tr.times.data(), tr.states.data(),
                                     tr.actions.data(), policy.feedback_gain.data(), policy.action_improvement.data(),
                                     linesearch_steps.data()));
  returns_.resize(num_trajectory_);
  failure_.resize(num_trajectory_);
  ctx_->Check(mjpcx_get_returns(ctx_->handle(), returns_.data(), failure_.data()));
  const int best = BestRollout(returns_, failure_);
  if (best == -1) {
    candidate_policy0.trajectory = policy.trajectory;
    feedback_scaling = 0.0;
  } else {
    TakeTrajectory(&candidate_policy0, best);
    feedback_scaling = linesearch_steps[best];
  }
  const size_t ng = (size_t)horizon * dim_action * dim_state_derivative, na = (size_t)horizon * dim_action;
  std::copy_n(policy.feedback_gain.begin(), ng, candidate_policy0.feedback_gain.begin());
  std::copy_n(policy.action_improvement.begin(), na, candidate_policy0.action_improvement.begin());
  candidate_policy0.representation = policy.representation;
  nominal_compute_time = GetDuration(start);
}

// ModelDerivatives::Compute with skip + linear interpolation (model_derivatives.cc:45-165)
void GpuILQGPlanner::ModelDerivatives(const Trajectory& tr, int T) {
  const int n = dim_state_derivative, m = dim_action, nr = dim_sensor, ds = dim_state;
  const int s = derivative_skip_ + 1;
  std::vector<int> evaluate;
  evaluate.push_back(0);
  for (int t = s; t < T - s; t += s) evaluate.push_back(t);
  evaluate.push_back(T - 2);
  evaluate.push_back(T - 1);
  std::sort(evaluate.begin(), evaluate.end());
  evaluate.erase(std::unique(evaluate.begin(), evaluate.end()), evaluate.end());
  evaluate.erase(std::remove_if(evaluate.begin(), evaluate.end(), [T](int e) { return e < 0 || e >= T; }), evaluate.end());
  const int E = (int)evaluate.size();
  etimes_.resize(E); estates_.resize((size_t)E * ds); eactions_.resize((size_t)E * m);
  for (int k = 0; k < E; k++) {
    const int t = evaluate[k];
    etimes_[k] = tr.times[t];
    std::copy_n(tr.states.begin() + (size_t)t * ds, ds, estates_.begin() + (size_t)k * ds);
    std::copy_n(tr.actions.begin() + (size_t)t * m, m, eactions_.begin() + (size_t)k * m);
  }
  const size_t sA = (size_t)n * n, sB = (size_t)n * m, sC = (size_t)nr * n, sD = (size_t)nr * m;
  eA_.resize(E * sA); eB_.resize(E * sB); eC_.resize(E * sC); eD_.resize(E * sD);
  ctx_->Check(mjpcx_transition_fd(ctx_->handle(), E, etimes_.data(), estates_.data(), eactions_.data(), settings.fd_tolerance,
                                  settings.fd_mode != 0, eA_.data(), eB_.data(), eC_.data(), eD_.data()));
  A_.assign(T * sA, 0.0); B_.assign(T * sB, 0.0); C_.assign(T * sC, 0.0); D_.assign(T * sD, 0.0);
  int k = 0;
  for (int t = 0; t < T; t++) {
    while (k + 1 < E && evaluate[k + 1] <= t) k++;
    const int e0 = k, e1 = std::min(k + 1, E - 1);
    const double tt = (evaluate[e0] == t || e0 == e1) ? 0.0 : double(t - evaluate[e0]) / double(evaluate[e1] - evaluate[e0]);
    auto mix = [&](std::vector<double>& full, const std::vector<double>& ev, size_t sz) {
      for (size_t i = 0; i < sz; i++) full[t * sz + i] = ev[e0 * sz + i] * (1.0 - tt) + ev[e1 * sz + i] * tt;
    };
    mix(A_, eA_, sA); mix(B_, eB_, sB); mix(C_, eC_, sC); mix(D_, eD_, sD);
  }
  // the last step has no transition: model_derivatives.cc:88-92 computes only C there
  std::fill(A_.begin() + (T - 1) * sA, A_.end(), 0.0);
  std::fill(B_.begin() + (T - 1) * sB, B_.end(), 0.0);
  std::fill(D_.begin() + (T - 1) * sD, D_.end(), 0.0);
}

// ilqg/planner.cc:377-627
void GpuILQGPlanner::Iteration(int horizon, ThreadPool& pool) {
  iLQGPolicy& c0 = candidate_policy0;
  Trajectory& tr = c0.trajectory;
  const int T = horizon, n = dim_state_derivative, m = dim_action;
  const double previous_return = tr.total_return;
  LineSearchSteps();
  ctx_->SyncTask(*task);

  auto start = std::chrono::steady_clock::now();
  ModelDerivatives(tr, T);
  model_derivative_compute_time = GetDuration(start);

  start = std::chrono::steady_clock::now();
  cx_.resize((size_t)T * n); cu_.resize((size_t)T * m); cxx_.resize((size_t)T * n * n); cxu_.resize((size_t)T * n * m);
  cuu_.resize((size_t)T * m * m);
  ctx_->Check(mjpcx_cost_derivatives(ctx_->handle(), T, tr.residual.data(), C_.data(), D_.data(), cx_.data(), cu_.data(),
                                     cxx_.data(), cxu_.data(), cuu_.data()));
  cost_derivative_compute_time = GetDuration(start);

  // ---- backward pass with regularisation retries (planner.cc:429-520)
  start = std::chrono::steady_clock::now();
  Vx_.resize((size_t)T * n); Vxx_.resize((size_t)T * n * n); K_.resize((size_t)T * m * n); du_.resize((size_t)T * m);
  bool ok = false;
  int reg_iter = 0;
  double dv[2] = {0, 0};
  while (reg_iter < settings.max_regularization_iterations && !ok) {
    std::int32_t status = 0;
    ctx_->Check(mjpcx_backward_pass(ctx_->handle(), n, m, T, regularization, settings.regularization_type, settings.action_limits,
                                    A_.data(), B_.data(), cx_.data(), cu_.data(), cxx_.data(), cxu_.data(), cuu_.data(),
                                    tr.actions.data(), model->actuator_ctrlrange, Vx_.data(), Vxx_.data(), K_.data(), du_.data(),
                                    dv, &status, nullptr));
    ok = status != 0;
    if (!ok && regularization <= settings.max_regularization) {
      ScaleRegularization(regularization_factor, settings.min_regularization, settings.max_regularization);
      reg_iter++;
    } else if (!ok) {
      break;
    }
  }
  backward_pass_compute_time = GetDuration(start);
  if (!ok) return;
  dV[0] = dv[0];
  dV[1] = dv[1];
  std::copy(K_.begin(), K_.end(), c0.feedback_gain.begin());
  std::copy(du_.begin(), du_.end(), c0.action_improvement.begin());

  // ---- ActionRollouts (planner.cc:630-692): line search over the improvement step
  start = std::chrono::steady_clock::now();
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state.data(), time, mocap.data(), userdata.data()));
  ctx_->Check(mjpcx_rollout_feedback(ctx_->handle(), num_trajectory_, T, /*mode=*/0, 0, 1, T, tr.times.data(), tr.states.data(),
                                     tr.actions.data(), c0.feedback_gain.data(), c0.action_improvement.data(),
                                     linesearch_steps.data()));
  returns_.resize(num_trajectory_);
  failure_.resize(num_trajectory_);
  ctx_->Check(mjpcx_get_returns(ctx_->handle(), returns_.data(), failure_.data()));
  const int best = BestRollout(returns_, failure_);
  if (best == -1) return;
  winner = best;
  // candidate_policy[winner]: the nominal trajectory with actions += step * improvement, NOT re-rolled
  // (planner.cc:556-572 copies candidate_policy[0] into every candidate before the rollouts overwrite trajectory[i])
  winner_policy_.CopyFrom(c0, T);
  winner_policy_.representation = c0.representation;
  for (size_t i = 0; i < (size_t)T * m; i++)
    winner_policy_.trajectory.actions[i] = tr.actions[i] + linesearch_steps[best] * c0.action_improvement[i];
  TakeTrajectory(&c0, best);
  if (best == 0) winner_policy_.trajectory = c0.trajectory;
  action_step = linesearch_steps[best];
  expected = -1.0 * action_step * (dV[0] + action_step * dV[1]) + 1.0e-16;
  improvement = previous_return - returns_[best];
  surprise = mju_min(mju_max(0.0, improvement / expected), 2.0);
  UpdateRegularization(settings.min_regularization, settings.max_regularization, surprise, action_step);
  rollouts_compute_time = GetDuration(start);

  start = std::chrono::steady_clock::now();
  {
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    previous_policy.CopyFrom(policy, T);
    previous_policy.feedback_scaling = policy.feedback_scaling;
    policy.CopyFrom(winner_policy_, T);
    policy.feedback_scaling = 1.0;
  }
  policy_update_compute_time = GetDuration(start);
}

void GpuILQGPlanner::ActionFromPolicy(double* action, const double* s, double t, bool use_previous) {
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  (use_previous ? previous_policy : policy).Action(action, s, t);
}

const Trajectory* GpuILQGPlanner::BestTrajectory() {
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  return &policy.trajectory;
}

}  // namespace mjpc
