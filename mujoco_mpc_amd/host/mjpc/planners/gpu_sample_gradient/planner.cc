#include "planner.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>

#include "../../utilities.h"

namespace mjpc {

using spline::SplineInterpolation;
using spline::TimeSpline;

void GpuSampleGradientPlanner::Initialize(mjModel* m, const Task& t) {
  model = m;
  task = &t;
  noise_exploration = GetNumberOrDefault(0.1, m, "sampling_exploration");
  num_trajectory_ = GetNumberOrDefault(10, m, "sampling_trajectories");
  interpolation_ = (SplineInterpolation)GetNumberOrDefault((int)spline::kCubicSpline, m, "sampling_representation");
  num_gradient_ = GetNumberOrDefault(0, m, "sample_gradient_trajectories");
  gradient_filter_ = GetNumberOrDefault(1.0, m, "sample_gradient_filter");
}

void GpuSampleGradientPlanner::EnsureCandidates(int n) {
  const size_t stride = (size_t)model->nu * kMaxTrajectoryHorizon;
  if ((int)candidate_policy.size() < n) {
    const size_t old = candidate_policy.size();
    candidate_policy.resize(n);
    for (size_t i = old; i < (size_t)n; i++) {
      candidate_policy[i].Allocate(model, *task, kMaxTrajectoryHorizon);
      candidate_policy[i].Reset(policy.num_spline_points > 0 ? policy.num_spline_points : 1);
    }
    noise.resize((size_t)n * stride, 0.0);
    const size_t oldo = trajectory_order.size();
    trajectory_order.resize(n);
    for (size_t i = oldo; i < (size_t)n; i++) trajectory_order[i] = (int)i;
    returns.resize(n, 0.0);
    failure.resize(n, 0);
  }
}

void GpuSampleGradientPlanner::Allocate() {
  state.resize(model->nq + model->nv + model->na);
  mocap.resize(7 * (size_t)model->nmocap);
  userdata.resize(model->nuserdata);
  const int num_max_parameter = model->nu * kMaxTrajectoryHorizon;
  policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  resampled_policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  previous_policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  plan_scratch = TimeSpline(model->nu);
  candidate_policy.clear(); noise.clear(); trajectory_order.clear(); returns.clear(); failure.clear();
  EnsureCandidates(std::max(num_trajectory_, 1));
  gradient.assign(num_max_parameter, 0.0);
  gradient_previous.assign(num_max_parameter, 0.0);
  best_.Initialize((int)state.size(), model->nu, task->num_residual, task->num_trace, kMaxTrajectoryHorizon);
  best_.Allocate(kMaxTrajectoryHorizon);
  ctx_ = std::make_unique<gpu::Context>(model, *task, device_, precision_);  // throws if no device kernel covers the model
}

void GpuSampleGradientPlanner::Reset(int horizon, const double* initial_repeated_action) {
  std::fill(state.begin(), state.end(), 0.0);
  std::fill(mocap.begin(), mocap.end(), 0.0);
  std::fill(userdata.begin(), userdata.end(), 0.0);
  time = 0.0;
  {
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    policy.Reset(horizon, initial_repeated_action);
    previous_policy.Reset(horizon, initial_repeated_action);
  }
  resampled_policy.Reset(horizon, initial_repeated_action);
  plan_scratch.Clear();
  std::fill(noise.begin(), noise.end(), 0.0);
  for (auto& c : candidate_policy) c.Reset(horizon);
  std::fill(returns.begin(), returns.end(), 0.0);
  std::fill(failure.begin(), failure.end(), 0);
  best_.Reset(kMaxTrajectoryHorizon);
  best_valid_ = false;
  improvement = 0.0;
  winner = 0;
  std::fill(gradient.begin(), gradient.end(), 0.0);
  std::fill(gradient_previous.begin(), gradient_previous.end(), 0.0);
  return_weight_.clear();
  step_size_.clear();
}

void GpuSampleGradientPlanner::SetState(const State& s) { s.CopyTo(state.data(), mocap.data(), userdata.data(), &time); }

// planner.cc:166-264
void GpuSampleGradientPlanner::OptimizePolicy(int horizon, ThreadPool& pool) {
  const int num_trajectory = num_trajectory_;
  num_gradient_ = std::min(num_gradient_, num_trajectory - 1);
  const int num_gradient = num_gradient_;
  const int num_noisy = num_trajectory - num_gradient;
  EnsureCandidates(num_trajectory);

  const int num_spline_points = policy.num_spline_points;
  policy.plan.SetInterpolation(interpolation_);
  {
    const std::shared_lock<std::shared_mutex> lock(mtx_);
    resampled_policy.CopyFrom(policy, num_spline_points);
  }
  ResamplePolicy(resampled_policy, horizon, num_spline_points);                      // nominal at the current time
  for (int i = 0; i < num_gradient; i++) ResamplePolicy(candidate_policy[num_noisy + i], horizon, num_spline_points);

  const auto rollouts_start = std::chrono::steady_clock::now();
  Rollouts(num_trajectory, num_gradient, horizon);  // p + s * N(0, 1), and the gradient candidates of the previous step
  rollouts_compute_time = GetDuration(rollouts_start);

  const auto update_start = std::chrono::steady_clock::now();
  for (int i = 0; i < num_trajectory; i++) trajectory_order[i] = i;
  std::partial_sort(trajectory_order.begin(), trajectory_order.begin() + num_trajectory, trajectory_order.begin() + num_trajectory,
                    [this](int a, int b) { return returns[a] < returns[b]; });
  winner = returns[trajectory_order[0]] < returns[idx_nominal] ? trajectory_order[0] : idx_nominal;
  if (winner > idx_nominal) winner_type_ = winner < num_trajectory - num_gradient ? kPerturb : kGradient;
  else winner_type_ = kNominal;
  {
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    previous_policy = policy;
    policy.SetPlan(candidate_policy[winner].plan);
  }
  best_valid_ = false;
  improvement = mju_max(returns[idx_nominal] - returns[winner], 0.0);
  policy_update_compute_time = GetDuration(update_start);

  const auto gradient_start = std::chrono::steady_clock::now();
  GradientCandidates(num_trajectory, num_gradient, horizon);  // evaluated at the NEXT planning iteration
  gradient_candidates_compute_time = GetDuration(gradient_start);
  iteration++;
}

void GpuSampleGradientPlanner::NominalTrajectory(int horizon, ThreadPool& pool) {
  const TimeSpline& plan = resampled_policy.plan;
  std::vector<double> times(plan.times()), values(plan.values());
  if (times.empty()) { times.assign(1, time); values.assign(model->nu, 0.0); }
  ctx_->SyncTask(*task);
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state.data(), time, mocap.data(), userdata.data()));
  ctx_->Check(mjpcx_rollout_splines(ctx_->handle(), 1, horizon, (int)times.size(), (int)plan.Interpolation(), times.data(), values.data()));
  num_rolled_ = 1;
  winner = idx_nominal;
  ctx_->FetchTrajectory(0, &best_);
  best_valid_ = true;
}

void GpuSampleGradientPlanner::ActionFromPolicy(double* action, const double* s, double t, bool use_previous) {
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  (use_previous ? previous_policy : policy).Action(action, s, t);
}

// planner.cc:290-318
void GpuSampleGradientPlanner::ResamplePolicy(SamplingPolicy& p, int horizon, int num_spline_points) {
  double nominal_time = time;
  const double time_shift = mju_max((horizon - 1) * model->opt.timestep / (num_spline_points - 1), 1.0e-5);
  plan_scratch.Clear();
  plan_scratch.Reserve(num_spline_points);
  plan_scratch.SetInterpolation(p.plan.Interpolation());
  for (int t = 0; t < num_spline_points; t++) {
    TimeSpline::Node node = plan_scratch.AddNode(nominal_time);
    p.Action(node.values().data(), /*state=*/nullptr, nominal_time);
    nominal_time += time_shift;
  }
  p.SetPlan(plan_scratch);
  p.num_spline_points = num_spline_points;
}

// planner.cc:321-352: standard normals per parameter (kept for the gradient estimate), scaled by noise_exploration, clamped
void GpuSampleGradientPlanner::AddNoiseToPolicy(int i) {
  const int num_spline_points = candidate_policy[i].num_spline_points;
  const size_t shift = (size_t)i * ((size_t)model->nu * kMaxTrajectoryHorizon);
  const int np = num_spline_points * model->nu;
  for (int k = 0; k < np; k += 2) {
    double z[2];
    HostGaussianPair(seed_, (std::uint32_t)i, (std::uint32_t)(k >> 1), iteration, z);
    noise[shift + k] = z[0];
    if (k + 1 < np) noise[shift + k + 1] = z[1];
  }
  for (int j = 0; j < num_spline_points; j++) {
    TimeSpline::Node node = candidate_policy[i].plan.NodeAt(j);
    for (int k = 0; k < model->nu; k++) node.values()[k] += noise_exploration * noise[shift + (size_t)j * model->nu + k];
    Clamp(node.values().data(), model->actuator_ctrlrange, model->nu);
  }
}

// planner.cc:355-400: candidates 0 .. num_noisy-1 = nominal (+ noise for i > 0), the rest are the gradient candidates;
// ONE device launch replaces the thread-pool fan-out
void GpuSampleGradientPlanner::Rollouts(int num_trajectory, int num_gradient, int horizon) {
  const auto noise_start = std::chrono::steady_clock::now();
  for (int i = 0; i < num_trajectory - num_gradient; i++) {
    candidate_policy[i].CopyFrom(resampled_policy, resampled_policy.num_spline_points);
    if (i > idx_nominal) AddNoiseToPolicy(i);
  }
  noise_compute_time = GetDuration(noise_start);
  const TimeSpline& nominal = resampled_policy.plan;
  const size_t np = nominal.values().size();
  std::vector<double> values((size_t)num_trajectory * np);
  for (int i = 0; i < num_trajectory; i++) {
    const std::vector<double>& v = candidate_policy[i].plan.values();
    if (v.size() != np) throw gpu::Error(MJPCX_EINVAL, "candidate splines of unequal size");
    std::copy(v.begin(), v.end(), values.begin() + (size_t)i * np);
  }
  ctx_->SyncTask(*task);
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state.data(), time, mocap.data(), userdata.data()));
  ctx_->Check(mjpcx_rollout_splines(ctx_->handle(), num_trajectory, horizon, (int)nominal.Size(), (int)nominal.Interpolation(),
                                    nominal.times().data(), values.data()));
  ctx_->Check(mjpcx_get_returns(ctx_->handle(), returns.data(), failure.data()));
  num_rolled_ = num_trajectory;
}

// planner.cc:403-480
void GpuSampleGradientPlanner::GradientCandidates(int num_trajectory, int num_gradient, int horizon) {
  if (num_gradient < 1) return;
  const int num_spline_points = resampled_policy.num_spline_points;
  const int num_parameters = num_spline_points * model->nu;
  mju_copy(gradient_previous.data(), gradient.data(), num_parameters);
  const int num_noisy = num_trajectory - num_gradient;
  // fitness shaping (Wierstra et al. 2014); the weights are computed when the number of noisy samples changes, from the
  // order of the noisy samples only -- as the reference does
  if ((int)return_weight_.size() != num_noisy) {
    return_weight_.resize(num_noisy);
    for (int i = 0; i < num_noisy; i++) trajectory_order[i] = i;
    std::partial_sort(trajectory_order.begin(), trajectory_order.begin() + num_noisy, trajectory_order.begin() + num_noisy,
                      [this](int a, int b) { return returns[a] < returns[b]; });
    const double f0 = std::log(0.5 * num_noisy + 1.0);
    double den = 0.0;
    for (int i = 0; i < num_noisy; i++) den += std::max(0.0, f0 - std::log(trajectory_order[i] + 1));
    for (int i = 0; i < num_noisy; i++)
      return_weight_[i] = std::max(0.0, f0 - std::log(trajectory_order[i] + 1)) / den - 1.0 / num_noisy;
  }
  std::fill(gradient.begin(), gradient.end(), 0.0);
  const size_t stride = (size_t)model->nu * kMaxTrajectoryHorizon;
  for (int i = 0; i < num_noisy; i++) {
    const double* noisei = noise.data() + (size_t)trajectory_order[i] * stride;
    for (int k = 0; k < num_parameters; k++) gradient[k] += noisei[k] * (return_weight_[i] / num_noisy);
  }
  if ((int)step_size_.size() != num_gradient) {
    step_size_.resize(num_gradient);
    LogScale(step_size_.data(), gradient_max_step_size, gradient_min_step_size, num_gradient);
  }
  const double gradient_filter = gradient_filter_;
  for (int i = num_noisy; i < num_trajectory; i++) {
    candidate_policy[i].CopyFrom(resampled_policy, num_spline_points);
    const double scaling = step_size_[i - num_noisy] / noise_exploration;
    for (int t = 0; t < (int)candidate_policy[i].plan.Size(); t++) {
      TimeSpline::Node n = candidate_policy[i].plan.NodeAt(t);
      for (int k = 0; k < model->nu; k++) {
        n.values()[k] += -scaling * gradient_filter * gradient[(size_t)t * model->nu + k];
        n.values()[k] += -scaling * (1.0 - gradient_filter) * gradient_previous[(size_t)t * model->nu + k];
      }
      Clamp(n.values().data(), model->actuator_ctrlrange, model->nu);
    }
  }
}

// the reference returns &trajectory[winner]; here the winner stays on the device until someone asks
const Trajectory* GpuSampleGradientPlanner::BestTrajectory() {
  if (!best_valid_) {
    if (num_rolled_ == 0 || winner < 0 || winner >= num_rolled_) return nullptr;
    ctx_->FetchTrajectory(winner, &best_);
    best_valid_ = true;
  }
  return &best_;
}

}  // namespace mjpc
