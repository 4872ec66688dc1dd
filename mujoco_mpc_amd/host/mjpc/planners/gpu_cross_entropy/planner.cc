#include "planner.h"

#include <algorithm>
#include <chrono>
#include <mutex>

#include "../../utilities.h"

namespace mjpc {

using spline::TimeSpline;

// cross_entropy/planner.cc:41-74
void GpuCrossEntropyPlanner::Initialize(mjModel* m, const Task& t) {
  model = m;
  task = &t;
  std_initial_ = GetNumberOrDefault(0.1, m, "sampling_exploration");
  std_min_ = GetNumberOrDefault(0.01, m, "std_min");
  explore_fraction_ = GetNumberOrDefault(0.0, m, "explore_fraction");
  num_trajectory_ = GetNumberOrDefault(10, m, "sampling_trajectories");
  n_elite_ = GetNumberOrDefault(std::max(num_trajectory_ / 10, 2), m, "n_elite");
}

// cross_entropy/planner.cc:77-119
void GpuCrossEntropyPlanner::Allocate() {
  state.resize(model->nq + model->nv + model->na);
  mocap.resize(7 * (size_t)model->nmocap);
  userdata.resize(model->nuserdata);
  policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  resampled_policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  previous_policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  parameters_scratch.assign((size_t)model->nu * kMaxTrajectoryHorizon, 0.0);
  times_scratch.assign(kMaxTrajectoryHorizon, 0.0);
  variance.assign((size_t)model->nu * kMaxTrajectoryHorizon, 0.0);
  nominal_.Initialize((int)state.size(), model->nu, task->num_residual, task->num_trace, kMaxTrajectoryHorizon);
  nominal_.Allocate(kMaxTrajectoryHorizon);
  ctx_ = std::make_unique<gpu::Context>(model, *task, device_, precision_);
}

// cross_entropy/planner.cc:122-160
void GpuCrossEntropyPlanner::Reset(int horizon, const double* initial_repeated_action) {
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
  std::fill(variance.begin(), variance.end(), std_initial_ * std_initial_);
  nominal_.Reset(kMaxTrajectoryHorizon);
  nominal_valid_ = false;
  improvement = 0.0;
}

double GpuCrossEntropyPlanner::PlanningTimestep() const {
  return GetNumberOrDefault(model->opt.timestep, model, "agent_timestep");
}

void GpuCrossEntropyPlanner::SetState(const State& s) { s.CopyTo(state.data(), mocap.data(), userdata.data(), &time); }

void GpuCrossEntropyPlanner::SetSharding(int rank, int world, MergeTopkFn merge, SumFn sum, void* user) {
  rank_ = rank;
  world_ = std::max(world, 1);
  merge_ = merge;
  sum_ = sum;
  user_ = user;
}

// cross_entropy/planner.cc:322-348
void GpuCrossEntropyPlanner::ResamplePolicy(int horizon) {
  const int P = resampled_policy.num_spline_points, nu = model->nu;
  double nominal_time = time;
  const double time_shift = mju_max((horizon - 1) * PlanningTimestep() / (P - 1), 1.0e-5);
  for (int t = 0; t < P; t++) {
    times_scratch[t] = nominal_time;
    resampled_policy.Action(parameters_scratch.data() + (size_t)t * nu, nullptr, nominal_time);
    nominal_time += time_shift;
  }
  const spline::SplineInterpolation interp = policy.plan.Interpolation();
  resampled_policy.plan.Clear();
  for (int t = 0; t < P; t++)
    resampled_policy.plan.AddNode(times_scratch[t], spline::Span<const double>(parameters_scratch.data() + (size_t)t * nu, nu));
  resampled_policy.plan.SetInterpolation(interp);
}

// cross_entropy/planner.cc:168-291
void GpuCrossEntropyPlanner::OptimizePolicy(int horizon, ThreadPool& pool) {
  resampled_policy.plan.SetInterpolation(interpolation_);
  const int num_trajectory = num_trajectory_;
  n_elite_ = std::min(n_elite_, num_trajectory);
  const int n_elite = n_elite_;
  {
    const std::shared_lock<std::shared_mutex> lock(mtx_);
    resampled_policy.CopyFrom(policy, policy.num_spline_points);
  }
  ResamplePolicy(horizon);
  const auto start = std::chrono::steady_clock::now();
  const int P = resampled_policy.num_spline_points, nu = model->nu, np = P * nu;

  // ---- Rollouts (:388-443): N noised candidates + the nominal as global candidate N (on the last rank)
  if (world_ > num_trajectory) throw gpu::Error(MJPCX_EINVAL, "more ranks than candidates: every rank needs at least one rollout");
  {
    const int q = num_trajectory / world_, r = num_trajectory % world_;  // contiguous ranges, the first (N % world) ranks take one more
    n_local_ = q + (rank_ < r ? 1 : 0);
    offset_ = rank_ * q + std::min(rank_, r);
    if (rank_ == world_ - 1) n_local_ += 1;  // the last rank also rolls out the nominal (global candidate N)
  }
  int explore_count = 0;  // candidates i < N * explore_fraction use std_initial instead of the fitted variance (:367-371)
  for (int i = 0; i < num_trajectory; i++) explore_count += i < num_trajectory * explore_fraction_;
  mjpcx_noise_spec ns{};
  ns.seed = seed_;
  ns.iteration = iteration;
  ns.mode = MJPCX_NOISE_CROSS_ENTROPY;
  ns.candidate_offset = offset_;
  ns.nominal_candidate = num_trajectory;
  ns.explore_count = explore_count;
  ns.std0 = std_initial_;
  ns.std1 = std_min_;
  ns.param_variance = variance.data();
  const TimeSpline& plan = resampled_policy.plan;
  ctx_->SyncTask(*task);  // the per-plan frozen ResidualFn copy (agent.cc:319)
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state.data(), time, mocap.data(), userdata.data()));
  ctx_->Check(mjpcx_rollout_noise(ctx_->handle(), n_local_, horizon, (int)plan.Size(), (int)plan.Interpolation(),
                                  plan.times().data(), plan.values().data(), &ns));

  // ---- the reference sorts all N returns (:206-211); only the n_elite best (+1 for the nominal) matter
  int k = std::min(n_elite + 1, n_local_);
  std::vector<std::int32_t> idx32(k);
  std::vector<double> ret(k);
  ctx_->Check(mjpcx_topk(ctx_->handle(), k, idx32.data(), ret.data()));
  std::vector<std::int64_t> idx(n_elite + 1, -1);
  ret.resize(n_elite + 1, 1.0e300);
  for (int i = 0; i < k; i++) idx[i] = (std::int64_t)idx32[i] + offset_;
  if (world_ > 1) {
    if (merge_) { if (merge_(user_, n_elite + 1, idx.data(), ret.data()) != 0) throw gpu::Error(MJPCX_EDEVICE, "top-k exchange failed"); }
    else {  // RCCL inside the library: it must hold a communicator of this world size, or the merge is silently the identity
      int comm_rank = 0, comm_world = 1;
      ctx_->Check(mjpcx_comm_info(ctx_->handle(), &comm_rank, &comm_world));
      if (comm_world != world_) throw gpu::Error(MJPCX_ESTATE, "sharded planner has neither exchange callbacks nor a communicator of its world size (mjpcx_comm_init)");
      ctx_->Check(mjpcx_merge_topk(ctx_->handle(), n_elite + 1, idx.data(), ret.data()));
    }
  }
  trajectory_order.clear();
  double best_return = 0;
  bool have_best = false;
  for (int i = 0; i < n_elite + 1 && (int)trajectory_order.size() < n_elite; i++) {
    if (idx[i] < 0 || idx[i] == num_trajectory) continue;  // the nominal rollout is not a candidate
    if (!have_best) { best_return = ret[i]; have_best = true; }
    trajectory_order.push_back((int)idx[i]);
  }
  rollouts_compute_time = GetDuration(start);

  // ---- elite mean / variance (:216-270): partial sums over this rank's elites, all-reduced when sharded
  const auto update_start = std::chrono::steady_clock::now();
  std::vector<std::int32_t> mine;
  for (int g : trajectory_order)
    if (g >= offset_ && g < offset_ + n_local_) mine.push_back(g - offset_);
  std::vector<double> sums(np + 1, 0.0), mean(np), sq(np, 0.0);
  ctx_->Check(mjpcx_elite_moments(ctx_->handle(), (int)mine.size(), mine.data(), nullptr, sums.data(), &sums[np]));
  auto all_sum = [&](double* v, int n) {
    if (world_ <= 1) return;
    if (sum_) { if (sum_(user_, v, n) != 0) throw gpu::Error(MJPCX_EDEVICE, "moment exchange failed"); }
    else ctx_->Check(mjpcx_elite_allreduce(ctx_->handle(), v, n));
  };
  all_sum(sums.data(), np + 1);
  for (int j = 0; j < np; j++) mean[j] = sums[j] / n_elite;
  const double avg_return = sums[np] / n_elite;
  double unused = 0;
  ctx_->Check(mjpcx_elite_moments(ctx_->handle(), (int)mine.size(), mine.data(), mean.data(), sq.data(), &unused));
  all_sum(sq.data(), np);
  std::fill(variance.begin(), variance.end(), 0.0);
  for (int j = 0; j < np; j++) variance[j] = sq[j] / (n_elite - 1);  // n_elite == 1: inf/nan, as the reference (:267)
  std::copy(mean.begin(), mean.end(), parameters_scratch.begin());
  {
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    previous_policy = policy;
    policy.plan.Clear();
    policy.plan.SetInterpolation(interpolation_);
    for (int t = 0; t < P; t++)
      policy.plan.AddNode(times_scratch[t], spline::Span<const double>(mean.data() + (size_t)t * nu, nu));
  }
  improvement = mju_max(avg_return - best_return, 0.0);
  nominal_valid_ = false;
  iteration++;
  policy_update_compute_time = GetDuration(update_start);
}

// cross_entropy/planner.cc:294-308
void GpuCrossEntropyPlanner::NominalTrajectory(int horizon, ThreadPool& pool) {
  const TimeSpline& plan = resampled_policy.plan;
  std::vector<double> times(plan.times()), values(plan.values());
  if (times.empty()) { times.assign(1, time); values.assign(model->nu, 0.0); }
  ctx_->SyncTask(*task);
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state.data(), time, mocap.data(), userdata.data()));
  ctx_->Check(mjpcx_rollout_splines(ctx_->handle(), 1, horizon, (int)times.size(), (int)plan.Interpolation(),
                                    times.data(), values.data()));
  ctx_->FetchTrajectory(0, &nominal_);
  nominal_valid_ = true;
  n_local_ = 0;
}

void GpuCrossEntropyPlanner::ActionFromPolicy(double* action, const double* s, double t, bool use_previous) {
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  (use_previous ? previous_policy : policy).Action(action, s, t);
}

const Trajectory* GpuCrossEntropyPlanner::BestTrajectory() {
  if (!nominal_valid_) {
    const int local = num_trajectory_ - offset_;  // the nominal rode along as global candidate N on the last rank
    if (n_local_ <= 0 || local < 0 || local >= n_local_) return nullptr;
    ctx_->FetchTrajectory(local, &nominal_);
    nominal_valid_ = true;
  }
  return &nominal_;
}

}  // namespace mjpc
