#include "planner.h"

#include <algorithm>
#include <chrono>
#include <mutex>

#include "../../utilities.h"

namespace mjpc {

using spline::SplineInterpolation;
using spline::TimeSpline;

void GpuSamplingPlanner::Initialize(mjModel* m, const Task& t) {
  model = m;
  task = &t;
  noise_exploration[0] = GetNumberOrDefault(0.1, m, "sampling_exploration");
  noise_exploration[1] = GetCustomNumericSize(m, "sampling_exploration") > 1
                             ? GetCustomNumericData(m, "sampling_exploration")[1] : 0.0;
  num_trajectory_ = GetNumberOrDefault(10, m, "sampling_trajectories");
  interpolation_ = (SplineInterpolation)GetNumberOrDefault((int)spline::kCubicSpline, m, "sampling_representation");
  sliding_plan_ = GetNumberOrDefault(0, m, "sampling_sliding_plan");
  winner = 0;
}

void GpuSamplingPlanner::Allocate() {
  state.resize(model->nq + model->nv + model->na);
  mocap.resize(7 * (size_t)model->nmocap);
  userdata.resize(model->nuserdata);
  policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  previous_policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  winner_policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  plan_scratch = TimeSpline(model->nu);
  best_.Initialize((int)state.size(), model->nu, task->num_residual, task->num_trace, kMaxTrajectoryHorizon);
  best_.Allocate(kMaxTrajectoryHorizon);
  ctx_ = std::make_unique<gpu::Context>(model, *task, device_, precision_);  // throws if no device kernel covers the model
}

void GpuSamplingPlanner::Reset(int horizon, const double* initial_repeated_action) {
  std::fill(state.begin(), state.end(), 0.0);
  std::fill(mocap.begin(), mocap.end(), 0.0);
  std::fill(userdata.begin(), userdata.end(), 0.0);
  time = 0.0;
  {
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    policy.Reset(horizon, initial_repeated_action);
    previous_policy.Reset(horizon, initial_repeated_action);
  }
  winner_policy.Reset(horizon, initial_repeated_action);
  plan_scratch.Clear();
  best_.Reset(kMaxTrajectoryHorizon);
  best_valid_ = false;
  improvement = 0.0;
  winner = 0;
}

double GpuSamplingPlanner::PlanningTimestep() const {
  return GetNumberOrDefault(model->opt.timestep, model, "agent_timestep");
}

void GpuSamplingPlanner::SetState(const State& s) { s.CopyTo(state.data(), mocap.data(), userdata.data(), &time); }

// resample the winner at P nodes starting at the current time, or slide the plan (sampling/planner.cc:240-323)
void GpuSamplingPlanner::UpdateNominalPolicy(int horizon) {
  const int P = winner_policy.num_spline_points;
  double nominal_time = time;
  const double time_horizon = (horizon - 1) * PlanningTimestep();
  if (sliding_plan_) {
    const int extra = interpolation_ == spline::kZeroSpline ? 1 : (interpolation_ == spline::kLinearSpline ? 2 : 4);
    const double shift = P > extra ? mju_max(time_horizon / (P - extra), 1.0e-5) : time_horizon;
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    TimeSpline& plan = policy.plan;
    if (plan.Size() && plan.NodeAt(0).time() > nominal_time) {  // simulation time was reset
      plan.ShiftTime(nominal_time);
      previous_policy.plan.ShiftTime(nominal_time);
    }
    plan.DiscardBefore(nominal_time);
    if (plan.Size() == 0) plan.AddNode(nominal_time);
    while ((int)plan.Size() < P) {
      const auto last = plan.NodeAt((int)plan.Size() - 1);
      std::vector<double> v(last.values().begin(), last.values().end());
      plan.AddNode(last.time() + shift, v);
    }
    return;
  }
  const double shift = interpolation_ == spline::kZeroSpline ? mju_max(time_horizon / P, 1.0e-5)
                                                             : mju_max(time_horizon / (P - 1), 1.0e-5);
  plan_scratch.Clear();
  plan_scratch.SetInterpolation(interpolation_);
  plan_scratch.Reserve(P);
  for (int k = 0; k < P; k++) {
    TimeSpline::Node node = plan_scratch.AddNode(nominal_time);
    winner_policy.Action(node.values().data(), /*state=*/nullptr, nominal_time);
    nominal_time += shift;
  }
  const std::unique_lock<std::shared_mutex> lock(mtx_);
  policy.plan = plan_scratch;
}

void GpuSamplingPlanner::SetSharding(int rank, int world, ExchangeFn exchange, void* user) {
  rank_ = rank;
  world_ = std::max(world, 1);
  exchange_ = exchange;
  exchange_user_ = user;
}

// the device fan-out: noise + N rollouts + returns, one launch (sampling/planner.cc:355-393)
void GpuSamplingPlanner::Rollouts(int num_trajectory, int horizon) {
  // this rank's slice of the global batch: contiguous ranges, the first (num_trajectory % world) ranks take one more
  if (world_ > num_trajectory) throw gpu::Error(MJPCX_EINVAL, "more ranks than candidates: every rank needs at least one rollout");
  const int q = num_trajectory / world_, r = num_trajectory % world_;
  const int n_local = q + (rank_ < r ? 1 : 0);
  offset_ = rank_ * q + std::min(rank_, r);
  mjpcx_noise_spec ns{};
  ns.seed = seed_;
  ns.iteration = iteration;
  ns.mode = MJPCX_NOISE_SAMPLING;
  ns.candidate_offset = offset_;
  ns.nominal_candidate = 0;  // `if (i != 0) AddNoiseToPolicy`: candidate 0 is the nominal
  ns.std0 = noise_exploration[0];
  ns.std1 = noise_exploration[1];
  const TimeSpline& plan = policy.plan;
  ctx_->SyncTask(*task);  // the per-plan frozen ResidualFn copy (agent.cc:319)
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state.data(), time, mocap.data(), userdata.data()));
  ctx_->Check(mjpcx_rollout_noise(ctx_->handle(), n_local, horizon, (int)plan.Size(), (int)plan.Interpolation(),
                                  plan.times().data(), plan.values().data(), &ns));
  num_rolled_ = n_local;
  best_valid_ = false;
  candidate_values_.clear();
}

void GpuSamplingPlanner::SetRankedSharding(MergeTopkFn merge, SumFn sum, void* user) {
  merge_ = merge;
  sum_ = sum;
  ranked_user_ = user;
}

// without a communicator of this world size the library's collectives are the identity: every rank would keep its own numbers silently
void GpuSamplingPlanner::RequireCommunicator() {
  int comm_rank = 0, comm_world = 1;
  ctx_->Check(mjpcx_comm_info(ctx_->handle(), &comm_rank, &comm_world));
  if (comm_world != world_) throw gpu::Error(MJPCX_ESTATE, "sharded planner has neither an exchange callback nor a communicator of its world size (mjpcx_comm_init)");
}

void GpuSamplingPlanner::AllReduceSum(double* values, int n) {
  if (world_ <= 1 || n <= 0) return;
  if (sum_) {
    if (sum_(ranked_user_, values, n) != 0) throw gpu::Error(MJPCX_EDEVICE, "sum over the ranks failed");
    return;
  }
  RequireCommunicator();
  ctx_->Check(mjpcx_elite_allreduce(ctx_->handle(), values, n));
}

int GpuSamplingPlanner::OptimizePolicyCandidates(int ncandidates, int horizon, ThreadPool& pool) {
  UpdateNominalPolicy(horizon);
  const int num_trajectory = num_trajectory_;
  ncandidates = std::min(ncandidates, num_trajectory);
  const auto start = std::chrono::steady_clock::now();
  policy.plan.SetInterpolation(interpolation_);
  Rollouts(num_trajectory, horizon);
  // this rank's best (at most its own share), then -- sharded -- the best of all ranks
  const int k_local = std::min(ncandidates, num_rolled_);
  std::vector<int32_t> idx(std::max(k_local, 1));
  std::vector<double> ret(std::max(k_local, 1), 0.0);
  if (k_local > 0) ctx_->Check(mjpcx_topk(ctx_->handle(), k_local, idx.data(), ret.data()));
  trajectory_order.assign(ncandidates, -1);
  scores_.assign(ncandidates, 0.0);
  if (world_ <= 1) {
    for (int i = 0; i < ncandidates; i++) { trajectory_order[i] = idx[i]; scores_[i] = ret[i]; }
  } else {
    std::vector<std::int64_t> gidx(ncandidates, -1);
    std::vector<double> gret(ncandidates, 1.0e300);
    for (int i = 0; i < k_local; i++) { gidx[i] = (std::int64_t)offset_ + idx[i]; gret[i] = ret[i]; }
    if (merge_) {
      if (merge_(ranked_user_, ncandidates, gidx.data(), gret.data()) != 0) throw gpu::Error(MJPCX_EDEVICE, "top-k exchange failed");
    } else {
      RequireCommunicator();
      ctx_->Check(mjpcx_merge_topk(ctx_->handle(), ncandidates, gidx.data(), gret.data()));
    }
    // the merged candidates' splines: every owner writes its own into a zeroed table, the sum over the ranks is the table
    const size_t np = policy.plan.Size() * (size_t)model->nu;
    candidate_values_.assign((size_t)ncandidates * np, 0.0);
    for (int i = 0; i < ncandidates; i++) {
      if (gidx[i] < 0) throw gpu::Error(MJPCX_ESTATE, "top-k exchange returned fewer candidates than the ranks hold");
      trajectory_order[i] = (int)gidx[i];
      scores_[i] = gret[i];
      const std::int64_t local = gidx[i] - offset_;
      if (local >= 0 && local < num_rolled_) ctx_->Check(mjpcx_fetch_spline(ctx_->handle(), (int)local, candidate_values_.data() + (size_t)i * np));
    }
    AllReduceSum(candidate_values_.data(), (int)candidate_values_.size());
  }
  rollouts_compute_time = GetDuration(start);
  iteration++;
  return ncandidates;
}

void GpuSamplingPlanner::OptimizePolicy(int horizon, ThreadPool& pool) {
  UpdateNominalPolicy(horizon);
  const auto start = std::chrono::steady_clock::now();
  policy.plan.SetInterpolation(interpolation_);
  Rollouts(num_trajectory_, horizon);
  // argmin + winner spline + trajectory[0].total_return in one launch and one sync
  int32_t index = 0;
  double best_return = 0, nominal_return = 0;
  std::vector<double> values(policy.plan.Size() * (size_t)model->nu);
  ctx_->Check(mjpcx_best(ctx_->handle(), /*ref_candidate=*/offset_ == 0 ? 0 : -1, &index, &best_return, &nominal_return,
                         values.data()));
  index += offset_;  // global candidate index
  if (world_ > 1) {
    if (exchange_) {  // a transport lent by the caller (gloo in the CPU-side tests)
      double record[3] = {best_return, (double)index, nominal_return};
      if (exchange_(exchange_user_, record, values.data(), (int)values.size()) != 0)
        throw gpu::Error(MJPCX_EDEVICE, "candidate exchange failed");
      best_return = record[0];
      index = (int32_t)record[1];
      nominal_return = record[2];
    } else {          // RCCL inside the library (mjpcx_comm_init on this planner's context)
      RequireCommunicator();
      ctx_->Check(mjpcx_exchange_best(ctx_->handle(), &index, &best_return, &nominal_return, values.data(), (int)values.size()));
    }
  }
  trajectory_order.assign(1, index);
  scores_.assign(1, best_return);
  rollouts_compute_time = GetDuration(start);
  iteration++;
  const auto update_start = std::chrono::steady_clock::now();
  SetWinner(index, values);
  improvement = mju_max(nominal_return - best_return, 0.0);
  policy_update_compute_time = GetDuration(update_start);
}

void GpuSamplingPlanner::SetWinner(int index, const std::vector<double>& values) {
  winner = index;
  const int nu = model->nu;
  TimeSpline plan(nu, policy.plan.Interpolation());
  for (size_t k = 0; k < policy.plan.Size(); k++)
    plan.AddNode(policy.plan.times()[k], spline::Span<const double>(values.data() + k * nu, nu));
  winner_policy.plan = plan;
  winner_policy.num_spline_points = policy.num_spline_points;
  best_valid_ = false;
  const std::unique_lock<std::shared_mutex> lock(mtx_);
  previous_policy = policy;
  policy = winner_policy;
}

void GpuSamplingPlanner::LoadCandidatePlan(int index, SamplingPolicy* out) {
  const int nu = model->nu;
  std::vector<double> values(policy.plan.Size() * (size_t)nu);
  ctx_->Check(mjpcx_fetch_spline(ctx_->handle(), index, values.data()));
  out->model = model;
  out->num_spline_points = policy.num_spline_points;
  out->plan = TimeSpline(nu, policy.plan.Interpolation());
  for (size_t k = 0; k < policy.plan.Size(); k++)
    out->plan.AddNode(policy.plan.times()[k], spline::Span<const double>(values.data() + k * nu, nu));
}

// ranked candidate -> its spline: from this process's device buffers, or (sharded) from the table OptimizePolicyCandidates summed
void GpuSamplingPlanner::LoadRankedPlan(int candidate, SamplingPolicy* out) {
  if (candidate_values_.empty()) return LoadCandidatePlan(trajectory_order[candidate], out);
  const int nu = model->nu;
  const size_t np = policy.plan.Size() * (size_t)nu;
  out->model = model;
  out->num_spline_points = policy.num_spline_points;
  out->plan = TimeSpline(nu, policy.plan.Interpolation());
  for (size_t k = 0; k < policy.plan.Size(); k++)
    out->plan.AddNode(policy.plan.times()[k], spline::Span<const double>(candidate_values_.data() + (size_t)candidate * np + k * nu, nu));
}

void GpuSamplingPlanner::NominalTrajectory(int horizon, ThreadPool& pool) {
  const TimeSpline& plan = winner_policy.plan.Size() ? winner_policy.plan : policy.plan;
  std::vector<double> times(plan.times()), values(plan.values());
  if (times.empty()) { times.assign(1, time); values.assign(model->nu, 0.0); }
  ctx_->SyncTask(*task);
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state.data(), time, mocap.data(), userdata.data()));
  ctx_->Check(mjpcx_rollout_splines(ctx_->handle(), 1, horizon, (int)times.size(), (int)plan.Interpolation(),
                                    times.data(), values.data()));
  num_rolled_ = 1;
  winner = 0;
  offset_ = 0;
  ctx_->FetchTrajectory(0, &best_);
  best_valid_ = true;
}

void GpuSamplingPlanner::ActionFromPolicy(double* action, const double* s, double t, bool use_previous) {
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  (use_previous ? previous_policy : policy).Action(action, s, t);
}

// the reference returns &trajectory[winner]; here the winner stays on the device until someone asks
const Trajectory* GpuSamplingPlanner::BestTrajectory() {
  if (!best_valid_) {
    const int local = winner - offset_;  // only the owner rank holds the winner's buffers
    if (num_rolled_ == 0 || local < 0 || local >= num_rolled_) return nullptr;
    ctx_->FetchTrajectory(local, &best_);
    best_valid_ = true;
  }
  return &best_;
}

double GpuSamplingPlanner::CandidateScore(int candidate) const { return scores_[candidate]; }

void GpuSamplingPlanner::ActionFromCandidatePolicy(double* action, int candidate, const double* s, double t) {
  SamplingPolicy p;
  LoadRankedPlan(candidate, &p);
  p.Action(action, s, t);
}

void GpuSamplingPlanner::CandidatePlan(int candidate, spline::TimeSpline* out) {
  SamplingPolicy p;
  LoadRankedPlan(candidate, &p);
  *out = p.plan;
}

void GpuSamplingPlanner::CopyCandidateToPolicy(int candidate) {
  SamplingPolicy p;
  LoadRankedPlan(candidate, &p);
  SetWinner(trajectory_order[candidate], p.plan.values());
}

}  // namespace mjpc
