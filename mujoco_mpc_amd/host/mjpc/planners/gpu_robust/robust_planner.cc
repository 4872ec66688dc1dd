#include "robust_planner.h"

#include <algorithm>

#include "../../utilities.h"

namespace mjpc {

void GpuRobustPlanner::Initialize(mjModel* model, const Task& task) {
  delegate_->Initialize(model, task);
  model_ = model;
  task_ = &task;
  nrepetitions_ = GetNumberOrDefault(5, model, "robust_repetitions");
  // if robust_candidates is not defined, derive it from the number of rollouts in the sampling config
  ncandidates_ = GetNumberOrDefault(-1, model, "robust_candidates");
  if (ncandidates_ == -1) ncandidates_ = (int)GetNumberOrDefault(10, model, "sampling_trajectories") / std::max(nrepetitions_, 1);
  xfrc_std_ = GetNumberOrDefault(0.1, model, "robust_xfrc");
  xfrc_rate_ = GetNumberOrDefault(0.1, model, "robust_xfrc_rate");
}

void GpuRobustPlanner::Allocate() {
  delegate_->Allocate();
  state_.resize(model_->nq + model_->nv + model_->na);
  mocap_.resize(7 * (size_t)model_->nmocap);
  userdata_.resize(model_->nuserdata);
  ctx_ = std::make_unique<gpu::Context>(model_, *task_, device_, precision_);  // throws if no device kernel covers the model
}

void GpuRobustPlanner::Reset(int horizon, const double* initial_repeated_action) {
  delegate_->Reset(horizon, initial_repeated_action);
  std::fill(state_.begin(), state_.end(), 0.0);
  std::fill(mocap_.begin(), mocap_.end(), 0.0);
  std::fill(userdata_.begin(), userdata_.end(), 0.0);
  time_ = 0.0;
  best_candidate = -1;
}

void GpuRobustPlanner::SetState(const State& state) {
  delegate_->SetState(state);
  state.CopyTo(state_.data(), mocap_.data(), userdata_.data(), &time_);
}

void GpuRobustPlanner::OptimizePolicy(int horizon, ThreadPool& pool) {
  // the best N candidates of the delegate
  const int ncandidates = delegate_->OptimizePolicyCandidates(ncandidates_, horizon, pool);
  if (!ncandidates) return;
  if (ncandidates == 1) {  // a single candidate: nothing to compare
    best_candidate = 0;
    delegate_->CopyCandidateToPolicy(0);
    return;
  }
  // every candidate `repetitions` times under force perturbations: one launch of ncandidates x repetitions rollouts
  const int repetitions = std::max(nrepetitions_, 1);
  const int nu = model_->nu;
  const int N = ncandidates * repetitions;
  std::vector<double> times, values;
  int interpolation = 0;
  size_t np = 0;
  for (int i = 0; i < ncandidates; i++) {
    spline::TimeSpline plan(nu);
    delegate_->CandidatePlan(i, &plan);
    if (i == 0) {
      times = plan.times();
      interpolation = (int)plan.Interpolation();
      np = plan.values().size();
      values.resize((size_t)N * np);
    }
    for (int j = 0; j < repetitions; j++) std::copy(plan.values().begin(), plan.values().end(), values.begin() + ((size_t)repetitions * i + j) * np);
  }
  // sharded (the delegate's ranks): the N perturbed rollouts are split into contiguous ranges like the delegate's candidates -- the
  // force noise is keyed on the GLOBAL rollout index --, every rank writes its returns into a zeroed table and the sum over the
  // ranks is the table; all ranks then select the same candidate
  const int world = delegate_->world(), rank = delegate_->rank();
  if (world > N) throw gpu::Error(MJPCX_EINVAL, "more ranks than perturbed rollouts: every rank needs at least one");
  const int q = N / world, r = N % world;
  const int n_local = q + (rank < r ? 1 : 0), lo = rank * q + std::min(rank, r);
  ctx_->SyncTask(*task_);
  ctx_->Check(mjpcx_set_state(ctx_->handle(), state_.data(), time_, mocap_.data(), userdata_.data()));
  ctx_->Check(mjpcx_rollout_splines_noisy(ctx_->handle(), n_local, horizon, (int)times.size(), interpolation, times.data(),
                                          values.data() + (size_t)lo * np, xfrc_std_, xfrc_rate_, seed_,
                                          /*candidate_offset=*/(int)(iteration * (std::uint32_t)N) + lo));
  iteration++;
  std::vector<double> returns(N, 0.0);
  std::vector<int32_t> failure(N, 0);
  ctx_->Check(mjpcx_get_returns(ctx_->handle(), returns.data() + lo, failure.data() + lo));
  if (world > 1) {
    std::vector<double> table(2 * (size_t)N, 0.0);
    for (int i = lo; i < lo + n_local; i++) { table[i] = failure[i] ? 0.0 : returns[i]; table[N + i] = failure[i] ? 1.0 : 0.0; }  // (a failed rollout's return is not used)
    delegate_->AllReduceSum(table.data(), (int)table.size());
    for (int i = 0; i < N; i++) { returns[i] = table[i]; failure[i] = table[N + i] != 0.0; }
  }
  // the candidate with the best mean perturbed return (failed rollouts do not count), robust_planner.cc:141-163
  best_candidate = -1;
  double best_score = 0;
  perturbed_score.assign(ncandidates, 0.0);
  for (int candidate = 0; candidate < ncandidates; candidate++) {
    double mean_return = delegate_->CandidateScore(candidate);
    int valid_rollouts = 0;
    for (int j = 0; j < repetitions; j++) {
      if (failure[repetitions * candidate + j]) continue;
      const double total_return = returns[repetitions * candidate + j];
      mean_return = (valid_rollouts * mean_return + total_return) / (valid_rollouts + 1);
      valid_rollouts++;
    }
    perturbed_score[candidate] = mean_return;
    if (best_candidate == -1 || mean_return < best_score) {
      best_candidate = candidate;
      best_score = mean_return;
    }
  }
  delegate_->CopyCandidateToPolicy(best_candidate);
}

}  // namespace mjpc
