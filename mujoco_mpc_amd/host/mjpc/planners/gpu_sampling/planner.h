// mjpc::GpuSamplingPlanner -- Predictive Sampling with the candidate fan-out on an MI355X.
//
// Drop-in for mjpc::SamplingPlanner (mjpc/planners/sampling/planner.h): same RankedPlanner interface,
// same public data members where they still make sense, same locking contract (ActionFromPolicy only
// takes mtx_ and never waits for the GPU). What moved to the device, behind include/mjpcx.h:
//   Rollouts()            sampling/planner.cc:355-393  -> mjpcx_rollout_noise (noise + N rollouts + returns)
//   std::partial_sort     sampling/planner.cc:184-188  -> mjpcx_best / mjpcx_topk
// What changed on purpose: the candidate count is not capped at kMaxTrajectory = 128 and
// the noise is a seedable counter-based generator instead of a function-local absl::BitGen (F4).
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <shared_mutex>
#include <vector>

#include "../../gpu/context.h"
#include "../planner.h"
#include "../sampling/policy.h"

namespace mjpc {

class GpuSamplingPlanner : public RankedPlanner {
 public:
  explicit GpuSamplingPlanner(int device = 0, int precision = 64, std::uint64_t seed = 0)
      : device_(device), precision_(precision), seed_(seed) {}
  ~GpuSamplingPlanner() override = default;

  void Initialize(mjModel* model, const Task& task) override;
  void Allocate() override;
  void Reset(int horizon, const double* initial_repeated_action = nullptr) override;
  void SetState(const State& state) override;
  void OptimizePolicy(int horizon, ThreadPool& pool) override;
  void NominalTrajectory(int horizon, ThreadPool& pool) override;
  void ActionFromPolicy(double* action, const double* state, double time, bool use_previous = false) override;
  const Trajectory* BestTrajectory() override;
  void Traces(mjvScene* scn) override {}
  void GUI(mjUI& ui) override {}
  void Plots(mjvFigure* fig_planner, mjvFigure* fig_timer, int planner_shift, int timer_shift, int planning,
             int* shift) override {}
  int NumParameters() override { return policy.num_spline_points * model->nu; }

  int OptimizePolicyCandidates(int ncandidates, int horizon, ThreadPool& pool) override;
  double CandidateScore(int candidate) const override;
  void ActionFromCandidatePolicy(double* action, int candidate, const double* state, double time) override;
  void CopyCandidateToPolicy(int candidate) override;

  // the spline of ranked candidate `candidate` (what ActionFromCandidatePolicy evaluates): for planners that roll the
  // candidates out again on the device (GpuRobustPlanner)
  void CandidatePlan(int candidate, spline::TimeSpline* out);
  void UpdateNominalPolicy(int horizon);
  void Rollouts(int num_trajectory, int horizon);

  // Multi-GPU: one process per GPU, rank r rolls out global candidates [r*n, (r+1)*n) of num_trajectory_
  // (noise is keyed on the GLOBAL index, so results do not depend on the rank count). After the local
  // selection the planner calls `exchange` once per plan step with record = {best return, global index,
  // nominal return} and the local winner's spline (np values); the callee all-gathers the records, picks
  // the global winner (lowest return, ties by index), overwrites record and spline with the winner's
  // (broadcast from its owner) and returns 0. The transport is the caller's: torch.distributed over
  // RCCL/xGMI in bench.py, gloo in the CPU tests.
  using ExchangeFn = int (*)(void* user, double record[3], double* spline, int np);
  void SetSharding(int rank, int world, ExchangeFn exchange, void* user);
  // The ranked interface (OptimizePolicyCandidates, RobustPlanner's delegate) sharded: every rank ranks its share, the k best of all
  // ranks are merged (in: this rank's k best as (global index, return), unused slots -1; out: the global k best, ties by global
  // index, identical on every rank) and their splines are summed into place from their owners. Callbacks as for the Cross-Entropy
  // planner (gloo in the CPU-side tests); without them the library's own communicator (mjpcx_merge_topk, mjpcx_elite_allreduce).
  using MergeTopkFn = int (*)(void* user, int k, std::int64_t* index, double* total_return);
  using SumFn = int (*)(void* user, double* values, int n);
  void SetRankedSharding(MergeTopkFn merge, SumFn sum, void* user);
  // in-place sum of a small vector over the ranks (identity on one rank): for planners built on this one (GpuRobustPlanner)
  void AllReduceSum(double* values, int n);
  int rank() const { return rank_; }
  int world() const { return world_; }
  gpu::Context* context() { return ctx_.get(); }

  // ----- members (names as in the reference) ----- //
  mjModel* model = nullptr;
  const Task* task = nullptr;
  std::vector<double> state, mocap, userdata;
  double time = 0;
  SamplingPolicy policy;  // guarded by mtx_
  SamplingPolicy previous_policy;
  SamplingPolicy winner_policy;  // candidate_policy[winner]
  spline::TimeSpline plan_scratch;
  std::vector<int> trajectory_order;
  double noise_exploration[2] = {0, 0};
  spline::SplineInterpolation interpolation_ = spline::kZeroSpline;
  int winner = 0;
  double improvement = 0;
  std::atomic<double> noise_compute_time{0};
  double rollouts_compute_time = 0, policy_update_compute_time = 0;
  std::uint8_t sliding_plan_ = false;
  int num_trajectory_ = 0;
  std::uint32_t iteration = 0;
  mutable std::shared_mutex mtx_;

 private:
  double PlanningTimestep() const;  // agent_timestep if the model defines it (agent.cc:288-291)
  void SetWinner(int index, const std::vector<double>& values);
  void LoadCandidatePlan(int index, SamplingPolicy* out);
  void LoadRankedPlan(int candidate, SamplingPolicy* out);
  int device_, precision_;
  std::uint64_t seed_;
  int rank_ = 0, world_ = 1, offset_ = 0;
  ExchangeFn exchange_ = nullptr;
  void* exchange_user_ = nullptr;
  MergeTopkFn merge_ = nullptr;
  SumFn sum_ = nullptr;
  void* ranked_user_ = nullptr;
  void RequireCommunicator();
  std::vector<double> candidate_values_;  // sharded ranked interface: the merged candidates' splines, k x (P nu); else empty
  std::unique_ptr<gpu::Context> ctx_;
  std::vector<double> scores_;
  Trajectory best_;
  bool best_valid_ = false;
  int num_rolled_ = 0;
};

}  // namespace mjpc
