// mjpc::GpuCrossEntropyPlanner -- Cross-Entropy Method with the candidate fan-out, the sort and the elite
// statistics on an MI355X.
//
// Drop-in for mjpc::CrossEntropyPlanner (mjpc/planners/cross_entropy/planner.{h,cc}): same Planner
// interface, member names and locking contract. On the device, behind include/mjpcx.h:
//   AddNoiseToPolicy + Rollouts   cross_entropy/planner.cc:351-443 -> mjpcx_rollout_noise (CROSS_ENTROPY mode;
//                                 the extra nominal rollout of :435 rides along as global candidate N)
//   std::sort of the returns      :206-211                          -> mjpcx_topk (only the elites matter)
//   elite mean / variance loops   :216-270                          -> mjpcx_elite_moments
// Changed on purpose: no kMaxTrajectory cap, seedable counter-based noise (F4).
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

class GpuCrossEntropyPlanner : public Planner {
 public:
  explicit GpuCrossEntropyPlanner(int device = 0, int precision = 64, std::uint64_t seed = 0)
      : device_(device), precision_(precision), seed_(seed) {}
  ~GpuCrossEntropyPlanner() override = default;

  void Initialize(mjModel* model, const Task& task) override;
  void Allocate() override;
  void Reset(int horizon, const double* initial_repeated_action = nullptr) override;
  void SetState(const State& state) override;
  void OptimizePolicy(int horizon, ThreadPool& pool) override;
  void NominalTrajectory(int horizon, ThreadPool& pool) override;
  void ActionFromPolicy(double* action, const double* state, double time, bool use_previous = false) override;
  const Trajectory* BestTrajectory() override;  // the NOMINAL trajectory, cross_entropy/planner.cc:446-448
  void Traces(mjvScene* scn) override {}
  void GUI(mjUI& ui) override {}
  void Plots(mjvFigure* fig_planner, mjvFigure* fig_timer, int planner_shift, int timer_shift, int planning,
             int* shift) override {}
  int NumParameters() override { return policy.num_spline_points * model->nu; }

  void ResamplePolicy(int horizon);

  // Multi-GPU candidate sharding (one process per GPU). Two exchanges per plan step, both tiny:
  //   merge_topk: in/out (index[k], return[k]) -- all-gather the ranks' local top-k, keep the global k best
  //               (ascending return, ties by global index), identical on every rank afterwards;
  //   sum:        in/out array -- all-reduce(sum) of the elite partial moments.
  // The transport is the caller's (torch.distributed over RCCL in bench scripts, gloo in the CPU tests).
  using MergeTopkFn = int (*)(void* user, int k, std::int64_t* index, double* total_return);
  using SumFn = int (*)(void* user, double* values, int n);
  void SetSharding(int rank, int world, MergeTopkFn merge, SumFn sum, void* user);
  gpu::Context* context() { return ctx_.get(); }

  // ----- members (names as in the reference) ----- //
  mjModel* model = nullptr;
  const Task* task = nullptr;
  std::vector<double> state, mocap, userdata;
  double time = 0;
  SamplingPolicy policy;            // guarded by mtx_
  SamplingPolicy resampled_policy;
  SamplingPolicy previous_policy;
  std::vector<double> parameters_scratch, times_scratch, variance;
  std::vector<int> trajectory_order;  // global indices of the elites, best first
  double std_initial_ = 0.1, std_min_ = 0.01, explore_fraction_ = 0;
  int n_elite_ = 2;
  spline::SplineInterpolation interpolation_ = spline::kZeroSpline;  // CE never reads sampling_representation
  double improvement = 0;
  std::atomic<double> noise_compute_time{0};
  double rollouts_compute_time = 0, policy_update_compute_time = 0;
  int num_trajectory_ = 0;
  std::uint32_t iteration = 0;
  mutable std::shared_mutex mtx_;

 private:
  double PlanningTimestep() const;
  int device_, precision_;
  std::uint64_t seed_;
  int rank_ = 0, world_ = 1, offset_ = 0, n_local_ = 0;
  MergeTopkFn merge_ = nullptr;
  SumFn sum_ = nullptr;
  void* user_ = nullptr;
  std::unique_ptr<gpu::Context> ctx_;
  Trajectory nominal_;
  bool nominal_valid_ = false;
};

}  // namespace mjpc
