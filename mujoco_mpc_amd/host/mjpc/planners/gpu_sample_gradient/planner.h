// mjpc::GpuSampleGradientPlanner -- SampleGradientPlanner (mjpc/planners/sample_gradient/planner.{h,cc}) with the rollout
// fan-out on the device.
//
// Per plan step: the nominal spline resampled at the current time (candidate 0), `num_noisy - 1` candidates = nominal +
// noise_exploration * N(0, 1) per parameter, and `num_gradient_` candidates along the (fitness-shaped, filtered) gradient
// estimate of the PREVIOUS step -- all rolled out by ONE mjpcx_rollout_splines call; winner selection, the natural-
// evolution-strategy gradient from the noise of the sorted samples (Wierstra et al.) and the log-spaced gradient steps stay
// on the host exactly as planner.cc:166-290, 388-480, including its caching of the shaping weights.
// Differences on purpose: kMaxTrajectory does not bound the candidate count; the standard normals come from a seedable
// counter-based generator (HostGaussianPair = the device's gaussian_pair) instead of a function-local absl::BitGen.
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

class GpuSampleGradientPlanner : public Planner {
 public:
  explicit GpuSampleGradientPlanner(int device = 0, int precision = 64, std::uint64_t seed = 0)
      : device_(device), precision_(precision), seed_(seed) {}
  ~GpuSampleGradientPlanner() override = default;

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
  void Plots(mjvFigure* fig_planner, mjvFigure* fig_timer, int planner_shift, int timer_shift, int planning, int* shift) override {}
  int NumParameters() override { return policy.num_spline_points * model->nu; }

  void ResamplePolicy(SamplingPolicy& policy, int horizon, int num_spline_points);
  void AddNoiseToPolicy(int i);
  void Rollouts(int num_trajectory, int num_gradient, int horizon);
  void GradientCandidates(int num_trajectory, int num_gradient, int horizon);
  gpu::Context* context() { return ctx_.get(); }

  // ----- members (names as in the reference) ----- //
  mjModel* model = nullptr;
  const Task* task = nullptr;
  std::vector<double> state, mocap, userdata;
  double time = 0;
  SamplingPolicy policy;  // guarded by mtx_
  std::vector<SamplingPolicy> candidate_policy;
  SamplingPolicy resampled_policy, previous_policy;
  spline::TimeSpline plan_scratch;
  std::vector<double> returns;   // trajectory[i].total_return
  std::vector<int32_t> failure;  // trajectory[i].failure
  std::vector<int> trajectory_order;
  double noise_exploration = 0;
  std::vector<double> noise;     // [candidate][parameter], stride = nu * kMaxTrajectoryHorizon as in the reference
  spline::SplineInterpolation interpolation_ = spline::kZeroSpline;
  double improvement = 0;
  int winner = 0;
  std::atomic<double> noise_compute_time{0};
  double rollouts_compute_time = 0, gradient_candidates_compute_time = 0, policy_update_compute_time = 0;
  int num_trajectory_ = 0;
  int num_gradient_ = 0;
  mutable std::shared_mutex mtx_;
  std::vector<double> gradient, gradient_previous;
  double gradient_filter_ = 1.0;
  std::vector<double> step_size_;
  double gradient_max_step_size = 2.0, gradient_min_step_size = 1.0e-3;
  std::vector<double> return_weight_;
  const int idx_nominal = 0;
  enum WinnerType : int { kNominal = 0, kPerturb, kGradient };
  int winner_type_ = 0;
  std::uint32_t iteration = 0;

 private:
  void EnsureCandidates(int n);
  int device_, precision_;
  std::uint64_t seed_;
  std::unique_ptr<gpu::Context> ctx_;
  Trajectory best_;
  bool best_valid_ = false;
  int num_rolled_ = 0;
};

}  // namespace mjpc
