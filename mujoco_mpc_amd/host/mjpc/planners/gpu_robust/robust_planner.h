// mjpc::GpuRobustPlanner -- RobustPlanner (mjpc/planners/robust/robust_planner.{h,cc}) with both fan-outs on the device.
//
// The delegate (a GpuSamplingPlanner) ranks its candidates (OptimizePolicyCandidates -> mjpcx_rollout_noise + mjpcx_topk);
// the best `ncandidates_` are then rolled out `nrepetitions_` times each under Ornstein-Uhlenbeck force perturbations
// (Trajectory::NoisyRollout -> ONE mjpcx_rollout_splines_noisy launch of ncandidates x repetitions wavefronts) and the
// candidate with the best mean perturbed return becomes the policy, exactly as robust_planner.cc:90-170 (including its
// running-mean quirk: the first valid perturbed rollout replaces the delegate's unperturbed score).
// Differences on purpose: the perturbed rollouts use their own device context (the delegate's trajectory buffers must
// survive for BestTrajectory()); the noise is a seedable counter-based stream instead of an unseeded absl::BitGen.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "../../gpu/context.h"
#include "../gpu_sampling/planner.h"
#include "../planner.h"

namespace mjpc {

class GpuRobustPlanner : public Planner {
 public:
  explicit GpuRobustPlanner(std::unique_ptr<GpuSamplingPlanner> delegate, int device = 0, int precision = 64, std::uint64_t seed = 0)
      : delegate_(std::move(delegate)), device_(device), precision_(precision), seed_(seed) {}
  ~GpuRobustPlanner() override = default;

  void Initialize(mjModel* model, const Task& task) override;
  void Allocate() override;
  void Reset(int horizon, const double* initial_repeated_action = nullptr) override;
  void SetState(const State& state) override;
  void OptimizePolicy(int horizon, ThreadPool& pool) override;
  void NominalTrajectory(int horizon, ThreadPool& pool) override { delegate_->NominalTrajectory(horizon, pool); }
  void ActionFromPolicy(double* action, const double* state, double time, bool use_previous = false) override {
    delegate_->ActionFromPolicy(action, state, time, use_previous);
  }
  const Trajectory* BestTrajectory() override { return delegate_->BestTrajectory(); }
  void Traces(mjvScene* scn) override { delegate_->Traces(scn); }
  void GUI(mjUI& ui) override { delegate_->GUI(ui); }
  void Plots(mjvFigure* fig_planner, mjvFigure* fig_timer, int planner_shift, int timer_shift, int planning, int* shift) override {
    delegate_->Plots(fig_planner, fig_timer, planner_shift, timer_shift, planning, shift);
  }
  int NumParameters() override { return delegate_->NumParameters(); }

  GpuSamplingPlanner* delegate() { return delegate_.get(); }
  gpu::Context* context() { return ctx_.get(); }

  // ----- members (names as in the reference) ----- //
  int ncandidates_ = 0;     // "robust_candidates" (default: sampling_trajectories / robust_repetitions)
  int nrepetitions_ = 0;    // "robust_repetitions" (5)
  double xfrc_std_ = 0;     // "robust_xfrc" (0.1)
  double xfrc_rate_ = 0;    // "robust_xfrc_rate" (0.1)
  int best_candidate = -1;  // of the last OptimizePolicy
  std::vector<double> perturbed_score;  // per candidate: the mean the selection used
  std::uint32_t iteration = 0;

 private:
  std::unique_ptr<GpuSamplingPlanner> delegate_;
  int device_, precision_;
  std::uint64_t seed_;
  mjModel* model_ = nullptr;
  const Task* task_ = nullptr;
  std::vector<double> state_, mocap_, userdata_;
  double time_ = 0;
  std::unique_ptr<gpu::Context> ctx_;
};

}  // namespace mjpc
