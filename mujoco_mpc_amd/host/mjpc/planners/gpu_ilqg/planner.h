// mjpc::GpuILQGPlanner -- iLQG with every data-parallel piece on an MI355X.
//
// Drop-in for mjpc::iLQGPlanner (mjpc/planners/ilqg/planner.{h,cc}). Behind include/mjpcx.h:
//   FeedbackRollouts / ActionRollouts  ilqg/planner.cc:630-724   -> mjpcx_rollout_feedback (all steps, one launch)
//   ModelDerivatives::Compute          model_derivatives.cc:45-106 -> mjpcx_transition_fd
//   CostDerivatives::Compute           cost_derivatives.cc:112-230 -> mjpcx_cost_derivatives
//   iLQGBackwardPass::Riccati          ilqg/backward_pass.cc:253-324 -> mjpcx_backward_pass (MFMA f64)
// On the host, as in the reference: the regularisation schedule (backward_pass.cc:327-356), BestRollout
// (planner.cc:727-740), derivative skip + interpolation (model_derivatives.cc:108-165), policy bookkeeping.
#pragma once
#include <memory>
#include <shared_mutex>
#include <vector>

#include "../../gpu/context.h"
#include "../planner.h"
#include "policy.h"
#include "settings.h"

namespace mjpc {

class GpuILQGPlanner : public Planner {
 public:
  explicit GpuILQGPlanner(int device = 0, int precision = 64) : device_(device), precision_(precision) {}
  ~GpuILQGPlanner() override = default;

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
  int NumParameters() override { return dim_action * kMaxTrajectoryHorizon; }

  void Iteration(int horizon, ThreadPool& pool);
  // index of the best non-failed rollout, scanning from the last index with strict < (planner.cc:727-740); -1 if none
  static int BestRollout(const std::vector<double>& total_return, const std::vector<std::int32_t>& failure);
  // iLQGBackwardPass::ScaleRegularization / UpdateRegularization (backward_pass.cc:327-356)
  void ScaleRegularization(double factor, double reg_min, double reg_max);
  void UpdateRegularization(double reg_min, double reg_max, double z, double s);
  gpu::Context* context() { return ctx_.get(); }

  // ----- members (names as in the reference) ----- //
  mjModel* model = nullptr;
  const Task* task = nullptr;
  std::vector<double> state, mocap, userdata;
  double time = 0;
  iLQGPolicy policy;            // guarded by mtx_
  iLQGPolicy previous_policy;
  iLQGPolicy candidate_policy0;  // candidate_policy[0]: the nominal for this iteration
  int dim_state = 0, dim_state_derivative = 0, dim_action = 0, dim_sensor = 0;
  iLQGSettings settings;
  std::vector<double> linesearch_steps;
  // iLQGBackwardPass state
  double regularization = 1.0, regularization_rate = 1.0, regularization_factor = 2.0;
  double dV[2] = {0, 0};
  double action_step = 0, feedback_scaling = 0, improvement = 0, expected = 0, surprise = 0;
  int winner = 0, num_trajectory_ = 0, num_rollouts_gui_ = 0, derivative_skip_ = 0;
  double nominal_compute_time = 0, model_derivative_compute_time = 0, cost_derivative_compute_time = 0,
         backward_pass_compute_time = 0, rollouts_compute_time = 0, policy_update_compute_time = 0;
  mutable std::shared_mutex mtx_;

 private:
  void LineSearchSteps();
  void ModelDerivatives(const Trajectory& tr, int T);
  void TakeTrajectory(iLQGPolicy* p, int index);
  int device_, precision_;
  std::unique_ptr<gpu::Context> ctx_;
  std::vector<double> A_, B_, C_, D_, cx_, cu_, cxx_, cxu_, cuu_, Vx_, Vxx_, K_, du_;
  std::vector<double> eA_, eB_, eC_, eD_, etimes_, estates_, eactions_;
  std::vector<double> returns_;
  std::vector<std::int32_t> failure_;
  iLQGPolicy winner_policy_;
};

}  // namespace mjpc
