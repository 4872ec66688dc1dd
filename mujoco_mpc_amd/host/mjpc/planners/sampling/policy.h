// mjpc::SamplingPolicy (mjpc/planners/sampling/policy.h): a TimeSpline of controls, clamped to ctrlrange.
#pragma once
#include "../../spline/spline.h"
#include "../policy.h"

namespace mjpc {
class SamplingPolicy : public Policy {
 public:
  void Allocate(const mjModel* model, const Task& task, int horizon) override;
  void Reset(int horizon, const double* initial_repeated_action = nullptr) override;
  void Action(double* action, const double* state, double time) const override;
  void CopyFrom(const SamplingPolicy& policy, int horizon);
  void SetPlan(const spline::TimeSpline& plan);

  const mjModel* model = nullptr;
  spline::TimeSpline plan;
  int num_spline_points = 0;
};
}  // namespace mjpc
