#include "policy.h"

#include <stdexcept>

#include "../../trajectory.h"
#include "../../utilities.h"

namespace mjpc {

void SamplingPolicy::Allocate(const mjModel* m, const Task& task, int horizon) {
  model = m;
  num_spline_points = GetNumberOrDefault(kMaxTrajectoryHorizon, m, "sampling_spline_points");
  plan = spline::TimeSpline(m->nu);
  plan.Reserve(num_spline_points);
}

void SamplingPolicy::Reset(int horizon, const double* initial_repeated_action) {
  plan.Clear();
  if (initial_repeated_action) plan.AddNode(0, spline::Span<const double>(initial_repeated_action, model->nu));
}

void SamplingPolicy::Action(double* action, const double* state, double time) const {
  if (!action) throw std::invalid_argument("SamplingPolicy::Action: null action");
  plan.Sample(time, spline::Span<double>(action, model->nu));
  Clamp(action, model->actuator_ctrlrange, model->nu);
}

void SamplingPolicy::CopyFrom(const SamplingPolicy& policy, int horizon) {
  plan = policy.plan;
  num_spline_points = policy.num_spline_points;
}

void SamplingPolicy::SetPlan(const spline::TimeSpline& p) { plan = p; }

}  // namespace mjpc
