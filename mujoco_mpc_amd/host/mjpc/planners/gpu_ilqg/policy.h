// mjpc::iLQGPolicy (mjpc/planners/ilqg/policy.{h,cc}): nominal trajectory + time-varying linear feedback.
#pragma once
#include <vector>

#include "../../trajectory.h"
#include "../policy.h"

namespace mjpc {

enum PolicyRepresentation : int { kZeroOrder = 0, kLinear = 1, kCubic = 2 };

class iLQGPolicy : public Policy {
 public:
  void Allocate(const mjModel* model, const Task& task, int horizon) override;
  void Reset(int horizon, const double* initial_repeated_action = nullptr) override;
  // action = interp(actions)(time) + feedback_scaling * interp(K)(time) * StateDiff(interp(states)(time), state), clamped
  void Action(double* action, const double* state, double time) const override;
  void CopyFrom(const iLQGPolicy& policy, int horizon);

  const mjModel* model = nullptr;
  Trajectory trajectory;
  std::vector<double> feedback_gain;       // (T-1) x nu x ndx
  std::vector<double> action_improvement;  // (T-1) x nu
  mutable std::vector<double> feedback_gain_scratch, state_scratch, action_scratch, state_interp;
  double feedback_scaling = 1.0;
  int representation = kLinear;
};

}  // namespace mjpc
