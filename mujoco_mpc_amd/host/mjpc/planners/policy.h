// mjpc::Policy (mjpc/planners/policy.h)
#pragma once
#include <mujoco/mujoco.h>
#include "../task.h"

namespace mjpc {
class Policy {
 public:
  virtual ~Policy() = default;
  virtual void Allocate(const mjModel* model, const Task& task, int horizon) = 0;
  virtual void Reset(int horizon, const double* initial_repeated_action = nullptr) = 0;
  virtual void Action(double* action, const double* state, double time) const = 0;
};
}  // namespace mjpc
