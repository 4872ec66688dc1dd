#include "trajectory.h"

#include <algorithm>

namespace mjpc {

void Trajectory::Initialize(int dim_state_, int dim_action_, int dim_residual_, int num_trace, int horizon_) {
  horizon = horizon_;
  dim_state = dim_state_;
  dim_action = dim_action_;
  dim_residual = dim_residual_;
  dim_trace = 3 * num_trace;
  failure = false;
}

void Trajectory::Allocate(int T) {
  states.resize((size_t)dim_state * T);
  actions.resize((size_t)dim_action * T);
  costs.resize(T);
  residual.resize((size_t)dim_residual * T);
  times.resize(T);
  trace.resize((size_t)dim_trace * T);
}

void Trajectory::Reset(int T, const double* initial_repeated_action) {
  std::fill_n(states.begin(), (size_t)dim_state * T, 0.0);
  for (int t = 0; t < T; t++)
    for (int k = 0; k < dim_action; k++)
      actions[(size_t)t * dim_action + k] = initial_repeated_action ? initial_repeated_action[k] : 0.0;
  std::fill_n(times.begin(), T, 0.0);
  std::fill_n(costs.begin(), T, 0.0);
  std::fill_n(residual.begin(), (size_t)dim_residual * T, 0.0);
  std::fill_n(trace.begin(), (size_t)dim_trace * T, 0.0);
  total_return = 0.0;
  failure = false;
}

void Trajectory::UpdateReturn(const Task* task) {
  total_return = 0;
  for (int t = 0; t < horizon; t++) {
    costs[t] = task->CostValue(residual.data() + (size_t)t * task->num_residual);
    total_return += costs[t];
  }
  total_return /= std::max(horizon, 1);
}

}  // namespace mjpc
