// mjpc::Trajectory buffers with the reference's public members (mjpc/trajectory.h:74-86). The
// reference's Rollout/NoisyRollout/RolloutDiscrete member functions drive mj_step on the CPU; in this
// build candidate rollouts run on the GPU behind the C ABI and a Trajectory is filled from there
// (GpuSamplingPlanner), so those members are intentionally absent: there is no CPU physics path.
#pragma once
#include <vector>

#include "task.h"

namespace mjpc {

inline constexpr int kMaxTrajectoryHorizon = 512;

class Trajectory {
 public:
  void Initialize(int dim_state, int dim_action, int dim_residual, int num_trace, int horizon);
  void Allocate(int T);
  void Reset(int T, const double* initial_repeated_action = nullptr);
  // total_return and costs from the stored residuals (same arithmetic as the device: mean of CostValue)
  void UpdateReturn(const Task* task);

  int horizon = 0, dim_state = 0, dim_action = 0, dim_residual = 0, dim_trace = 0;
  std::vector<double> states, actions, times, residual, costs, trace;
  double total_return = 0;
  bool failure = false;
};

}  // namespace mjpc
