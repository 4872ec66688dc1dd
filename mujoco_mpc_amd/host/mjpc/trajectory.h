// mjpc::Trajectory with the reference's public members and member functions (mjpc/trajectory.h:31-91).
//
// The planners of this build never call Rollout: they fan candidates out on the device in one launch (GpuSamplingPlanner,
// include/mjpcx.h) and fill Trajectory buffers from there. Rollout / RolloutDiscrete / NoisyRollout keep the reference's
// signatures for code written against mjpc::Trajectory: the policy is an arbitrary std::function, so it is evaluated on the
// host, step by step, in the reference's order of operations (trajectory.cc:100-210), and every mj_step runs on the device (a
// one-candidate, two-step launch per step -- there is no CPU physics in this build). That is a compatibility path, three
// orders of magnitude slower per step than the batched launch; see trajectory.cc for the two documented deviations.
#pragma once
#include <functional>
#include <vector>

#include <mujoco/mujoco.h>
#include "task.h"

namespace mjpc {

inline constexpr int kMaxTrajectoryHorizon = 512;

class Trajectory {
 public:
  void Initialize(int dim_state, int dim_action, int dim_residual, int num_trace, int horizon);
  void Allocate(int T);
  void Reset(int T, const double* initial_repeated_action = nullptr);
  // simulate model forward in time with continuous-time indexed policy (trajectory.h:43-49)
  void Rollout(std::function<void(double* action, const double* state, double time)> policy, const Task* task, const mjModel* model,
               mjData* data, const double* state, double time, const double* mocap, const double* userdata, int steps);
  // (trajectory.h:51-57) xfrc_std > 0 is served by the device's own noise stream (mjpcx_rollout_splines_noisy through
  // GpuRobustPlanner), not by this host-policy path: it aborts with a message, as the reference's mju_error would
  void NoisyRollout(std::function<void(double* action, const double* state, double time)> policy, const Task* task, const mjModel* model,
                    mjData* data, const double* state, double time, const double* mocap, const double* userdata, double xfrc_std,
                    double xfrc_rate, int steps);
  // simulate model forward in time with discrete-time indexed policy (trajectory.h:59-65)
  void RolloutDiscrete(std::function<void(double* action, const double* state, int index)> policy, const Task* task, const mjModel* model,
                       mjData* data, const double* state, double time, const double* mocap, const double* userdata, int steps);
  // total_return and costs from the stored residuals (same arithmetic as the device: mean of CostValue)
  void UpdateReturn(const Task* task);

  int horizon = 0, dim_state = 0, dim_action = 0, dim_residual = 0, dim_trace = 0;
  std::vector<double> states, actions, times, residual, costs, trace;
  double total_return = 0;
  bool failure = false;
};

// The host-policy rollouts (Rollout / RolloutDiscrete) keep one device context per (model, task) they were asked for; this frees them
// (they are re-created on the next use). Call it before unloading the library or the HIP runtime; a model re-loaded at the same
// address is detected on its own.
void ReleaseRolloutContexts();

}  // namespace mjpc
