// Headless closed-loop harness with the interface of mjpc/testspeed.h: simulate + plan synchronously and
// return the average cost. Both the planning rollouts AND the "physics thread" step run on the GPU (this
// build has no CPU physics): the simulation step is a 1-candidate, 2-step rollout at the model's own timestep.
#pragma once
#include <string>

namespace mjpc {

struct TestSpeedOptions {
  std::string model_dir;         // directory holding <Task>.mjpx blobs
  int num_candidates = 0;        // 0: the task's sampling_trajectories
  int device = 0;
  bool verbose = true;
};

// returns the average stage cost over the run, or -1 on error (mjpc/testspeed.cc:44-126)
double SynchronousPlanningCost(std::string task_name, int planner_thread_count, int steps_per_planning_iteration,
                               double total_time, const TestSpeedOptions& options);

}  // namespace mjpc
