// GPU: port of mjpc/test/sampling_planner/sampling_planner_test.cc (particle reaches the mocap goal, actions in
// limits) plus the RankedPlanner quartet, use_previous, and the closed-loop harness (task_test.cc:101-109).
// argv[1] = directory with Particle.mjpx / Cartpole.mjpx
#include <cmath>
#include <string>

#include "check.h"
#include "mjpc/planners/gpu_sampling/planner.h"
#include "mjpc/tasks/tasks.h"
#include "mjpc/testspeed.h"
#include "model_io.h"
using namespace mjpc;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  {
    auto storage = ModelStorage::Load(dir + "/Particle.mjpx");
    mjModel* model = storage->model();
    ParticleTestTask task;
    task.Reset(model);
    GpuSamplingPlanner planner(0, 64, /*seed=*/1);
    planner.Initialize(model, task);
    CHECK(planner.noise_exploration[0] == 0.01 && planner.policy.num_spline_points == 0);
    planner.num_trajectory_ = 256;  // beyond the reference's kMaxTrajectory
    planner.noise_exploration[0] = 0.1;
    planner.Allocate();
    CHECK(planner.policy.num_spline_points == 11 && planner.NumParameters() == 22);
    const int horizon = 11;  // agent_horizon 1 / agent_timestep 0.1 + 1
    planner.Reset(horizon);
    State state;
    state.Allocate(model);
    double qpos[2] = {0, 0}, qvel[2] = {0, 0}, mpos[3] = {0.25, 0, 0.01}, mquat[4] = {1, 0, 0, 0};
    state.Set(model, qpos, qvel, nullptr, mpos, mquat, nullptr, 0.0);
    planner.SetState(state);
    ThreadPool pool(1);
    CHECK(planner.BestTrajectory() == nullptr);
    double a_before[2] = {9, 9}, a_prev[2];
    for (int it = 0; it < 60; it++) {
      if (it == 59) planner.ActionFromPolicy(a_before, nullptr, 0.3);
      planner.OptimizePolicy(horizon, pool);
    }
    planner.ActionFromPolicy(a_prev, nullptr, 0.3, /*use_previous=*/true);
    CHECK(a_prev[0] == a_before[0] && a_prev[1] == a_before[1]);  // agent_test.cc: use_previous is the pre-update policy
    const Trajectory* best = planner.BestTrajectory();
    CHECK(best != nullptr && best->horizon == horizon && !best->failure);
    const double* last = best->states.data() + (size_t)(horizon - 1) * 4;
    CHECK_NEAR(last[0], 0.25, 0.1);  // sampling_planner_test.cc:91-98
    CHECK_NEAR(last[1], 0.0, 0.1);
    for (int t = 0; t < horizon; t++)
      for (int k = 0; k < 2; k++) CHECK(std::fabs(best->actions[t * 2 + k]) <= 1.0);
    CHECK(planner.improvement >= 0 && planner.winner == planner.trajectory_order[0]);
    // total_return is the mean of costs, costs = CostValue(residual) (trajectory.cc:312-326)
    Trajectory copy = *best;
    copy.UpdateReturn(&task);
    CHECK_NEAR(copy.total_return, best->total_return, 1e-12);
    // RankedPlanner
    const int n = planner.OptimizePolicyCandidates(4, horizon, pool);
    CHECK(n == 4);
    for (int i = 1; i < n; i++) CHECK(planner.CandidateScore(i - 1) <= planner.CandidateScore(i));
    double a0[2], a1[2];
    planner.ActionFromCandidatePolicy(a0, 1, nullptr, 0.2);
    planner.CopyCandidateToPolicy(1);
    planner.ActionFromPolicy(a1, nullptr, 0.2);
    CHECK(a0[0] == a1[0] && a0[1] == a1[1] && planner.winner == planner.trajectory_order[1]);
    planner.NominalTrajectory(horizon, pool);
    CHECK(planner.BestTrajectory()->horizon == horizon);
  }
  {  // a model/task pair with no device kernel must fail loudly, not fall back
    auto storage = ModelStorage::Load(dir + "/Cartpole.mjpx");
    ParticleTestTask wrong;  // particle residual id with the cartpole topology
    wrong.Reset(storage->model());
    GpuSamplingPlanner planner;
    planner.Initialize(storage->model(), wrong);
    bool threw = false;
    try { planner.Allocate(); } catch (const gpu::Error& e) { threw = e.code == MJPCX_EUNSUPPORTED; }
    CHECK(threw);
  }
  {  // StepAllTasksTest: closed loop produces a positive cost for every registered task
    TestSpeedOptions opt;
    opt.model_dir = dir;
    opt.verbose = false;
    opt.num_candidates = 128;
    for (auto& t : GetTasks()) {
      const double cost = SynchronousPlanningCost(t->Name(), 1, 100, 0.1, opt);
      CHECK(cost > 0);
    }
    CHECK(SynchronousPlanningCost("NoSuchTask", 1, 100, 0.1, opt) == -1);
  }
  TEST_MAIN_END();
}
