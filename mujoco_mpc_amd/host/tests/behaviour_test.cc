// GPU: the reference's behavioural planner suites with their own settings, statement for statement, on the device planners:
//   mjpc/test/sampling_planner/sampling_planner_test.cc:44-115   SamplingPlannerTest.RandomSearch
//   mjpc/test/planners/robust/robust_planner_test.cc:46-133      RobustPlannerTest.RandomSearch
//   mjpc/test/ilqg_planner/ilqg_test.cc:48-120                   iLQGTest.Particle
// The planner classes are the GPU ones (GpuSamplingPlanner for SamplingPlanner, ...); everything else -- the model
// (particle_task.xml), keyframes, exploration noise 0.01, 1000 / 25 iterations, horizon 2.5 s at 0.1 s, the default number
// of trajectories, the tolerances -- is the reference's. mjcb_sensor is not installed: the residual runs inside the kernels.
// argv[1] = directory with Particle.mjpx
#include <cmath>
#include <memory>
#include <string>

#include "check.h"
#include "mjpc/planners/gpu_ilqg/planner.h"
#include "mjpc/planners/gpu_robust/robust_planner.h"
#include "mjpc/planners/gpu_sampling/planner.h"
#include "mjpc/tasks/tasks.h"
#include "model_io.h"
using namespace mjpc;

namespace {
void ExpectReachedGoalWithinLimits(const mjModel* model, const Trajectory* best, const State& state, int steps, double pos_tol) {
  const int final_state_index = (steps - 1) * (model->nq + model->nv);
  CHECK((int)best->states.size() >= final_state_index);
  CHECK_NEAR(best->states[final_state_index], state.mocap()[0], pos_tol);
  CHECK_NEAR(best->states[final_state_index + 1], state.mocap()[1], pos_tol);
  CHECK_NEAR(best->states[final_state_index + 2], 0.0, 1.0e-1);
  CHECK_NEAR(best->states[final_state_index + 3], 0.0, 1.0e-1);
  for (int t = 0; t < steps - 1; t++)
    for (int i = 0; i < model->nu; i++) {
      CHECK(best->actions[t * model->nu + i] <= model->actuator_ctrlrange[2 * i + 1]);
      CHECK(best->actions[t * model->nu + i] >= model->actuator_ctrlrange[2 * i]);
    }
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  const int iterations = 1000;
  const double horizon = 2.5, timestep = 0.1;
  const int steps = (int)mju_max(mju_min(horizon / timestep + 1, kMaxTrajectoryHorizon), 1);
  CHECK(steps == 26);

  {  // ---------------- SamplingPlannerTest.RandomSearch
    auto storage = ModelStorage::Load(dir + "/Particle.mjpx");
    mjModel* model = storage->model();
    ParticleTestTask task;
    task.Reset(model);
    mjData* data = mj_makeData(model);
    State state;
    state.Initialize(model);
    state.Allocate(model);
    state.Reset();
    state.Set(model, data);
    CHECK_NEAR(state.mocap()[0], 0.25, 1e-15);  // particle.xml: the goal body

    GpuSamplingPlanner planner;
    planner.Initialize(model, task);
    planner.Allocate();
    planner.Reset(kMaxTrajectoryHorizon);
    planner.noise_exploration[0] = 0.01;
    CHECK(planner.num_trajectory_ == 10);  // the reference's default ("sampling_trajectories" absent)
    model->opt.timestep = timestep;
    ThreadPool pool(1);
    planner.SetState(state);
    for (int i = 0; i < iterations; i++) planner.OptimizePolicy(steps, pool);
    ExpectReachedGoalWithinLimits(model, planner.BestTrajectory(), state, steps, 1.0e-1);
    mj_deleteData(data);
  }

  {  // ---------------- RobustPlannerTest.RandomSearch
    auto storage = ModelStorage::Load(dir + "/Particle.mjpx");
    mjModel* model = storage->model();
    ParticleTestTask task;
    task.Reset(model);
    mjData* data = mj_makeData(model);
    const int home_id = NameToId(model, mjOBJ_KEY, "ctrl_test");
    CHECK(home_id >= 0);
    mj_resetDataKeyframe(model, data, home_id);
    State state;
    state.Initialize(model);
    state.Allocate(model);
    state.Reset();
    state.Set(model, data);

    GpuRobustPlanner planner(std::make_unique<GpuSamplingPlanner>());
    planner.Initialize(model, task);
    planner.Allocate();
    planner.Reset(kMaxTrajectoryHorizon, data->ctrl);
    double res[2];
    planner.ActionFromPolicy(res, state.state().data(), 2);
    CHECK_NEAR(res[0], 0.1, 1.0e-4);  // the keyframe's ctrl
    CHECK_NEAR(res[1], 0.2, 1.0e-4);
    model->opt.timestep = timestep;
    ThreadPool pool(1);
    planner.SetState(state);
    for (int i = 0; i < iterations; i++) planner.OptimizePolicy(steps, pool);
    ExpectReachedGoalWithinLimits(model, planner.BestTrajectory(), state, steps, 1.0e-1);
    mj_deleteData(data);
  }

  {  // ---------------- iLQGTest.Particle
    auto storage = ModelStorage::Load(dir + "/Particle.mjpx");
    mjModel* model = storage->model();
    ParticleTestTask task;
    task.Reset(model);
    mjData* data = mj_makeData(model);
    State state;
    state.Initialize(model);
    state.Allocate(model);
    state.Reset();
    state.Set(model, data);
    GpuILQGPlanner planner;
    planner.Initialize(model, task);
    planner.Allocate();
    planner.Reset(kMaxTrajectoryHorizon);
    model->opt.timestep = timestep;
    ThreadPool pool(1);
    planner.SetState(state);
    for (int i = 0; i < 25; i++) planner.OptimizePolicy(steps, pool);
    ExpectReachedGoalWithinLimits(model, &planner.candidate_policy0.trajectory, state, steps, 1.0e-2);
    mj_deleteData(data);
  }
  TEST_MAIN_END();
}
