// GPU: port of mjpc/test/agent/rollout_test.cc:67-153 -- Trajectory::Rollout with a PD feedback policy (an arbitrary
// std::function, evaluated on the host; every mj_step on the device) on the particle task whose residual copies the
// state. Plus RolloutDiscrete against Rollout, NoisyRollout with zero noise, and Planner::data_ / ResizeMjData.
// argv[1] = directory with ParticleCopy.mjpx
#include <cmath>
#include <string>

#include "check.h"
#include "mjpc/planners/gpu_sampling/planner.h"
#include "mjpc/tasks/tasks.h"
#include "mjpc/trajectory.h"
#include "model_io.h"
using namespace mjpc;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  auto storage = ModelStorage::Load(dir + "/ParticleCopy.mjpx");
  mjModel* model = storage->model();
  ParticleCopyTestTask task;
  task.Reset(model);
  mjData* data = mj_makeData(model);
  CHECK(model->nq + model->nv == task.num_residual);
  const int nx = model->nq + model->nv;

  const double position_goal[2] = {0.1, 0.1}, velocity_goal[2] = {0.0, 0.0};
  auto feedback_policy = [&](double* action, const double* state, double time) {
    const double P = 10.0, D = 2.5;
    for (int k = 0; k < 2; k++) action[k] = -P * (state[k] - position_goal[k]) - D * (state[2 + k] - velocity_goal[k]);
  };
  Trajectory trajectory;
  const int horizon = 100;
  trajectory.Initialize(nx, model->nu, task.num_residual, 1, horizon);
  trajectory.Allocate(horizon);
  const double state[4] = {0, 0, 0, 0};
  double mocap[7];
  mju_copy(mocap, data->mocap_pos, 3);
  mju_copy(mocap + 3, data->mocap_quat, 4);
  mocap[3] = 1;  // (mj_makeData of the stand-in leaves the quaternion at zero)
  trajectory.Rollout(feedback_policy, &task, model, data, state, 0.0, mocap, nullptr, horizon);
  CHECK(!trajectory.failure && trajectory.horizon == horizon);
  // rollout_test.cc:137-139: the PD controller reaches the goal
  const double* last = trajectory.states.data() + (size_t)(horizon - 1) * nx;
  CHECK_NEAR(std::fabs(last[0] - 0.1) + std::fabs(last[1] - 0.1), 0.0, 0.1);
  CHECK_NEAR(std::fabs(last[2]) + std::fabs(last[3]), 0.0, 0.1);
  // rollout_test.cc:141-145: residual[t] pairs with states[t]
  double l1 = 0;
  for (int i = 0; i < horizon * nx; i++) l1 += std::fabs(trajectory.states[i] - trajectory.residual[i]);
  CHECK_NEAR(l1, 0.0, 1e-5);
  // bookkeeping of trajectory.cc:100-210: times advance by the planning timestep, the last action repeats, the return is the
  // mean of CostValue over the rows, mjData ends at the final state
  CHECK_NEAR(trajectory.times[horizon - 1] - trajectory.times[0], (horizon - 1) * (trajectory.times[1] - trajectory.times[0]), 1e-9);
  for (int k = 0; k < model->nu; k++)
    CHECK(trajectory.actions[(size_t)(horizon - 1) * model->nu + k] == trajectory.actions[(size_t)(horizon - 2) * model->nu + k]);
  Trajectory copy = trajectory;
  copy.UpdateReturn(&task);
  CHECK_NEAR(copy.total_return, trajectory.total_return, 1e-12);
  CHECK_NEAR(data->qpos[0], last[0], 0.0);
  // the first control is clamped to actuator_ctrlrange (10 * 0.1 = 1 sits on the limit; a larger gain must clip)
  for (int t = 0; t < horizon; t++) for (int k = 0; k < model->nu; k++) CHECK(std::fabs(trajectory.actions[(size_t)t * model->nu + k]) <= 1.0 + 1e-12);

  // RolloutDiscrete with the same control sequence replays the same trajectory
  Trajectory replay;
  replay.Initialize(nx, model->nu, task.num_residual, 1, horizon);
  replay.Allocate(horizon);
  auto index_policy = [&](double* action, const double* /*state*/, int index) {
    mju_copy(action, trajectory.actions.data() + (size_t)index * model->nu, model->nu);
  };
  replay.RolloutDiscrete(index_policy, &task, model, data, state, 0.0, mocap, nullptr, horizon);
  for (int i = 0; i < horizon * nx; i++) CHECK_NEAR(replay.states[i], trajectory.states[i], 1e-12);
  CHECK_NEAR(replay.total_return, trajectory.total_return, 1e-12);
  // NoisyRollout without noise is Rollout
  Trajectory quiet;
  quiet.Initialize(nx, model->nu, task.num_residual, 1, horizon);
  quiet.Allocate(horizon);
  quiet.NoisyRollout(feedback_policy, &task, model, data, state, 0.0, mocap, nullptr, 0.0, 1.0, horizon);
  CHECK_NEAR(quiet.total_return, trajectory.total_return, 1e-12);
  // a one-row trajectory is a single mj_forward
  Trajectory single;
  single.Initialize(nx, model->nu, task.num_residual, 1, 1);
  single.Allocate(1);
  const double s1[4] = {0.05, -0.02, 0.3, 0.1};
  single.Rollout(feedback_policy, &task, model, data, s1, 0.5, mocap, nullptr, 1);
  for (int i = 0; i < nx; i++) CHECK_NEAR(single.residual[i], s1[i], 1e-12);

  // the host-policy rollouts keep a device context per (model, task) address pair: a different model at the same address (here the
  // same struct with a heavier particle, as after mj_deleteModel + a reload) must not be served by the stale context, and the
  // contexts can be released and come back
  {
    const double mass = model->body_mass[model->nbody - 1];
    model->body_mass[model->nbody - 1] = 3.0 * mass;
    Trajectory heavy;
    heavy.Initialize(nx, model->nu, task.num_residual, 1, horizon);
    heavy.Allocate(horizon);
    heavy.Rollout(feedback_policy, &task, model, data, state, 0.0, mocap, nullptr, horizon);
    CHECK(!heavy.failure && std::fabs(heavy.total_return - trajectory.total_return) > 1e-6);
    model->body_mass[model->nbody - 1] = mass;
    ReleaseRolloutContexts();
    Trajectory again;
    again.Initialize(nx, model->nu, task.num_residual, 1, horizon);
    again.Allocate(horizon);
    again.Rollout(feedback_policy, &task, model, data, state, 0.0, mocap, nullptr, horizon);
    CHECK_NEAR(again.total_return, trajectory.total_return, 1e-12);
  }

  // Planner::data_ / ResizeMjData (planners/planner.cc:23-33)
  GpuSamplingPlanner planner(0, 64, 1);
  planner.ResizeMjData(model, 3);
  CHECK(planner.data_.size() == 3 && planner.data_[2]->qpos != nullptr);
  planner.ResizeMjData(model, 0);
  CHECK(planner.data_.size() == 1);
  mj_deleteData(data);
  TEST_MAIN_END();
}
