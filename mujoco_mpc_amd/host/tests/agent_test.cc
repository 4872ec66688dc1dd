// mjpc/test/agent/agent_test.cc in spirit: an Agent constructed on a task plans (PlanIteration), its action becomes
// non-trivial, and the by-name setters reach the task. argv[1] = directory with Particle.mjpx / Cartpole.mjpx /
// QuadrupedFlat.mjpx. Needs a GPU (the planners have no CPU path).
#include <algorithm>
#include <cmath>
#include <string>

#include "check.h"
#include "mjpc/agent.h"
#include "mjpc/tasks/tasks.h"
#include "mjpc/utilities.h"
#include "model_io.h"

using namespace mjpc;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  {  // planner registry: reference enum slots
    auto planners = LoadPlanners();
    CHECK((int)planners.size() == kNumPlannerTypes);
    CHECK(planners[kSamplingPlanner] && planners[kILQGPlanner] && planners[kCrossEntropyPlanner]);
    CHECK(!planners[kGradientPlanner] && !planners[kILQSPlanner]);
    CHECK(planners[kSampleGradientPlanner] != nullptr);
    CHECK(planners[kRobustPlanner] != nullptr);  // RobustPlanner over a GpuSamplingPlanner delegate
  }
  for (const char* name : {"Particle", "Cartpole", "Quadruped Flat"}) {
    std::string file = name;
    file.erase(std::remove(file.begin(), file.end(), ' '), file.end());
    auto storage = ModelStorage::Load(dir + "/" + file + ".mjpx");
    std::shared_ptr<Task> task;
    for (auto& t : GetTasks()) if (t->Name() == name) task = t;
    CHECK(task != nullptr);
    Agent agent;
    agent.SetTaskList({task});
    agent.Initialize(storage->model());
    if (std::string(name) == "Quadruped Flat") agent.SetPlanner(kCrossEntropyPlanner);  // (the XML asks for iLQG; this loop exercises the sampling-family planners)
    agent.Allocate();
    agent.Reset();
    const mjModel* m = storage->model();
    CHECK(agent.PlanSteps() == (int)std::fmin(std::fmax(agent.Horizon() / GetNumberOrDefault(0.01, m, "agent_timestep") + 1, 1), 512));
    // state: home keyframe if present
    std::vector<double> qpos(m->qpos0, m->qpos0 + m->nq), qvel(m->nv, 0.0);
    if (const double* home = KeyQPosByName(m, "home")) qpos.assign(home, home + m->nq);
    std::vector<double> mp(3 * (size_t)m->nmocap), mq(4 * (size_t)m->nmocap);
    for (int b = 0; b < m->nbody; b++)
      if (m->body_mocapid[b] >= 0) { mju_copy(mp.data() + 3 * m->body_mocapid[b], m->body_pos + 3 * b, 3); mju_copy(mq.data() + 4 * m->body_mocapid[b], m->body_quat + 4 * b, 4); }
    agent.state.Set(m, qpos.data(), qvel.data(), nullptr, mp.data(), mq.data(), nullptr, 0.0);
    ThreadPool pool(1);
    for (int k = 0; k < 3; k++) agent.PlanIteration(&pool);
    std::vector<double> action(m->nu, 0.0);
    agent.ActivePlanner().ActionFromPolicy(action.data(), agent.state.state().data(), 0.0);
    bool finite = true;
    for (double a : action) finite = finite && std::isfinite(a) && std::fabs(a) <= 1.0 + 1e-12;
    CHECK(finite);
    CHECK(agent.ComputeTime() > 0);
    CHECK(agent.SetWeightByName(task->weight_names[0], 0.123) == 0 && task->weight[0] == 0.123);
    CHECK(agent.SetWeightByName("no such term", 1.0) == -1);
    agent.plan_enabled = false;  // NominalTrajectory path
    agent.PlanIteration(&pool);
    CHECK(agent.ActivePlanner().BestTrajectory() != nullptr);
  }
  {  // SetPlanner: out-of-range and unfilled slots fall back to Sampling; a real switch allocates the new planner before use
    auto storage = ModelStorage::Load(dir + "/Particle.mjpx");
    std::shared_ptr<Task> task;
    for (auto& t : GetTasks()) if (t->Name() == "Particle") task = t;
    CHECK(task != nullptr);
    Agent agent;
    agent.SetTaskList({task});
    agent.Initialize(storage->model());
    agent.Allocate();
    agent.Reset();
    const mjModel* m = storage->model();
    std::vector<double> qpos(m->qpos0, m->qpos0 + m->nq), qvel(m->nv, 0.0);
    const double mp[3] = {0.25, 0, 0.01}, mq[4] = {1, 0, 0, 0};
    agent.state.Set(m, qpos.data(), qvel.data(), nullptr, mp, mq, nullptr, 0.0);
    agent.SetPlanner(99);
    CHECK(agent.planner_id() == kSamplingPlanner);
    agent.SetPlanner(kGradientPlanner);  // no device implementation behind this slot
    CHECK(agent.planner_id() == kSamplingPlanner);
    agent.SetPlanner(kCrossEntropyPlanner);
    CHECK(agent.planner_id() == kCrossEntropyPlanner);
    ThreadPool pool(1);
    agent.PlanIteration(&pool);  // would dereference an unallocated planner without the lazy Allocate
    CHECK(agent.ActivePlanner().BestTrajectory() != nullptr);
    // switch -> Reset -> switch back: Agent::Reset resets EVERY planner that holds a context (the reference's Agent::Reset does), so
    // the planner the caller returns to starts from the zero policy, not from the one it had before the simulation was reset
    std::vector<double> action(m->nu, 0.0), zero(m->nu, 0.0);
    for (int k = 0; k < 3; k++) agent.PlanIteration(&pool);
    agent.ActivePlanner().ActionFromPolicy(action.data(), agent.state.state().data(), 0.05);
    bool moved = false;
    for (int i = 0; i < m->nu; i++) moved |= action[i] != 0.0;
    CHECK(moved);                               // the cross-entropy planner holds a non-trivial policy now
    agent.SetPlanner(kSamplingPlanner);         // away ...
    agent.Reset();                              // ... the simulation is reset while another planner is active ...
    agent.SetPlanner(kCrossEntropyPlanner);     // ... and back
    agent.ActivePlanner().ActionFromPolicy(action.data(), agent.state.state().data(), 0.05);
    for (int i = 0; i < m->nu; i++) CHECK(action[i] == 0.0);
  }
  TEST_MAIN_END();
}
