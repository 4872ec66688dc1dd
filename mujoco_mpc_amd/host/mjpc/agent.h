// mjpc::Agent (mjpc/agent.{h,cc}), the plan loop around the GPU planners: task list, planner registry, the planning
// copy's timestep/horizon (`agent_timestep`, `agent_horizon`, `agent_planner`), `PlanIteration` with the per-plan frozen
// residual copy, and the by-name setters the gRPC/UI layers call. GUI, plotting, estimators and model loading from
// files are front-end concerns and are not part of this class (out of scope).
#pragma once
#include <atomic>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "planners/include.h"
#include "states/state.h"
#include "task.h"
#include "threadpool.h"

namespace mjpc {

class Agent {
 public:
  explicit Agent(int device = 0, int precision = 64) : planners_(LoadPlanners(device, precision)) {}
  Agent(mjModel* model, std::shared_ptr<Task> task, int device = 0, int precision = 64);

  void SetTaskList(std::vector<std::shared_ptr<Task>> tasks) { tasks_ = std::move(tasks); }
  void Initialize(mjModel* model);  // agent.cc:71-139 (the agent plans on `model` with its own timestep: FlatModel override)
  void Allocate();                  // agent.cc:142-156
  void Reset(const double* initial_repeated_action = nullptr);  // agent.cc:159-182
  void PlanIteration(ThreadPool* pool);                         // agent.cc:283-357
  void Plan(std::atomic<bool>& exitrequest, std::atomic<int>& uiloadrequest);  // agent.cc:360-371

  Planner& ActivePlanner() const { return *planners_[planner_]; }
  Task* ActiveTask() const { return tasks_[active_task_id_].get(); }
  int PlanSteps() const { return steps_; }
  double Horizon() const { return horizon_; }
  double ComputeTime() const { return agent_compute_time_; }
  int GetActionDim() const { return model_->nu; }
  int GetTaskIdByName(std::string_view name) const;
  void SetTaskByIndex(int id) { active_task_id_ = id; }
  // an index without a planner behind it (kGradient / iLQS have no device implementation, see Initialize) falls back to
  // Sampling, as Initialize does for the agent_planner numeric; the newly active planner is (re)allocated before it plans
  void SetPlanner(int planner);
  int planner_id() const { return planner_; }
  int SetParamByName(std::string_view name, double value);   // "residual_<name>" numerics (agent.cc:1016-1030)
  int SetWeightByName(std::string_view name, double value);  // cost-term weights (agent.cc:1061-1075)
  int SetModeByName(std::string_view name);                  // task_transition entries (agent.cc:472-490)
  int SetSelectionParamByName(std::string_view name, std::string_view value);  // drop-down parameters (agent.cc:421-444)
  std::vector<std::string> GetAllModeNames() const;          // agent.cc:458-465
  std::string GetModeName() const;                           // agent.cc:467-470
  mjModel* GetModel() const { return model_; }

  State state;
  bool plan_enabled = true, action_enabled = true, allocate_enabled = false;
  std::vector<bool> allocated_;  // per planner: holds a device context for the current model / task (allocated on first activation)
  int gui_task_id = 0;

 private:
  mjModel* model_ = nullptr;
  std::vector<std::shared_ptr<Task>> tasks_;
  std::vector<std::unique_ptr<Planner>> planners_;
  std::unique_ptr<ResidualFn> residual_fn_;
  int active_task_id_ = 0, planner_ = 0, steps_ = 1, count_ = 0;
  double horizon_ = 0.5, timestep_ = 1.0e-2, agent_compute_time_ = 0;
  int integrator_ = 0;
};

}  // namespace mjpc
