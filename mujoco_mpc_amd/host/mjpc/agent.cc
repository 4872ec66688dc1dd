#include "agent.h"

#include <cctype>
#include <chrono>
#include <cstdio>
#include <sstream>
#include <stdexcept>

#include "trajectory.h"
#include "utilities.h"

namespace mjpc {

Agent::Agent(mjModel* model, std::shared_ptr<Task> task, int device, int precision) : Agent(device, precision) {
  SetTaskList({std::move(task)});
  Initialize(model);
  Allocate();
  Reset();
}

void Agent::Initialize(mjModel* model) {
  model_ = model;
  int num_missing = 0;  // ctrl limits are required for all actuators (agent.cc:76-88)
  for (int i = 0; i < model_->nu; i++)
    if (!model_->actuator_ctrllimited[i]) { num_missing++; std::printf("actuator %i missing limits\n", i); }
  if (num_missing > 0) throw std::runtime_error("Ctrl limits required for all actuators.");
  planner_ = GetNumberOrDefault(0, model, "agent_planner");
  if (planner_ < 0 || planner_ >= (int)planners_.size() || !planners_[planner_]) {  // a slot this build does not fill (Gradient, iLQS)
    std::fprintf(stderr, "agent_planner %d is not available on the GPU path; using the Sampling planner\n", planner_);
    planner_ = kSamplingPlanner;
  }
  integrator_ = GetNumberOrDefault(model->opt.integrator, model, "agent_integrator");
  horizon_ = GetNumberOrDefault(0.5, model, "agent_horizon");
  timestep_ = GetNumberOrDefault(1.0e-2, model, "agent_timestep");
  steps_ = (int)mju_max(mju_min(horizon_ / timestep_ + 1, kMaxTrajectoryHorizon), 1);
  active_task_id_ = gui_task_id;
  ActiveTask()->Reset(model);
  for (const auto& planner : planners_)
    if (planner) planner->Initialize(model_, *ActiveTask());
  state.Allocate(model);
  state.Reset();
  plan_enabled = true;
  action_enabled = true;
  allocate_enabled = true;
  count_ = 0;
}

void Agent::SetPlanner(int planner) {
  if (planner < 0 || planner >= (int)planners_.size() || !planners_[planner]) {
    std::fprintf(stderr, "planner %d is not available on the GPU path; using the Sampling planner\n", planner);
    planner = kSamplingPlanner;
  }
  if (planner == planner_) return;
  planner_ = planner;
  // Allocate() only ever gave the then-active planner a device context: a planner gets its own the FIRST time it becomes active and
  // keeps it (and its policy) when the caller switches away and back, as the reference's planners keep their buffers
  if (allocated_.size() != planners_.size()) allocated_.assign(planners_.size(), false);
  if (model_ && !allocate_enabled && !allocated_[planner_]) {
    ActivePlanner().Allocate();
    ActivePlanner().Reset(kMaxTrajectoryHorizon);
    allocated_[planner_] = true;
  }
}

void Agent::Allocate() {
  // only the active planner gets a device context here (the reference allocates every planner's buffers; on the
  // device that would mean one model upload and LDS/scratch budget per unused planner)
  ActivePlanner().Allocate();
  allocated_.assign(planners_.size(), false);  // (a new model / task: every other planner's context is stale)
  allocated_[planner_] = true;
  allocate_enabled = false;
}

void Agent::Reset(const double* initial_repeated_action) {
  // every planner that holds a device context is reset, as the reference resets every planner (agent.cc Reset): a planner the caller
  // switches back to after a simulation reset must not resume the policy it had before it
  for (size_t i = 0; i < planners_.size(); i++) {
    if ((int)i == planner_ || !planners_[i]) continue;
    if (i < allocated_.size() && allocated_[i]) planners_[i]->Reset(kMaxTrajectoryHorizon, initial_repeated_action);
  }
  ActivePlanner().Reset(kMaxTrajectoryHorizon, initial_repeated_action);
  state.Reset();
  count_ = 0;
}

void Agent::PlanIteration(ThreadPool* pool) {
  const auto agent_start = std::chrono::steady_clock::now();
  // the planning copy's opt.timestep / opt.integrator = agent_timestep / agent_integrator: applied where the model is
  // flattened for the device (gpu::Context), the host mjModel is left untouched
  steps_ = (int)mju_max(mju_min(horizon_ / timestep_ + 1, kMaxTrajectoryHorizon), 1);
  if (allocate_enabled) return;
  ActivePlanner().SetState(state);
  // a frozen copy of the residual parameters for this plan (agent.cc:319); the device copy is refreshed from the task by
  // the planner itself (gpu::Context::SyncTask) right before its rollouts
  residual_fn_ = ActiveTask()->Residual();
  if (plan_enabled) {
    ActivePlanner().OptimizePolicy(steps_, *pool);
    agent_compute_time_ = GetDuration(agent_start);
    count_ += 1;
  } else {
    ActivePlanner().NominalTrajectory(steps_, *pool);
    agent_compute_time_ = 0.0;
  }
  residual_fn_.reset();
}

void Agent::Plan(std::atomic<bool>& exitrequest, std::atomic<int>& uiloadrequest) {
  ThreadPool pool(1);
  while (!exitrequest.load())
    if (model_ && uiloadrequest.load() == 0) PlanIteration(&pool);
}

int Agent::GetTaskIdByName(std::string_view name) const {
  for (size_t i = 0; i < tasks_.size(); i++)
    if (tasks_[i]->Name() == name) return (int)i;
  for (size_t i = 0; i < tasks_.size(); i++)  // the spelling without spaces (model file names, the Python registry)
    if (SameTaskName(tasks_[i]->Name(), name)) return (int)i;
  return -1;
}

namespace {
bool StartsWith(std::string_view s, std::string_view p) { return s.substr(0, p.size()) == p; }
bool EqualsIgnoreCase(std::string_view a, std::string_view b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i++)
    if (std::tolower((unsigned char)a[i]) != std::tolower((unsigned char)b[i])) return false;
  return true;
}
}  // namespace

// agent.cc:395-419: name with or without the "residual_" prefix, compared ignoring case; returns the numeric's id
int Agent::SetParamByName(std::string_view name, double value) {
  if (StartsWith(name, "residual_")) name.remove_prefix(9);
  if (StartsWith(name, "selection_")) return -1;  // SetSelectionParamByName is the interface for those
  int shift = 0;
  for (int i = 0; i < model_->nnumeric; i++) {
    const std::string_view n(model_->names + model_->name_numericadr[i]);
    if (!StartsWith(n, "residual_")) continue;
    if (EqualsIgnoreCase(n.substr(9), name)) { ActiveTask()->parameters[shift] = value; return i; }
    shift++;
  }
  return -1;
}

// agent.cc:421-444. The parameter slot is the position among ALL "residual_" numerics, which is how Task::parameters is laid
// out (task.cc:225-243); the reference counts only the "residual_select_" ones before it and so writes the wrong slot when a
// plain parameter precedes the selection.
int Agent::SetSelectionParamByName(std::string_view name, std::string_view value) {
  if (StartsWith(name, "residual_select_")) name.remove_prefix(16);
  if (StartsWith(name, "selection_")) name.remove_prefix(10);
  int shift = 0;
  for (int i = 0; i < model_->nnumeric; i++) {
    const std::string_view n(model_->names + model_->name_numericadr[i]);
    if (!StartsWith(n, "residual_")) continue;
    if (StartsWith(n, "residual_select_") && EqualsIgnoreCase(n.substr(16), name)) {
      ActiveTask()->parameters[shift] = ResidualParameterFromSelection(model_, n.substr(16), value);
      return i;
    }
    shift++;
  }
  return -1;
}

int Agent::SetWeightByName(std::string_view name, double value) {
  Task* t = ActiveTask();
  for (int i = 0; i < t->num_term; i++)
    if (EqualsIgnoreCase(t->weight_names[i], name)) { t->weight[i] = value; return i; }
  return -1;
}

std::vector<std::string> Agent::GetAllModeNames() const {
  if (const char* transition = GetCustomTextData(model_, "task_transition")) return SplitBar(transition, true);
  return {"default_mode"};
}

std::string Agent::GetModeName() const {
  const std::vector<std::string> names = GetAllModeNames();
  const int mode = ActiveTask()->mode;
  return mode >= 0 && mode < (int)names.size() ? names[mode] : "";
}

int Agent::SetModeByName(std::string_view name) {
  if (GetCustomTextData(model_, "task_transition")) {
    const std::vector<std::string> names = GetAllModeNames();
    for (size_t i = 0; i < names.size(); i++)
      if (names[i] == name) { ActiveTask()->mode = (int)i; return (int)i; }
    return -1;
  }
  if (name == "default_mode") { ActiveTask()->mode = 0; return 0; }
  return -1;
}

}  // namespace mjpc
