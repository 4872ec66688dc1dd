#include "agent_service.h"

#include <algorithm>
#include <cmath>
#include <sstream>

#include "../../../../include/mjpcx.h"
#include "../planners/gpu_cross_entropy/planner.h"
#include "../planners/gpu_sampling/planner.h"
#include "../trajectory.h"
#include "../utilities.h"

namespace mjpc::agent_grpc {

namespace {
const Status kNotInitialized{kFailedPrecondition, "Init not called."};
bool StartsWith(std::string_view s, std::string_view p) { return s.substr(0, p.size()) == p; }
Status SizeError(const char* name, int model_size, int vector_size) {  // grpc_agent_util.cc:95-115
  std::ostringstream e;
  e << "INVALID_ARGUMENT: expected " << name << " size " << model_size << ", got " << vector_size;
  return {kInvalidArgument, e.str()};
}
}  // namespace

struct AgentService::Data {
  double time = 0;
  std::vector<double> qpos, qvel, act, ctrl, mocap_pos, mocap_quat, userdata;
  mjData View() {
    mjData d{};
    d.time = time;
    d.qpos = qpos.data(); d.qvel = qvel.data(); d.mocap_pos = mocap_pos.data(); d.mocap_quat = mocap_quat.data();
    return d;
  }
};

AgentService::AgentService(std::vector<std::shared_ptr<Task>> tasks, std::string model_dir, int device, int precision,
                           int num_candidates)
    : tasks_(std::move(tasks)), model_dir_(std::move(model_dir)), device_(device), precision_(precision),
      num_candidates_(num_candidates), agent_(device, precision), pool_(1) {}

AgentService::~AgentService() = default;

// mj_resetDataKeyframe(model, data, "home") if the model has that key, else mj_resetData (agent_service.cc:118-123)
void AgentService::ResetData(Data* d) const {
  const mjModel* m = storage_->model();
  d->time = 0;
  d->qpos.assign(m->qpos0, m->qpos0 + m->nq);
  d->qvel.assign(m->nv, 0.0);
  d->act.assign(m->na, 0.0);
  d->ctrl.assign(m->nu, 0.0);
  d->userdata.assign(m->nuserdata, 0.0);
  d->mocap_pos.assign(3 * (size_t)m->nmocap, 0.0);
  d->mocap_quat.assign(4 * (size_t)m->nmocap, 0.0);
  for (int b = 0; b < m->nbody; b++)
    if (m->body_mocapid[b] >= 0) {
      mju_copy(d->mocap_pos.data() + 3 * m->body_mocapid[b], m->body_pos + 3 * b, 3);
      mju_copy(d->mocap_quat.data() + 4 * m->body_mocapid[b], m->body_quat + 4 * b, 4);
    }
  const int home = NameToId(m, mjOBJ_KEY, "home");
  if (home >= 0) {
    mju_copy(d->qpos.data(), m->key_qpos + (size_t)home * m->nq, m->nq);
    mju_copy(d->qvel.data(), m->key_qvel + (size_t)home * m->nv, m->nv);
    if (m->nmocap) mju_copy(d->mocap_pos.data(), m->key_mpos + (size_t)home * 3 * m->nmocap, 3 * m->nmocap);
  }
}

Status AgentService::Init(const std::string& task_id) {
  // agent_service.cc:86-132, grpc_agent_util.cc:535-560. A model override (InitRequest.model) is a wire-layer concern: the
  // Python front end compiles the XML to a blob and points model_dir at it; an .mjb is refused there.
  try {
    agent_.SetTaskList(tasks_);
    const int task_index = agent_.GetTaskIdByName(task_id);
    if (task_index == -1) return {kInvalidArgument, "Invalid task_id: '" + task_id + "'"};
    agent_.gui_task_id = task_index;
    agent_.SetTaskByIndex(task_index);
    std::string file = task_id;
    file.erase(std::remove(file.begin(), file.end(), ' '), file.end());
    try {
      storage_ = ModelStorage::Load(model_dir_ + "/" + file + ".mjpx");
    } catch (const std::exception& e) {
      return {kInternal, std::string("Failed to load model: ") + e.what()};
    }
    mjModel* model = storage_->model();
    agent_.Initialize(model);
    if (num_candidates_ > 0) {
      if (auto* ps = dynamic_cast<GpuSamplingPlanner*>(&agent_.ActivePlanner())) ps->num_trajectory_ = num_candidates_;
      if (auto* ce = dynamic_cast<GpuCrossEntropyPlanner*>(&agent_.ActivePlanner())) ce->num_trajectory_ = num_candidates_;
    }
    agent_.Allocate();
    agent_.Reset();
    // the service's own copy of the model steps at the model's timestep (the agent plans at agent_timestep)
    sim_model_ = std::make_unique<mjModel>(*model);
    sim_model_->nnumeric = 0;
    sim_ = std::make_unique<gpu::Context>(sim_model_.get(), *agent_.ActiveTask(), device_);
    kin_.Allocate(model);
    data_ = std::make_unique<Data>();
    ResetData(data_.get());
    agent_.state.Set(model, data_->qpos.data(), data_->qvel.data(), data_->act.data(), data_->mocap_pos.data(),
                     data_->mocap_quat.data(), data_->userdata.data(), data_->time);
    agent_.plan_enabled = true;
    agent_.action_enabled = true;
    return {};
  } catch (const std::exception& e) {
    data_.reset();
    return {kInternal, e.what()};
  }
}

// the part of mj_forward Task::Transition reads: body / site poses, subtree centres of mass and velocities
void AgentService::Forward(Data* d, mjData* view) {
  const mjModel* m = storage_->model();
  State s;
  s.Allocate(m);
  s.Set(m, d->qpos.data(), d->qvel.data(), d->act.data(), d->mocap_pos.data(), d->mocap_quat.data(), d->userdata.data(), d->time);
  std::vector<double> mocap7(7 * (size_t)m->nmocap);
  for (int k = 0; k < m->nmocap; k++) {
    mju_copy(mocap7.data() + 7 * k, d->mocap_pos.data() + 3 * k, 3);
    mju_copy(mocap7.data() + 7 * k + 3, d->mocap_quat.data() + 4 * k, 4);
  }
  sim_->Check(mjpcx_set_state(sim_->handle(), s.state().data(), d->time, mocap7.data(), nullptr));
  have_kinematics_ = sim_->Kinematics(&kin_);
  if (have_kinematics_ && view) kin_.Attach(view);
}

// mj_step with data->ctrl; `residual` (may be null) receives the task residual of the pre-step state
void AgentService::StepPhysics(Data* d, double* residual) {
  const mjModel* m = storage_->model();
  const int ds = m->nq + m->nv + m->na;
  Task* task = agent_.ActiveTask();
  Forward(d, nullptr);  // uploads the state
  sim_->SyncTask(*task);
  Trajectory one;
  one.Initialize(ds, m->nu, task->num_residual, task->num_trace, 2);
  one.Allocate(2);
  sim_->Check(mjpcx_rollout_splines(sim_->handle(), 1, 2, 1, MJPCX_SPLINE_ZERO, &d->time, d->ctrl.data()));
  sim_->FetchTrajectory(0, &one);
  if (one.failure) throw std::runtime_error("simulation diverged");
  if (residual) mju_copy(residual, one.residual.data(), task->num_residual);
  mju_copy(d->qpos.data(), one.states.data() + ds, m->nq);
  mju_copy(d->qvel.data(), one.states.data() + ds + m->nq, m->nv);
  d->time = one.times[1];
}

Status AgentService::GetState(StateMsg* out) {
  if (!Initialized()) return kNotInitialized;
  const mjModel* m = storage_->model();
  const State& s = agent_.state;  // agent_.state.CopyTo(model, data_); then the fields of data_ (agent_service.cc:150-157)
  out->has_time = true;
  out->time = s.time();
  out->qpos.assign(s.state().begin(), s.state().begin() + m->nq);
  out->qvel.assign(s.state().begin() + m->nq, s.state().begin() + m->nq + m->nv);
  out->act.assign(s.state().begin() + m->nq + m->nv, s.state().begin() + m->nq + m->nv + m->na);
  out->mocap_pos.resize(3 * (size_t)m->nmocap);
  out->mocap_quat.resize(4 * (size_t)m->nmocap);
  for (int k = 0; k < m->nmocap; k++) {
    mju_copy(out->mocap_pos.data() + 3 * k, s.mocap().data() + 7 * k, 3);
    mju_copy(out->mocap_quat.data() + 4 * k, s.mocap().data() + 7 * k + 3, 4);
  }
  out->userdata = s.userdata();
  return {};
}

// grpc_agent_util.cc:117-155
Status AgentService::SetStateFields(const StateMsg& st) {
  const mjModel* m = storage_->model();
  Data& d = *data_;
  if (st.has_time) d.time = st.time;
#define FIELD(name, n)                                                           \
  if (!st.name.empty()) {                                                        \
    if ((int)st.name.size() != (n)) return SizeError(#name, (n), (int)st.name.size()); \
    d.name = st.name;                                                            \
  }
  FIELD(qpos, m->nq) FIELD(qvel, m->nv) FIELD(act, m->na) FIELD(mocap_pos, 3 * m->nmocap) FIELD(mocap_quat, 4 * m->nmocap)
  FIELD(userdata, m->nuserdata)
#undef FIELD
  agent_.state.Set(m, d.qpos.data(), d.qvel.data(), d.act.data(), d.mocap_pos.data(), d.mocap_quat.data(), d.userdata.data(), d.time);
  return {};
}

Status AgentService::SetState(const StateMsg& st) {
  if (!Initialized()) return kNotInitialized;
  try {
    Status s = SetStateFields(st);
    if (!s.ok()) return s;
    // mj_forward; task->Transition(model, data_); agent_.SetState(data_) (agent_service.cc:168-172)
    Data& d = *data_;
    mjModel* m = storage_->model();
    mjData view = d.View();
    Forward(&d, &view);
    agent_.ActiveTask()->Transition(m, &view);
    agent_.state.Set(m, d.qpos.data(), d.qvel.data(), d.act.data(), d.mocap_pos.data(), d.mocap_quat.data(), d.userdata.data(), d.time);
    return {};
  } catch (const std::exception& e) {
    return {kInternal, e.what()};
  }
}

// grpc_agent_util.cc:159-221
Status AgentService::GetAction(bool has_time, double time_in, double averaging_duration, bool nominal_action,
                               std::vector<double>* action) {
  if (!Initialized()) return kNotInitialized;
  try {
    const mjModel* m = storage_->model();
    Planner& planner = agent_.ActivePlanner();
    double time = has_time ? time_in : agent_.state.time();
    std::vector<double> ret(m->nu, 0.0);
    if (averaging_duration > 0) {
      int nactions = 0;
      const double end_time = time + averaging_duration;
      if (nominal_action) {
        std::vector<double> a(m->nu, 0.0);
        while (time < end_time) {
          planner.ActionFromPolicy(a.data(), nullptr, time);
          mju_addTo(ret.data(), a.data(), m->nu);
          time += m->opt.timestep;
          nactions++;
        }
      } else {  // roll the physics out under the policy (no Task::Transition during the rollout)
        Data r = *data_;
        {
          StateMsg s;
          GetState(&s);
          r.qpos = s.qpos; r.qvel = s.qvel; r.act = s.act; r.mocap_pos = s.mocap_pos; r.mocap_quat = s.mocap_quat; r.userdata = s.userdata;
        }
        r.time = time;
        State rs;
        rs.Allocate(m);
        while (r.time <= end_time) {
          rs.Set(m, r.qpos.data(), r.qvel.data(), r.act.data(), r.mocap_pos.data(), r.mocap_quat.data(), r.userdata.data(), r.time);
          planner.ActionFromPolicy(r.ctrl.data(), rs.state().data(), r.time);
          mju_addTo(ret.data(), r.ctrl.data(), m->nu);
          StepPhysics(&r, nullptr);
          nactions++;
        }
      }
      mju_scl(ret.data(), ret.data(), 1.0 / nactions, m->nu);
    } else {
      planner.ActionFromPolicy(ret.data(), nominal_action ? nullptr : agent_.state.state().data(), time);
    }
    *action = ret;
    data_->ctrl = ret;  // agent_service.cc:186-189
    return {};
  } catch (const std::exception& e) {
    return {kInternal, e.what()};
  }
}

// grpc_agent_util.cc:223-275: task->Residual(model, data, ...) at the service's data, split per user sensor
Status AgentService::GetCostTerms(std::vector<CostTerm>* out) {
  if (!Initialized()) return kNotInitialized;
  try {
    Task* task = agent_.ActiveTask();
    std::vector<double> residual(task->num_residual, 0.0), terms(task->num_term, 0.0);
    Data scratch = *data_;  // the residual is the sensor stage of mj_forward at data_: evaluate it without advancing data_
    StepPhysics(&scratch, residual.data());
    task->UnweightedCostTerms(terms.data(), residual.data());
    out->clear();
    int shift = 0;
    for (int i = 0; i < task->num_term; i++) {
      CostTerm t;
      t.name = task->weight_names[i];
      t.value = terms[i];
      t.weight = task->weight[i];
      t.residual.assign(residual.begin() + shift, residual.begin() + shift + task->dim_norm_residual[i]);
      shift += task->dim_norm_residual[i];
      out->push_back(std::move(t));
    }
    return {};
  } catch (const std::exception& e) {
    return {kInternal, e.what()};
  }
}

Status AgentService::PlannerStep() {
  if (!Initialized()) return kNotInitialized;
  try {
    agent_.plan_enabled = true;
    agent_.PlanIteration(&pool_);
    return {};
  } catch (const std::exception& e) {
    return {kInternal, e.what()};
  }
}

// agent_service.cc:228-244
Status AgentService::Step(bool use_previous_policy) {
  if (!Initialized()) return kNotInitialized;
  try {
    mjModel* m = storage_->model();
    Data& d = *data_;
    StateMsg s;
    GetState(&s);  // state.CopyTo(model, data_)
    d.time = s.time; d.qpos = s.qpos; d.qvel = s.qvel; d.act = s.act; d.mocap_pos = s.mocap_pos; d.mocap_quat = s.mocap_quat;
    d.userdata = s.userdata;
    mjData view = d.View();
    Forward(&d, &view);
    agent_.ActiveTask()->Transition(m, &view);
    agent_.ActivePlanner().ActionFromPolicy(d.ctrl.data(), agent_.state.state().data(), agent_.state.time(), use_previous_policy);
    StepPhysics(&d, nullptr);
    agent_.state.Set(m, d.qpos.data(), d.qvel.data(), d.act.data(), d.mocap_pos.data(), d.mocap_quat.data(), d.userdata.data(), d.time);
    return {};
  } catch (const std::exception& e) {
    return {kInternal, e.what()};
  }
}

// grpc_agent_util.cc:277-287
Status AgentService::Reset() {
  if (!Initialized()) return kNotInitialized;
  try {
    const mjModel* m = storage_->model();
    agent_.Reset();
    ResetData(data_.get());
    Data& d = *data_;
    agent_.state.Set(m, d.qpos.data(), d.qvel.data(), d.act.data(), d.mocap_pos.data(), d.mocap_quat.data(), d.userdata.data(), d.time);
    return {};
  } catch (const std::exception& e) {
    return {kInternal, e.what()};
  }
}

// grpc_agent_util.cc:289-343
Status AgentService::SetTaskParameters(const std::map<std::string, TaskParameter>& parameters) {
  if (!Initialized()) return kNotInitialized;
  const mjModel* m = agent_.GetModel();
  for (const auto& [name, value] : parameters) {
    const int found = value.is_selection ? agent_.SetSelectionParamByName(name, value.selection) : agent_.SetParamByName(name, value.numeric);
    if (found == -1) {
      std::ostringstream e;
      e << "Parameter " << name << " not found in task.  Available names are:\n";
      for (int i = 0; i < m->nnumeric; i++) {
        const std::string_view n(m->names + m->name_numericadr[i]);
        const bool select = StartsWith(n, "residual_select_");
        if (value.is_selection ? select : (StartsWith(n, "residual_") && !select)) e << n.substr(value.is_selection ? 16 : 9) << "\n";
      }
      return {kInvalidArgument, e.str()};
    }
  }
  agent_.ActiveTask()->UpdateResidual();
  return {};
}

// grpc_agent_util.cc:345-369
Status AgentService::GetTaskParameters(std::vector<std::pair<std::string, TaskParameter>>* out) {
  if (!Initialized()) return kNotInitialized;
  const mjModel* m = agent_.GetModel();
  out->clear();
  int shift = 0;
  for (int i = 0; i < m->nnumeric; i++) {
    const std::string_view n(m->names + m->name_numericadr[i]);
    if (!StartsWith(n, "residual_")) continue;
    TaskParameter p;
    std::string name;
    if (StartsWith(n, "residual_select_")) {
      name = std::string(n.substr(16));
      p.is_selection = true;
      p.selection = ResidualSelection(m, name, agent_.ActiveTask()->parameters[shift]);
    } else {
      name = std::string(n.substr(9));
      p.numeric = agent_.ActiveTask()->parameters[shift];
    }
    out->emplace_back(std::move(name), std::move(p));
    shift++;
  }
  return {};
}

// grpc_agent_util.cc:371-401
Status AgentService::SetCostWeights(bool reset_to_defaults, const std::map<std::string, double>& cost_weights) {
  if (!Initialized()) return kNotInitialized;
  try {
    if (reset_to_defaults) agent_.ActiveTask()->Reset(agent_.GetModel());
  } catch (const std::exception& e) {
    return {kInternal, e.what()};
  }
  for (const auto& [name, weight] : cost_weights)
    if (agent_.SetWeightByName(name, weight) == -1) {
      std::ostringstream e;
      e << "Weight '" << name << "' not found in task. Available names are:\n";
      for (const std::string& n : agent_.ActiveTask()->weight_names) e << "  " << n << "\n";
      return {kInvalidArgument, e.str()};
    }
  return {};
}

// grpc_agent_util.cc:404-437
Status AgentService::SetMode(const std::string& mode) {
  if (!Initialized()) return kNotInitialized;
  if (agent_.SetModeByName(mode) == -1) {
    std::ostringstream e;
    e << "Mode '" << mode << "' not found in task. Available names are:\n";
    for (const std::string& n : agent_.GetAllModeNames()) e << "  " << n << "\n";
    return {kInvalidArgument, e.str()};
  }
  return {};
}
Status AgentService::GetMode(std::string* mode) {
  if (!Initialized()) return kNotInitialized;
  *mode = agent_.GetModeName();
  return {};
}
Status AgentService::GetAllModes(std::vector<std::string>* modes) {
  if (!Initialized()) return kNotInitialized;
  *modes = agent_.GetAllModeNames();
  return {};
}

// agent_service.cc:312-347: states and times for every plan step, actions for all but the last
Status AgentService::GetBestTrajectory(std::vector<double>* states, std::vector<double>* actions, std::vector<double>* times,
                                       int* steps) {
  if (!Initialized()) return kNotInitialized;
  try {
    const Trajectory* tr = agent_.ActivePlanner().BestTrajectory();
    // no plan yet (right after Init / Reset, or after NominalTrajectory on the Cross-Entropy planner): the planners return
    // nullptr where the reference returns its pre-allocated buffer -- report it instead of dereferencing
    if (tr == nullptr) return {kFailedPrecondition, "No trajectory has been planned yet: call PlannerStep first."};
    const int ns = tr->dim_state, na = tr->dim_action;
    *steps = agent_.PlanSteps() < tr->horizon ? agent_.PlanSteps() : tr->horizon;  // never past the rows this plan wrote
    states->clear(); actions->clear(); times->clear();
    for (int t = 0; t < *steps; t++) {
      states->insert(states->end(), tr->states.begin() + (size_t)t * ns, tr->states.begin() + (size_t)(t + 1) * ns);
      times->push_back(tr->times[t]);
      if (t >= *steps - 1) continue;
      actions->insert(actions->end(), tr->actions.begin() + (size_t)t * na, tr->actions.begin() + (size_t)(t + 1) * na);
    }
    return {};
  } catch (const std::exception& e) {
    return {kInternal, e.what()};
  }
}

// grpc_agent_util.cc:439-520 (state, cost weights, mode, mocap poses by body name -- in that order; the request's
// `parameters` map is not read by the reference either)
Status AgentService::SetAnything(const StateMsg* state, const std::map<std::string, double>& cost_weights,
                                 const std::string& mode, const std::map<std::string, Pose>& mocap) {
  if (!Initialized()) return kNotInitialized;
  const mjModel* m = storage_->model();
  if (state) {
    Status s = SetStateFields(*state);
    if (!s.ok()) return s;
  }
  if (!cost_weights.empty()) {
    Status s = SetCostWeights(false, cost_weights);
    if (!s.ok()) return s;
  }
  if (!mode.empty()) {
    Status s = SetMode(mode);
    if (!s.ok()) return s;
  }
  if (!mocap.empty()) {
    for (const auto& [name, pose] : mocap) {
      const int id = NameToId(m, mjOBJ_BODY, name);
      if (id < 0) return {kInvalidArgument, "Body '" + name + "' not found."};
      if (m->body_mocapid[id] < 0) return {kInvalidArgument, "Body '" + name + "' is not a mocap body."};
      if (!pose.pos.empty() && pose.pos.size() != 3)
        return {kInvalidArgument, "Mocap '" + name + "' has invalid pose size " + std::to_string(pose.pos.size()) + "."};
      if (!pose.quat.empty() && pose.quat.size() != 4)
        return {kInvalidArgument, "Mocap '" + name + "' has invalid quat size " + std::to_string(pose.quat.size()) + "."};
    }
    Data& d = *data_;
    for (const auto& [name, pose] : mocap) {
      const int mid = m->body_mocapid[NameToId(m, mjOBJ_BODY, name)];
      for (size_t i = 0; i < pose.pos.size(); i++) d.mocap_pos[3 * mid + i] = pose.pos[i];
      if (pose.quat.size() == 4) {
        double n = 0;
        for (int i = 0; i < 4; i++) n += pose.quat[i] * pose.quat[i];
        n = std::sqrt(n);
        for (int i = 0; i < 4; i++) d.mocap_quat[4 * mid + i] = n > 1e-15 ? pose.quat[i] / n : (i == 0 ? 1.0 : 0.0);
      }
    }
    agent_.state.Set(m, d.qpos.data(), d.qvel.data(), d.act.data(), d.mocap_pos.data(), d.mocap_quat.data(), d.userdata.data(), d.time);
  }
  return {};
}

}  // namespace mjpc::agent_grpc
