// C ABI over mjpc::agent_grpc::AgentService for the Python gRPC front end (mujoco_mpc_amd/grpc_service.py): one function per
// RPC of agent.proto, plain pointers and sizes, the return value is the gRPC status code (0 = OK) and
// mjpc_agent_service_error() the status message. Variable-length outputs are written into caller buffers of stated capacity;
// strings are zero-terminated.
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mjpc/grpc/agent_service.h"
#include "mjpc/tasks/tasks.h"

namespace {
using mjpc::agent_grpc::AgentService;
using mjpc::agent_grpc::Status;
struct Handle {
  AgentService service;
  std::string error;
  std::vector<mjpc::agent_grpc::CostTerm> terms;
  std::vector<std::pair<std::string, mjpc::agent_grpc::TaskParameter>> params;
  Handle(const char* dir, int device, int precision, int n) : service(mjpc::GetTasks(), dir, device, precision, n) {}
};
int Done(Handle* h, const Status& s) {
  h->error = s.message;
  return s.code;
}
void CopyStr(char* dst, int cap, const std::string& s) {
  if (!dst || cap <= 0) return;
  std::strncpy(dst, s.c_str(), (size_t)cap - 1);
  dst[cap - 1] = 0;
}
mjpc::agent_grpc::StateMsg MakeState(int has_time, double time, const double* qpos, int nqpos, const double* qvel, int nqvel,
                                     const double* act, int nact, const double* mpos, int nmpos, const double* mquat, int nmquat,
                                     const double* user, int nuser) {
  mjpc::agent_grpc::StateMsg s;
  s.has_time = has_time != 0;
  s.time = time;
  if (nqpos > 0) s.qpos.assign(qpos, qpos + nqpos);
  if (nqvel > 0) s.qvel.assign(qvel, qvel + nqvel);
  if (nact > 0) s.act.assign(act, act + nact);
  if (nmpos > 0) s.mocap_pos.assign(mpos, mpos + nmpos);
  if (nmquat > 0) s.mocap_quat.assign(mquat, mquat + nmquat);
  if (nuser > 0) s.userdata.assign(user, user + nuser);
  return s;
}
}  // namespace

#define H static_cast<Handle*>(h)

extern "C" {

void* mjpc_agent_service_create(const char* model_dir, int device, int precision, int num_candidates) {
  try {
    return new Handle(model_dir, device, precision, num_candidates);
  } catch (...) {
    return nullptr;
  }
}
void mjpc_agent_service_destroy(void* h) { delete H; }
const char* mjpc_agent_service_error(void* h) { return H->error.c_str(); }

int mjpc_agent_service_init(void* h, const char* task_id) { return Done(H, H->service.Init(task_id)); }

// sizes[10] = nq, nv, na, nmocap, nuserdata, nu, num_term, num_residual, number of task parameters, plan steps
int mjpc_agent_service_sizes(void* h, int* sizes) {
  if (!H->service.Initialized()) return Done(H, {mjpc::agent_grpc::kFailedPrecondition, "Init not called."});
  const mjModel* m = H->service.model();
  const mjpc::Task* t = H->service.agent().ActiveTask();
  const int v[10] = {m->nq, m->nv, m->na, m->nmocap, m->nuserdata, m->nu, t->num_term, t->num_residual, (int)t->parameters.size(),
                     H->service.agent().PlanSteps()};
  std::memcpy(sizes, v, sizeof v);
  return 0;
}

int mjpc_agent_service_get_state(void* h, double* time, double* qpos, double* qvel, double* act, double* mocap_pos,
                                 double* mocap_quat, double* userdata) {
  mjpc::agent_grpc::StateMsg s;
  const Status st = H->service.GetState(&s);
  if (st.ok()) {
    *time = s.time;
    auto put = [](double* dst, const std::vector<double>& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * 8); };
    put(qpos, s.qpos); put(qvel, s.qvel); put(act, s.act); put(mocap_pos, s.mocap_pos); put(mocap_quat, s.mocap_quat);
    put(userdata, s.userdata);
  }
  return Done(H, st);
}

// a count of 0 = "field not set" (agent.proto State: repeated fields, optional time)
int mjpc_agent_service_set_state(void* h, int has_time, double time, const double* qpos, int nqpos, const double* qvel, int nqvel,
                                 const double* act, int nact, const double* mocap_pos, int nmocap_pos, const double* mocap_quat,
                                 int nmocap_quat, const double* userdata, int nuserdata) {
  return Done(H, H->service.SetState(MakeState(has_time, time, qpos, nqpos, qvel, nqvel, act, nact, mocap_pos, nmocap_pos,
                                               mocap_quat, nmocap_quat, userdata, nuserdata)));
}

int mjpc_agent_service_get_action(void* h, int has_time, double time, double averaging_duration, int nominal_action,
                                  double* action) {
  std::vector<double> a;
  const Status st = H->service.GetAction(has_time != 0, time, averaging_duration, nominal_action != 0, &a);
  if (st.ok() && !a.empty()) std::memcpy(action, a.data(), a.size() * 8);
  return Done(H, st);
}

int mjpc_agent_service_planner_step(void* h) { return Done(H, H->service.PlannerStep()); }
int mjpc_agent_service_step(void* h, int use_previous_policy) { return Done(H, H->service.Step(use_previous_policy != 0)); }
int mjpc_agent_service_reset(void* h) { return Done(H, H->service.Reset()); }

int mjpc_agent_service_set_task_parameter(void* h, const char* name, int is_selection, double numeric, const char* selection) {
  mjpc::agent_grpc::TaskParameter p;
  p.is_selection = is_selection != 0;
  p.numeric = numeric;
  p.selection = selection ? selection : "";
  return Done(H, H->service.SetTaskParameters({{name, p}}));
}
// refreshes the parameter list and returns its length through *count; then read entries with ..._task_parameter_at
int mjpc_agent_service_get_task_parameters(void* h, int* count) {
  const Status st = H->service.GetTaskParameters(&H->params);
  *count = (int)H->params.size();
  return Done(H, st);
}
int mjpc_agent_service_task_parameter_at(void* h, int index, char* name, int name_cap, int* is_selection, double* numeric,
                                         char* selection, int selection_cap) {
  if (index < 0 || index >= (int)H->params.size()) return Done(H, {mjpc::agent_grpc::kInvalidArgument, "parameter index out of range"});
  const auto& [n, p] = H->params[index];
  CopyStr(name, name_cap, n);
  *is_selection = p.is_selection;
  *numeric = p.numeric;
  CopyStr(selection, selection_cap, p.selection);
  return 0;
}

int mjpc_agent_service_reset_cost_weights(void* h) { return Done(H, H->service.SetCostWeights(true, {})); }
int mjpc_agent_service_set_cost_weight(void* h, const char* name, double weight) {
  return Done(H, H->service.SetCostWeights(false, {{name, weight}}));
}
// evaluates residuals and cost terms at the service's state; *count = number of cost terms; read with ..._cost_term_at
int mjpc_agent_service_get_cost_terms(void* h, int* count) {
  const Status st = H->service.GetCostTerms(&H->terms);
  *count = (int)H->terms.size();
  return Done(H, st);
}
int mjpc_agent_service_cost_term_at(void* h, int index, char* name, int name_cap, double* value, double* weight, double* residual,
                                    int residual_cap, int* dim) {
  if (index < 0 || index >= (int)H->terms.size()) return Done(H, {mjpc::agent_grpc::kInvalidArgument, "cost term index out of range"});
  const auto& t = H->terms[index];
  CopyStr(name, name_cap, t.name);
  *value = t.value;
  *weight = t.weight;
  *dim = (int)t.residual.size();
  for (int i = 0; i < *dim && i < residual_cap; i++) residual[i] = t.residual[i];
  return 0;
}

int mjpc_agent_service_set_mode(void* h, const char* mode) { return Done(H, H->service.SetMode(mode)); }
int mjpc_agent_service_get_mode(void* h, char* mode, int cap) {
  std::string m;
  const Status st = H->service.GetMode(&m);
  CopyStr(mode, cap, m);
  return Done(H, st);
}
// '|'-separated, as the model's task_transition text holds them
int mjpc_agent_service_get_all_modes(void* h, char* modes, int cap) {
  std::vector<std::string> v;
  const Status st = H->service.GetAllModes(&v);
  std::string joined;
  for (size_t i = 0; i < v.size(); i++) joined += (i ? "|" : "") + v[i];
  CopyStr(modes, cap, joined);
  return Done(H, st);
}

// states: steps x (nq + nv + na), actions: (steps - 1) x nu, times: steps
int mjpc_agent_service_best_trajectory(void* h, double* states, int states_cap, double* actions, int actions_cap, double* times,
                                       int times_cap, int* steps) {
  std::vector<double> s, a, t;
  const Status st = H->service.GetBestTrajectory(&s, &a, &t, steps);
  if (st.ok()) {
    if ((int)s.size() > states_cap || (int)a.size() > actions_cap || (int)t.size() > times_cap)
      return Done(H, {mjpc::agent_grpc::kInternal, "trajectory buffers too small"});
    std::memcpy(states, s.data(), s.size() * 8);
    std::memcpy(actions, a.data(), a.size() * 8);
    std::memcpy(times, t.data(), t.size() * 8);
  }
  return Done(H, st);
}

// SetAnything, decomposed by the front end into its parts in the reference's order: state (no Transition), cost weights,
// mode, then mocap poses; these two entries cover the parts that differ from the plain setters
int mjpc_agent_service_set_anything_state(void* h, int has_time, double time, const double* qpos, int nqpos, const double* qvel,
                                          int nqvel, const double* act, int nact, const double* mocap_pos, int nmocap_pos,
                                          const double* mocap_quat, int nmocap_quat, const double* userdata, int nuserdata) {
  const auto s = MakeState(has_time, time, qpos, nqpos, qvel, nqvel, act, nact, mocap_pos, nmocap_pos, mocap_quat, nmocap_quat,
                           userdata, nuserdata);
  return Done(H, H->service.SetAnything(&s, {}, "", {}));
}
int mjpc_agent_service_set_mocap(void* h, const char* body, const double* pos, int npos, const double* quat, int nquat) {
  mjpc::agent_grpc::Pose p;
  if (npos > 0) p.pos.assign(pos, pos + npos);
  if (nquat > 0) p.quat.assign(quat, quat + nquat);
  return Done(H, H->service.SetAnything(nullptr, {}, "", {{body, p}}));
}

}  // extern "C"
