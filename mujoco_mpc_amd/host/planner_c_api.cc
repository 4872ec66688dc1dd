// planner_c_api.cc -- a thin extern "C" view of the C++ GPU planners (mjpc::GpuSamplingPlanner,
// mjpc::GpuCrossEntropyPlanner, mjpc::GpuILQGPlanner) for non-C++ drivers: bench.py and the pytest suite drive the
// C++ planners through it with ctypes; the planner logic itself stays in C++.
#include <cstring>
#include <memory>
#include <string>

#include "mjpc/planners/gpu_cross_entropy/planner.h"
#include "mjpc/planners/gpu_ilqg/planner.h"
#include "mjpc/planners/gpu_sampling/planner.h"
#include "mjpc/planners/gpu_robust/robust_planner.h"
#include "mjpc/planners/gpu_sample_gradient/planner.h"
#include "mjpc/tasks/tasks.h"
#include "mjpc/utilities.h"
#include "model_io.h"

namespace {
struct Handle {
  std::unique_ptr<mjpc::ModelStorage> storage;
  std::shared_ptr<mjpc::Task> task;
  std::unique_ptr<mjpc::Planner> planner;
  mjpc::GpuSamplingPlanner* ps = nullptr;       // exactly one of these three is set
  mjpc::GpuCrossEntropyPlanner* ce = nullptr;
  mjpc::GpuILQGPlanner* ilqg = nullptr;
  mjpc::GpuRobustPlanner* robust = nullptr;
  mjpc::GpuSampleGradientPlanner* sg = nullptr;     // wraps a GpuSamplingPlanner: `ps` then points at its delegate
  mjpc::State state;
  mjpc::ThreadPool pool{1};
  std::string error;
  mjpc::gpu::Context* context() { return ps ? ps->context() : ce ? ce->context() : sg ? sg->context() : ilqg->context(); }
};
thread_local std::string g_error;
}  // namespace

extern "C" {

const char* mjpc_planner_last_error(void* h) { return h ? static_cast<Handle*>(h)->error.c_str() : g_error.c_str(); }

// kind: "sampling" | "cross_entropy" | "ilqg". num_trajectory > 0 overrides the model's custom numeric.
void* mjpc_planner_create_kind(const char* kind, const char* blob_path, const char* task_name, int device, int precision,
                               unsigned long long seed, int num_trajectory) {
  try {
    auto h = std::make_unique<Handle>();
    h->storage = mjpc::ModelStorage::Load(blob_path);
    for (auto& t : mjpc::GetTasks())
      if (mjpc::SameTaskName(t->Name(), task_name)) h->task = t;
    if (!h->task) { g_error = std::string("unknown task ") + task_name; return nullptr; }
    h->task->Reset(h->storage->model());
    const std::string k = kind ? kind : "sampling";
    if (k == "sampling") {
      h->ps = new mjpc::GpuSamplingPlanner(device, precision, seed);
      h->planner.reset(h->ps);
    } else if (k == "cross_entropy") {
      h->ce = new mjpc::GpuCrossEntropyPlanner(device, precision, seed);
      h->planner.reset(h->ce);
    } else if (k == "ilqg") {
      h->ilqg = new mjpc::GpuILQGPlanner(device, precision);
      h->planner.reset(h->ilqg);
    } else if (k == "sample_gradient") {
      h->sg = new mjpc::GpuSampleGradientPlanner(device, precision, seed);
      h->planner.reset(h->sg);
    } else if (k == "robust") {
      h->robust = new mjpc::GpuRobustPlanner(std::make_unique<mjpc::GpuSamplingPlanner>(device, precision, seed), device, precision, seed);
      h->planner.reset(h->robust);
      h->ps = h->robust->delegate();
    } else {
      g_error = "unknown planner kind " + k;
      return nullptr;
    }
    h->planner->Initialize(h->storage->model(), *h->task);
    if (num_trajectory > 0) {
      if (h->ps) h->ps->num_trajectory_ = num_trajectory;
      if (h->sg) h->sg->num_trajectory_ = num_trajectory;
      if (h->ce) { h->ce->num_trajectory_ = num_trajectory; h->ce->n_elite_ = std::max(num_trajectory / 10, 2); }
      if (h->ilqg) h->ilqg->num_rollouts_gui_ = h->ilqg->num_trajectory_ = num_trajectory;
    }
    h->planner->Allocate();
    h->state.Allocate(h->storage->model());
    h->state.Reset();
    return h.release();
  } catch (const std::exception& e) {
    g_error = e.what();
    return nullptr;
  }
}
void* mjpc_planner_create(const char* blob_path, const char* task_name, int device, int precision,
                          unsigned long long seed, int num_trajectory) {
  return mjpc_planner_create_kind("sampling", blob_path, task_name, device, precision, seed, num_trajectory);
}
void mjpc_planner_destroy(void* h) { delete static_cast<Handle*>(h); }

#define GUARD(h, ...)                                    \
  Handle* H = static_cast<Handle*>(h);                   \
  try { __VA_ARGS__; return 0; } catch (const std::exception& e) { H->error = e.what(); return -1; }

int mjpc_planner_set_sharding(void* h, int rank, int world, mjpc::GpuSamplingPlanner::ExchangeFn fn, void* user) {
  GUARD(h, { if (!H->ps) throw std::runtime_error("not a sampling planner"); H->ps->SetSharding(rank, world, fn, user); });
}
// the ranked interface of a sampling planner (the robust planner's delegate): top-k merge + sum callbacks, as the Cross-Entropy planner's
int mjpc_planner_set_sharding_ranked(void* h, int rank, int world, mjpc::GpuSamplingPlanner::MergeTopkFn merge,
                                     mjpc::GpuSamplingPlanner::SumFn sum, void* user) {
  GUARD(h, {
    if (!H->ps) throw std::runtime_error("not a sampling planner");
    H->ps->SetSharding(rank, world, nullptr, nullptr);
    H->ps->SetRankedSharding(merge, sum, user);
  });
}
int mjpc_planner_set_sharding_ce(void* h, int rank, int world, mjpc::GpuCrossEntropyPlanner::MergeTopkFn merge,
                                 mjpc::GpuCrossEntropyPlanner::SumFn sum, void* user) {
  GUARD(h, { if (!H->ce) throw std::runtime_error("not a cross-entropy planner"); H->ce->SetSharding(rank, world, merge, sum, user); });
}
// RCCL inside libmjpcx.so (include/mjpcx.h, mjpcx_comm_*): rank 0 makes the id, every rank joins with it; the planner then shards
// its candidates by rank and exchanges through the library (no transport callback)
int mjpc_comm_unique_id(void* id_out) { return mjpcx_comm_unique_id(id_out); }
int mjpc_planner_comm_init(void* h, const void* unique_id, int rank, int world) {
  GUARD(h, {
    mjpc::gpu::Context* ctx = H->ps ? H->ps->context() : (H->ce ? H->ce->context() : nullptr);
    if (!ctx) throw std::runtime_error("this planner kind is not sharded (replicas only)");
    ctx->Check(mjpcx_comm_init(ctx->handle(), unique_id, rank, world));
    if (H->ps) H->ps->SetSharding(rank, world, nullptr, nullptr);
    else H->ce->SetSharding(rank, world, nullptr, nullptr, nullptr);
  });
}
int mjpc_planner_comm_barrier(void* h) {
  GUARD(h, {
    mjpc::gpu::Context* ctx = H->ps ? H->ps->context() : (H->ce ? H->ce->context() : nullptr);
    if (!ctx) throw std::runtime_error("this planner kind has no communicator");
    ctx->Check(mjpcx_comm_barrier(ctx->handle()));
  });
}
int mjpc_planner_reset(void* h, int horizon) { GUARD(h, H->planner->Reset(horizon)); }
int mjpc_planner_set_state(void* h, const double* qpos, const double* qvel, const double* mocap_pos,
                           const double* mocap_quat, double time) {
  GUARD(h, {
    const mjModel* m = H->storage->model();
    std::vector<double> mp(3 * (size_t)m->nmocap), mq(4 * (size_t)m->nmocap);
    for (int b = 0; b < m->nbody; b++)
      if (m->body_mocapid[b] >= 0) {
        mju_copy(mp.data() + 3 * m->body_mocapid[b], m->body_pos + 3 * b, 3);
        mju_copy(mq.data() + 4 * m->body_mocapid[b], m->body_quat + 4 * b, 4);
      }
    H->state.Set(m, qpos, qvel, nullptr, mocap_pos ? mocap_pos : mp.data(), mocap_quat ? mocap_quat : mq.data(), nullptr, time);
    H->planner->SetState(H->state);
  });
}
// Task::Transition at `time` (mode: < 0 keeps the task's current mode); parameters/weights the transition changed
// reach the device with the next plan
int mjpc_planner_task_transition(void* h, double time, int mode) {
  GUARD(h, {
    mjData d{};
    d.time = time;
    if (mode >= 0) H->task->mode = mode;
    H->task->Transition(H->storage->model(), &d);
  });
}
// Task::Transition with simulation state: tasks that edit mjData (humanoid::Tracking: mocap_pos, and qpos / qvel on a motion
// switch) write into the caller's arrays (any of them may be NULL)
int mjpc_planner_task_transition_state(void* h, double time, int mode, double* qpos, double* qvel, double* mocap_pos) {
  GUARD(h, {
    mjData d{};
    d.time = time;
    d.qpos = qpos; d.qvel = qvel; d.mocap_pos = mocap_pos;
    if (mode >= 0) H->task->mode = mode;
    H->task->Transition(H->storage->model(), &d);
  });
}
// the host-side normal generator (utilities.h HostGaussianPair), exported for the tests that pin it to the device stream
void mjpc_host_gaussian_pair(unsigned long long seed, unsigned cand, unsigned pair, unsigned iter, double* z) {
  mjpc::HostGaussianPair(seed, cand, pair, iter, z);
}
// SampleGradientPlanner: number of gradient candidates / filter; its last gradient estimate and winner type
int mjpc_planner_sample_gradient_config(void* h, int num_gradient, double gradient_filter) {
  GUARD(h, {
    if (!H->sg) throw std::runtime_error("not a sample-gradient planner");
    if (num_gradient >= 0) H->sg->num_gradient_ = num_gradient;
    if (gradient_filter >= 0) H->sg->gradient_filter_ = gradient_filter;
  });
}
int mjpc_planner_sample_gradient_result(void* h, int* winner_type, double* gradient, int n, double* returns, int nret) {
  GUARD(h, {
    if (!H->sg) throw std::runtime_error("not a sample-gradient planner");
    *winner_type = H->sg->winner_type_;
    for (int i = 0; i < n && i < (int)H->sg->gradient.size(); i++) gradient[i] = H->sg->gradient[i];
    for (int i = 0; i < nret && i < (int)H->sg->returns.size(); i++) returns[i] = H->sg->returns[i];
  });
}
// RobustPlanner knobs and the outcome of its last OptimizePolicy (scores: ncandidates mean perturbed returns)
int mjpc_planner_robust_config(void* h, int ncandidates, int nrepetitions, double xfrc_std, double xfrc_rate) {
  GUARD(h, {
    if (!H->robust) throw std::runtime_error("not a robust planner");
    if (ncandidates > 0) H->robust->ncandidates_ = ncandidates;
    if (nrepetitions > 0) H->robust->nrepetitions_ = nrepetitions;
    if (xfrc_std >= 0) H->robust->xfrc_std_ = xfrc_std;
    if (xfrc_rate > 0) H->robust->xfrc_rate_ = xfrc_rate;
  });
}
int mjpc_planner_robust_result(void* h, int* best_candidate, double* scores, int capacity) {
  GUARD(h, {
    if (!H->robust) throw std::runtime_error("not a robust planner");
    *best_candidate = H->robust->best_candidate;
    for (int i = 0; i < capacity && i < (int)H->robust->perturbed_score.size(); i++) scores[i] = H->robust->perturbed_score[i];
  });
}
int mjpc_planner_task_set_parameter(void* h, int index, double value) {
  GUARD(h, { if (index < 0 || index >= (int)H->task->parameters.size()) throw std::runtime_error("parameter index"); // callers pass numbers; "residual_select_*" slots keep an integer's bits (utilities.h)
    const bool sel = index < (int)H->task->parameter_is_selection.size() && H->task->parameter_is_selection[index];
    H->task->parameters[index] = sel ? mjpc::ReinterpretAsDouble((std::int64_t)value) : value; });
}
int mjpc_planner_optimize(void* h, int horizon) { GUARD(h, H->planner->OptimizePolicy(horizon, H->pool)); }
int mjpc_planner_nominal(void* h, int horizon) { GUARD(h, H->planner->NominalTrajectory(horizon, H->pool)); }
// state may be NULL (open-loop action)
int mjpc_planner_action(void* h, double time, int use_previous, double* action) {
  GUARD(h, H->planner->ActionFromPolicy(action, nullptr, time, use_previous != 0));
}
int mjpc_planner_action_state(void* h, const double* state, double time, int use_previous, double* action) {
  GUARD(h, H->planner->ActionFromPolicy(action, state, time, use_previous != 0));
}
int mjpc_planner_num_parameters(void* h) { return static_cast<Handle*>(h)->planner->NumParameters(); }
int mjpc_planner_num_spline_points(void* h) {
  Handle* H = static_cast<Handle*>(h);
  return H->ps ? H->ps->policy.num_spline_points : H->ce ? H->ce->policy.num_spline_points : H->sg ? H->sg->policy.num_spline_points : 0;
}
int mjpc_planner_winner(void* h) {
  Handle* H = static_cast<Handle*>(h);
  if (H->sg) return H->sg->winner;
  return H->ps ? H->ps->winner : H->ilqg ? H->ilqg->winner : (H->ce->trajectory_order.empty() ? -1 : H->ce->trajectory_order[0]);
}
double mjpc_planner_improvement(void* h) {
  Handle* H = static_cast<Handle*>(h);
  if (H->sg) return H->sg->improvement;
  return H->ps ? H->ps->improvement : H->ce ? H->ce->improvement : H->ilqg->improvement;
}
double mjpc_planner_best_score(void* h) {
  Handle* H = static_cast<Handle*>(h);
  if (H->sg) return H->sg->returns.empty() ? 0.0 : H->sg->returns[H->sg->winner];
  return H->ps ? H->ps->CandidateScore(0) : 0.0;
}
// policy spline nodes (sampling / cross-entropy): returns the node count; copies up to `cap` nodes
int mjpc_planner_policy(void* h, double* times, double* values, int cap) {
  Handle* H = static_cast<Handle*>(h);
  if (H->ilqg) return 0;
  const auto& plan = H->ps ? H->ps->policy.plan : H->sg ? H->sg->policy.plan : H->ce->policy.plan;
  const int n = (int)plan.Size(), nu = H->storage->model()->nu;
  for (int k = 0; k < n && k < cap; k++) {
    times[k] = plan.times()[k];
    std::memcpy(values + (size_t)k * nu, plan.values().data() + (size_t)k * nu, sizeof(double) * nu);
  }
  return n;
}
// cross-entropy: fitted variance of the P*nu spline parameters, elite indices (global), n_elite
int mjpc_planner_ce_variance(void* h, double* variance, int n) {
  Handle* H = static_cast<Handle*>(h);
  if (!H->ce) return -1;
  for (int i = 0; i < n && i < (int)H->ce->variance.size(); i++) variance[i] = H->ce->variance[i];
  return 0;
}
int mjpc_planner_ce_elites(void* h, int* index, int cap) {
  Handle* H = static_cast<Handle*>(h);
  if (!H->ce) return -1;
  const int n = (int)H->ce->trajectory_order.size();
  for (int i = 0; i < n && i < cap; i++) index[i] = H->ce->trajectory_order[i];
  return n;
}
int mjpc_planner_ce_set(void* h, int n_elite, double std_initial, double std_min, double explore_fraction, int interpolation) {
  Handle* H = static_cast<Handle*>(h);
  if (!H->ce) return -1;
  if (n_elite > 0) H->ce->n_elite_ = n_elite;
  if (std_initial >= 0) H->ce->std_initial_ = std_initial;
  if (std_min >= 0) H->ce->std_min_ = std_min;
  if (explore_fraction >= 0) H->ce->explore_fraction_ = explore_fraction;
  if (interpolation >= 0) H->ce->interpolation_ = (mjpc::spline::SplineInterpolation)interpolation;
  return 0;
}
// iLQG: scalars out[0..9] = {regularization, dV0, dV1, action_step, feedback_scaling, improvement, expected, surprise,
// winner, policy.trajectory.total_return}
int mjpc_planner_ilqg_info(void* h, double* out) {
  Handle* H = static_cast<Handle*>(h);
  if (!H->ilqg) return -1;
  auto* p = H->ilqg;
  const double v[10] = {p->regularization, p->dV[0], p->dV[1], p->action_step, p->feedback_scaling, p->improvement,
                        p->expected, p->surprise, (double)p->winner, p->policy.trajectory.total_return};
  std::memcpy(out, v, sizeof v);
  return 0;
}
int mjpc_planner_ilqg_set(void* h, int regularization_type, int action_limits, int fd_mode, int derivative_skip,
                          int representation) {
  Handle* H = static_cast<Handle*>(h);
  if (!H->ilqg) return -1;
  auto* p = H->ilqg;
  if (regularization_type >= 0) p->settings.regularization_type = regularization_type;
  if (action_limits >= 0) p->settings.action_limits = action_limits;
  if (fd_mode >= 0) p->settings.fd_mode = fd_mode;
  if (derivative_skip >= 0) p->derivative_skip_ = derivative_skip;
  if (representation >= 0) p->policy.representation = p->previous_policy.representation = p->candidate_policy0.representation = representation;
  return 0;
}
// iLQG policy arrays for T steps: actions (T x nu), states (T x dim_state), times (T), gains (T x nu x ndx)
int mjpc_planner_ilqg_policy(void* h, int T, double* times, double* states, double* actions, double* gains) {
  Handle* H = static_cast<Handle*>(h);
  if (!H->ilqg) return -1;
  const auto& p = H->ilqg->policy;
  const int nu = H->ilqg->dim_action, ds = H->ilqg->dim_state, ndx = H->ilqg->dim_state_derivative;
  if (times) std::memcpy(times, p.trajectory.times.data(), sizeof(double) * T);
  if (states) std::memcpy(states, p.trajectory.states.data(), sizeof(double) * T * ds);
  if (actions) std::memcpy(actions, p.trajectory.actions.data(), sizeof(double) * T * nu);
  if (gains) std::memcpy(gains, p.feedback_gain.data(), sizeof(double) * T * nu * ndx);
  return p.trajectory.horizon;
}
// BestTrajectory(): horizon or -1; any output may be NULL
int mjpc_planner_best_trajectory(void* h, double* states, double* actions, double* times, double* costs, double* total_return) {
  Handle* H = static_cast<Handle*>(h);
  try {
    const mjpc::Trajectory* tr = H->planner->BestTrajectory();
    if (!tr) return -1;
    const int T = tr->horizon;
    if (states) std::memcpy(states, tr->states.data(), sizeof(double) * T * tr->dim_state);
    if (actions) std::memcpy(actions, tr->actions.data(), sizeof(double) * T * tr->dim_action);
    if (times) std::memcpy(times, tr->times.data(), sizeof(double) * T);
    if (costs) std::memcpy(costs, tr->costs.data(), sizeof(double) * T);
    if (total_return) *total_return = tr->total_return;
    return T;
  } catch (const std::exception& e) {
    H->error = e.what();
    return -1;
  }
}
// the underlying mjpcx context (timing, algorithmic bytes, kernel name)
mjpcx_ctx* mjpc_planner_ctx(void* h) { return static_cast<Handle*>(h)->context()->handle(); }

}  // extern "C"
