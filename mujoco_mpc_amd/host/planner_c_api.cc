// planner_c_api.cc -- a thin extern "C" view of mjpc::GpuSamplingPlanner for non-C++ drivers (bench.py and the
// pytest suite drive the C++ planner through it with ctypes; the planner logic itself stays in C++).
#include <cstring>
#include <memory>
#include <string>

#include "mjpc/planners/gpu_sampling/planner.h"
#include "mjpc/tasks/tasks.h"
#include "model_io.h"

namespace {
struct Handle {
  std::unique_ptr<mjpc::ModelStorage> storage;
  std::shared_ptr<mjpc::Task> task;
  std::unique_ptr<mjpc::GpuSamplingPlanner> planner;
  mjpc::State state;
  mjpc::ThreadPool pool{1};
  std::string error;
};
thread_local std::string g_error;
}  // namespace

extern "C" {

const char* mjpc_planner_last_error(void* h) { return h ? static_cast<Handle*>(h)->error.c_str() : g_error.c_str(); }

void* mjpc_planner_create(const char* blob_path, const char* task_name, int device, int precision,
                          unsigned long long seed, int num_trajectory) {
  try {
    auto h = std::make_unique<Handle>();
    h->storage = mjpc::ModelStorage::Load(blob_path);
    for (auto& t : mjpc::GetTasks())
      if (t->Name() == task_name) h->task = t;
    if (!h->task) { g_error = std::string("unknown task ") + task_name; return nullptr; }
    h->task->Reset(h->storage->model());
    h->planner = std::make_unique<mjpc::GpuSamplingPlanner>(device, precision, seed);
    h->planner->Initialize(h->storage->model(), *h->task);
    if (num_trajectory > 0) h->planner->num_trajectory_ = num_trajectory;
    h->planner->Allocate();
    h->state.Allocate(h->storage->model());
    h->state.Reset();
    return h.release();
  } catch (const std::exception& e) {
    g_error = e.what();
    return nullptr;
  }
}
void mjpc_planner_destroy(void* h) { delete static_cast<Handle*>(h); }

#define GUARD(h, ...)                                    \
  Handle* H = static_cast<Handle*>(h);                   \
  try { __VA_ARGS__; return 0; } catch (const std::exception& e) { H->error = e.what(); return -1; }

int mjpc_planner_set_sharding(void* h, int rank, int world, mjpc::GpuSamplingPlanner::ExchangeFn fn, void* user) {
  GUARD(h, H->planner->SetSharding(rank, world, fn, user));
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
int mjpc_planner_optimize(void* h, int horizon) { GUARD(h, H->planner->OptimizePolicy(horizon, H->pool)); }
int mjpc_planner_action(void* h, double time, int use_previous, double* action) {
  GUARD(h, H->planner->ActionFromPolicy(action, nullptr, time, use_previous != 0));
}
int mjpc_planner_num_spline_points(void* h) { return static_cast<Handle*>(h)->planner->policy.num_spline_points; }
int mjpc_planner_winner(void* h) { return static_cast<Handle*>(h)->planner->winner; }
double mjpc_planner_improvement(void* h) { return static_cast<Handle*>(h)->planner->improvement; }
double mjpc_planner_best_score(void* h) { return static_cast<Handle*>(h)->planner->CandidateScore(0); }
// policy spline nodes: returns the node count; copies up to `cap` nodes
int mjpc_planner_policy(void* h, double* times, double* values, int cap) {
  Handle* H = static_cast<Handle*>(h);
  const auto& plan = H->planner->policy.plan;
  const int n = (int)plan.Size(), nu = H->storage->model()->nu;
  for (int k = 0; k < n && k < cap; k++) {
    times[k] = plan.times()[k];
    std::memcpy(values + (size_t)k * nu, plan.values().data() + (size_t)k * nu, sizeof(double) * nu);
  }
  return n;
}
// the underlying mjpcx context (timing, algorithmic bytes, kernel name)
mjpcx_ctx* mjpc_planner_ctx(void* h) { return static_cast<Handle*>(h)->planner->context()->handle(); }

}  // extern "C"
