#include "trajectory.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <utility>

#include "gpu/context.h"

namespace mjpc {

namespace {
constexpr double kMaxReturnValue = 1.0e6;  // trajectory.cc:29

// One device context per (model, task) the host-policy rollouts are asked for, created on first use. The key is the pair of
// addresses; a model deleted and another loaded at the same address (or a task re-created there, or the model edited in place) is
// told apart by a fingerprint of every numeric array the context is built from, and gets a fresh context. The reference calls
// Trajectory::Rollout from ThreadPool workers concurrently: a slot's context is shared (shared_ptr: a replacement or
// ReleaseRolloutContexts() cannot destroy it under a running rollout) and a rollout holds the slot's own mutex from set_state to the
// last fetch, so two threads rolling out the same (model, task) take turns on its one stream instead of interleaving. The table is
// never destroyed (a static destructor would call into the HIP runtime after it may have shut down); ReleaseRolloutContexts() drops
// the contexts while the runtime is alive.
struct RolloutSlot {
  std::uint64_t fingerprint = 0;
  std::shared_ptr<gpu::Context> ctx;
  std::shared_ptr<std::mutex> busy;
};
struct RolloutLease {
  std::shared_ptr<gpu::Context> ctx;
  std::shared_ptr<std::mutex> busy;
};
using RolloutTable = std::map<std::pair<const mjModel*, const Task*>, RolloutSlot>;
std::mutex& RolloutMutex() { static std::mutex* m = new std::mutex; return *m; }
RolloutTable& RolloutContexts() { static RolloutTable* t = new RolloutTable; return *t; }

std::uint64_t Fingerprint(const mjModel* m, const Task* task) {
  std::uint64_t h = 1469598103934665603ull;  // FNV-1a, eight bytes at a time, over the dimensions, the options and every numeric model array
  auto mix = [&h](const void* p, size_t bytes) {
    if (!p) return;
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { std::uint64_t w; std::memcpy(&w, b + i, 8); h ^= w; h *= 1099511628211ull; }
    for (; i < bytes; i++) { h ^= b[i]; h *= 1099511628211ull; }
  };
  const int dims[] = {m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->nsite, m->nmocap, m->nsensor, m->ntendon, m->nwrap, m->nexclude,
                      task->num_residual, task->num_term, task->num_trace};
  mix(dims, sizeof dims);
  const mjtNum opt[] = {m->opt.timestep, m->opt.gravity[0], m->opt.gravity[1], m->opt.gravity[2], m->opt.tolerance, m->opt.impratio,
                        (mjtNum)m->opt.integrator, (mjtNum)m->opt.iterations, (mjtNum)m->opt.cone, (mjtNum)m->opt.disableflags};
  mix(opt, sizeof opt);
  const size_t N = sizeof(mjtNum), nb = (size_t)m->nbody, nj = (size_t)m->njnt, nv = (size_t)m->nv, ng = (size_t)m->ngeom, nu = (size_t)m->nu;
  mix(m->body_mass, N * nb); mix(m->body_pos, N * 3 * nb); mix(m->body_quat, N * 4 * nb); mix(m->body_ipos, N * 3 * nb);
  mix(m->body_iquat, N * 4 * nb); mix(m->body_inertia, N * 3 * nb);
  mix(m->jnt_pos, N * 3 * nj); mix(m->jnt_axis, N * 3 * nj); mix(m->jnt_stiffness, N * nj); mix(m->jnt_range, N * 2 * nj);
  mix(m->jnt_margin, N * nj); mix(m->jnt_solref, N * 2 * nj); mix(m->jnt_solimp, N * 5 * nj); mix(m->jnt_limited, nj);
  mix(m->dof_armature, N * nv); mix(m->dof_damping, N * nv); mix(m->dof_frictionloss, N * nv); mix(m->dof_solref, N * 2 * nv);
  mix(m->dof_solimp, N * 5 * nv);
  mix(m->geom_size, N * 3 * ng); mix(m->geom_pos, N * 3 * ng); mix(m->geom_quat, N * 4 * ng); mix(m->geom_friction, N * 3 * ng);
  mix(m->geom_solref, N * 2 * ng); mix(m->geom_solimp, N * 5 * ng); mix(m->geom_margin, N * ng); mix(m->geom_gap, N * ng);
  mix(m->geom_solmix, N * ng); mix(m->geom_type, sizeof(int) * ng); mix(m->geom_contype, sizeof(int) * ng);
  mix(m->geom_conaffinity, sizeof(int) * ng); mix(m->geom_condim, sizeof(int) * ng); mix(m->geom_priority, sizeof(int) * ng);
  mix(m->qpos0, N * (size_t)m->nq); mix(m->qpos_spring, N * (size_t)m->nq);
  mix(m->site_pos, N * 3 * (size_t)m->nsite); mix(m->site_quat, N * 4 * (size_t)m->nsite);
  mix(m->actuator_gear, N * 6 * nu); mix(m->actuator_gainprm, N * mjNGAIN * nu); mix(m->actuator_biasprm, N * mjNBIAS * nu);
  mix(m->actuator_ctrlrange, N * 2 * nu); mix(m->actuator_forcerange, N * 2 * nu); mix(m->actuator_ctrllimited, nu);
  mix(m->actuator_forcelimited, nu);
  mix(m->tendon_range, N * 2 * (size_t)m->ntendon); mix(m->wrap_prm, N * (size_t)m->nwrap);
  return h;
}

RolloutLease RolloutContext(const mjModel* model, const Task* task) {
  const std::uint64_t fp = Fingerprint(model, task);   // (hashed outside the table's lock: other (model, task) pairs are not held up)
  const std::lock_guard<std::mutex> lock(RolloutMutex());
  RolloutSlot& slot = RolloutContexts()[{model, task}];
  if (!slot.ctx || slot.fingerprint != fp) {
    slot.ctx = std::make_shared<gpu::Context>(model, *task, /*device=*/0, /*precision=*/64);
    slot.busy = std::make_shared<std::mutex>();
    slot.fingerprint = fp;
  }
  return {slot.ctx, slot.busy};
}
}  // namespace

void ReleaseRolloutContexts() {
  const std::lock_guard<std::mutex> lock(RolloutMutex());
  RolloutContexts().clear();
}

namespace {
// The body shared by Rollout and RolloutDiscrete: `policy(action, state, t)` with t the step index.
template <class Policy>
void HostPolicyRollout(Trajectory* tr, Policy policy, const Task* task, const mjModel* model, mjData* data, const double* state, double time,
                       const double* mocap, const double* userdata, int steps) {
  tr->failure = false;
  tr->horizon = steps;
  const int nx = tr->dim_state, nu = tr->dim_action, nr = tr->dim_residual, ntr = tr->dim_trace;
  const RolloutLease lease = RolloutContext(model, task);
  const std::lock_guard<std::mutex> turn(*lease.busy);  // one rollout at a time on this (model, task)'s context and stream
  gpu::Context* ctx = lease.ctx.get();
  ctx->SyncTask(*task);
  Trajectory one;  // the two rows a single mj_step produces
  one.Initialize(nx, nu, nr, ntr / 3, 2);
  one.Allocate(2);
  // trajectory.cc:118-132: mocap, userdata, state, time
  if (data) {
    if (mocap) { for (int i = 0; i < model->nmocap; i++) { mju_copy(data->mocap_pos + 3 * i, mocap + 7 * i, 3); mju_copy(data->mocap_quat + 4 * i, mocap + 7 * i + 3, 4); } }
    if (userdata) mju_copy(data->userdata, userdata, model->nuserdata);
    data->time = time;
  }
  mju_copy(tr->states.data(), state, nx);
  tr->times[0] = time;
  const double node_time = 0;  // a one-node, zero-order spline IS a constant control
  for (int t = 0; t < steps - 1; t++) {
    double* action = tr->actions.data() + (size_t)t * nu;
    const double* x = tr->states.data() + (size_t)t * nx;
    policy(action, x, t, tr->times[t]);                     // policy(DataAt(actions, t * nu), DataAt(states, t * nx), data->time)
    if (data) mju_copy(data->ctrl, action, nu);
    // mj_step + the sensor callback at x (+ mj_forward at the next state, unused except at the end) on the device
    ctx->Check(mjpcx_set_state(ctx->handle(), x, tr->times[t], mocap, userdata));
    ctx->Check(mjpcx_rollout_splines(ctx->handle(), 1, 2, 1, MJPCX_SPLINE_ZERO, &node_time, action));
    one.horizon = 2;
    ctx->FetchTrajectory(0, &one);
    mju_copy(action, one.actions.data(), nu);               // the device clamps the control as mj_step's ctrl clamp does not: see (2) below
    mju_copy(tr->residual.data() + (size_t)t * nr, one.residual.data(), nr);
    mju_copy(tr->trace.data() + (size_t)t * ntr, one.trace.data(), ntr);
    if (one.failure) {                                      // trajectory.cc:169-173
      tr->failure = true;
      tr->total_return = kMaxReturnValue;
      std::fprintf(stderr, "Rollout divergence at step %d\n", t);
      return;
    }
    mju_copy(tr->states.data() + (size_t)(t + 1) * nx, one.states.data() + nx, nx);
    tr->times[t + 1] = one.times[1];
    if (data) { data->time = one.times[1]; mju_copy(data->qpos, one.states.data() + nx, model->nq); mju_copy(data->qvel, one.states.data() + nx + model->nq, model->nv); }
    if (t == steps - 2) {
      // the last launch also ran mj_forward at the final state with the last control: trajectory.cc:194-206
      mju_copy(tr->residual.data() + (size_t)(steps - 1) * nr, one.residual.data() + nr, nr);
      mju_copy(tr->trace.data() + (size_t)(steps - 1) * ntr, one.trace.data() + ntr, ntr);
    }
  }
  if (steps > 1) {
    mju_copy(tr->actions.data() + (size_t)(steps - 1) * nu, tr->actions.data() + (size_t)(steps - 2) * nu, nu);  // trajectory.cc:190-192
  } else {
    // a single row: mj_forward at the given state with the controls the caller left in `data`
    std::vector<double> u(nu, 0.0);
    if (data) mju_copy(u.data(), data->ctrl, nu);
    ctx->Check(mjpcx_set_state(ctx->handle(), state, time, mocap, userdata));
    ctx->Check(mjpcx_rollout_splines(ctx->handle(), 1, 1, 1, MJPCX_SPLINE_ZERO, &node_time, u.data()));
    one.horizon = 1;
    ctx->FetchTrajectory(0, &one);
    mju_copy(tr->residual.data(), one.residual.data(), nr);
    mju_copy(tr->trace.data(), one.trace.data(), ntr);
  }
  tr->UpdateReturn(task);
}
}  // namespace

// Deviations from trajectory.cc:100-210, both confined to this compatibility path:
//  (1) every step is its own device launch, so the constraint solver starts each step from the unconstrained acceleration
//      instead of the previous step's solution (mj_step's warm start); the solutions agree to the solver tolerance (1e-8);
//  (2) actions[t] is stored after the device's clamp to actuator_ctrlrange, which every policy of this code base applies
//      itself (sampling/policy.cc:58, ilqg/policy.cc:160).
void Trajectory::Rollout(std::function<void(double* action, const double* state, double time)> policy, const Task* task, const mjModel* model,
                         mjData* data, const double* state, double time, const double* mocap, const double* userdata, int steps) {
  HostPolicyRollout(this, [&](double* a, const double* x, int, double t) { policy(a, x, t); }, task, model, data, state, time, mocap, userdata, steps);
}

void Trajectory::RolloutDiscrete(std::function<void(double* action, const double* state, int index)> policy, const Task* task,
                                 const mjModel* model, mjData* data, const double* state, double time, const double* mocap,
                                 const double* userdata, int steps) {
  HostPolicyRollout(this, [&](double* a, const double* x, int index, double) { policy(a, x, index); }, task, model, data, state, time, mocap, userdata, steps);
}

void Trajectory::NoisyRollout(std::function<void(double* action, const double* state, double time)> policy, const Task* task,
                              const mjModel* model, mjData* data, const double* state, double time, const double* mocap,
                              const double* userdata, double xfrc_std, double xfrc_rate, int steps) {
  if (xfrc_std > 0) {
    // the Ornstein-Uhlenbeck force noise lives in the device's rollout kernels (one stream per candidate and step); a host
    // policy cannot be interleaved with it. Fatal configuration error, reported the way the reference reports them.
    std::fprintf(stderr, "Trajectory::NoisyRollout with xfrc_std > 0 and a host policy: use GpuRobustPlanner (mjpcx_rollout_splines_noisy)\n");
    std::abort();
  }
  (void)xfrc_rate;
  Rollout(std::move(policy), task, model, data, state, time, mocap, userdata, steps);
}

void Trajectory::Initialize(int dim_state_, int dim_action_, int dim_residual_, int num_trace, int horizon_) {
  horizon = horizon_;
  dim_state = dim_state_;
  dim_action = dim_action_;
  dim_residual = dim_residual_;
  dim_trace = 3 * num_trace;
  failure = false;
}

void Trajectory::Allocate(int T) {
  states.resize((size_t)dim_state * T);
  actions.resize((size_t)dim_action * T);
  costs.resize(T);
  residual.resize((size_t)dim_residual * T);
  times.resize(T);
  trace.resize((size_t)dim_trace * T);
}

void Trajectory::Reset(int T, const double* initial_repeated_action) {
  std::fill_n(states.begin(), (size_t)dim_state * T, 0.0);
  for (int t = 0; t < T; t++)
    for (int k = 0; k < dim_action; k++)
      actions[(size_t)t * dim_action + k] = initial_repeated_action ? initial_repeated_action[k] : 0.0;
  std::fill_n(times.begin(), T, 0.0);
  std::fill_n(costs.begin(), T, 0.0);
  std::fill_n(residual.begin(), (size_t)dim_residual * T, 0.0);
  std::fill_n(trace.begin(), (size_t)dim_trace * T, 0.0);
  total_return = 0.0;
  failure = false;
}

void Trajectory::UpdateReturn(const Task* task) {
  total_return = 0;
  for (int t = 0; t < horizon; t++) {
    costs[t] = task->CostValue(residual.data() + (size_t)t * task->num_residual);
    total_return += costs[t];
  }
  total_return /= std::max(horizon, 1);
}

}  // namespace mjpc
