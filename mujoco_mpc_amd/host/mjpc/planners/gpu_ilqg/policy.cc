#include "policy.h"

#include <algorithm>
#include <stdexcept>

#include "../../utilities.h"

namespace mjpc {

// ilqg/policy.cc:29-60
void iLQGPolicy::Allocate(const mjModel* m, const Task& task, int horizon) {
  model = m;
  const int ds = m->nq + m->nv + m->na, ndx = 2 * m->nv + m->na;
  trajectory.Initialize(ds, m->nu, task.num_residual, task.num_trace, horizon);
  trajectory.Allocate(horizon);
  feedback_gain.assign((size_t)horizon * m->nu * ndx, 0.0);
  action_improvement.assign((size_t)horizon * m->nu, 0.0);
  feedback_gain_scratch.assign((size_t)m->nu * ndx, 0.0);
  state_scratch.assign(ndx, 0.0);
  action_scratch.assign(m->nu, 0.0);
  state_interp.assign(ds, 0.0);
  representation = GetNumberOrDefault((int)kLinear, m, "ilqg_representation");
}

// ilqg/policy.cc:63-79
void iLQGPolicy::Reset(int horizon, const double* initial_repeated_action) {
  trajectory.Reset(horizon, initial_repeated_action);
  const int ndx = 2 * model->nv + model->na;
  std::fill(feedback_gain.begin(), feedback_gain.begin() + (size_t)horizon * model->nu * ndx, 0.0);
  std::fill(action_improvement.begin(), action_improvement.begin() + (size_t)horizon * model->nu, 0.0);
  feedback_scaling = 1.0;
}

// ilqg/policy.cc:82-161
void iLQGPolicy::Action(double* action, const double* state, double time) const {
  const int nu = model->nu, ds = model->nq + model->nv + model->na, ndx = 2 * model->nv + model->na;
  const int H = trajectory.horizon;
  int bounds[2];
  FindInterval(bounds, trajectory.times.data(), time, H);
  const bool zero = bounds[0] == bounds[1] || representation == kZeroOrder;
  auto interp = [&](double* out, const double* ys, int dim, int length) {
    if (zero) ZeroInterpolation(out, time, trajectory.times.data(), ys, dim, length);
    else if (representation == kCubic) CubicInterpolation(out, time, trajectory.times.data(), ys, dim, length);
    else LinearInterpolation(out, time, trajectory.times.data(), ys, dim, length);
  };
  interp(action, trajectory.actions.data(), nu, H - 1);
  if (state) {
    interp(state_interp.data(), trajectory.states.data(), ds, H);
    if (model->nq != model->nv) NormalizeStateQuaternions(model, state_interp.data());
    interp(feedback_gain_scratch.data(), feedback_gain.data(), nu * ndx, H - 1);
    StateDiff(model, state_scratch.data(), state_interp.data(), state, 1.0);
    for (int i = 0; i < nu; i++) {
      double s = 0;
      for (int j = 0; j < ndx; j++) s += feedback_gain_scratch[(size_t)i * ndx + j] * state_scratch[j];
      action[i] += feedback_scaling * s;
    }
  }
  Clamp(action, model->actuator_ctrlrange, nu);
}

// ilqg/policy.cc:164-175
void iLQGPolicy::CopyFrom(const iLQGPolicy& policy, int horizon) {
  trajectory = policy.trajectory;
  const int ndx = 2 * model->nv + model->na;
  std::copy_n(policy.feedback_gain.begin(), (size_t)horizon * model->nu * ndx, feedback_gain.begin());
  std::copy_n(policy.action_improvement.begin(), (size_t)horizon * model->nu, action_improvement.begin());
}

}  // namespace mjpc
