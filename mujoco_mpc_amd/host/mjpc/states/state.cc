#include "state.h"

#include <algorithm>
#include <mutex>

namespace mjpc {

void State::Resize(const mjModel* m) {
  state_.resize(m->nq + m->nv + m->na);
  mocap_.resize(7 * (size_t)m->nmocap);
  userdata_.resize(m->nuserdata);
}
void State::Allocate(const mjModel* model) {
  const std::unique_lock<std::shared_mutex> lock(mtx_);
  Resize(model);
}
void State::Reset() {
  const std::unique_lock<std::shared_mutex> lock(mtx_);
  std::fill(state_.begin(), state_.end(), 0.0);
  std::fill(mocap_.begin(), mocap_.end(), 0.0);
  std::fill(userdata_.begin(), userdata_.end(), 0.0);
  time_ = 0.0;
}
void State::Set(const mjModel* model, const mjData* data) {
  if (!model || !data) return;
  Set(model, data->qpos, data->qvel, data->act, data->mocap_pos, data->mocap_quat, data->userdata, data->time);
}
void State::Set(const mjModel* model, const double* qpos, const double* qvel, const double* act,
                const double* mocap_pos, const double* mocap_quat, const double* userdata, double time) {
  const std::unique_lock<std::shared_mutex> lock(mtx_);
  Resize(model);
  PutPosition(model, qpos);
  PutVelocity(model, qvel);
  PutAct(model, act);
  PutMocap(model, mocap_pos, mocap_quat);
  PutUserData(model, userdata);
  time_ = time;
}
// the single-field setters are public (the reference's simulation thread uses them one by one) and take the lock themselves; Set()
// holds it once around the unlocked Put* helpers
void State::PutPosition(const mjModel* m, const double* qpos) { mju_copy(state_.data(), qpos, m->nq); }
void State::PutVelocity(const mjModel* m, const double* qvel) { mju_copy(state_.data() + m->nq, qvel, m->nv); }
void State::PutAct(const mjModel* m, const double* act) { if (m->na) mju_copy(state_.data() + m->nq + m->nv, act, m->na); }
void State::PutMocap(const mjModel* m, const double* mocap_pos, const double* mocap_quat) {
  for (int i = 0; i < m->nmocap; i++) {
    mju_copy(mocap_.data() + 7 * i, mocap_pos + 3 * i, 3);
    mju_copy(mocap_.data() + 7 * i + 3, mocap_quat + 4 * i, 4);
  }
}
void State::PutUserData(const mjModel* m, const double* userdata) { if (m->nuserdata) mju_copy(userdata_.data(), userdata, m->nuserdata); }
void State::SetPosition(const mjModel* m, const double* qpos) { const std::unique_lock<std::shared_mutex> lock(mtx_); PutPosition(m, qpos); }
void State::SetVelocity(const mjModel* m, const double* qvel) { const std::unique_lock<std::shared_mutex> lock(mtx_); PutVelocity(m, qvel); }
void State::SetAct(const mjModel* m, const double* act) { const std::unique_lock<std::shared_mutex> lock(mtx_); PutAct(m, act); }
void State::SetMocap(const mjModel* m, const double* mocap_pos, const double* mocap_quat) {
  const std::unique_lock<std::shared_mutex> lock(mtx_);
  PutMocap(m, mocap_pos, mocap_quat);
}
void State::SetUserData(const mjModel* m, const double* userdata) { const std::unique_lock<std::shared_mutex> lock(mtx_); PutUserData(m, userdata); }
void State::SetTime(const mjModel*, double time) { const std::unique_lock<std::shared_mutex> lock(mtx_); time_ = time; }
void State::CopyTo(double* dst_state, double* dst_mocap, double* dst_userdata, double* dst_time) const {
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  mju_copy(dst_state, state_.data(), (int)state_.size());
  *dst_time = time_;
  mju_copy(dst_mocap, mocap_.data(), (int)mocap_.size());
  mju_copy(dst_userdata, userdata_.data(), (int)userdata_.size());
}
void State::CopyTo(const mjModel* m, mjData* d) const {
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  mju_copy(d->qpos, state_.data(), m->nq);
  mju_copy(d->qvel, state_.data() + m->nq, m->nv);
  if (m->na) mju_copy(d->act, state_.data() + m->nq + m->nv, m->na);
  for (int i = 0; i < m->nmocap; i++) {
    mju_copy(d->mocap_pos + 3 * i, mocap_.data() + 7 * i, 3);
    mju_copy(d->mocap_quat + 4 * i, mocap_.data() + 7 * i + 3, 4);
  }
  if (m->nuserdata) mju_copy(d->userdata, userdata_.data(), m->nuserdata);
  d->time = time_;
}

}  // namespace mjpc
