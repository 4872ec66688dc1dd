// mjpc::State (mjpc/states/state.h): the thread-safe snapshot handed to Planner::SetState.
#pragma once
#include <shared_mutex>
#include <vector>

#include <mujoco/mujoco.h>

namespace mjpc {

class State {
 public:
  void Initialize(const mjModel* model) {}
  void Allocate(const mjModel* model);
  void Reset();
  void Set(const mjModel* model, const mjData* data);
  void Set(const mjModel* model, const double* qpos, const double* qvel, const double* act, const double* mocap_pos,
           const double* mocap_quat, const double* userdata, double time);
  void SetPosition(const mjModel* model, const double* qpos);
  void SetVelocity(const mjModel* model, const double* qvel);
  void SetAct(const mjModel* model, const double* act);
  void SetMocap(const mjModel* model, const double* mocap_pos, const double* mocap_quat);
  void SetUserData(const mjModel* model, const double* userdata);
  void SetTime(const mjModel* model, double time);
  void CopyTo(double* dst_state, double* dst_mocap, double* dst_userdata, double* time) const;
  void CopyTo(const mjModel* model, mjData* data) const;

  const std::vector<double>& state() const { return state_; }
  const std::vector<double>& mocap() const { return mocap_; }
  const std::vector<double>& userdata() const { return userdata_; }
  double time() const { return time_; }

 private:
  void Resize(const mjModel* model);
  // unlocked bodies of the setters (callers hold mtx_)
  void PutPosition(const mjModel* model, const double* qpos);
  void PutVelocity(const mjModel* model, const double* qvel);
  void PutAct(const mjModel* model, const double* act);
  void PutMocap(const mjModel* model, const double* mocap_pos, const double* mocap_quat);
  void PutUserData(const mjModel* model, const double* userdata);
  std::vector<double> state_, mocap_, userdata_;
  double time_ = 0;
  mutable std::shared_mutex mtx_;
};

}  // namespace mjpc
