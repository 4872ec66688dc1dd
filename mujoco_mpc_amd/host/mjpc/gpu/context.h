// mjpc::gpu -- the binding between MJPC's host types and the C ABI of libmjpcx.so (include/mjpcx.h):
// repacks an mjModel (MuJoCo strides) and a Task into the flat structs mjpcx_create takes, and wraps the
// context in RAII. This is the code a maintainer adds to a real MJPC checkout.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../../include/mjpcx.h"
#include <mujoco/mujoco.h>
#include "../task.h"
#include "../trajectory.h"

namespace mjpc::gpu {

class Error : public std::runtime_error {
 public:
  Error(int code, const std::string& what) : std::runtime_error(what), code(code) {}
  int code;
};

// Owns the repacked arrays behind an mjpcx_model.
class FlatModel {
 public:
  // `timestep` / `integrator`: Agent::PlanIteration's overrides of the planning copy (agent.cc:288-291)
  // `differentiable`: MakeDifferentiable (utilities.cc:60-75) on the planning copy: solimp[0] = 0 for joints and geoms
  FlatModel(const mjModel* m, double timestep, int integrator, bool differentiable = false);
  const mjpcx_model* get() const { return &flat_; }

 private:
  mjpcx_model flat_{};
  std::vector<int32_t> jnt_limited_, trnid_, ctrllimited_, forcelimited_, tendon_limited_;
  std::vector<double> gear_, gainprm_, biasprm_, jnt_solimp_, geom_solimp_;
};

class FlatTask {
 public:
  explicit FlatTask(const Task& task);
  const mjpcx_task* get() const { return &flat_; }

 private:
  mjpcx_task flat_{};
  std::vector<int32_t> norm_, residual_int_;
  std::vector<double> residual_real_, parameters_;
};

// host copies of the mjData kinematics a Task::Transition may read (quadruped.cc:229-391), filled by Context::Kinematics
struct KinematicsBuffers {
  std::vector<double> xpos, xquat, xmat, xipos, site_xpos, subtree_com, subtree_linvel;
  void Allocate(const mjModel* m);
  void Attach(mjData* d);  // points the mjData kinematic fields at these buffers
};

class Context {
 public:
  Context(const mjModel* model, const Task& task, int device, int precision = 64, bool differentiable = false);
  ~Context();
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;

  mjpcx_ctx* handle() { return ctx_; }
  std::string KernelName() const { return mjpcx_kernel_name(ctx_); }
  void Check(int rc) const;  // throws gpu::Error with mjpcx_last_error on rc != 0
  void SyncTask(const Task& task);  // weights, norm/residual parameters, risk and the frozen ResidualFn state
  // gathers candidate `index` into a (pre-allocated) reference-layout Trajectory
  void FetchTrajectory(int index, Trajectory* trajectory);
  // kinematics of the state last given to mjpcx_set_state; false when this model's kernel family has no such query
  // (lane-per-candidate models: none of their tasks' Transition reads kinematics), throws on any other error
  bool Kinematics(KinematicsBuffers* out);

 private:
  mjpcx_ctx* ctx_ = nullptr;
};

}  // namespace mjpc::gpu
