// mjpc::humanoid::Tracking (mjpc/tasks/humanoid/tracking/tracking.{h,cc}) for the GPU planners.
//
// The 141-entry residual runs inside the rollout kernels (MJPCX_RESIDUAL_HUMANOID_TRACK, csrc/wave_residual.h) from the
// model's key_mpos table; this class owns what the reference's Task owns on the host: the motion bookkeeping of
// TransitionLocked (tracking.cc:219-264: motion switch, reference time, the interpolated mocap marker positions, the
// reset to the motion's first keyframe) and the frozen ResidualFn copy (current_mode_, reference_time_) handed to the
// device as mjpcx_task::residual_int / residual_real.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../../task.h"

namespace mjpc::humanoid {

class Tracking : public Task {
 public:
  class ResidualFn : public mjpc::BaseResidualFn {
   public:
    explicit ResidualFn(const Tracking* task, int current_mode = 0, double reference_time = 0)
        : mjpc::BaseResidualFn(task), current_mode_(current_mode), reference_time_(reference_time) {}
    // evaluated on the device; the host entry point has no kinematics or sensors to read and throws
    void Residual(const mjModel* model, const mjData* data, double* residual) const override;

   private:
    friend class Tracking;
    int current_mode_;
    double reference_time_;
  };

  Tracking() : residual_(this) {}

  std::string Name() const override;
  std::string XmlPath() const override;
  int DeviceResidualId() const override;
  // [first key, last key, 16 tracking-site ids, 16 mocap ids], [reference_time]
  void ResidualState(std::vector<int32_t>* ints, std::vector<double>* reals) const override;

  // sets data->mocap_pos from data->time; on a motion switch or at time 0 also data->qpos / qvel (null pointers are skipped)
  void TransitionLocked(mjModel* model, mjData* data) override;

 protected:
  void ResetLocked(const mjModel* model) override;
  std::unique_ptr<mjpc::ResidualFn> ResidualLocked() const override {
    return std::make_unique<ResidualFn>(this, residual_.current_mode_, residual_.reference_time_);
  }
  ResidualFn* InternalResidual() override { return &residual_; }

 private:
  ResidualFn residual_;
  std::vector<int32_t> site_ids_, mocap_ids_;
};

}  // namespace mjpc::humanoid
