// mjpc::QuadrupedFlat (mjpc/tasks/quadruped/quadruped.{h,cc}) for the GPU planners.
//
// The residual itself runs inside the rollout kernels (MJPCX_RESIDUAL_QUADRUPED_FLAT, csrc/wave_residual.h); this
// class owns what the reference's Task owns on the host: ResetLocked (ids, flip kinematics, quadruped.cc:520-607),
// the task state TransitionLocked manages (quadruped.cc:229-391, complete: phase clock, manual and automatic gait
// switching, the Walk goal motion, the Flip bookkeeping) and its frozen copy for the planner (ResidualState ->
// mjpcx_task::residual_int / residual_real). TransitionLocked reads the mjData kinematics the reference reads (torso pose,
// subtree centre of mass and linear velocity, head site, mocap goal); a host without physics fills them from
// mjpcx_kinematics (host/mjpc/testspeed.cc does). With those pointers NULL the kinematics-dependent parts are skipped.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../task.h"
#include "../../utilities.h"

namespace mjpc {

class QuadrupedFlat : public Task {
 public:
  std::string Name() const override;
  std::string XmlPath() const override;
  int DeviceResidualId() const override;
  void ResidualState(std::vector<int32_t>* ints, std::vector<double>* reals) const override;

  class ResidualFn : public BaseResidualFn {
   public:
    explicit ResidualFn(const QuadrupedFlat* task) : BaseResidualFn(task) {}
    // evaluated on the device; the host entry point has no kinematics to read and throws
    void Residual(const mjModel* model, const mjData* data, double* residual) const override;

    enum A1Mode { kModeQuadruped = 0, kModeBiped, kModeWalk, kModeScramble, kModeFlip, kNumMode };
    enum A1Gait { kGaitStand = 0, kGaitWalk, kGaitTrot, kGaitCanter, kGaitGallop, kNumGait };
    constexpr static int kGaitAll[kNumGait] = {kGaitStand, kGaitWalk, kGaitTrot, kGaitCanter, kGaitGallop};
    // velocity ranges for automatic gait switching, m/s; time constant of the com-speed filter; minimum time between switches
    constexpr static double kGaitAuto[kNumGait] = {0, 0.02, 0.02, 0.6, 2};
    constexpr static double kAutoGaitFilter = 0.2, kAutoGaitMinTime = 1;
    constexpr static double kMinAngvel = 0.01;  // below this target yaw velocity, walk straight
    // gait parameters, set when switching into gait (quadruped.h:99-108): duty ratio, cadence, amplitude,
    // balance, upright, height
    constexpr static double kGaitParam[kNumGait][6] = {{1, 1, 0, 0, 1, 1},          {0.75, 1, 0.03, 0, 1, 1},
                                                       {0.45, 2, 0.03, 0.2, 1, 1},  {0.4, 4, 0.05, 0.03, 0.5, 0.2},
                                                       {0.3, 3.5, 0.10, 0.03, 0.2, 0.1}};
    constexpr static double kHeightQuadruped = 0.25, kCrouchHeight = 0.15, kLeapHeight = 0.5, kMaxHeight = 0.8;

    double GetPhase(double time) const { return phase_start_ + (time - phase_start_time_) * phase_velocity_; }
    A1Gait GetGait() const { return current_mode_ == kModeBiped ? kGaitTrot : static_cast<A1Gait>(ReinterpretAsInt(current_gait_)); }
    void Walk(double pos[2], double time) const;  // horizontal Walk trajectory (quadruped.cc:633-649)

    // task state, managed by Transition (quadruped.h:186-214)
    A1Mode current_mode_ = kModeQuadruped;
    double last_transition_time_ = -1;
    double mode_start_time_ = 0, position_[3] = {0}, heading_[2] = {0}, speed_ = 0, angvel_ = 0;
    double ground_ = 0, orientation_[4] = {0};
    double current_gait_ = 0 /* bits of kGaitStand */, phase_start_ = 0, phase_start_time_ = 0, phase_velocity_ = 0;
    double com_vel_[2] = {0, 0}, gait_switch_time_ = 0;
    std::vector<double> save_weight_;
    double save_gait_switch_ = 0;
    // constants, computed in Reset
    int torso_body_id_ = -1, head_site_id_ = -1, goal_mocap_id_ = -1;
    int gait_param_id_ = -1, gait_switch_param_id_ = -1, flip_dir_param_id_ = -1, biped_type_param_id_ = -1;
    int cadence_param_id_ = -1, amplitude_param_id_ = -1, duty_param_id_ = -1, arm_posture_param_id_ = -1, heading_param_id_ = -1;
    int upright_cost_id_ = -1, balance_cost_id_ = -1, height_cost_id_ = -1;
    int foot_geom_id_[4] = {-1, -1, -1, -1};
    int key_home_ = -1, key_crouch_ = -1;
    double gravity_ = 0, jump_vel_ = 0, flight_time_ = 0, jump_acc_ = 0, crouch_time_ = 0, leap_time_ = 0, jump_time_ = 0,
           crouch_vel_ = 0, land_time_ = 0, land_acc_ = 0, flight_rot_vel_ = 0, jump_rot_vel_ = 0, jump_rot_acc_ = 0, land_rot_acc_ = 0;
  };

  QuadrupedFlat() : residual_(this) {}

 protected:
  void TransitionLocked(mjModel* model, mjData* data) override;
  void ResetLocked(const mjModel* model) override;
  std::unique_ptr<mjpc::ResidualFn> ResidualLocked() const override { return std::make_unique<ResidualFn>(residual_); }
  ResidualFn* InternalResidual() override { return &residual_; }

 private:
  ResidualFn residual_;
};

}  // namespace mjpc
