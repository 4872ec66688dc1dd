#include "quadruped.h"

#include <cmath>
#include <stdexcept>

#include "../../../../../include/mjpcx.h"
#include "../../../model_io.h"

namespace mjpc {

namespace {
constexpr double kPi = 3.14159265358979323846;
int ParameterIndex(const mjModel* model, const std::string& name) {  // utilities.cc:207-223
  int first = -1, id = -1;
  const std::string full = "residual_" + name;
  for (int i = 0; i < model->nnumeric; i++) {
    const std::string n = model->names + model->name_numericadr[i];
    if (first < 0 && n.rfind("residual_", 0) == 0) first = i;
    if (n == full) id = i;
  }
  if (id < 0) throw std::runtime_error("Parameter '" + name + "' not found");
  return id - first;
}
}  // namespace

std::string QuadrupedFlat::Name() const { return "QuadrupedFlat"; }
std::string QuadrupedFlat::XmlPath() const { return "quadruped/task_flat.xml"; }
int QuadrupedFlat::DeviceResidualId() const { return MJPCX_RESIDUAL_QUADRUPED_FLAT; }

void QuadrupedFlat::ResidualFn::Residual(const mjModel*, const mjData*, double*) const {
  throw std::runtime_error("QuadrupedFlat::Residual runs on the device (MJPCX_RESIDUAL_QUADRUPED_FLAT); no CPU physics in this build");
}

// quadruped.cc:520-607
void QuadrupedFlat::ResetLocked(const mjModel* model) {
  ResidualFn& r = residual_;
  r.gait_param_id_ = ParameterIndex(model, "select_Gait");
  r.gait_switch_param_id_ = ParameterIndex(model, "select_Gait switch");
  r.flip_dir_param_id_ = ParameterIndex(model, "select_Flip dir");
  r.biped_type_param_id_ = ParameterIndex(model, "select_Biped type");
  r.cadence_param_id_ = ParameterIndex(model, "Cadence");
  r.amplitude_param_id_ = ParameterIndex(model, "Amplitude");
  r.duty_param_id_ = ParameterIndex(model, "Duty ratio");
  r.arm_posture_param_id_ = ParameterIndex(model, "Arm posture");
  r.heading_param_id_ = ParameterIndex(model, "Heading");
  auto cost_term = [&](const char* name) {
    for (int i = 0; i < num_term; i++) if (weight_names[i] == name) return i;
    throw std::runtime_error(std::string("cost term '") + name + "' not found");
  };
  r.balance_cost_id_ = cost_term("Balance");
  r.upright_cost_id_ = cost_term("Upright");
  r.height_cost_id_ = cost_term("Height");
  r.torso_body_id_ = NameToId(model, mjOBJ_XBODY, "trunk");
  if (r.torso_body_id_ < 0) throw std::runtime_error("body 'trunk' not found");
  r.head_site_id_ = NameToId(model, mjOBJ_SITE, "head");
  if (r.head_site_id_ < 0) throw std::runtime_error("site 'head' not found");
  const int goal_id = NameToId(model, mjOBJ_XBODY, "goal");
  if (goal_id < 0) throw std::runtime_error("body 'goal' not found");
  r.goal_mocap_id_ = model->body_mocapid[goal_id];
  if (r.goal_mocap_id_ < 0) throw std::runtime_error("body 'goal' is not mocap");
  int foot_index = 0;
  for (const char* footname : {"FL", "HL", "FR", "HR"}) {
    const int foot_id = NameToId(model, mjOBJ_GEOM, footname);
    if (foot_id < 0) throw std::runtime_error(std::string("geom '") + footname + "' not found");
    r.foot_geom_id_[foot_index++] = foot_id;
  }
  r.key_home_ = NameToId(model, mjOBJ_KEY, "home");
  r.key_crouch_ = NameToId(model, mjOBJ_KEY, "crouch");
  if (r.key_home_ < 0 || r.key_crouch_ < 0) throw std::runtime_error("keyframes 'home' / 'crouch' not found");
  // task state back to its defaults
  r.current_mode_ = ResidualFn::kModeQuadruped;
  r.last_transition_time_ = -1;
  r.mode_start_time_ = 0; r.speed_ = r.angvel_ = r.ground_ = 0;
  for (double& v : r.position_) v = 0;
  for (double& v : r.heading_) v = 0;
  for (double& v : r.orientation_) v = 0;
  r.current_gait_ = ResidualFn::kGaitStand;
  r.phase_start_ = r.phase_start_time_ = r.phase_velocity_ = 0;
  // derived kinematic quantities for Flip
  r.gravity_ = std::sqrt(model->opt.gravity[0] * model->opt.gravity[0] + model->opt.gravity[1] * model->opt.gravity[1] +
                         model->opt.gravity[2] * model->opt.gravity[2]);
  r.jump_vel_ = std::sqrt(2 * r.gravity_ * (ResidualFn::kMaxHeight - ResidualFn::kLeapHeight));
  r.flight_time_ = 2 * r.jump_vel_ / r.gravity_;
  r.jump_acc_ = r.jump_vel_ * r.jump_vel_ / (2 * (ResidualFn::kLeapHeight - ResidualFn::kCrouchHeight));
  r.crouch_time_ = std::sqrt(2 * (ResidualFn::kHeightQuadruped - ResidualFn::kCrouchHeight) / r.jump_acc_);
  r.leap_time_ = r.jump_vel_ / r.jump_acc_;
  r.jump_time_ = r.crouch_time_ + r.leap_time_;
  r.crouch_vel_ = -r.jump_acc_ * r.crouch_time_;
  r.land_time_ = 2 * (ResidualFn::kLeapHeight - ResidualFn::kHeightQuadruped) / r.jump_vel_;
  r.land_acc_ = r.jump_vel_ / r.land_time_;
  r.flight_rot_vel_ = 1.25 * kPi / r.flight_time_;
  r.jump_rot_vel_ = kPi / r.leap_time_ - r.flight_rot_vel_;
  r.jump_rot_acc_ = (r.flight_rot_vel_ - r.jump_rot_vel_) / r.leap_time_;
  r.land_rot_acc_ = 2 * (r.flight_rot_vel_ * r.land_time_ - kPi / 4) / (r.land_time_ * r.land_time_);
}

// quadruped.cc:229-391, the parts that need no kinematics (see the header)
void QuadrupedFlat::TransitionLocked(mjModel* model, mjData* data) {
  ResidualFn& r = residual_;
  const double time = data->time;
  if (time < r.last_transition_time_ || r.last_transition_time_ == -1) {
    if (mode != ResidualFn::kModeQuadruped && mode != ResidualFn::kModeBiped) mode = ResidualFn::kModeQuadruped;
    r.last_transition_time_ = r.phase_start_time_ = r.phase_start_ = time;
  }
  if (mode != r.current_mode_ && r.current_mode_ != ResidualFn::kModeQuadruped) {
    if (mode == ResidualFn::kModeWalk || mode == ResidualFn::kModeFlip) mode = ResidualFn::kModeQuadruped;
  }
  const double phase_velocity = 2 * kPi * parameters[r.cadence_param_id_];
  if (phase_velocity != r.phase_velocity_) {
    r.phase_start_ = r.GetPhase(time);
    r.phase_start_time_ = time;
    r.phase_velocity_ = phase_velocity;
  }
  if (mode == ResidualFn::kModeBiped) parameters[r.gait_param_id_] = ResidualFn::kGaitTrot;
  const double gait_selection = parameters[r.gait_param_id_];
  if (gait_selection != r.current_gait_) {
    r.current_gait_ = gait_selection;
    const int gait = r.current_mode_ == ResidualFn::kModeBiped ? ResidualFn::kGaitTrot : (int)r.current_gait_;
    parameters[r.duty_param_id_] = ResidualFn::kGaitParam[gait][0];
    parameters[r.cadence_param_id_] = ResidualFn::kGaitParam[gait][1];
    parameters[r.amplitude_param_id_] = ResidualFn::kGaitParam[gait][2];
    weight[r.balance_cost_id_] = ResidualFn::kGaitParam[gait][3];
    weight[r.upright_cost_id_] = ResidualFn::kGaitParam[gait][4];
    weight[r.height_cost_id_] = ResidualFn::kGaitParam[gait][5];
  }
  r.current_mode_ = static_cast<ResidualFn::A1Mode>(mode);
  r.last_transition_time_ = time;
}

// layout: csrc/wave_residual.h
void QuadrupedFlat::ResidualState(std::vector<int32_t>* ints, std::vector<double>* reals) const {
  std::lock_guard<std::mutex> lock(mutex_);
  const ResidualFn& r = residual_;
  *ints = {(int32_t)r.current_mode_, r.torso_body_id_, r.head_site_id_, r.goal_mocap_id_, r.foot_geom_id_[0], r.foot_geom_id_[1],
           r.foot_geom_id_[2], r.foot_geom_id_[3], (int32_t)r.current_gait_, (int32_t)parameters[r.flip_dir_param_id_],
           (int32_t)parameters[r.biped_type_param_id_], r.amplitude_param_id_, r.duty_param_id_, r.arm_posture_param_id_,
           r.heading_param_id_, r.key_home_, r.key_crouch_};
  *reals = {r.mode_start_time_, r.position_[0], r.position_[1], r.position_[2], r.heading_[0], r.heading_[1], r.speed_, r.angvel_,
            r.ground_, r.orientation_[0], r.orientation_[1], r.orientation_[2], r.orientation_[3], r.phase_start_,
            r.phase_start_time_, r.phase_velocity_, r.gravity_, r.jump_vel_, r.flight_time_, r.jump_acc_, r.crouch_time_,
            r.leap_time_, r.jump_time_, r.crouch_vel_, r.land_time_, r.land_acc_, r.flight_rot_vel_, r.jump_rot_vel_,
            r.jump_rot_acc_, r.land_rot_acc_};
}

}  // namespace mjpc
