#include "quadruped.h"

#include <cmath>
#include <stdexcept>

#include "../../../../../include/mjpcx.h"
#include "../../../model_io.h"
#include "../../utilities.h"

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

std::string QuadrupedFlat::Name() const { return "Quadruped Flat"; }  // quadruped.cc:31
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
  r.current_gait_ = ReinterpretAsDouble(ResidualFn::kGaitStand);
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

// quadruped.cc:633-649
void QuadrupedFlat::ResidualFn::Walk(double pos[2], double time) const {
  if (std::fabs(angvel_) < kMinAngvel) {  // no rotation, go in a straight line
    double forward[2] = {heading_[0], heading_[1]};
    const double n = std::sqrt(forward[0] * forward[0] + forward[1] * forward[1]);
    if (n > 1e-15) { forward[0] /= n; forward[1] /= n; }
    pos[0] = position_[0] + heading_[0] + time * speed_ * forward[0];
    pos[1] = position_[1] + heading_[1] + time * speed_ * forward[1];
  } else {  // walk on a circle
    const double angle = time * angvel_, cs = std::cos(angle), sn = std::sin(angle);
    pos[0] = cs * heading_[0] - sn * heading_[1] + position_[0];
    pos[1] = sn * heading_[0] + cs * heading_[1] + position_[1];
  }
}

// quadruped.cc:229-391. Kinematic inputs (data->xpos / xmat / xquat / site_xpos / subtree_com / subtree_linvel, mocap_pos) are
// the mjData fields the reference reads (it reaches subtree_com / subtree_linvel through the torso_subtreecom /
// torso_subtreelinvel sensors); NULL pointers skip the part that needs them.
void QuadrupedFlat::TransitionLocked(mjModel* model, mjData* data) {
  ResidualFn& r = residual_;
  const double time = data->time;
  // ---------- handle mjData reset ----------
  if (time < r.last_transition_time_ || r.last_transition_time_ == -1) {
    if (mode != ResidualFn::kModeQuadruped && mode != ResidualFn::kModeBiped) mode = ResidualFn::kModeQuadruped;
    r.last_transition_time_ = r.phase_start_time_ = r.phase_start_ = time;
  }
  // ---------- prevent forbidden mode transitions ----------
  if (mode != r.current_mode_ && r.current_mode_ != ResidualFn::kModeQuadruped) {
    if (mode == ResidualFn::kModeWalk || mode == ResidualFn::kModeFlip) mode = ResidualFn::kModeQuadruped;
  }
  // ---------- handle phase velocity change ----------
  const double phase_velocity = 2 * kPi * parameters[r.cadence_param_id_];
  if (phase_velocity != r.phase_velocity_) {
    r.phase_start_ = r.GetPhase(time);
    r.phase_start_time_ = time;
    r.phase_velocity_ = phase_velocity;
  }
  // ---------- automatic gait switching ----------
  const int torso = r.torso_body_id_;
  if (data->subtree_linvel) {
    const double* comvel = data->subtree_linvel + 3 * torso;
    const double beta = std::exp(-(time - r.last_transition_time_) / ResidualFn::kAutoGaitFilter);
    r.com_vel_[0] = beta * r.com_vel_[0] + (1 - beta) * comvel[0];
    r.com_vel_[1] = beta * r.com_vel_[1] + (1 - beta) * comvel[1];
  }
  const int auto_switch = ReinterpretAsInt(parameters[r.gait_switch_param_id_]);  // quadruped.cc:267
  if (mode == ResidualFn::kModeBiped) {
    parameters[r.gait_param_id_] = ReinterpretAsDouble(ResidualFn::kGaitTrot);  // biped always trots
  } else if (auto_switch && data->subtree_linvel) {
    const double com_speed = std::sqrt(r.com_vel_[0] * r.com_vel_[0] + r.com_vel_[1] * r.com_vel_[1]);
    for (int gait : ResidualFn::kGaitAll) {
      if (mode == ResidualFn::kModeScramble && gait == ResidualFn::kGaitStand) continue;  // scramble requires a non-static gait
      const bool lower = com_speed > ResidualFn::kGaitAuto[gait];
      const bool upper = gait == ResidualFn::kGaitGallop || com_speed <= ResidualFn::kGaitAuto[gait + 1];
      const bool wait = std::fabs(r.gait_switch_time_ - time) > ResidualFn::kAutoGaitMinTime;
      if (lower && upper && wait) {
        parameters[r.gait_param_id_] = ReinterpretAsDouble(gait);
        r.gait_switch_time_ = time;
      }
    }
  }
  // ---------- handle gait switch, manual or auto ----------
  const double gait_selection = parameters[r.gait_param_id_];
  if (gait_selection != r.current_gait_) {
    r.current_gait_ = gait_selection;
    const int gait = r.GetGait();
    parameters[r.duty_param_id_] = ResidualFn::kGaitParam[gait][0];
    parameters[r.cadence_param_id_] = ResidualFn::kGaitParam[gait][1];
    parameters[r.amplitude_param_id_] = ResidualFn::kGaitParam[gait][2];
    weight[r.balance_cost_id_] = ResidualFn::kGaitParam[gait][3];
    weight[r.upright_cost_id_] = ResidualFn::kGaitParam[gait][4];
    weight[r.height_cost_id_] = ResidualFn::kGaitParam[gait][5];
  }
  // ---------- Walk ----------
  double* goal_pos = data->mocap_pos ? data->mocap_pos + 3 * r.goal_mocap_id_ : nullptr;
  if (mode == ResidualFn::kModeWalk && goal_pos && data->xmat && data->xpos) {
    const double angvel = parameters[ParameterIndex(model, "Walk turn")];
    const double speed = parameters[ParameterIndex(model, "Walk speed")];
    const double* torso_xmat = data->xmat + 9 * torso;
    double forward[2] = {torso_xmat[0], torso_xmat[3]};  // current torso direction
    const double fn = std::sqrt(forward[0] * forward[0] + forward[1] * forward[1]);
    if (fn > 1e-15) { forward[0] /= fn; forward[1] /= fn; }
    const double leftward[2] = {-forward[1], forward[0]};
    // switching into Walk or parameters changed: reset the task state
    if (mode != r.current_mode_ || r.angvel_ != angvel || r.speed_ != speed) {
      r.mode_start_time_ = time;
      r.speed_ = speed;
      r.angvel_ = angvel;
      double axis[2] = {data->xpos[3 * torso], data->xpos[3 * torso + 1]};  // rotation axis / walk origin
      if (std::fabs(angvel) > ResidualFn::kMinAngvel) {
        const double dd = speed / angvel;
        axis[0] += dd * leftward[0];
        axis[1] += dd * leftward[1];
      }
      r.position_[0] = axis[0];
      r.position_[1] = axis[1];
      r.heading_[0] = goal_pos[0] - axis[0];  // vector from the axis to the initial goal position
      r.heading_[1] = goal_pos[1] - axis[1];
    }
    r.Walk(goal_pos, time - r.mode_start_time_);  // move the goal
  }
  // ---------- Flip ----------
  if (mode == ResidualFn::kModeFlip && data->xquat && data->subtree_com) {
    if (mode != r.current_mode_) {  // switching into Flip: reset the task state
      r.mode_start_time_ = time;
      mju_copy(r.orientation_, data->xquat + 4 * torso, 4);
      r.ground_ = Ground(model, data, data->subtree_com + 3 * torso);
      r.save_weight_ = weight;
      r.save_gait_switch_ = parameters[r.gait_switch_param_id_];
      auto term = [&](const char* name) {
        for (int i = 0; i < num_term; i++) if (weight_names[i] == name) return i;
        throw std::runtime_error(std::string("cost term '") + name + "' not found");
      };
      weight[term("Upright")] = 0.2;
      weight[term("Height")] = 5;
      weight[term("Position")] = 0;
      weight[term("Gait")] = 0;
      weight[term("Balance")] = 0;
      weight[term("Effort")] = 0.005;
      weight[term("Posture")] = 0.1;
      parameters[r.gait_switch_param_id_] = ReinterpretAsDouble(0);
    }
    const double flip_time = time - r.mode_start_time_;
    if (flip_time >= r.jump_time_ + r.flight_time_ + r.land_time_) {  // Flip ended: back to Quadruped, restore the values
      mode = ResidualFn::kModeQuadruped;
      weight = r.save_weight_;
      parameters[r.gait_switch_param_id_] = r.save_gait_switch_;
      if (goal_pos && data->site_xpos) {
        goal_pos[0] = data->site_xpos[3 * r.head_site_id_ + 0];
        goal_pos[1] = data->site_xpos[3 * r.head_site_id_ + 1];
      }
    }
  }
  // save mode
  r.current_mode_ = static_cast<ResidualFn::A1Mode>(mode);
  r.last_transition_time_ = time;
}

// layout: csrc/wave_residual.h
void QuadrupedFlat::ResidualState(std::vector<int32_t>* ints, std::vector<double>* reals) const {
  std::lock_guard<std::mutex> lock(mutex_);
  const ResidualFn& r = residual_;
  *ints = {(int32_t)r.current_mode_, r.torso_body_id_, r.head_site_id_, r.goal_mocap_id_, r.foot_geom_id_[0], r.foot_geom_id_[1],
           r.foot_geom_id_[2], r.foot_geom_id_[3], (int32_t)ReinterpretAsInt(r.current_gait_), (int32_t)ReinterpretAsInt(parameters[r.flip_dir_param_id_]),
           (int32_t)ReinterpretAsInt(parameters[r.biped_type_param_id_]), r.amplitude_param_id_, r.duty_param_id_, r.arm_posture_param_id_,
           r.heading_param_id_, r.key_home_, r.key_crouch_};
  *reals = {r.mode_start_time_, r.position_[0], r.position_[1], r.position_[2], r.heading_[0], r.heading_[1], r.speed_, r.angvel_,
            r.ground_, r.orientation_[0], r.orientation_[1], r.orientation_[2], r.orientation_[3], r.phase_start_,
            r.phase_start_time_, r.phase_velocity_, r.gravity_, r.jump_vel_, r.flight_time_, r.jump_acc_, r.crouch_time_,
            r.leap_time_, r.jump_time_, r.crouch_vel_, r.land_time_, r.land_acc_, r.flight_rot_vel_, r.jump_rot_vel_,
            r.jump_rot_acc_, r.land_rot_acc_};
}

}  // namespace mjpc
