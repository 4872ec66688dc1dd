#include "tracking.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <stdexcept>
#include <tuple>

#include "../../../../../../include/mjpcx.h"
#include "../../../../model_io.h"

namespace {
// linear interpolation between two consecutive keyframes (tracking.cc:29-38)
std::tuple<int, int, double, double> InterpolationValues(double index, int max_index) {
  const double clamped = std::clamp(index, 0.0, (double)max_index);
  const int index_0 = (int)std::floor(clamped);
  const int index_1 = std::min(index_0 + 1, max_index);
  const double weight_1 = clamped - index_0;
  return {index_0, index_1, 1.0 - weight_1, weight_1};
}

constexpr double kFps = 30.0;  // CMU mocap keyframes (tracking.cc:41)
// frames per motion, in the <include> order of the task XML (tracking.cc:43-54)
constexpr int kMotionLengths[] = {121, 154, 115, 78, 145, 188, 260, 279, 39, 510};
constexpr int kNumMotion = sizeof(kMotionLengths) / sizeof(kMotionLengths[0]);

int MotionLength(int id) { return kMotionLengths[id]; }
int MotionStartIndex(int id) {
  int start = 0;
  for (int i = 0; i < id; i++) start += MotionLength(i);
  return start;
}

// marker order of the residual (tracking.cc:71-75)
const std::array<std::string, 16> kBodyNames = {"pelvis", "head",   "ltoe",  "rtoe",   "lheel",  "rheel",     "lknee",     "rknee",
                                                "lhand",  "rhand",  "lelbow", "relbow", "lshoulder", "rshoulder", "lhip",  "rhip"};
}  // namespace

namespace mjpc::humanoid {

std::string Tracking::Name() const { return "Humanoid Track"; }  // as the reference; "HumanoidTrack" is accepted as an alias by the C API
std::string Tracking::XmlPath() const { return "humanoid/tracking/task.xml"; }
int Tracking::DeviceResidualId() const { return MJPCX_RESIDUAL_HUMANOID_TRACK; }

void Tracking::ResidualFn::Residual(const mjModel*, const mjData*, double*) const {
  throw std::runtime_error("humanoid::Tracking residual is evaluated on the device (MJPCX_RESIDUAL_HUMANOID_TRACK)");
}

void Tracking::ResetLocked(const mjModel* model) {
  site_ids_.clear();
  mocap_ids_.clear();
  for (const std::string& name : kBodyNames) {
    const int site = NameToId(model, mjOBJ_SITE, "tracking[" + name + "]");
    const int body = NameToId(model, mjOBJ_BODY, "mocap[" + name + "]");
    if (site < 0 || body < 0 || model->body_mocapid[body] < 0) throw std::runtime_error("humanoid tracking: missing marker " + name);
    site_ids_.push_back(site);
    mocap_ids_.push_back(model->body_mocapid[body]);
  }
  int total = 0;
  for (int i = 0; i < kNumMotion; i++) total += MotionLength(i);
  if (model->nkey < total) throw std::runtime_error("humanoid tracking: the model lacks the mocap keyframes");
}

void Tracking::ResidualState(std::vector<int32_t>* ints, std::vector<double>* reals) const {
  const int start = MotionStartIndex(residual_.current_mode_);
  ints->assign({start, start + MotionLength(residual_.current_mode_) - 1});
  ints->insert(ints->end(), site_ids_.begin(), site_ids_.end());
  ints->insert(ints->end(), mocap_ids_.begin(), mocap_ids_.end());
  reals->assign({residual_.reference_time_});
}

void Tracking::TransitionLocked(mjModel* model, mjData* d) {
  if (mode < 0 || mode >= kNumMotion) throw std::runtime_error("humanoid tracking: motion id out of range");
  const int start = MotionStartIndex(mode);
  const int length = MotionLength(mode);
  if (residual_.current_mode_ != mode || d->time == 0.0) {  // motion switch: restart the clip from its first keyframe
    residual_.current_mode_ = mode;
    residual_.reference_time_ = d->time;
    if (d->qpos) mju_copy(d->qpos, model->key_qpos + (size_t)model->nq * start, model->nq);
    if (d->qvel) mju_copy(d->qvel, model->key_qvel + (size_t)model->nv * start, model->nv);
  }
  const double current_index = (d->time - residual_.reference_time_) * kFps + start;
  const auto [key_0, key_1, weight_0, weight_1] = InterpolationValues(current_index, start + length - 1);
  if (d->mocap_pos) {
    const double* frame_0 = model->key_mpos + (size_t)model->nmocap * 3 * key_0;
    const double* frame_1 = model->key_mpos + (size_t)model->nmocap * 3 * key_1;
    for (int i = 0; i < 3 * model->nmocap; i++) {
      d->mocap_pos[i] = frame_0[i] * weight_0;
      d->mocap_pos[i] += frame_1[i] * weight_1;
    }
  }
}

}  // namespace mjpc::humanoid
