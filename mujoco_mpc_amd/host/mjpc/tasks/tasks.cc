#include "tasks.h"

#include "humanoid/tracking/tracking.h"
#include "quadruped/quadruped.h"

#include <cmath>

#include "../../../../include/mjpcx.h"

namespace mjpc {

// ---- Cartpole: vertical, centered, velocity, control
std::string Cartpole::Name() const { return "Cartpole"; }
std::string Cartpole::XmlPath() const { return "cartpole/task.xml"; }
int Cartpole::DeviceResidualId() const { return MJPCX_RESIDUAL_CARTPOLE; }
void Cartpole::ResidualFn::Residual(const mjModel* model, const mjData* data, double* residual) const {
  residual[0] = std::cos(data->qpos[1]) - 1;
  residual[1] = data->qpos[0] - parameters_[0];
  residual[2] = data->qvel[1];
  residual[3] = data->ctrl[0];
}

// ---- particle: position error w.r.t. the mocap goal, velocity
std::string ParticleTestTask::Name() const { return "Particle"; }
std::string ParticleTestTask::XmlPath() const { return "particle/task.xml"; }
int ParticleTestTask::DeviceResidualId() const { return MJPCX_RESIDUAL_PARTICLE; }
void ParticleTestTask::ResidualFn::Residual(const mjModel* model, const mjData* data, double* residual) const {
  mju_copy(residual, data->qpos, model->nq);
  residual[0] -= data->mocap_pos[0];
  residual[1] -= data->mocap_pos[1];
  mju_copy(residual + 2, data->qvel, model->nv);
}

// ---- particle copy: residual = [qpos, qvel]
std::string ParticleCopyTestTask::Name() const { return "ParticleCopy"; }
std::string ParticleCopyTestTask::XmlPath() const { return "particle/task.xml"; }
int ParticleCopyTestTask::DeviceResidualId() const { return MJPCX_RESIDUAL_PARTICLE_COPY; }
void ParticleCopyTestTask::ResidualFn::Residual(const mjModel* model, const mjData* data, double* residual) const {
  mju_copy(residual, data->qpos, model->nq);
  mju_copy(residual + model->nq, data->qvel, model->nv);
}

std::vector<std::shared_ptr<Task>> GetTasks() {
  return {std::make_shared<Cartpole>(), std::make_shared<ParticleTestTask>(), std::make_shared<ParticleCopyTestTask>(),
          std::make_shared<QuadrupedFlat>(), std::make_shared<humanoid::Tracking>()};
}

}  // namespace mjpc
