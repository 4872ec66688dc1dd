// mjpc/test/tasks/task_test.cc:49-99 on the compiled particle task (argv[1] = Particle.mjpx, argv[2] = Cartpole.mjpx).
#include <cmath>

#include "check.h"
#include "mjpc/tasks/tasks.h"
#include "mjpc/utilities.h"
#include "model_io.h"
using namespace mjpc;

int main(int argc, char** argv) {
  if (argc < 3) { std::printf("usage: task_test Particle.mjpx Cartpole.mjpx\n"); return 2; }
  auto storage = ModelStorage::Load(argv[1]);
  const mjModel* model = storage->model();
  ParticleTestTask task;
  task.Reset(model);
  CHECK_NEAR(task.risk, 1.0, 1e-5);
  CHECK(task.mode == 0 && task.parameters.size() == 2);
  CHECK_NEAR(task.parameters[0], 0.05, 1e-5);
  CHECK_NEAR(task.parameters[1], -0.1, 1e-5);
  CHECK(task.num_residual == 4 && task.num_term == 2 && task.num_trace == 1);
  CHECK(task.dim_norm_residual[0] == 2 && task.dim_norm_residual[1] == 2);
  CHECK(task.num_norm_parameter[0] == 0 && task.num_norm_parameter[1] == 0);
  CHECK(task.norm[0] == kQuadratic && task.norm[1] == kQuadratic);
  CHECK_NEAR(task.weight[0], 5.0, 1e-5);
  CHECK_NEAR(task.weight[1], 0.1, 1e-5);
  CHECK(task.trace_site.size() == 1 && task.trace_site[0] == 0);
  double terms[2];
  const double residual[] = {1.0e-3, 2.0e-3, 3.0e-3, 4.0e-3};
  task.CostTerms(terms, residual);
  const double c = 5.0 * 0.5 * (residual[0] * residual[0] + residual[1] * residual[1]) +
                   0.1 * 0.5 * (residual[2] * residual[2] + residual[3] * residual[3]);
  CHECK_NEAR(terms[0] + terms[1], c, 1e-12);
  task.risk = 0.2;
  task.UpdateResidual();
  CHECK_NEAR(task.CostValue(residual), (std::exp(0.2 * c) - 1.0) / 0.2, 1e-12);
  // GetNumberOrDefault (mjpc/test/agent/agent_utilities_test.cc)
  CHECK_NEAR(GetNumberOrDefault(7.0, model, "test_double"), 0.1, 1e-15);
  CHECK(GetNumberOrDefault(0, model, "test_int") == 1 && GetNumberOrDefault(3, model, "absent") == 3);
  CHECK(GetCustomNumericSize(model, "test_doubles") == 2);
  double xs[3] = {-2.0, 0.5, 2.0};
  const double bounds[6] = {-1, 1, -1, 1, -1, 1};
  Clamp(xs, bounds, 3);
  CHECK(xs[0] == -1.0 && xs[1] == 0.5 && xs[2] == 1.0);
  CHECK(KeyQPosByName(model, "home") != nullptr && KeyQPosByName(model, "home")[1] == 2.0);

  auto cstorage = ModelStorage::Load(argv[2]);
  Cartpole cart;
  cart.Reset(cstorage->model());
  CHECK(cart.num_term == 4 && cart.norm[0] == kSmoothAbsLoss && cart.norm[2] == kQuadratic);
  CHECK(cart.norm_parameter.size() == 2 && cart.norm_parameter[0] == 0.01 && cart.norm_parameter[1] == 0.1);
  // residual function on a hand-built mjData
  double qpos[2] = {0.3, 1.0}, qvel[2] = {0.1, -0.2}, ctrl[1] = {0.7}, r[4];
  mjData d{};
  d.qpos = qpos; d.qvel = qvel; d.ctrl = ctrl;
  cart.Residual(cstorage->model(), &d, r);
  CHECK_NEAR(r[0], std::cos(1.0) - 1, 1e-15);
  CHECK(r[1] == 0.3 && r[2] == -0.2 && r[3] == 0.7);
  CHECK(GetTasks().size() == 5);
  if (argc > 3) {  // QuadrupedFlat: ResetLocked ids, Transition state and the frozen residual copy (quadruped.cc:229-391, 520-607)
    auto qstorage = ModelStorage::Load(argv[3]);
    std::shared_ptr<Task> quad;
    for (auto& t : GetTasks()) if (t->Name() == "QuadrupedFlat") quad = t;
    CHECK(quad != nullptr);
    quad->Reset(qstorage->model());
    CHECK(quad->num_term == 9 && quad->num_residual == 42 && quad->num_trace == 1 && quad->parameters.size() == 11);
    std::vector<int32_t> ri; std::vector<double> rr;
    quad->ResidualState(&ri, &rr);
    CHECK(ri.size() == 17 && rr.size() == 30);
    CHECK(ri[0] == 0 && ri[8] == 0 && rr[15] == 0.0);  // Quadruped mode, Stand, phase clock not started
    mjData d{};
    d.time = 0.5;
    quad->parameters[0] = 2;  // select_Gait = Trot
    quad->Transition(qstorage->model(), &d);
    quad->ResidualState(&ri, &rr);
    CHECK(ri[8] == 2);                                  // gait switched
    CHECK_NEAR(rr[15], 2 * 3.14159265358979323846 * 2, 1e-12);  // phase velocity from the XML cadence (2 Hz) at the first transition
    CHECK_NEAR(rr[13], 0.5, 0); CHECK_NEAR(rr[14], 0.5, 0);   // phase_start_ / phase_start_time_ = time of the first transition
    CHECK(quad->parameters[4] == 0.45 && quad->parameters[2] == 2 && quad->parameters[3] == 0.03);  // duty, cadence, amplitude of Trot
    CHECK(quad->weight[4] == 0.2 && quad->weight[0] == 1 && quad->weight[1] == 1);                   // balance, upright, height
    CHECK_NEAR(rr[18], 2 * std::sqrt(2 * 9.81 * 0.3) / 9.81, 1e-12);                               // flight_time_
  }
  if (argc > 4) {  // humanoid::Tracking: cost parse, marker ids, Transition (tracking.cc:219-264) and the frozen residual copy
    auto hstorage = ModelStorage::Load(argv[4]);
    mjModel* hm = hstorage->model();
    std::shared_ptr<Task> track;
    for (auto& t : GetTasks()) if (t->Name() == "Humanoid Track") track = t;
    CHECK(track != nullptr);
    track->Reset(hm);
    CHECK(track->num_term == 21 && track->num_residual == 141 && track->num_trace == 1);
    CHECK(track->trace_site[0] == -1 - NameToId(hm, mjOBJ_BODY, "torso"));  // trace0 is the frame of a body
    CHECK(hm->nkey == 1889 && hm->nmocap == 16 && hm->ntendon == 2 && hm->nexclude == 2);
    std::vector<double> qpos(hm->nq, 0.0), qvel(hm->nv, 1.0), mpos(3 * hm->nmocap, 0.0);
    mjData d{};
    d.qpos = qpos.data(); d.qvel = qvel.data(); d.mocap_pos = mpos.data();
    d.time = 0.0;
    track->mode = 8;  // Run: keys 1340 .. 1378
    track->Transition(hm, &d);
    std::vector<int32_t> ri; std::vector<double> rr;
    track->ResidualState(&ri, &rr);
    CHECK(ri.size() == 34 && rr.size() == 1 && ri[0] == 1340 && ri[1] == 1378 && rr[0] == 0.0);
    CHECK(ri[2] == NameToId(hm, mjOBJ_SITE, "tracking[pelvis]") && ri[18] == hm->body_mocapid[NameToId(hm, mjOBJ_BODY, "mocap[pelvis]")]);
    for (int i = 0; i < hm->nq; i++) CHECK(qpos[i] == hm->key_qpos[(size_t)hm->nq * 1340 + i]);   // reset to the first keyframe
    CHECK(qvel[0] == hm->key_qvel[(size_t)hm->nv * 1340] && qvel[10] == 0.0);
    for (int i = 0; i < 48; i++) CHECK(mpos[i] == hm->key_mpos[(size_t)48 * 1340 + i]);
    d.time = 0.25;  // index 1340 + 7.5: halfway between two keyframes; no reset
    qpos[0] = 123.0;
    track->Transition(hm, &d);
    CHECK(qpos[0] == 123.0);
    for (int i = 0; i < 48; i++) CHECK_NEAR(mpos[i], 0.5 * hm->key_mpos[(size_t)48 * 1347 + i] + 0.5 * hm->key_mpos[(size_t)48 * 1348 + i], 1e-15);
    d.time = 100.0;  // past the end of the clip: clamps to the last key
    track->Transition(hm, &d);
    for (int i = 0; i < 48; i++) CHECK(mpos[i] == hm->key_mpos[(size_t)48 * 1378 + i]);
    track->mode = 9;  // motion switch re-references the clock
    track->Transition(hm, &d);
    track->ResidualState(&ri, &rr);
    CHECK(ri[0] == 1379 && ri[1] == 1888 && rr[0] == 100.0);
  }
  TEST_MAIN_END();
}
