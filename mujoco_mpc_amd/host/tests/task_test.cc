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
    for (auto& t : GetTasks()) if (t->Name() == "Quadruped Flat") quad = t;
    CHECK(quad != nullptr);
    quad->Reset(qstorage->model());
    CHECK(quad->num_term == 9 && quad->num_residual == 42 && quad->num_trace == 1 && quad->parameters.size() == 11);
    std::vector<int32_t> ri; std::vector<double> rr;
    quad->ResidualState(&ri, &rr);
    CHECK(ri.size() == 17 && rr.size() == 30);
    CHECK(ri[0] == 0 && ri[8] == 0 && rr[15] == 0.0);  // Quadruped mode, Stand, phase clock not started
    mjData d{};
    d.time = 0.5;
    quad->parameters[0] = ReinterpretAsDouble(2);  // select_Gait = Trot
    quad->Transition(qstorage->model(), &d);
    quad->ResidualState(&ri, &rr);
    CHECK(ri[8] == 2);                                  // gait switched
    CHECK_NEAR(rr[15], 2 * 3.14159265358979323846 * 2, 1e-12);  // phase velocity from the XML cadence (2 Hz) at the first transition
    CHECK_NEAR(rr[13], 0.5, 0); CHECK_NEAR(rr[14], 0.5, 0);   // phase_start_ / phase_start_time_ = time of the first transition
    CHECK(quad->parameters[4] == 0.45 && quad->parameters[2] == 2 && quad->parameters[3] == 0.03);  // duty, cadence, amplitude of Trot
    CHECK(quad->weight[4] == 0.2 && quad->weight[0] == 1 && quad->weight[1] == 1);                   // balance, upright, height
    CHECK_NEAR(rr[18], 2 * std::sqrt(2 * 9.81 * 0.3) / 9.81, 1e-12);                               // flight_time_
    // ---- the kinematics-dependent parts of TransitionLocked, on synthetic mjData kinematics ----
    mjModel* qm = qstorage->model();
    const int torso = ri[1], head = ri[2], goal = ri[3];
    std::vector<double> xpos(3 * qm->nbody, 0.0), xquat(4 * qm->nbody, 0.0), xmat(9 * qm->nbody, 0.0), com(3 * qm->nbody, 0.0),
        linvel(3 * qm->nbody, 0.0), site(3 * qm->nsite, 0.0), mpos(3 * qm->nmocap, 0.0), mquat(4 * qm->nmocap, 0.0);
    for (int b = 0; b < qm->nbody; b++) { xquat[4 * b] = 1; xmat[9 * b] = xmat[9 * b + 4] = xmat[9 * b + 8] = 1; }
    for (int b = 0; b < qm->nbody; b++) if (qm->body_mocapid[b] >= 0) {
      for (int k = 0; k < 3; k++) mpos[3 * qm->body_mocapid[b] + k] = qm->body_pos[3 * b + k];
      mquat[4 * qm->body_mocapid[b]] = 1;
    }
    d.xpos = xpos.data(); d.xquat = xquat.data(); d.xmat = xmat.data(); d.subtree_com = com.data(); d.subtree_linvel = linvel.data();
    d.site_xpos = site.data(); d.mocap_pos = mpos.data(); d.mocap_quat = mquat.data();
    // Ground (utilities.cc:556-574) against the scene's group-0 geoms: floor plane, mocap box, tilted ramp, hill sphere
    { const double p0[3] = {20, -20, 1}; CHECK_NEAR(Ground(qm, &d, p0), -0.01, 1e-12); }
    { const double p0[3] = {-2.5, 0.2, 1}; CHECK_NEAR(Ground(qm, &d, p0), 0.3, 1e-12); }
    { const double p0[3] = {3.13, 2.5, 0.6}; CHECK_NEAR(Ground(qm, &d, p0), -0.18 + 0.5 / std::cos(0.2), 1e-9); }
    { const double p0[3] = {6, 6, 1}; CHECK_NEAR(Ground(qm, &d, p0), 0.5, 1e-12); }
    // automatic gait switching: filtered COM speed 0 -> 0.632 -> 0.970 m/s selects Canter once kAutoGaitMinTime has passed
    quad->Reset(qm);
    CHECK(ReinterpretAsInt(quad->parameters[1]) == 1);  // Gait switch = Automatic in the XML
    // the reference's encoding of drop-down fields (task.cc:57, utilities.cc:118-124,225): an integer's BITS in the double; the
    // device is handed plain numbers (Task::NumericParameters)
    CHECK(quad->parameters[1] == ReinterpretAsDouble(1) && quad->parameters[1] < 1e-300 && quad->parameter_is_selection[1] == 1);
    CHECK(quad->NumericParameters()[1] == 1.0 && quad->NumericParameters()[4] == quad->parameters[4] && quad->parameter_is_selection[4] == 0);
    linvel[3 * torso] = 1.0;
    d.time = 0.5; quad->Transition(qm, &d);
    CHECK(ReinterpretAsInt(quad->parameters[0]) == 0);
    d.time = 0.7; quad->Transition(qm, &d);
    CHECK(ReinterpretAsInt(quad->parameters[0]) == 0);  // in the Canter range, but less than 1 s since the last switch (t = 0)
    d.time = 1.2; quad->Transition(qm, &d);
    quad->ResidualState(&ri, &rr);
    CHECK(ReinterpretAsInt(quad->parameters[0]) == 3 && ri[8] == 3);
    CHECK(quad->parameters[4] == 0.4 && quad->parameters[2] == 4 && quad->parameters[3] == 0.05);  // Canter duty, cadence, amplitude
    d.time = 1.3; linvel[3 * torso] = 0.0; quad->Transition(qm, &d);
    CHECK(ReinterpretAsInt(quad->parameters[0]) == 3);  // still waiting
    // Walk: straight line, then a circle about the axis speed / angvel to the left of the torso
    quad->parameters[1] = ReinterpretAsDouble(0);
    xpos[3 * torso] = 1; xpos[3 * torso + 1] = 2; xpos[3 * torso + 2] = 0.3;
    mpos[3 * goal] = 3; mpos[3 * goal + 1] = 2;
    quad->parameters[5] = 0.5; quad->parameters[6] = 0;
    quad->mode = 2;
    d.time = 2.0; quad->Transition(qm, &d);
    quad->ResidualState(&ri, &rr);
    CHECK(ri[0] == 2 && rr[0] == 2.0 && rr[1] == 1 && rr[2] == 2 && rr[4] == 2 && rr[5] == 0 && rr[6] == 0.5 && rr[7] == 0);
    CHECK_NEAR(mpos[3 * goal], 3, 1e-15); CHECK_NEAR(mpos[3 * goal + 1], 2, 1e-15);
    d.time = 3.0; quad->Transition(qm, &d);
    CHECK_NEAR(mpos[3 * goal], 3.5, 1e-15); CHECK_NEAR(mpos[3 * goal + 1], 2, 1e-15);
    quad->parameters[6] = 0.5;  // turn: the walk state resets, axis = torso + (speed / angvel) leftward = (1, 3)
    d.time = 4.0; quad->Transition(qm, &d);
    quad->ResidualState(&ri, &rr);
    CHECK(rr[0] == 4.0 && rr[1] == 1 && rr[2] == 3 && rr[4] == 2.5 && rr[5] == -1 && rr[7] == 0.5);
    d.time = 5.0; quad->Transition(qm, &d);
    CHECK_NEAR(mpos[3 * goal], 1 + std::cos(0.5) * 2.5 + std::sin(0.5), 1e-14);
    CHECK_NEAR(mpos[3 * goal + 1], 3 + std::sin(0.5) * 2.5 - std::cos(0.5), 1e-14);
    // forbidden transition: Walk -> Flip falls back to Quadruped
    quad->mode = 4;
    d.time = 5.1; quad->Transition(qm, &d);
    quad->ResidualState(&ri, &rr);
    CHECK(quad->mode == 0 && ri[0] == 0);
    // Flip from Quadruped: saves the weights and the gait switch, records orientation and ground height, and ends after
    // jump + flight + land time with the goal under the head
    quad->parameters[1] = ReinterpretAsDouble(1);
    d.time = 5.9; quad->Transition(qm, &d);  // automatic switching settles on Stand (filtered speed ~ 0) before the weights are saved
    CHECK(ReinterpretAsInt(quad->parameters[0]) == 0);
    std::vector<double> w0 = quad->weight;
    com[3 * torso] = 6; com[3 * torso + 1] = 6; com[3 * torso + 2] = 0.8;
    xquat[4 * torso] = 0.6; xquat[4 * torso + 3] = 0.8;
    site[3 * head] = 6.2; site[3 * head + 1] = 6.1;
    quad->mode = 4;
    d.time = 6.0; quad->Transition(qm, &d);
    quad->ResidualState(&ri, &rr);
    CHECK(ri[0] == 4 && rr[0] == 6.0 && rr[9] == 0.6 && rr[12] == 0.8);
    CHECK_NEAR(rr[8], 0.5, 1e-12);  // on top of the hill
    CHECK(quad->weight[0] == 0.2 && quad->weight[1] == 5 && quad->weight[2] == 0 && quad->weight[3] == 0 && quad->weight[4] == 0);
    CHECK(quad->weight[5] == 0.005 && quad->weight[6] == 0.1 && ReinterpretAsInt(quad->parameters[1]) == 0);
    const double flip_total = rr[22] + rr[18] + rr[24];
    d.time = 6.0 + 0.5 * flip_total; quad->Transition(qm, &d);
    CHECK(quad->mode == 4);
    d.time = 6.0 + flip_total + 1e-9; quad->Transition(qm, &d);
    quad->ResidualState(&ri, &rr);
    CHECK(quad->mode == 0 && ri[0] == 0 && quad->weight == w0 && ReinterpretAsInt(quad->parameters[1]) == 1);
    CHECK(mpos[3 * goal] == 6.2 && mpos[3 * goal + 1] == 6.1);
    // mjData reset (time going backwards) restarts the phase clock
    d.time = 0.1; quad->Transition(qm, &d);
    quad->ResidualState(&ri, &rr);
    CHECK(rr[13] == 0.1 && rr[14] == 0.1);
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
