#include "testspeed.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <memory>
#include <vector>

#include "../model_io.h"
#include "planners/gpu_sampling/planner.h"
#include "tasks/tasks.h"
#include "utilities.h"

namespace mjpc {

double SynchronousPlanningCost(std::string task_name, int planner_thread_count, int steps_per_planning_iteration,
                               double total_time, const TestSpeedOptions& opt) {
  std::shared_ptr<Task> task;
  for (auto& t : GetTasks())
    if (SameTaskName(t->Name(), task_name)) task = t;
  if (!task) {
    std::cerr << "Invalid --task flag: '" << task_name << "'. Valid values:\n";
    for (auto& t : GetTasks()) std::cerr << "  " << t->Name() << "\n";
    return -1;
  }
  std::string file = task_name == "ParticleCopy" ? std::string("Particle") : task_name;
  file.erase(std::remove(file.begin(), file.end(), ' '), file.end());  // "Humanoid Track" -> HumanoidTrack.mjpx
  const std::string blob = opt.model_dir + "/" + file + ".mjpx";
  std::unique_ptr<ModelStorage> storage;
  try {
    storage = ModelStorage::Load(blob);
  } catch (const std::exception& e) {
    std::cerr << e.what() << "\n";
    return -1;
  }
  mjModel* model = storage->model();
  task->Reset(model);

  // initial condition: keyframe "home" if present (testspeed.cc:71-76)
  const int ds = model->nq + model->nv + model->na;
  std::vector<double> qpos(model->qpos0, model->qpos0 + model->nq), qvel(model->nv, 0.0), ctrl(model->nu, 0.0);
  if (const double* home = KeyQPosByName(model, "home")) qpos.assign(home, home + model->nq);
  std::vector<double> mocap_pos(3 * (size_t)model->nmocap), mocap_quat(4 * (size_t)model->nmocap);
  for (int b = 0; b < model->nbody; b++)
    if (model->body_mocapid[b] >= 0) {
      mju_copy(mocap_pos.data() + 3 * model->body_mocapid[b], model->body_pos + 3 * b, 3);
      mju_copy(mocap_quat.data() + 4 * model->body_mocapid[b], model->body_quat + 4 * b, 4);
    }
  double time = 0.0;

  try {
    // planner (Agent::Initialize/Allocate/Reset, agent.cc:74-130)
    GpuSamplingPlanner planner(opt.device);
    planner.Initialize(model, *task);
    if (opt.num_candidates > 0) planner.num_trajectory_ = opt.num_candidates;
    planner.Allocate();
    const double agent_timestep = GetNumberOrDefault(model->opt.timestep, model, "agent_timestep");
    const double agent_horizon = GetNumberOrDefault(0.5, model, "agent_horizon");
    const int steps = std::max(1, std::min((int)(agent_horizon / agent_timestep + 1), kMaxTrajectoryHorizon));  // agent.cc:288-293
    planner.Reset(steps, ctrl.data());
    // the simulated "real world": same model at its own timestep, stepped on the device
    mjModel sim_model = *model;
    sim_model.nnumeric = 0;  // no agent_timestep override for the simulation copy
    gpu::Context sim(&sim_model, *task, opt.device);
    Trajectory one;
    one.Initialize(ds, model->nu, task->num_residual, task->num_trace, 2);
    one.Allocate(2);

    ThreadPool pool(std::max(1, planner_thread_count));  // accepted for interface parity; rollouts run on the GPU
    State state;
    state.Allocate(model);
    const int total_steps = (int)std::ceil(total_time / model->opt.timestep);
    const auto loop_start = std::chrono::steady_clock::now();
    double total_cost = 0;
    int plans = 0;
    double plan_us = 0;
    std::vector<double> full_state(ds), mocap7(7 * (size_t)model->nmocap);
    gpu::KinematicsBuffers kin;
    kin.Allocate(model);
    auto pack_mocap = [&]() {
      for (int k = 0; k < model->nmocap; k++) {
        mju_copy(mocap7.data() + 7 * k, mocap_pos.data() + 3 * k, 3);
        mju_copy(mocap7.data() + 7 * k + 3, mocap_quat.data() + 4 * k, 4);
      }
    };
    for (int i = 0; i < total_steps; i++) {
      {  // agent.ActiveTask()->Transition(model, data) (testspeed.cc:97): tasks may edit qpos / qvel / mocap_pos and read the
         // kinematics of the current state, which the reference's mjData holds after mj_step and this host asks the device for
        mjData d{};
        d.time = time;
        d.qpos = qpos.data(); d.qvel = qvel.data(); d.mocap_pos = mocap_pos.data(); d.mocap_quat = mocap_quat.data();
        state.Set(model, qpos.data(), qvel.data(), nullptr, mocap_pos.data(), mocap_quat.data(), nullptr, time);
        pack_mocap();
        sim.Check(mjpcx_set_state(sim.handle(), state.state().data(), time, mocap7.data(), nullptr));
        if (sim.Kinematics(&kin)) kin.Attach(&d);
        task->Transition(model, &d);
      }
      state.Set(model, qpos.data(), qvel.data(), nullptr, mocap_pos.data(), mocap_quat.data(), nullptr, time);
      planner.ActionFromPolicy(ctrl.data(), state.state().data(), time);
      // mj_step of the simulation copy + the stage cost at the pre-step state
      pack_mocap();
      sim.SyncTask(*task);
      sim.Check(mjpcx_set_state(sim.handle(), state.state().data(), time, mocap7.data(), nullptr));
      sim.Check(mjpcx_rollout_splines(sim.handle(), 1, 2, 1, MJPCX_SPLINE_ZERO, &time, ctrl.data()));
      sim.FetchTrajectory(0, &one);
      if (one.failure) { std::cerr << "simulation diverged at step " << i << "\n"; return -1; }
      total_cost += one.costs[0];
      mju_copy(qpos.data(), one.states.data() + ds, model->nq);
      mju_copy(qvel.data(), one.states.data() + ds + model->nq, model->nv);
      time = one.times[1];
      if (i % steps_per_planning_iteration == 0) {  // Agent::PlanIteration (agent.cc:283-357)
        const auto plan_start = std::chrono::steady_clock::now();
        state.Set(model, qpos.data(), qvel.data(), nullptr, mocap_pos.data(), mocap_quat.data(), nullptr, time);
        planner.SetState(state);
        planner.OptimizePolicy(steps, pool);
        plan_us += GetDuration(plan_start);
        plans++;
      }
    }
    const double wall = GetDuration(loop_start) * 1e-6;
    if (opt.verbose) {
      std::printf("Total wall time (%d simulation steps, %d plan iterations of %d candidates x %d steps): %.3f s (%.2fx realtime)\n",
                  total_steps, plans, planner.num_trajectory_, steps, wall, total_time / wall);
      std::printf("Mean plan iteration: %.1f us; kernel: %s\n", plans ? plan_us / plans : 0.0, "see mjpcx_kernel_name");
      std::printf("Average cost per step (lower is better): %.6f\n", total_cost / total_steps);
    }
    return total_cost / total_steps;
  } catch (const gpu::Error& e) {
    std::cerr << "GPU path failed: " << e.what() << "\n";
    return -1;
  }
}

}  // namespace mjpc
