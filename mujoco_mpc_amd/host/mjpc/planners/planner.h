// mjpc::Planner / mjpc::RankedPlanner (mjpc/planners/planner.h:32-102). The GUI hooks (Traces, GUI,
// Plots) take MuJoCo visualisation types; they are declared over opaque forward declarations so the
// interface is complete without the GUI being in scope (interactive front-end out of scope).
#pragma once
#include <vector>

#include <mujoco/mujoco.h>
#include "../states/state.h"
#include "../task.h"
#include "../threadpool.h"
#include "../trajectory.h"
#include "../utilities.h"

struct mjvScene_;  typedef struct mjvScene_ mjvScene;
struct mjUI_;      typedef struct mjUI_ mjUI;
struct mjvFigure_; typedef struct mjvFigure_ mjvFigure;

namespace mjpc {

inline constexpr int kMaxTrajectory = 128;  // the CPU planners' cap; the GPU planners are not bound by it

class Planner {
 public:
  virtual ~Planner() = default;
  virtual void Initialize(mjModel* model, const Task& task) = 0;
  virtual void Allocate() = 0;
  virtual void Reset(int horizon, const double* initial_repeated_action = nullptr) = 0;
  virtual void SetState(const State& state) = 0;
  virtual void OptimizePolicy(int horizon, ThreadPool& pool) = 0;
  virtual void NominalTrajectory(int horizon, ThreadPool& pool) = 0;
  virtual void ActionFromPolicy(double* action, const double* state, double time, bool use_previous = false) = 0;
  virtual const Trajectory* BestTrajectory() = 0;
  virtual void Traces(mjvScene* scn) = 0;
  virtual void GUI(mjUI& ui) = 0;
  virtual void Plots(mjvFigure* fig_planner, mjvFigure* fig_timer, int planner_shift, int timer_shift, int planning,
                     int* shift) = 0;
  virtual int NumParameters() = 0;

  // planners/planner.h:78-79: one mjData per pool thread. The GPU planners roll nothing out on the host, so a single
  // mjData (the state / control scratch a caller of Trajectory::Rollout hands over) is all they keep; the members stay
  // so that code written against the reference's Planner compiles unchanged.
  std::vector<UniqueMjData> data_;
  void ResizeMjData(const mjModel* model, int num_threads);
};

class RankedPlanner : public Planner {
 public:
  virtual int OptimizePolicyCandidates(int ncandidates, int horizon, ThreadPool& pool) = 0;
  virtual double CandidateScore(int candidate) const = 0;
  virtual void ActionFromCandidatePolicy(double* action, int candidate, const double* state, double time) = 0;
  virtual void CopyCandidateToPolicy(int candidate) = 0;
};

}  // namespace mjpc
