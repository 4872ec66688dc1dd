// mjpc/planners/include.{h,cc}: the planner registry. Slots follow the reference's PlannerType enum so that a task
// XML's `agent_planner` numeric selects the same algorithm; slots whose planner has no GPU port are empty (nullptr).
#pragma once
#include <memory>
#include <vector>

#include "planner.h"

namespace mjpc {

enum PlannerType : int {
  kSamplingPlanner = 0,   // -> GpuSamplingPlanner
  kGradientPlanner,       // not ported (empty slot)
  kILQGPlanner,           // -> GpuILQGPlanner
  kILQSPlanner,           // not ported
  kRobustPlanner,         // -> GpuRobustPlanner(GpuSamplingPlanner)
  kCrossEntropyPlanner,   // -> GpuCrossEntropyPlanner
  kSampleGradientPlanner, // -> GpuSampleGradientPlanner
  kNumPlannerTypes
};
extern const char kPlannerNames[];

std::vector<std::unique_ptr<Planner>> LoadPlanners(int device = 0, int precision = 64);

}  // namespace mjpc
