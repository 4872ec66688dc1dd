#include "include.h"

#include "gpu_cross_entropy/planner.h"
#include "gpu_ilqg/planner.h"
#include "gpu_robust/robust_planner.h"
#include "gpu_sample_gradient/planner.h"
#include "gpu_sampling/planner.h"

namespace mjpc {

const char kPlannerNames[] = "Sampling\nGradient\niLQG\niLQS\nRobust Sampling\nCross Entropy\nSample Gradient";

std::vector<std::unique_ptr<Planner>> LoadPlanners(int device, int precision) {
  std::vector<std::unique_ptr<Planner>> planners(kNumPlannerTypes);
  planners[kSamplingPlanner] = std::make_unique<GpuSamplingPlanner>(device, precision);
  planners[kILQGPlanner] = std::make_unique<GpuILQGPlanner>(device, precision);
  planners[kRobustPlanner] = std::make_unique<GpuRobustPlanner>(std::make_unique<GpuSamplingPlanner>(device, precision), device, precision);
  planners[kCrossEntropyPlanner] = std::make_unique<GpuCrossEntropyPlanner>(device, precision);
  planners[kSampleGradientPlanner] = std::make_unique<GpuSampleGradientPlanner>(device, precision);
  return planners;
}

}  // namespace mjpc
