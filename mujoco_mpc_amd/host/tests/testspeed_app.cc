// Headless CLI with the flags of mjpc/testspeed_app.cc (--task, --planner_thread, --steps_per_planning_iteration,
// --total_time) plus --model_dir / --candidates.
#include <cstdlib>
#include <cstring>
#include <string>

#include "mjpc/testspeed.h"

int main(int argc, char** argv) {
  std::string task = "Cartpole";
  int planner_thread = 1, spi = 4, candidates = 0;
  double total_time = 1.0;
  mjpc::TestSpeedOptions opt;
  opt.model_dir = ".";
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&](const char* key) -> const char* {
      const size_t n = std::strlen(key);
      if (a.compare(0, n, key) == 0 && a.size() > n && a[n] == '=') return argv[i] + n + 1;
      return nullptr;
    };
    if (const char* v = val("--task")) task = v;
    else if (const char* v = val("--planner_thread")) planner_thread = std::atoi(v);
    else if (const char* v = val("--steps_per_planning_iteration")) spi = std::atoi(v);
    else if (const char* v = val("--total_time")) total_time = std::atof(v);
    else if (const char* v = val("--model_dir")) opt.model_dir = v;
    else if (const char* v = val("--candidates")) candidates = std::atoi(v);
  }
  opt.num_candidates = candidates;
  const double cost = mjpc::SynchronousPlanningCost(task, planner_thread, spi, total_time, opt);
  return cost < 0 ? 1 : 0;
}
