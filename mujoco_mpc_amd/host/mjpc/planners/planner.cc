#include "planner.h"

#include <algorithm>

namespace mjpc {

// planners/planner.cc:23-33
void Planner::ResizeMjData(const mjModel* model, int num_threads) {
  const size_t want = (size_t)std::max(1, num_threads);
  if (data_.size() > want) data_.erase(data_.begin() + (std::ptrdiff_t)want, data_.end());
  data_.reserve(want);
  while (data_.size() < want) data_.push_back(MakeUniqueMjData(mj_makeData(model)));
}

}  // namespace mjpc
