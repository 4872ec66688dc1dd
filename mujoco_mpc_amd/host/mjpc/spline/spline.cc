#include "spline.h"

#include <algorithm>
#include <stdexcept>

namespace mjpc::spline {

TimeSpline::TimeSpline(int dim, SplineInterpolation interpolation, int initial_capacity)
    : interpolation_(interpolation), dim_(dim) {
  Reserve(initial_capacity);
}

void TimeSpline::Reserve(int num_nodes) {
  if (num_nodes > 0) {
    times_.reserve(num_nodes);
    values_.reserve((std::size_t)num_nodes * dim_);
  }
}

TimeSpline::Node TimeSpline::NodeAt(int index) {
  return Node(times_[index], values_.data() + (std::size_t)index * dim_, dim_);
}
TimeSpline::ConstNode TimeSpline::NodeAt(int index) const {
  return ConstNode(times_[index], values_.data() + (std::size_t)index * dim_, dim_);
}

void TimeSpline::Clear() {
  times_.clear();
  values_.clear();
}

TimeSpline::Node TimeSpline::AddNode(double time) { return AddNode(time, Span<const double>()); }

TimeSpline::Node TimeSpline::AddNode(double time, Span<const double> v) {
  if (!(v.empty() || (int)v.size() == dim_)) throw std::invalid_argument("TimeSpline::AddNode: wrong number of values");
  const bool at_back = times_.empty() || time > times_.back();
  if (!at_back && !(time < times_.front()))
    throw std::invalid_argument("Adding nodes to the middle of the spline isn't supported.");
  const std::size_t pos = at_back ? times_.size() : 0;
  times_.insert(times_.begin() + pos, time);
  values_.insert(values_.begin() + pos * dim_, (std::size_t)dim_, 0.0);
  if (!v.empty()) std::copy(v.begin(), v.end(), values_.begin() + pos * dim_);
  return NodeAt((int)pos);
}

int TimeSpline::UpperBound(double time) const {
  return (int)(std::upper_bound(times_.begin(), times_.end(), time) - times_.begin());
}

// one-sided difference at the ends, mean of the two adjacent secants inside
double TimeSpline::Slope(int node, int k) const {
  const int last = (int)times_.size() - 1;
  auto secant = [&](int a, int b) {
    return (values_[(std::size_t)b * dim_ + k] - values_[(std::size_t)a * dim_ + k]) / (times_[b] - times_[a]);
  };
  if (node == 0) return secant(0, 1);
  if (node == last) return secant(last - 1, last);
  return 0.5 * secant(node, node + 1) + 0.5 * secant(node - 1, node);
}

void TimeSpline::Sample(double time, Span<double> out) const {
  if ((int)out.size() != dim_) throw std::invalid_argument("TimeSpline::Sample: wrong output size");
  if (times_.empty()) {
    std::fill(out.begin(), out.end(), 0.0);
    return;
  }
  const int up = UpperBound(time);
  if (up == (int)times_.size() || up == 0) {
    const int n = up == 0 ? 0 : up - 1;
    std::copy_n(values_.begin() + (std::size_t)n * dim_, dim_, out.begin());
    return;
  }
  const int lo = up - 1;
  const double span = times_[up] - times_[lo];
  const double t = (time - times_[lo]) / span;
  const double* a = values_.data() + (std::size_t)lo * dim_;
  const double* b = values_.data() + (std::size_t)up * dim_;
  switch (interpolation_) {
    case kZeroSpline:
      std::copy_n(a, dim_, out.begin());
      return;
    case kLinearSpline:
      for (int i = 0; i < dim_; i++) out[i] = a[i] * (1 - t) + b[i] * t;
      return;
    case kCubicSpline: {
      const double h00 = 2.0 * t * t * t - 3.0 * t * t + 1.0;
      const double h10 = (t * t * t - 2.0 * t * t + t) * span;
      const double h01 = -2.0 * t * t * t + 3 * t * t;
      const double h11 = (t * t * t - t * t) * span;
      for (int i = 0; i < dim_; i++) out[i] = h00 * a[i] + h10 * Slope(lo, i) + h01 * b[i] + h11 * Slope(up, i);
      return;
    }
  }
  throw std::logic_error("Unknown interpolation");
}

std::vector<double> TimeSpline::Sample(double time) const {
  std::vector<double> v(dim_);
  Sample(time, Span<double>(v.data(), v.size()));
  return v;
}

int TimeSpline::DiscardBefore(double time) {
  int first_kept = UpperBound(time);
  if (first_kept == 0) return 0;
  first_kept -= 1;                                                   // last node at or before `time`
  if (interpolation_ == kCubicSpline && first_kept > 0) first_kept -= 1;  // its slope needs one more
  times_.erase(times_.begin(), times_.begin() + first_kept);
  values_.erase(values_.begin(), values_.begin() + (std::size_t)first_kept * dim_);
  return first_kept;
}

void TimeSpline::ShiftTime(double start_time) {
  if (times_.empty()) return;
  const double shift = start_time - times_.front();
  for (double& t : times_) t += shift;
}

}  // namespace mjpc::spline
