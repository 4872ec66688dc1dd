// mjpc::spline::TimeSpline with the reference's public interface (mjpc/spline/spline.h): a set of
// time-stamped nodes of `dim` values, sampled with zero-order / linear / cubic-Hermite interpolation.
// This implementation keeps nodes in two plain vectors (times, values) in time order; the reference's
// ring buffer is an allocation detail that no caller can observe.
#pragma once
#include <cstddef>
#include <vector>

namespace mjpc::spline {

enum SplineInterpolation : int { kZeroSpline, kLinearSpline, kCubicSpline };

// Minimal contiguous view (stands in for absl::Span in this dependency-free build).
template <typename T>
class Span {
 public:
  Span() = default;
  Span(T* p, std::size_t n) : p_(p), n_(n) {}
  T* data() const { return p_; }
  std::size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T* begin() const { return p_; }
  T* end() const { return p_ + n_; }
  T& operator[](std::size_t i) const { return p_[i]; }
  T& at(std::size_t i) const { return p_[i]; }

 private:
  T* p_ = nullptr;
  std::size_t n_ = 0;
};

class TimeSpline {
 public:
  explicit TimeSpline(int dim = 0, SplineInterpolation interpolation = kZeroSpline, int initial_capacity = 1);

  template <typename T>
  class NodeT {
   public:
    NodeT() = default;
    NodeT(double time, T* values, int dim) : time_(time), values_(values, dim) {}
    double time() const { return time_; }
    Span<T> values() const { return values_; }

   private:
    double time_ = 0;
    Span<T> values_;
  };
  using Node = NodeT<double>;
  using ConstNode = NodeT<const double>;

  std::size_t Size() const { return times_.size(); }
  Node NodeAt(int index);
  ConstNode NodeAt(int index) const;

  void SetInterpolation(SplineInterpolation interpolation) { interpolation_ = interpolation; }
  SplineInterpolation Interpolation() const { return interpolation_; }
  int Dim() const { return dim_; }
  void Reserve(int num_nodes);

  // interpolated values at `time`; constant extrapolation outside the node range
  void Sample(double time, Span<double> values) const;
  std::vector<double> Sample(double time) const;

  // drops nodes that cannot influence samples at or after `time`; returns how many
  int DiscardBefore(double time);
  // shifts all node times so that the first node is at `start_time`
  void ShiftTime(double start_time);
  void Clear();
  // nodes may only be added before the first or after the last node; throws std::invalid_argument otherwise
  Node AddNode(double time);
  Node AddNode(double time, Span<const double> values);
  Node AddNode(double time, const std::vector<double>& values) {
    return AddNode(time, Span<const double>(values.data(), values.size()));
  }

  const std::vector<double>& times() const { return times_; }
  const std::vector<double>& values() const { return values_; }  // Size() x Dim(), time order

 private:
  int UpperBound(double time) const;
  double Slope(int node, int k) const;
  SplineInterpolation interpolation_;
  int dim_;
  std::vector<double> times_;
  std::vector<double> values_;
};

}  // namespace mjpc::spline
