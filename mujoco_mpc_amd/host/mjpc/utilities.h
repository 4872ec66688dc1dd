// Subset of mjpc/utilities.{h,cc} on the rollout path.
#pragma once
#include <chrono>
#include <cmath>
#include <optional>
#include <string_view>
#include <vector>

#include "../mujoco_min.h"

namespace mjpc {

// custom <numeric> lookup, utilities.h:40-68
inline double* GetCustomNumericData(const mjModel* m, std::string_view name) {
  for (int i = 0; i < m->nnumeric; i++)
    if (name == std::string_view(m->names + m->name_numericadr[i])) return m->numeric_data + m->numeric_adr[i];
  return nullptr;
}
inline int GetCustomNumericSize(const mjModel* m, std::string_view name) {
  for (int i = 0; i < m->nnumeric; i++)
    if (name == std::string_view(m->names + m->name_numericadr[i])) return m->numeric_size[i];
  return 0;
}
template <typename T>
std::optional<T> GetNumber(const mjModel* m, std::string_view name) {
  const double* d = GetCustomNumericData(m, name);
  if (!d) return std::nullopt;
  return static_cast<T>(d[0]);
}
template <typename T>
T GetNumberOrDefault(T fallback, const mjModel* m, std::string_view name) {
  return GetNumber<T>(m, name).value_or(fallback);
}

// bounds = [lo0, hi0, lo1, hi1, ...], utilities.cc:112-116
inline void Clamp(double* x, const double* bounds, int n) {
  for (int i = 0; i < n; i++) x[i] = mju_clip(x[i], bounds[2 * i], bounds[2 * i + 1]);
}

template <typename T>
T* DataAt(std::vector<T>& v, typename std::vector<T>::size_type i) { return v.data() + i; }
template <typename T>
const T* DataAt(const std::vector<T>& v, typename std::vector<T>::size_type i) { return v.data() + i; }

// microseconds since `start`, utilities.cc:1298
inline double GetDuration(std::chrono::steady_clock::time_point start) {
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - start).count();
}

// keyframe qpos by name, utilities.cc:288-296
inline double* KeyQPosByName(const mjModel* m, std::string_view name) {
  for (int i = 0; i < m->nkey; i++)
    if (name == std::string_view(m->names + m->name_keyadr[i])) return m->key_qpos + (size_t)m->nq * i;
  return nullptr;
}

// ---- interval search and interpolation over a sorted time grid, utilities.h:124-144, utilities.cc:304-344
template <typename T>
void FindInterval(int* bounds, const T* sequence, double value, int length) {
  // upper_bound: first element strictly greater than value
  int upper = 0;
  while (upper < length && !(value < sequence[upper])) upper++;
  const int lower = upper - 1;
  if (lower < 0) { bounds[0] = 0; bounds[1] = 0; }
  else if (lower > length - 1) { bounds[0] = length - 1; bounds[1] = length - 1; }
  else { bounds[0] = mjMAX(lower, 0); bounds[1] = mjMIN(upper, length - 1); }
}
inline void ZeroInterpolation(double* output, double x, const double* xs, const double* ys, int dim, int length) {
  int bounds[2];
  FindInterval(bounds, xs, x, length);
  mju_copy(output, ys + (size_t)dim * bounds[0], dim);
}
inline void LinearInterpolation(double* output, double x, const double* xs, const double* ys, int dim, int length) {
  int bounds[2];
  FindInterval(bounds, xs, x, length);
  if (bounds[0] == bounds[1]) { mju_copy(output, ys + (size_t)dim * bounds[0], dim); return; }
  const double t = (x - xs[bounds[0]]) / (xs[bounds[1]] - xs[bounds[0]]);
  for (int i = 0; i < dim; i++) output[i] = ys[(size_t)dim * bounds[0] + i] * (1.0 - t) + ys[(size_t)dim * bounds[1] + i] * t;
}
// log-spaced values from min_value up to max_value, utilities.cc:819-826
inline void LogScale(double* values, double max_value, double min_value, int steps) {
  const double step = (std::log(max_value) - std::log(min_value)) / (steps > 1 ? steps - 1 : 1);
  for (int i = 0; i < steps; i++) values[i] = std::exp(std::log(min_value) + i * step);
}
// state difference for models without quaternion joints (nq == nv), utilities.cc:543-553
inline void StateDiff(const mjModel* m, double* dx, const double* s1, const double* s2, double h) {
  const int n = m->nq + m->nv + m->na;  // == 2 nv + na
  for (int i = 0; i < n; i++) dx[i] = (s2[i] - s1[i]) / h;
}

}  // namespace mjpc
