// Subset of mjpc/utilities.{h,cc} on the rollout path.
#pragma once
#include <chrono>
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

}  // namespace mjpc
