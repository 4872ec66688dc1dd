// Subset of mjpc/utilities.{h,cc} on the rollout path.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <cstdint>
#include <chrono>
#include <cmath>
#include <memory>
#include <optional>
#include <string_view>
#include <vector>

#include <mujoco/mujoco.h>

namespace mjpc {

// custom <numeric> lookup, utilities.h:40-68
// task names compared with spaces removed: "QuadrupedFlat" (model file / Python registry spelling) == "Quadruped Flat"
inline bool SameTaskName(std::string_view a, std::string_view b) {
  std::string x(a), y(b);
  x.erase(std::remove(x.begin(), x.end(), ' '), x.end());
  y.erase(std::remove(y.begin(), y.end(), ' '), y.end());
  return x == y;
}
// custom text field by name (utilities.cc:186-198); nullptr if the model has none of that name
inline char* GetCustomTextData(const mjModel* m, std::string_view name) {
  for (int i = 0; i < m->ntext; i++)
    if (name == std::string_view(m->names + m->name_textadr[i])) return m->text_data + m->text_adr[i];
  return nullptr;
}
inline std::vector<std::string> SplitBar(std::string_view s, bool skip_empty) {
  std::vector<std::string> out;
  size_t b = 0;
  while (b <= s.size()) {
    size_t e = s.find('|', b);
    if (e == std::string_view::npos) e = s.size();
    if (e > b || !skip_empty) out.emplace_back(s.substr(b, e - b));
    b = e + 1;
  }
  return out;
}
// A double field of Task::parameters that stores an integer (utilities.cc:118-124): the reference keeps the BITS of an int64 in
// the double ("residual_select_*" drop-downs), read back through its low 32 bits. Same encoding here, so that code written
// against the reference (ReinterpretAsInt(parameters[...]) in a task's Residual / Transition) sees the same values.
inline int ReinterpretAsInt(double value) { int i; std::memcpy(&i, &value, sizeof i); return i; }
inline double ReinterpretAsDouble(std::int64_t value) { double d; std::memcpy(&d, &value, sizeof d); return d; }
// utilities.cc:225-229: the XML's numeric value of a "residual_select_*" field, stored as an integer's bits
inline double DefaultResidualSelection(const mjModel* m, int numeric_index) {
  return ReinterpretAsDouble((std::int64_t)m->numeric_data[m->numeric_adr[numeric_index]]);
}
// drop-down selections (utilities.cc:142-181): "residual_select_<name>" holds the index into the '|'-separated custom text
// "residual_list_<name>"
inline std::string ResidualSelection(const mjModel* m, std::string_view name, double residual_parameter) {
  const char* options = GetCustomTextData(m, "residual_list_" + std::string(name));
  if (!options) return "";
  const std::vector<std::string> v = SplitBar(options, false);
  const int i = ReinterpretAsInt(residual_parameter);
  return i >= 0 && i < (int)v.size() ? v[i] : "";
}
inline double ResidualParameterFromSelection(const mjModel* m, std::string_view name, std::string_view value) {
  const char* options = GetCustomTextData(m, "residual_list_" + std::string(name));
  if (!options) return 0;
  const std::vector<std::string> v = SplitBar(options, false);
  for (size_t i = 0; i < v.size(); i++) if (v[i] == value) return ReinterpretAsDouble((std::int64_t)i);
  return 0;
}

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

// utilities.h:261-265
using UniqueMjData = std::unique_ptr<mjData, void (*)(mjData*)>;
inline UniqueMjData MakeUniqueMjData(mjData* d) { return UniqueMjData(d, mj_deleteData); }

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
// cubic Hermite interpolation with finite-difference slopes (utilities.cc:336-422): the slope at a grid point is the mean of
// the two neighbouring secants, one-sided at the ends of the grid and zero on a two-point grid
inline double FiniteDifferenceSlope(double x, const double* xs, const double* ys, int dim, int length, int i) {
  int b[2];
  FindInterval(b, xs, x, length);
  const auto secant = [&](int hi, int lo) { return (ys[(size_t)dim * hi + i] - ys[(size_t)dim * lo + i]) / (xs[hi] - xs[lo]); };
  if (b[0] == 0 && b[1] == 0) return length > 2 ? secant(1, 0) : 0.0;                                        // below the grid
  if (b[0] == length - 1 && b[1] == length - 1) return length > 2 ? secant(length - 1, length - 2) : 0.0;   // above it
  if (b[0] == 0) return secant(b[1], 0);                                                                    // first interval
  return 0.5 * secant(b[1], b[0]) + 0.5 * secant(b[0], b[0] - 1);
}
inline void CubicInterpolation(double* output, double x, const double* xs, const double* ys, int dim, int length) {
  int b[2];
  FindInterval(b, xs, x, length);
  if (b[0] == b[1]) { mju_copy(output, ys + (size_t)dim * b[0], dim); return; }
  const double span = xs[b[1]] - xs[b[0]], t = (x - xs[b[0]]) / span;
  const double t2 = t * t, t3 = t2 * t;
  const double c0 = 2.0 * t3 - 3.0 * t2 + 1.0, c1 = (t3 - 2.0 * t2 + t) * span, c2 = -2.0 * t3 + 3 * t2, c3 = (t3 - t2) * span;
  for (int i = 0; i < dim; i++) {
    const double m0 = FiniteDifferenceSlope(xs[b[0]], xs, ys, dim, length, i), m1 = FiniteDifferenceSlope(xs[b[1]], xs, ys, dim, length, i);
    output[i] = c0 * ys[(size_t)dim * b[0] + i] + c1 * m0 + c2 * ys[(size_t)dim * b[1] + i] + c3 * m1;
  }
}
// Ground (utilities.cc:556-574): global height of the nearest group-0 geom under `pos`, by a ray cast straight down from
// 0.5 m above it. Host stand-in for mj_ray over the geoms a planning scene has below the robot: planes, spheres and boxes on
// the world body or on mocap bodies (their pose = mocap pose o geom pose). Returns pos[2] + 0.5 - distance; throws if
// nothing is hit, where the reference calls mju_error.
inline double Ground(const mjModel* m, const mjData* d, const double pos[3]) {
  const double height_offset = 0.5;
  const double o[3] = {pos[0], pos[1], pos[2] + height_offset};
  double best = -1;
  auto quat2mat = [](const double* q, double* R) {
    const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q11 = q[1] * q[1], q12 = q[1] * q[2],
                 q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
    R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
    R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02); R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01); R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
  };
  for (int g = 0; g < m->ngeom; g++) {
    if (m->geom_group[g] != 0) continue;
    const int b = m->geom_bodyid[g];
    const bool world = b == 0, mocap = b > 0 && m->body_mocapid[b] >= 0;
    if (!world && !mocap) continue;
    // geom frame in the world
    double bp[3] = {0, 0, 0}, bq[4] = {1, 0, 0, 0};
    if (mocap) {
      if (!d->mocap_pos || !d->mocap_quat) continue;
      for (int k = 0; k < 3; k++) bp[k] = d->mocap_pos[3 * m->body_mocapid[b] + k];
      for (int k = 0; k < 4; k++) bq[k] = d->mocap_quat[4 * m->body_mocapid[b] + k];
    }
    double Rb[9], Rg[9], R[9], gp[3];
    quat2mat(bq, Rb);
    quat2mat(m->geom_quat + 4 * g, Rg);
    for (int i = 0; i < 3; i++) {
      gp[i] = bp[i] + Rb[3 * i] * m->geom_pos[3 * g] + Rb[3 * i + 1] * m->geom_pos[3 * g + 1] + Rb[3 * i + 2] * m->geom_pos[3 * g + 2];
      for (int j = 0; j < 3; j++) R[3 * i + j] = Rb[3 * i] * Rg[j] + Rb[3 * i + 1] * Rg[3 + j] + Rb[3 * i + 2] * Rg[6 + j];
    }
    const double* size = m->geom_size + 3 * g;
    // ray in the geom frame: origin lo, direction ld (world direction (0, 0, -1))
    double rel[3] = {o[0] - gp[0], o[1] - gp[1], o[2] - gp[2]}, lo[3], ld[3];
    for (int j = 0; j < 3; j++) {
      lo[j] = R[j] * rel[0] + R[3 + j] * rel[1] + R[6 + j] * rel[2];
      ld[j] = -R[6 + j];
    }
    double dist = -1;
    if (m->geom_type[g] == 0) {  // plane z = 0 (infinite when size is 0, else bounded)
      if (ld[2] < -1e-15 && lo[2] > 0) {
        const double t = -lo[2] / ld[2], x = lo[0] + t * ld[0], y = lo[1] + t * ld[1];
        if ((size[0] <= 0 || std::fabs(x) <= size[0]) && (size[1] <= 0 || std::fabs(y) <= size[1])) dist = t;
      }
    } else if (m->geom_type[g] == 2) {  // sphere
      const double bq2 = lo[0] * ld[0] + lo[1] * ld[1] + lo[2] * ld[2], c = lo[0] * lo[0] + lo[1] * lo[1] + lo[2] * lo[2] - size[0] * size[0];
      const double disc = bq2 * bq2 - c;
      if (disc >= 0) { const double t = -bq2 - std::sqrt(disc); if (t >= 0) dist = t; }
    } else if (m->geom_type[g] == 6) {  // box: slabs
      double tmin = -1e300, tmax = 1e300;
      bool hit = true;
      for (int j = 0; j < 3 && hit; j++) {
        if (std::fabs(ld[j]) < 1e-15) { if (std::fabs(lo[j]) > size[j]) hit = false; continue; }
        double t1 = (-size[j] - lo[j]) / ld[j], t2 = (size[j] - lo[j]) / ld[j];
        if (t1 > t2) std::swap(t1, t2);
        tmin = std::max(tmin, t1); tmax = std::min(tmax, t2);
        if (tmin > tmax) hit = false;
      }
      if (hit && tmin >= 0) dist = tmin;
    }
    if (dist >= 0 && (best < 0 || dist < best)) best = dist;
  }
  if (best < 0) throw std::runtime_error("no group 0 geom detected by raycast");
  return pos[2] + height_offset - best;
}
// Philox4x32-10 + Box-Muller as specified in include/mjpcx.h (the generator the device kernels use for candidate noise):
// two standard normals for (seed, candidate, pair, iteration). Host planners that need the raw noise (SampleGradient) draw
// it here instead of from the reference's function-local absl::BitGen.
inline void HostGaussianPair(std::uint64_t seed, std::uint32_t cand, std::uint32_t pair, std::uint32_t iter, double z[2]) {
  std::uint32_t c0 = cand, c1 = pair, c2 = iter, c3 = 0u, k0 = (std::uint32_t)seed, k1 = (std::uint32_t)(seed >> 32);
  for (int r = 0; r < 10; r++) {
    const std::uint64_t p0 = (std::uint64_t)0xD2511F53u * c0, p1 = (std::uint64_t)0xCD9E8D57u * c2;
    const std::uint32_t n0 = (std::uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (std::uint32_t)p1, n2 = (std::uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (std::uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  auto u53 = [](std::uint32_t hi, std::uint32_t lo) {
    const std::uint64_t k = (((std::uint64_t)hi << 32) | lo) >> 11;
    return ((double)k + 0.5) * (1.0 / 9007199254740992.0);
  };
  const double u1 = u53(c0, c1), u2 = u53(c2, c3);
  const double r = std::sqrt(-2.0 * std::log(u1));
  z[0] = r * std::cos(6.283185307179586476925286766559 * u2);
  z[1] = r * std::sin(6.283185307179586476925286766559 * u2);
}
// log-spaced values from min_value up to max_value, utilities.cc:819-826
inline void LogScale(double* values, double max_value, double min_value, int steps) {
  const double step = (std::log(max_value) - std::log(min_value)) / (steps > 1 ? steps - 1 : 1);
  for (int i = 0; i < steps; i++) values[i] = std::exp(std::log(min_value) + i * step);
}
// StateDiff (utilities.cc:543-553): dx = (s2 - s1) / h in the tangent space: mj_differentiatePos for the positions
// (free-joint translations and scalar joints subtract; quaternions give the rotation vector of q1^-1 q2), velocities
// subtract. dx has 2 nv + na entries.
inline void StateDiff(const mjModel* m, double* dx, const double* s1, const double* s2, double h) {
  const int nq = m->nq, nv = m->nv;
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    const int t = m->jnt_type[j];
    if (t == mjJNT_FREE) { for (int k = 0; k < 3; k++) dx[da + k] = (s2[qa + k] - s1[qa + k]) / h; qa += 3; da += 3; }
    if (t == mjJNT_FREE || t == mjJNT_BALL) {  // mju_subQuat + mju_quat2Vel
      const double* b = s1 + qa; const double* a = s2 + qa;
      const double n[4] = {b[0], -b[1], -b[2], -b[3]};
      const double qd[4] = {n[0] * a[0] - n[1] * a[1] - n[2] * a[2] - n[3] * a[3], n[0] * a[1] + n[1] * a[0] + n[2] * a[3] - n[3] * a[2],
                            n[0] * a[2] - n[1] * a[3] + n[2] * a[0] + n[3] * a[1], n[0] * a[3] + n[1] * a[2] - n[2] * a[1] + n[3] * a[0]};
      double ax[3] = {qd[1], qd[2], qd[3]};
      const double sn = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      if (sn > 1e-15) for (double& v : ax) v /= sn;
      double speed = 2 * std::atan2(sn, qd[0]);
      if (speed > 3.14159265358979323846) speed -= 2 * 3.14159265358979323846;
      for (int k = 0; k < 3; k++) dx[da + k] = ax[k] * speed / h;
    } else {
      dx[da] = (s2[qa] - s1[qa]) / h;
    }
  }
  for (int i = 0; i < nv + m->na; i++) dx[nv + i] = (s2[nq + i] - s1[nq + i]) / h;
}
// mj_normalizeQuat on every quaternion of a qpos-layout vector (ilqg/policy.cc:118-125)
inline void NormalizeStateQuaternions(const mjModel* m, double* qpos) {
  for (int j = 0; j < m->njnt; j++) {
    const int t = m->jnt_type[j];
    if (t != mjJNT_FREE && t != mjJNT_BALL) continue;
    double* q = qpos + m->jnt_qposadr[j] + (t == mjJNT_FREE ? 3 : 0);
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else for (int k = 0; k < 4; k++) q[k] /= n;
  }
}

}  // namespace mjpc
