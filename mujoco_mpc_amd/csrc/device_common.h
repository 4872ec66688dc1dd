// device_common.h -- device-side model/task structs and small math helpers shared by
// the gfx950 rollout kernels. Everything here is written for CDNA4 directly (64-wide
// wavefronts, wave-uniform model constants through the scalar cache); no CUDA/host
// dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mjpcx {

constexpr int kJntFree = 0, kJntBall = 1, kJntSlide = 2, kJntHinge = 3;
constexpr double kMinVal = 1e-15;  // mjMINVAL
constexpr double kMaxVal = 1e10;   // mjMAXVAL
constexpr double kMinImp = 0.0001, kMaxImp = 0.9999;  // mjMINIMP / mjMAXIMP: clip range of solimp's d0, d_width, midpoint
constexpr double kLsTolerance = 0.01;  // mjOption.ls_tolerance default: line-search gradient tolerance relative to the solver tolerance
constexpr double kMaxReturn = 1.0e6;  // kMaxReturnValue, mjpc/trajectory.cc:29

// Capacity of the lane-per-candidate ("small model") kernel family.
// LaneModel + LaneTask travel BY VALUE in the kernel-argument segment (<= 4 KiB), so the
// capacities are sized to keep sizeof(LaneModel<double>) + sizeof(LaneTask<double>) ~ 3 KiB.
constexpr int kLaneMaxBody = 6, kLaneMaxDof = 6, kLaneMaxAct = 6, kLaneMaxSite = 4;
constexpr int kLaneMaxMocap = 2, kLaneMaxTerm = 8, kLaneMaxParam = 8;

// Model constants of a slide/hinge tree, in the compute precision T. Passed by value as a
// kernel argument: the kernarg segment is constant address space, so every access is an
// s_load through the scalar cache into SGPRs (no VGPRs, no LDS, no aliasing with the
// kernel's own global stores -- which is what forces vector re-loads through a pointer).
template <typename T>
struct LaneModel {
  T timestep, gravity[3], solver_tolerance, meaninertia;
  int disableflags, solver_iterations;
  T body_pos[kLaneMaxBody][3], body_quat[kLaneMaxBody][4];
  T body_ipos[kLaneMaxBody][3], body_iquat[kLaneMaxBody][4];
  T body_mass[kLaneMaxBody], body_inertia[kLaneMaxBody][3];
  T root_invmass[kLaneMaxBody];  // 1 / subtree mass, for tree roots
  T jnt_pos[kLaneMaxDof][3], jnt_axis[kLaneMaxDof][3];
  T jnt_stiffness[kLaneMaxDof], jnt_range[kLaneMaxDof][2], jnt_margin[kLaneMaxDof];
  T jnt_solref[kLaneMaxDof][2], jnt_solimp[kLaneMaxDof][5];
  T qpos0[kLaneMaxDof], qpos_spring[kLaneMaxDof];
  T dof_armature[kLaneMaxDof], dof_damping[kLaneMaxDof], dof_invweight0[kLaneMaxDof];
  int any_damping;
  int integrator;  // MJPCX_INT_EULER / MJPCX_INT_RK4
  T site_pos[kLaneMaxSite][3];
  T act_gear[kLaneMaxAct], act_gain[kLaneMaxAct], act_bias[kLaneMaxAct][3];
  T act_ctrlrange[kLaneMaxAct][2], act_forcerange[kLaneMaxAct][2];
  int act_biastype[kLaneMaxAct], act_ctrllimited[kLaneMaxAct], act_forcelimited[kLaneMaxAct];
};

// Per-plan task parameters (the frozen ResidualFn copy, mjpc/agent.cc:319) and the
// per-plan initial condition (Planner::SetState).
template <typename T>
struct LaneTask {
  int norm[kLaneMaxTerm];
  T weight[kLaneMaxTerm], norm_p[kLaneMaxTerm], norm_q[kLaneMaxTerm];
  T parameters[kLaneMaxParam];
  T risk;
  // initial condition
  T qpos[kLaneMaxDof], qvel[kLaneMaxDof];
  T time;
  T mocap_pos[kLaneMaxMocap][3], mocap_quat[kLaneMaxMocap][4];
};

// ---- small fixed-size math; all indices are compile-time after unrolling ----
template <typename T> __device__ __forceinline__ void quat_mul(T (&r)[4], const T (&a)[4], const T (&b)[4]) {
  T t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  T t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  T t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  T t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
template <typename T> __device__ __forceinline__ void quat_to_mat(T (&m)[9], const T (&q)[4]) {
  T q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  T q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  T q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03); m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
template <typename T> __device__ __forceinline__ void mat_vec(T (&r)[3], const T (&m)[9], const T (&v)[3]) {
  T x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  T y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  T z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> __device__ __forceinline__ void cross3(T (&r)[3], const T* a, const T* b) {
  T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> __device__ __forceinline__ void normalize4(T (&q)[4]) {
  T n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < T(kMinVal)) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  T s = T(1) / n;
  q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
// spatial inertia about the subtree-com reference point: [Ixx Iyy Izz Ixy Ixz Iyz, m*d, m]
template <typename T>
__device__ __forceinline__ void inert_com(T (&res)[10], const T (&inert)[3], const T (&mat)[9], const T (&dif)[3], T mass) {
  T t0 = inert[0] * mat[0], t1 = inert[0] * mat[3], t2 = inert[0] * mat[6];
  T t3 = inert[1] * mat[1], t4 = inert[1] * mat[4], t5 = inert[1] * mat[7];
  T t6 = inert[2] * mat[2], t7 = inert[2] * mat[5], t8 = inert[2] * mat[8];
  res[0] = mat[0] * t0 + mat[1] * t3 + mat[2] * t6 + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] = mat[3] * t1 + mat[4] * t4 + mat[5] * t7 + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] = mat[6] * t2 + mat[7] * t5 + mat[8] * t8 + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] = mat[0] * t1 + mat[1] * t4 + mat[2] * t7 - mass * dif[0] * dif[1];
  res[4] = mat[0] * t2 + mat[1] * t5 + mat[2] * t8 - mass * dif[0] * dif[2];
  res[5] = mat[3] * t2 + mat[4] * t5 + mat[5] * t8 - mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2];
  res[9] = mass;
}
template <typename T> __device__ __forceinline__ void mul_inert_vec(T (&res)[6], const T (&i)[10], const T (&v)[6]) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
template <typename T> __device__ __forceinline__ void cross_motion(T (&res)[6], const T (&vel)[6], const T (&v)[6]) {
  T a[3], b[3], c[3];
  cross3(a, &vel[0], &v[0]);
  cross3(b, &vel[0], &v[3]);
  cross3(c, &vel[3], &v[0]);
  res[0] = a[0]; res[1] = a[1]; res[2] = a[2];
  res[3] = b[0] + c[0]; res[4] = b[1] + c[1]; res[5] = b[2] + c[2];
}
template <typename T> __device__ __forceinline__ void cross_force(T (&res)[6], const T (&vel)[6], const T (&f)[6]) {
  T a[3], b[3], c[3];
  cross3(a, &vel[0], &f[0]);
  cross3(b, &vel[3], &f[3]);
  cross3(c, &vel[0], &f[3]);
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  res[3] = c[0]; res[4] = c[1]; res[5] = c[2];
}
template <typename T> __device__ __forceinline__ T dot6(const T (&a)[6], const T (&b)[6]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
// mju_isBad (NaN, or |x| > mjMAXVAL) as an integer test on the bit pattern: for non-negative
// IEEE values the bit patterns order like the values, and Inf/NaN patterns are above every
// finite one. Immune to the compiler's floating-point assumptions (finite-math TUs).
__device__ __forceinline__ bool is_bad(double x) {
  const uint64_t mag = (uint64_t)__double_as_longlong(x) & 0x7fffffffffffffffull;
  return mag > 0x4202A05F20000000ull;  // bits of 1e10
}
__device__ __forceinline__ bool is_bad(float x) {
  const uint32_t mag = (uint32_t)__float_as_int(x) & 0x7fffffffu;
  return mag > 0x501502F9u;  // bits of 1e10f
}
__device__ __forceinline__ void sincos_t(double x, double& s, double& c) { sincos(x, &s, &c); }
__device__ __forceinline__ void sincos_t(float x, float& s, float& c) { sincosf(x, &s, &c); }
template <typename T> __device__ __forceinline__ T clampv(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ---- Philox4x32-10 + Box-Muller, exactly as specified in include/mjpcx.h ----
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
  uint64_t k = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)k + 0.5) * (1.0 / 9007199254740992.0);
}
// always evaluated in fp64 so that fp32 and fp64 kernels draw the same candidates
__device__ __forceinline__ void gaussian_pair(uint64_t seed, uint32_t cand, uint32_t pair, uint32_t iter, double (&z)[2]) {
  uint32_t o[4];
  philox4x32_10(cand, pair, iter, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  double u1 = u53(o[0], o[1]), u2 = u53(o[2], o[3]);
  double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincos(6.283185307179586476925286766559 * u2, &s, &c);
  z[0] = r * c; z[1] = r * s;
}
__device__ __forceinline__ double bernoulli_uniform(uint64_t seed, uint32_t cand, uint32_t iter) {
  uint32_t o[4];
  philox4x32_10(cand, 0u, iter, 1u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return u53(o[0], o[1]);
}

// ---- mjpc::Norm value path (mjpc/norm.cc:50-210) over a compile-time-sized slice ----
template <typename T, int N>
__device__ __forceinline__ T norm_value(const T* x, int type, T p, T q) {
  T y = 0;
  switch (type) {  // wave-uniform: a scalar branch
    case -1: y = x[0]; break;
    case 0:
#pragma unroll
      for (int i = 0; i < N; i++) y += x[i] * x[i];
      y *= T(0.5);
      break;
    case 1: {
      T c = 0;
#pragma unroll
      for (int i = 0; i < N; i++) c += x[i] * x[i];
      T a = pow(c, q / 2) + pow(p, q);
      y = pow(a, 1 / q) - p;
      break;
    }
    case 2: {
      T c = 0;
#pragma unroll
      for (int i = 0; i < N; i++) c += x[i] * x[i];
      y = sqrt(c + p * p) - p;
      break;
    }
    case 3:
#pragma unroll
      for (int i = 0; i < N; i++) y += p * p * (cosh(x[i] / p) - T(1));
      break;
    case 5:
#pragma unroll
      for (int i = 0; i < N; i++) y += pow(fabs(x[i]), p);
      break;
    case 6:
#pragma unroll
      for (int i = 0; i < N; i++) y += sqrt(x[i] * x[i] + p * p) - p;
      break;
    case 7:
#pragma unroll
      for (int i = 0; i < N; i++) y += pow(pow(fabs(x[i]), q) + pow(p, q), 1 / q) - p;
      break;
    case 8:
#pragma unroll
      for (int i = 0; i < N; i++) y += p > 0 ? p * log(1 + exp(x[i] / p)) : (x[i] > 0 ? x[i] : T(0));
      break;
    default: break;
  }
  return y;
}

// ---- M = L D L' for a small dense SPD matrix held in registers (lower triangle of M valid).
// Dinv = 1/D is kept so that every later solve is multiply/fma only (fp64 division is ~12
// instructions on CDNA4; the solve is on the per-step dependent chain).
template <int NV, typename T>
__device__ __forceinline__ void ldl_factor(T (&L)[NV][NV], T (&Dinv)[NV], const T (&M)[NV][NV]) {
  T D[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) {
    T d = M[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
    D[j] = d;
    Dinv[j] = T(1) / d;
#pragma unroll
    for (int i = j + 1; i < NV; i++) {
      T v = M[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
      L[i][j] = v * Dinv[j];
    }
  }
}
template <int NV, typename T>
__device__ __forceinline__ void ldl_solve(T (&x)[NV], const T (&L)[NV][NV], const T (&Dinv)[NV], const T (&b)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; i++) {
    T v = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) v -= L[i][k] * x[k];
    x[i] = v;
  }
#pragma unroll
  for (int i = 0; i < NV; i++) x[i] *= Dinv[i];
#pragma unroll
  for (int i = NV - 1; i >= 0; i--) {
    T v = x[i];
#pragma unroll
    for (int k = i + 1; k < NV; k++) v -= L[k][i] * x[k];
    x[i] = v;
  }
}

}  // namespace mjpcx
