// ilqg_dense.h -- the dense-algebra half of iLQG on gfx950:
//   * cost_derivatives_kernel : CostDerivatives::Compute (mjpc/planners/cost_derivatives.cc:112-230):
//       per timestep Gauss-Newton cx, cu, cxx, cxu, cuu from the norm gradient/Hessian and the residual
//       Jacobians C (= rx), D (= ru); one workgroup per timestep
//   * backward_pass_kernel : iLQGBackwardPass::RiccatiStep over the horizon (backward_pass.cc:65-250) as ONE
//       persistent wavefront walking t = T-2..0; the n x n x n products (A'W, (A'W)A, (A'W)B, B'W, (B'W)B)
//       run on the matrix cores with v_mfma_f64_16x16x4_f64, the box-QP (mju_boxQP) and the small
//       triangular solves on the vector ALU. Matrices live in LDS, row-major, zero-padded to 16.
#pragma once
#include "device_common.h"

namespace mjpcx {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- norm value + gradient + Hessian
// mjpc::Norm with g and H (mjpc/norm.cc:50-210); n <= 32; H row-major n x n (zeroed here)
__device__ inline double norm_grad_hess(double* g, double* H, const double* x, double p, double q, int n, int type) {
  double y = 0;
  for (int i = 0; i < n * n; i++) H[i] = 0;
  switch (type) {
    case -1: y = x[0]; g[0] = 1; break;
    case 0:
      for (int i = 0; i < n; i++) { y += x[i] * x[i]; g[i] = x[i]; H[i * n + i] = 1; }
      y *= 0.5;
      break;
    case 1: {
      double c = 0;
      for (int i = 0; i < n; i++) c += x[i] * x[i];
      const double a = pow(c, q / 2) + pow(p, q), s = pow(a, 1 / q), d = pow(c, q / 2 - 1), b = s / a * d;
      y = s - p;
      for (int i = 0; i < n; i++) g[i] = b * x[i];
      const double e = (1 - q) * d / a + (q - 2) / (c > kMinVal ? c : kMinVal);
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) H[i + j * n] = b * ((i == j ? 1.0 : 0.0) + x[i] * x[j] * e);
      break;
    }
    case 2: {
      double c = p * p;
      for (int i = 0; i < n; i++) c += x[i] * x[i];
      const double s = sqrt(c);
      y = s - p;
      for (int i = 0; i < n; i++) g[i] = s ? x[i] * (1 / s) : 0.0;
      if (s)
        for (int i = 0; i < n; i++)
          for (int j = 0; j < n; j++) H[i + j * n] = ((i == j ? 1 : 0) - g[i] * g[j]) / s;
      break;
    }
    case 3:
      for (int i = 0; i < n; i++) { y += p * p * (cosh(x[i] / p) - 1.0); g[i] = p * sinh(x[i] / p); H[i * n + i] = cosh(x[i] / p); }
      break;
    case 5:
      for (int i = 0; i < n; i++) {
        const double s = fabs(x[i]);
        y += pow(s, p);
        g[i] = (x[i] > 0 ? 1 : (x[i] < 0 ? -1 : 0)) * p * pow(s, p - 1);
        H[i * n + i] = (p - 1) * p * pow(s, p - 2);
      }
      break;
    case 6:
      for (int i = 0; i < n; i++) {
        const double s = sqrt(x[i] * x[i] + p * p);
        y += s - p;
        g[i] = s ? x[i] / s : 0;
        H[i * n + i] = s ? (1 - g[i] * g[i]) / s : 0;
      }
      break;
    case 7:
      for (int i = 0; i < n; i++) {
        const double a = fabs(x[i]), d = pow(a, q), e = d + pow(p, q), s = pow(e, 1 / q);
        y += s - p;
        const double c = s * pow(a, q - 2) / e;
        g[i] = c * x[i];
        H[i * n + i] = c * (q - 1) * (1 - d / e);
      }
      break;
    case 8:
      for (int i = 0; i < n; i++) {
        if (p > 0) {
          const double s = exp(x[i] / p);
          y += p * log(1 + s);
          g[i] = s / (1 + s);
          H[i * n + i] = s / (p * (1 + s) * (1 + s));
        } else {
          y += x[i] > 0 ? x[i] : 0;
          g[i] = x[i] > 0 ? 1 : 0;
        }
      }
      break;
    default: break;
  }
  return y;
}

struct CostSpec {  // Task cost specification, by value in the kernarg segment
  int num_term, num_residual;
  int dim[32], norm[32];
  double weight[32], p[32], q[32];
  double risk;
};

// one workgroup (64 lanes) per timestep
__global__ __launch_bounds__(64) void cost_derivatives_kernel(const CostSpec cs, const double* __restrict__ r,
                                                               const double* __restrict__ C, const double* __restrict__ D,
                                                               int T, int ndx, int nu, double* cx, double* cu,
                                                               double* cxx, double* cxu, double* cuu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* g = reinterpret_cast<double*>(smem_raw);  // [32]
  double* Hn = g + 32;                              // [32*32]
  double* Hrx = Hn + 32 * 32;                       // [32 * ndx]
  double* Hru = Hrx + 32 * ndx;                     // [32 * nu]
  __shared__ double cval;
  const int t = blockIdx.x, lane = threadIdx.x;
  const int nr = cs.num_residual;
  double* cx_t = cx + (size_t)t * ndx; double* cu_t = cu + (size_t)t * nu;
  double* cxx_t = cxx + (size_t)t * ndx * ndx; double* cxu_t = cxu + (size_t)t * ndx * nu; double* cuu_t = cuu + (size_t)t * nu * nu;
  for (int i = lane; i < ndx; i += 64) cx_t[i] = 0;
  for (int i = lane; i < nu; i += 64) cu_t[i] = 0;
  for (int i = lane; i < ndx * ndx; i += 64) cxx_t[i] = 0;
  for (int i = lane; i < ndx * nu; i += 64) cxu_t[i] = 0;
  for (int i = lane; i < nu * nu; i += 64) cuu_t[i] = 0;
  if (lane == 0) cval = 0;
  __syncthreads();
  int shift = 0;
  for (int k = 0; k < cs.num_term; k++) {
    const int nk = cs.dim[k];
    const double w = cs.weight[k] / T;  // weights[i] / T, cost_derivatives.cc:151
    const double* rk = r + (size_t)t * nr + shift;
    const double* rx = C + ((size_t)t * nr + shift) * ndx;  // nk x ndx
    const double* ru = D + ((size_t)t * nr + shift) * nu;   // nk x nu
    if (lane == 0) cval += w * norm_grad_hess(g, Hn, rk, cs.p[k], cs.q[k], nk, cs.norm[k]);
    __syncthreads();
    for (int e = lane; e < nk * ndx; e += 64) {  // Hrx = H rx
      const int a = e / ndx, j = e % ndx;
      double s = 0;
      for (int b = 0; b < nk; b++) s += Hn[a * nk + b] * rx[b * ndx + j];
      Hrx[e] = s;
    }
    for (int e = lane; e < nk * nu; e += 64) {
      const int a = e / nu, j = e % nu;
      double s = 0;
      for (int b = 0; b < nk; b++) s += Hn[a * nk + b] * ru[b * nu + j];
      Hru[e] = s;
    }
    __syncthreads();
    for (int i = lane; i < ndx; i += 64) { double s = 0; for (int a = 0; a < nk; a++) s += rx[a * ndx + i] * g[a]; cx_t[i] += w * s; }
    for (int i = lane; i < nu; i += 64) { double s = 0; for (int a = 0; a < nk; a++) s += ru[a * nu + i] * g[a]; cu_t[i] += w * s; }
    for (int e = lane; e < ndx * ndx; e += 64) {  // cxx += w (H rx)' rx
      const int i = e / ndx, j = e % ndx;
      double s = 0;
      for (int a = 0; a < nk; a++) s += Hrx[a * ndx + i] * rx[a * ndx + j];
      cxx_t[e] += w * s;
    }
    for (int e = lane; e < ndx * nu; e += 64) {  // cxu += w (H rx)' ru
      const int i = e / nu, j = e % nu;
      double s = 0;
      for (int a = 0; a < nk; a++) s += Hrx[a * ndx + i] * ru[a * nu + j];
      cxu_t[e] += w * s;
    }
    for (int e = lane; e < nu * nu; e += 64) {  // cuu += w (H ru)' ru
      const int i = e / nu, j = e % nu;
      double s = 0;
      for (int a = 0; a < nk; a++) s += Hru[a * nu + i] * ru[a * nu + j];
      cuu_t[e] += w * s;
    }
    __syncthreads();
    shift += nk;
  }
  // exponential risk transformation, cost_derivatives.cc:156-226 (including its use of the ALREADY scaled
  // cx / cu in the rank-one terms)
  if (fabs(cs.risk) >= 1.0e-6) {
    const double s = exp(cs.risk * cval);
    __syncthreads();
    for (int i = lane; i < ndx; i += 64) cx_t[i] *= s;
    for (int i = lane; i < nu; i += 64) cu_t[i] *= s;
    __syncthreads();
    for (int e = lane; e < ndx * ndx; e += 64) cxx_t[e] = cxx_t[e] * s + cs.risk * s * cx_t[e / ndx] * cx_t[e % ndx];
    for (int e = lane; e < ndx * nu; e += 64) cxu_t[e] = cxu_t[e] * s + cs.risk * s * cx_t[e / nu] * cu_t[e % nu];
    for (int e = lane; e < nu * nu; e += 64) cuu_t[e] = cuu_t[e] * s + cs.risk * s * cu_t[e / nu] * cu_t[e % nu];
  }
}

// ---------------------------------------------------------------- one-wave MFMA GEMM on LDS matrices
// Cm[M x N] = op(Am) * Bm (+ Dm if given), row-major with leading dimensions; op(A) = A' when transA.
// M, N, K are the LOGICAL sizes; out-of-range operand elements read as zero, out-of-range results are dropped.
// v_mfma_f64_16x16x4_f64 fragment layout (gfx950): A: lane l holds A[i = l&15][k = l>>4]; B: B[k = l>>4][j = l&15];
// C/D: 4 regs per lane, reg r -> row (l>>4) + 4r, col l&15.
// The operands are LDS pointers by TYPE (address space 3): ds_read/ds_write whatever the inliner decides.
// KSTEPS = ceil(Kmax / 4): ALL operand fragments of an output tile are fetched first (2*KSTEPS independent LDS
// reads, one wait), then the MFMAs issue back to back; a read pair per MFMA would expose the LDS latency KSTEPS
// times per tile. Returns the advanced tile counter (16x16 output tiles are dealt round-robin over the waves).
typedef __attribute__((address_space(3))) double lds_f64;
#define MJPCX_LDS(p) ((lds_f64*)(p))
template <int KSTEPS>
__device__ __forceinline__ int wave_gemm(lds_f64* Cm, int ldc, const lds_f64* Am, int lda, bool transA, const lds_f64* Bm,
                                         int ldb, int M, int N, int K, const lds_f64* Dm, int ldd, int lane, int wave = 0,
                                         int nwave = 1, int tile = 0) {
  const int li = lane & 15, lk = lane >> 4;
  const int astep = transA ? lda : 1;
  for (int ti = 0; ti < M; ti += 16)
    for (int tj = 0; tj < N; tj += 16) {
      if ((tile++) % nwave != wave) continue;
      v4f64 acc = {0, 0, 0, 0};
      const int i = ti + li, j = tj + li;
      const bool iv = i < M, jv = j < N;
      const int ic = iv ? i : M - 1, jc = jv ? j : N - 1;  // clamped: every lane reads a valid address
      const lds_f64* ap = transA ? Am + ic : Am + ic * lda;
      const lds_f64* bp = Bm + jc;
      double av[KSTEPS], bv[KSTEPS];
      int aoff = lk * astep, boff = lk * ldb;
#pragma unroll
      for (int u = 0; u < KSTEPS; u++) {
        const bool kv = 4 * u + lk < K;
        av[u] = ap[kv ? aoff : 0];
        bv[u] = bp[kv ? boff : 0];
        if (!(iv && kv)) av[u] = 0.0;
        if (!(jv && kv)) bv[u] = 0.0;
        aoff += 4 * astep; boff += 4 * ldb;
      }
#pragma unroll
      for (int u = 0; u < KSTEPS; u++)
        if (4 * u < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int ci = ti + lk + 4 * rg;
        if (ci < M && jv) Cm[ci * ldc + j] = acc[rg] + (Dm ? Dm[ci * ldd + j] : 0.0);
      }
    }
  return tile;
}

// ---------------------------------------------------------------- serial helpers (one lane)
__device__ inline int chol_factor_serial(double* a, int n) {
  int rank = n;
  for (int j = 0; j < n; j++) {
    double s = a[j * n + j];
    for (int k = 0; k < j; k++) s -= a[j * n + k] * a[j * n + k];
    if (s < 1e-15) { s = 1e-15; rank--; }
    s = sqrt(s);
    a[j * n + j] = s;
    for (int i = j + 1; i < n; i++) {
      double v = a[i * n + j];
      for (int k = 0; k < j; k++) v -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = v / s;
    }
  }
  return rank;
}
__device__ inline void chol_solve_serial(double* x, const double* L, const double* b, int n) {
  for (int i = 0; i < n; i++) {
    double v = b[i];
    for (int k = 0; k < i; k++) v -= L[i * n + k] * x[k];
    x[i] = v / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double v = x[i];
    for (int k = i + 1; k < n; k++) v -= L[k * n + i] * x[k];
    x[i] = v / L[i * n + i];
  }
}
// mju_boxQP (projected Newton, Tassa et al. 2014); scratch: 5n doubles. Returns nfree or -1.
__device__ inline int boxqp_serial(double* res, double* R, int* index, const double* H, const double* g, int n,
                                   const double* lower, const double* upper, double* scratch) {
  double *grad = scratch, *search = grad + n, *cand = search + n, *tmp = cand + n, *rhs = tmp + n;
  int nfree = 0;
  for (int i = 0; i < n; i++) res[i] = res[i] < lower[i] ? lower[i] : (res[i] > upper[i] ? upper[i] : res[i]);
  for (int iter = 0; iter < 100; iter++) {
    double value = 0;
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += H[i * n + k] * res[k];
      value += 0.5 * res[i] * s + g[i] * res[i];
      grad[i] = s + g[i];
    }
    nfree = 0;
    for (int i = 0; i < n; i++) {
      const bool clamped = (res[i] <= lower[i] && grad[i] > 0) || (res[i] >= upper[i] && grad[i] < 0);
      if (!clamped) index[nfree++] = i;
    }
    if (nfree == 0) break;
    for (int a = 0; a < nfree; a++)
      for (int b = 0; b < nfree; b++) R[a * nfree + b] = H[index[a] * n + index[b]];
    if (chol_factor_serial(R, nfree) < nfree) return -1;
    double gn = 0;
    for (int a = 0; a < nfree; a++) gn += grad[index[a]] * grad[index[a]];
    if (sqrt(gn) < 1e-16) break;
    for (int i = 0; i < n; i++) tmp[i] = res[i];
    for (int a = 0; a < nfree; a++) tmp[index[a]] = 0;  // x_clamped
    for (int a = 0; a < nfree; a++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += H[index[a] * n + k] * tmp[k];
      rhs[a] = -(g[index[a]] + s);
    }
    chol_solve_serial(cand, R, rhs, nfree);
    for (int i = 0; i < n; i++) search[i] = 0;
    for (int a = 0; a < nfree; a++) search[index[a]] = cand[a] - res[index[a]];
    double sdotg = 0;
    for (int i = 0; i < n; i++) sdotg += search[i] * grad[i];
    if (sdotg >= 0) break;
    double step = 1;
    bool ok = false;
    while (step > 1e-22) {
      double vc = 0;
      for (int i = 0; i < n; i++) {
        const double c = res[i] + step * search[i];
        cand[i] = c < lower[i] ? lower[i] : (c > upper[i] ? upper[i] : c);
      }
      for (int i = 0; i < n; i++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += H[i * n + k] * cand[k];
        vc += 0.5 * cand[i] * s + g[i] * cand[i];
      }
      if ((vc - value) / (step * sdotg) >= 0.1) { ok = true; break; }
      step *= 0.5;
    }
    if (!ok) break;
    for (int i = 0; i < n; i++) res[i] = cand[i];
  }
  return nfree;
}

// ---------------------------------------------------------------- wave-cooperative small dense algebra (m <= 16)
// Matrices live in LDS with leading dimension 16; lane i < m owns ROW i of wave 0; dependent phases are separated
// by wave_sync() (in-order DS execution within a wave), never by a workgroup barrier.
// Clamped (box-constrained) coordinates are handled by MASKING instead of compressing: their row/column of the
// Hessian is replaced by the identity and their right-hand side by zero, which yields exactly the free-subspace
// solution of mju_boxQP with zeros at the clamped coordinates.
// LDS ordering point inside ONE wavefront: DS operations of a wave execute in order, so only the compiler has to
// be stopped from moving LDS accesses across it (no s_barrier: the other waves of the workgroup are not involved)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum16(double v, int lane) {  // sum over lanes 0..15 (others contribute 0)
  v = lane < 16 ? v : 0.0;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return __shfl(v, 0, 64);
}
// in-place Cholesky of the masked matrix L (ld 16); returns false if a pivot is not positive
__device__ __forceinline__ bool coop_chol16(double* L, int m, int lane) {
  bool ok = true;
  for (int j = 0; j < m; j++) {
    if (lane == j) {
      double s = L[j * 16 + j];
      for (int k = 0; k < j; k++) s -= L[j * 16 + k] * L[j * 16 + k];
      L[j * 16 + j] = s > 1e-15 ? sqrt(s) : -1.0;
    }
    wave_sync();
    const double d = L[j * 16 + j];
    if (d < 0) { ok = false; break; }
    if (lane > j && lane < m) {
      double v = L[lane * 16 + j];
      for (int k = 0; k < j; k++) v -= L[lane * 16 + k] * L[j * 16 + k];
      L[lane * 16 + j] = v / d;
    }
    wave_sync();
  }
  return ok;
}
// x = (L L')^-1 b for one right-hand side held in LDS (x, length m); lanes 0..m-1 cooperate
__device__ __forceinline__ void coop_solve16(double* x, const double* L, int m, int lane) {
  for (int j = 0; j < m; j++) {  // forward
    if (lane == j) x[j] /= L[j * 16 + j];
    wave_sync();
    if (lane > j && lane < m) x[lane] -= L[lane * 16 + j] * x[j];
    wave_sync();
  }
  for (int j = m - 1; j >= 0; j--) {  // backward
    if (lane == j) x[j] /= L[j * 16 + j];
    wave_sync();
    if (lane < j) x[lane] -= L[j * 16 + lane] * x[j];
    wave_sync();
  }
}
// mju_boxQP (projected Newton), cooperative: res in/out (LDS, m), H (LDS ld 16), g/lower/upper (LDS).
// On return Lm holds the Cholesky factor of the masked Hessian and free_mask the free set. Returns nfree or -1.
__device__ __forceinline__ int coop_boxqp16(double* res, double* Lm, unsigned* free_mask, const double* H, const double* g, int m,
                                             const double* lower, const double* upper, double* work, int lane) {
  double* search = work;       // [16]
  double* cand = work + 16;    // [16]
  int nfree = 0;
  if (lane < m) res[lane] = fmin(fmax(res[lane], lower[lane]), upper[lane]);
  wave_sync();
  double oldvalue = 0;
  for (int iter = 0; iter < 100; iter++) {
    double hx = 0, xi = 0, gi = 0;
    if (lane < m) {
      xi = res[lane]; gi = g[lane];
      for (int k = 0; k < m; k++) hx += H[lane * 16 + k] * res[k];
    }
    const double value = wave_sum16(0.5 * xi * hx + gi * xi, lane);
    if (iter > 0 && (oldvalue - value) < 1e-8 * fabs(oldvalue)) break;  // no further relative improvement
    oldvalue = value;
    const double grad = hx + gi;
    const bool clamped = lane < m && ((xi <= lower[lane] && grad > 0) || (xi >= upper[lane] && grad < 0));
    const bool is_free = lane < m && !clamped;
    const unsigned long long fm = __ballot(is_free);
    *free_mask = (unsigned)fm;
    nfree = __popcll(fm);
    if (nfree == 0) break;
    // masked Hessian and its factor
    if (lane < m)
      for (int k = 0; k < m; k++) {
        const bool fk = (fm >> k) & 1;
        Lm[lane * 16 + k] = (is_free && fk) ? H[lane * 16 + k] : (lane == k ? 1.0 : 0.0);
      }
    wave_sync();
    if (!coop_chol16(Lm, m, lane)) return -1;
    const double gn = wave_sum16(is_free ? grad * grad : 0.0, lane);
    if (sqrt(gn) < 1e-16) break;
    // Newton step in the free subspace: rhs = -(g + H x_clamped) on free rows, 0 on clamped rows
    if (lane < m) {
      double s = 0;
      for (int k = 0; k < m; k++) s += ((fm >> k) & 1) ? 0.0 : H[lane * 16 + k] * res[k];
      search[lane] = is_free ? -(gi + s) : 0.0;
    }
    wave_sync();
    coop_solve16(search, Lm, m, lane);
    const double sd = (lane < m && is_free) ? search[lane] - xi : 0.0;
    const double sdotg = wave_sum16(sd * grad, lane);
    if (sdotg >= 0) break;
    double step = 1;
    bool ok = false;
    while (step > 1e-22) {
      if (lane < m) cand[lane] = fmin(fmax(xi + step * sd, lower[lane]), upper[lane]);
      wave_sync();
      double hc = 0, ci = 0;
      if (lane < m) {
        ci = cand[lane];
        for (int k = 0; k < m; k++) hc += H[lane * 16 + k] * cand[k];
      }
      const double vc = wave_sum16(0.5 * ci * hc + gi * ci, lane);
      if ((vc - value) / (step * sdotg) >= 0.1) { ok = true; break; }
      step *= 0.5;
      wave_sync();
    }
    if (!ok) break;
    wave_sync();
    if (lane < m) res[lane] = cand[lane];
    wave_sync();
  }
  return nfree;
}

// ---- register-resident variants (wave 0): lane i < 16 keeps ROW i of the m x m matrix in 16 registers; values of
// other rows arrive through v_readlane (uniform source lane), so a factorisation costs no LDS round trips at all
__device__ __forceinline__ double bcast_lane(double v, int src) {  // src must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// 1 / sqrt(x) and sqrt(x) for x > 0 from the hardware estimate (v_rsq_f64, ~2^-26) refined by two Newton steps and one correction of the
// root: an ulp or two from the correctly rounded values, a fifth of the instructions of sqrt() followed by a division -- the pivots of
// a 12 x 12 factorisation are a chain of sixteen of these with nothing to overlap them
__device__ __forceinline__ void bp_sqrt_rsqrt(double x, double& s, double& r) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = fma(y, fma(-hx * y, y, 0.5), y);
  y = fma(y, fma(-hx * y, y, 0.5), y);
  double g = x * y;
  g = fma(fma(-g, g, x), 0.5 * y, g);
  s = g; r = y;
}
// sum over lanes 0..15 (the others contribute nothing), the same value in every lane: four DPP row shifts instead of five LDS-crossbar
// shuffles (the box-QP evaluates two to four of these per iteration on its critical path)
__device__ __forceinline__ double row_sum16(double v, int lane) {
  v = lane < 16 ? v : 0.0;
  // row_shr:1,2,4,8 (bound_ctrl: lanes shifted in from outside the row read 0): after the four steps lane 15 of the row holds the total
#define MJPCX_ROW_SHR(ctrl) do { const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true); \
    const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true); v += __hiloint2double(hi_, lo_); } while (0)
  MJPCX_ROW_SHR(0x111); MJPCX_ROW_SHR(0x112); MJPCX_ROW_SHR(0x114); MJPCX_ROW_SHR(0x118);
#undef MJPCX_ROW_SHR
  return bcast_lane(v, 15);
}
// in-place Cholesky of the M x M matrix (M = 12 or 16 at compile time: the run-time m <= M, rows / columns beyond m are the identity); on
// exit row[k], k <= lane, holds L[lane][k] and inv[j] = 1 / L[j][j] (the same in every lane). Same operation order as the left-looking
// serial algorithm (each entry has its products subtracted for k = 0, 1, ...), with the division by the pivot a multiplication by its
// reciprocal. Returns false on a non-positive pivot.
template <int M>
__device__ __forceinline__ bool reg_chol16(double (&row)[16], double (&inv)[16], int lane) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < M; j++) {
    const double djj = bcast_lane(row[j], j);
    ok = ok && djj > 1e-15;
    double d, id;
    bp_sqrt_rsqrt(ok ? djj : 1.0, d, id);
    inv[j] = id;
    const double lij = lane == j ? d : row[j] * id;
    row[j] = lij;
#pragma unroll
    for (int k = j + 1; k < M; k++) {
      const double lkj = bcast_lane(lij, k);
      row[k] = lane >= k ? row[k] - lij * lkj : row[k];
    }
  }
#pragma unroll
  for (int j = M; j < 16; j++) inv[j] = 1.0;
  return ok;
}
// x = (L L')^-1 b; row[k] = L[lane][k], col[k] = L[k][lane], inv[j] = 1 / L[j][j]; b is this lane's right-hand side entry
template <int M>
__device__ __forceinline__ double reg_solve16(const double (&row)[16], const double (&col)[16], const double (&inv)[16], double b, int lane) {
#pragma unroll
  for (int j = 0; j < M; j++) {
    const double yj = bcast_lane(b, j) * inv[j];
    b = lane == j ? yj : (lane > j ? b - row[j] * yj : b);
  }
#pragma unroll
  for (int j = M - 1; j >= 0; j--) {
    const double xj = bcast_lane(b, j) * inv[j];
    b = lane == j ? xj : (lane < j ? b - col[j] * xj : b);
  }
  return b;
}
// L rows (registers) -> LDS (ld 16) -> L columns (registers)
__device__ __forceinline__ void reg_transpose16(double (&col)[16], const double (&row)[16], double* lds, int lane) {
  if (lane < 16) {
#pragma unroll
    for (int k = 0; k < 16; k++) lds[lane * 16 + k] = row[k];
  }
  wave_sync();
#pragma unroll
  for (int k = 0; k < 16; k++) col[k] = lane < 16 ? lds[k * 16 + lane] : 0.0;
}
template <int M>
__device__ __forceinline__ double reg_matvec16(const double (&row)[16], double x, int m, int lane) {
  double s = 0;
#pragma unroll
  for (int k = 0; k < M; k++) s += row[k] * bcast_lane(x, k);
  return lane < m ? s : 0.0;
}
// mju_boxQP (projected Newton) with everything in registers. res: this lane's coordinate (in/out, warm start);
// Hrow: row of H (zero beyond m); on return Lrow/Lcol hold the factor of the masked Hessian of the last evaluated free set (also left
// in Llds, ld 16, its reciprocal pivots in Linvlds) and fmask that free set. Returns nfree or -1 (factorisation failed).
// boxed == false is the unconstrained branch of the backward pass (one factorisation, res = -H^-1 g): it shares this
// body so that the unrolled Cholesky and triangular solves exist ONCE in the instruction stream.
template <int M>
__device__ __forceinline__ int reg_boxqp16(double& res, double (&Lrow)[16], unsigned& fmask, const double (&Hrow)[16], double gi,
                                            int m, double lower, double upper, double* Llds, double* Linvlds, int lane, bool boxed) {
  int nfree = 0;
  if (boxed) res = lane < m ? fmin(fmax(res, lower), upper) : 0.0;
  double oldvalue = 0;
  double Lcol[16], Linv[16];
#pragma unroll
  for (int k = 0; k < 16; k++) Linv[k] = 1.0;
  unsigned long long prev_fm = 0;
  for (int iter = 0; iter < 100; iter++) {
    const double xi = res;
    double value = 0, grad = 0;
    bool is_free = lane < m;
    if (boxed) {
      const double hx = reg_matvec16<M>(Hrow, xi, m, lane);
      value = row_sum16(0.5 * xi * hx + gi * xi, lane);
      grad = hx + gi;
      const bool clamped = lane < m && ((xi <= lower && grad > 0) || (xi >= upper && grad < 0));
      is_free = lane < m && !clamped;
    }
    const unsigned long long fm = __ballot(is_free);
    fmask = (unsigned)fm;
    nfree = __popcll(fm);
    if (nfree == 0) break;
    if (iter == 0 || fm != prev_fm) {  // re-factorise only when the clamped set changed
#pragma unroll
      for (int k = 0; k < 16; k++) Lrow[k] = (is_free && ((fm >> k) & 1)) ? Hrow[k] : (lane == k ? 1.0 : 0.0);
      if (!reg_chol16<M>(Lrow, Linv, lane)) return -1;
      wave_sync();
      reg_transpose16(Lcol, Lrow, Llds, lane);
      if (lane == 0) {  // the reciprocal pivots for the K columns (backward_pass_kernel)
#pragma unroll
        for (int k = 0; k < 16; k++) Linvlds[k] = Linv[k];
      }
      prev_fm = fm;
    }
    double rhs = -gi;
    if (boxed) {
      if (iter > 0 && (oldvalue - value) < 1e-8 * fabs(oldvalue)) break;  // no further relative improvement
      oldvalue = value;
      const double gn = row_sum16(is_free ? grad * grad : 0.0, lane);
      if (sqrt(gn) < 1e-16) break;
      // Newton step in the free subspace: rhs = -(g + H x_clamped) on free rows, 0 on clamped rows
      double sc = 0;
#pragma unroll
      for (int k = 0; k < M; k++) sc += ((fm >> k) & 1) ? 0.0 : Hrow[k] * bcast_lane(xi, k);
      rhs = is_free ? -(gi + sc) : 0.0;
    }
    const double search = reg_solve16<M>(Lrow, Lcol, Linv, rhs, lane);
    if (!boxed) { res = search; break; }
    const double sd = is_free ? search - xi : 0.0;
    const double sdotg = row_sum16(sd * grad, lane);
    if (sdotg >= 0) break;
    double step = 1, ci = 0;
    bool ok = false;
    while (step > 1e-22) {
      ci = lane < m ? fmin(fmax(xi + step * sd, lower), upper) : 0.0;
      const double hc = reg_matvec16<M>(Hrow, ci, m, lane);
      const double vc = row_sum16(0.5 * ci * hc + gi * ci, lane);
      if ((vc - value) / (step * sdotg) >= 0.1) { ok = true; break; }
      step *= 0.5;
    }
    if (!ok) break;
    res = ci;
  }
  return nfree;
}

// ---- strided global <-> LDS block moves for one wave: no integer division in the loop, four independent
// global accesses in flight per lane (a one-element-per-iteration loop exposes the full HBM latency each time)
template <int MODE>  // 0: dst = src, 1: dst += src (global -> LDS); 2: LDS -> global
__device__ __forceinline__ void wave_block_move(double* lds, int ld, double* glob, int rows, int cols, int lane, int nthr = 64) {
  const int total = rows * cols;
  int e = lane, r = lane / cols, c = lane % cols;
  while (e < total) {
    double v[4];
    int rr[4], cc[4], cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (e < total) {
        rr[u] = r; cc[u] = c; cnt = u + 1;
        if (MODE != 2) v[u] = glob[e];
        e += nthr; r += nthr / cols; c += nthr % cols;
        if (c >= cols) { c -= cols; r++; }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (u < cnt) {
        if (MODE == 0) lds[rr[u] * ld + cc[u]] = v[u];
        else if (MODE == 1) lds[rr[u] * ld + cc[u]] += v[u];
        else glob[rr[u] * cols + cc[u]] = lds[rr[u] * ld + cc[u]];
      }
    }
  }
}
// f(row, col) over a rows x cols index space, one element per thread per trip, without integer division per element
template <class F>
__device__ __forceinline__ void block_for_each(int rows, int cols, int tid, int nthr, F f) {
  const int total = rows * cols, dr = nthr / cols, dc = nthr % cols;
  int r = tid / cols, c = tid % cols;
  for (int e = tid; e < total; e += nthr) {
    f(r, c);
    r += dr; c += dc;
    if (c >= cols) { c -= cols; r++; }
  }
}

struct BackwardArgs {
  int n, m, T;
  double mu;
  int reg_type, use_limits;
  const double *A, *B, *cx, *cu, *cxx, *cxu, *cuu, *actions, *limits;
  double *Vx, *Vxx, *K, *du, *dV;
  int* status;  // 1 ok, 0 failed (Quu not PD at some step)
  long long* stamps;  // optional: 16 phase timestamps of the first step (s_memtime), for tuning
};

constexpr int kBackwardWaves = 4;
constexpr int kBackwardThreads = 64 * kBackwardWaves;
// contiguous global -> registers -> LDS staging, K elements per thread (total <= K * kBackwardThreads)
template <int K>
__device__ __forceinline__ void flat_load(double (&v)[K], const double* g, int total, int tid) {
#pragma unroll
  for (int k = 0; k < K; k++) { const int e = tid + k * kBackwardThreads; v[k] = g[e < total ? e : total - 1]; }
}
template <int K>
__device__ __forceinline__ void flat_store(double* l, const double (&v)[K], int total, int tid) {
#pragma unroll
  for (int k = 0; k < K; k++) { const int e = tid + k * kBackwardThreads; if (e < total) l[e] = v[k]; }
}

// ONE workgroup of 4 wavefronts walks the horizon backwards (the recursion is sequential in t). Per step the
// MFMA tiles of the GEMMs and all element-wise phases are spread over the 256 threads; the m x m (m <= 16)
// factorisation / box-QP runs register-resident on wave 0 while the others wait at the next barrier. The inputs of
// step t-1 (A, B, c*) are fetched into registers while step t computes, so HBM latency is off the critical path.
// All LDS matrices are DENSE (leading dimension = their column count), which makes every staging copy flat:
//   W[n*n] Wx[n] At[n*n] Bt[n*m] tmp[n*n] tmp2[m*n] Qxx[n*n] Qxu[n*m] Quu[m*m] Quur[m*m] Qx[n] Qu[m]
//   Kt[m*n] KQ[m*n] dut[m] qsum[m] L[16*16] cxl[n] cul[m] actl[m] lim[2m]      (carved with NP = 16*ceil(n/16), 16)
template <int M>
__global__ __launch_bounds__(kBackwardThreads) void backward_pass_kernel(const BackwardArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = a.n, m = a.m, T = a.T, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kBackwardWaves, NT = kBackwardThreads;
  const int NP = (n + 15) & ~15, MP = 16;
  const int nn = n * n, nm = n * m, mm = m * m;
  double* W = reinterpret_cast<double*>(smem_raw);
  double* Wx = W + NP * NP; double* At = Wx + NP; double* Bt = At + NP * NP; double* tmp = Bt + NP * MP;
  double* tmp2 = tmp + NP * NP; double* Qxx = tmp2 + MP * NP; double* Qxu = Qxx + NP * NP; double* Quu = Qxu + NP * MP;
  double* Qxur = Quu + MP * MP; double* Quur = Qxur + NP * MP; double* Qx = Quur + MP * MP; double* Qu = Qx + NP;
  double* Kt = Qu + MP; double* KQ = Kt + MP * NP; double* dut = KQ + MP * NP; double* qsum = dut + MP;
  double* Llds = qsum + MP; double* cxl = Llds + MP * MP; double* cul = cxl + NP; double* actl = cul + MP;
  double* lim = actl + MP;  // 2*MP
  double* Linvl = lim + 2 * MP;  // MP: reciprocal pivots of the factor in Llds
  __shared__ unsigned fmask_s;
  __shared__ int ok_s;
  __shared__ double dV0, dV1;
  if (tid == 0) { dV0 = 0; dV1 = 0; ok_s = 1; }
  double boxres = 0;  // BoxQP::res warm start (wave 0, lane i), reset per sweep
  // terminal condition: V = c at T-1
  for (int e = tid; e < nn; e += NT) { const double v = a.cxx[(size_t)(T - 1) * nn + e]; W[e] = v; a.Vxx[(size_t)(T - 1) * nn + e] = v; }
  for (int i = tid; i < n; i += NT) { const double v = a.cx[(size_t)(T - 1) * n + i]; Wx[i] = v; a.Vx[(size_t)(T - 1) * n + i] = v; }
  for (int i = tid; i < 2 * m; i += NT) lim[i] = a.limits[i];
  // register staging of the next step's inputs
  double rA[9], rB[3], rCxx[9], rCxu[3], rCuu[1], rCx[1], rCu[1], rAct[1];
  auto prefetch = [&](int t) {
    flat_load(rA, a.A + (size_t)t * nn, nn, tid);
    flat_load(rB, a.B + (size_t)t * nm, nm, tid);
    flat_load(rCxx, a.cxx + (size_t)t * nn, nn, tid);
    flat_load(rCxu, a.cxu + (size_t)t * nm, nm, tid);
    flat_load(rCuu, a.cuu + (size_t)t * mm, mm, tid);
    flat_load(rCx, a.cx + (size_t)t * n, n, tid);
    flat_load(rCu, a.cu + (size_t)t * m, m, tid);
    flat_load(rAct, a.actions + (size_t)t * m, m, tid);
  };
  prefetch(T - 2);
#define MJPCX_STAMP(k) do { if (a.stamps && t == T - 3 && tid == 0) a.stamps[k] = (long long)__builtin_readcyclecounter(); } while (0)
  for (int t = T - 2; t >= 0; t--) {
    MJPCX_STAMP(0);
    // the cost Hessians land directly in the Qxx/Qxu/Quu buffers: they are the "D" operand of the GEMMs below
    flat_store(At, rA, nn, tid); flat_store(Bt, rB, nm, tid); flat_store(Qxx, rCxx, nn, tid); flat_store(Qxu, rCxu, nm, tid);
    flat_store(Quu, rCuu, mm, tid); flat_store(cxl, rCx, n, tid); flat_store(cul, rCu, m, tid); flat_store(actl, rAct, m, tid);
    __syncthreads();
    if (t > 0) prefetch(t - 1);
    MJPCX_STAMP(1);
    // tmp = A' W ; tmp2 = B' W ; Qxx = tmp A + cxx ; Qxu = tmp B + cxu ; Quu = tmp2 B + cuu   (matrix cores)
    int tc = 0;
    tc = wave_gemm<12>(MJPCX_LDS(tmp), n, MJPCX_LDS(At), n, true, MJPCX_LDS(W), n, n, n, n, nullptr, 0, lane, wave, NW, tc);
    tc = wave_gemm<12>(MJPCX_LDS(tmp2), n, MJPCX_LDS(Bt), m, true, MJPCX_LDS(W), n, m, n, n, nullptr, 0, lane, wave, NW, tc);
    // Qx = cx + A' Vx ; Qu = cu + B' Vx
    if (wave == NW - 1) {  // the last wave has the fewest GEMM tiles
      const int i = lane;  // n + m <= 64
      if (i < n) {
        double sx = cxl[i];
#pragma unroll 4
        for (int k = 0; k < n; k++) sx += At[k * n + i] * Wx[k];
        Qx[i] = sx;
      } else if (i < n + m) {
        const int u = i - n;
        double su = cul[u];
#pragma unroll 4
        for (int k = 0; k < n; k++) su += Bt[k * m + u] * Wx[k];
        Qu[u] = su;
      }
    }
    __syncthreads();
    tc = 0;
    tc = wave_gemm<12>(MJPCX_LDS(Qxx), n, MJPCX_LDS(tmp), n, false, MJPCX_LDS(At), n, n, n, n, MJPCX_LDS(Qxx), n, lane, wave, NW, tc);
    tc = wave_gemm<12>(MJPCX_LDS(Qxu), m, MJPCX_LDS(tmp), n, false, MJPCX_LDS(Bt), m, n, m, n, MJPCX_LDS(Qxu), m, lane, wave, NW, tc);
    tc = wave_gemm<12>(MJPCX_LDS(Quu), m, MJPCX_LDS(tmp2), n, false, MJPCX_LDS(Bt), m, m, m, n, MJPCX_LDS(Quu), m, lane, wave, NW, tc);
    __syncthreads();
    MJPCX_STAMP(2);
    // ---- regularisation
    if (a.reg_type == 2) {  // value: recompute with W + mu I
      for (int i = tid; i < n; i += NT) W[i * n + i] += a.mu;
      __syncthreads();
      tc = 0;
      tc = wave_gemm<12>(MJPCX_LDS(tmp2), n, MJPCX_LDS(Bt), m, true, MJPCX_LDS(W), n, m, n, n, nullptr, 0, lane, wave, NW, tc);
      __syncthreads();
      // + cxu / cuu: recovered as (unregularised Q) - (unregularised product) would lose bits; re-read them instead
      const double* cuug = a.cuu + (size_t)t * mm;
      for (int e = tid; e < mm; e += NT) Quur[e] = cuug[e];
      for (int i = tid; i < n; i += NT) W[i * n + i] -= a.mu;
      __syncthreads();
      tc = 0;
      tc = wave_gemm<12>(MJPCX_LDS(Quur), m, MJPCX_LDS(tmp2), n, false, MJPCX_LDS(Bt), m, m, m, n, MJPCX_LDS(Quur), m, lane, wave, NW, tc);
    } else {
      for (int e = tid; e < mm; e += NT) Quur[e] = Quu[e];
      if (a.mu != 0 && a.reg_type == 1) {
        tc = 0;
        tc = wave_gemm<12>(MJPCX_LDS(tmp2), m, MJPCX_LDS(Bt), m, true, MJPCX_LDS(Bt), m, m, m, n, nullptr, 0, lane, wave, NW, tc);  // B'B (m x m)
        __syncthreads();
        const double mu = a.mu;
        for (int e = tid; e < mm; e += NT) Quur[e] += mu * tmp2[e];
      } else if (a.mu != 0 && a.reg_type == 0) {
        __syncthreads();
        for (int i = tid; i < m; i += NT) Quur[i * m + i] += a.mu;
      }
    }
    __syncthreads();
    MJPCX_STAMP(3);
    // ---- du (wave 0, register-resident: lane i owns row i of the m x m problem)
    if (wave == 0) {
      double Hrow[16], Lrow[16];
#pragma unroll
      for (int k = 0; k < 16; k++) Hrow[k] = (lane < m && k < m) ? Quur[(lane < m ? lane : 0) * m + (k < m ? k : 0)] : (lane == k ? 1.0 : 0.0);
      const double qu = lane < m ? Qu[lane] : 0.0;
      unsigned fmask = (1u << m) - 1u;
      const bool boxed = a.use_limits != 0;
      const double act = (boxed && lane < m) ? actl[lane] : 0.0;
      const double lo = (boxed && lane < m) ? lim[2 * lane] - act : 0.0, hi = (boxed && lane < m) ? lim[2 * lane + 1] - act : 0.0;
      double x0 = boxed ? boxres : 0.0;
      const int mf = reg_boxqp16<M>(x0, Lrow, fmask, Hrow, qu, m, lo, hi, Llds, Linvl, lane, boxed);
      if (boxed) boxres = x0;
      const bool ok = mf >= 0;
      const double du_i = lane < m ? x0 : 0.0;
      if (ok) {
        if (mf == 0 && lane < 16) {  // everything clamped: identity factor, K = 0
#pragma unroll
          for (int k = 0; k < 16; k++) Llds[lane * 16 + k] = lane == k ? 1.0 : 0.0;
          Linvl[lane] = 1.0;
        }
        if (lane < m) dut[lane] = du_i;
        // dV and Quu du + Qu
        double quu_row[16];
#pragma unroll
        for (int k = 0; k < 16; k++) quu_row[k] = (lane < m && k < m) ? Quu[(lane < m ? lane : 0) * m + (k < m ? k : 0)] : 0.0;
        const double qd = reg_matvec16<M>(quu_row, du_i, m, lane);
        if (lane < m) qsum[lane] = qd + qu;
        const double d0 = row_sum16(lane < m ? du_i * qu : 0.0, lane), d1 = row_sum16(lane < m ? 0.5 * du_i * qd : 0.0, lane);
        if (lane == 0) { dV0 += d0; dV1 += d1; }
      }
      if (lane == 0) { fmask_s = fmask; if (!ok) ok_s = 0; }
    }
    __syncthreads();
    if (!ok_s) break;
    MJPCX_STAMP(4);
    {
      // K = -H_masked^-1 Qux with the UNregularised Qxu, as backward_pass.cc:176-206 does (Qxu_reg is computed there
      // but never used, so it is not computed here); rows of clamped controls are zero. One thread per state column, unrolled to 16 so
      // that the per-thread solution vector stays in registers; L is read from LDS (broadcast across lanes)
      const unsigned fmask = fmask_s;
      for (int j = tid; j < n; j += NT) {
        // (rows / columns m .. M-1 of the factor are the identity and their right-hand sides zero: no run-time guards in the unrolled loops)
        double x[M];
#pragma unroll
        for (int i = 0; i < M; i++) x[i] = (i < m && ((fmask >> i) & 1)) ? Qxu[j * m + (i < m ? i : 0)] : 0.0;
#pragma unroll
        for (int i = 0; i < M; i++) {
          double v = x[i];
#pragma unroll
          for (int k = 0; k < i; k++) v -= Llds[i * 16 + k] * x[k];
          x[i] = v * Linvl[i];
        }
#pragma unroll
        for (int i = M - 1; i >= 0; i--) {
          double v = x[i];
#pragma unroll
          for (int k = i + 1; k < M; k++) v -= Llds[k * 16 + i] * x[k];
          x[i] = v * Linvl[i];
        }
#pragma unroll
        for (int i = 0; i < M; i++) if (i < m) Kt[i * n + j] = -x[i];
      }
    }
    __syncthreads();
    MJPCX_STAMP(5);
    // ---- cost-to-go update
    tc = 0;
    tc = wave_gemm<4>(MJPCX_LDS(KQ), n, MJPCX_LDS(Quu), m, false, MJPCX_LDS(Kt), n, m, n, m, nullptr, 0, lane, wave, NW, tc);  // Quu K  (m x n)
    tc = wave_gemm<4>(MJPCX_LDS(W), n, MJPCX_LDS(Qxu), m, false, MJPCX_LDS(Kt), n, n, n, m, nullptr, 0, lane, wave, NW, tc);   // Qxu K -> W (the old W is dead now)
    // Vx = Qx + K'(Quu du + Qu) + Qxu du
    for (int i = tid; i < n; i += NT) {
      double sv = Qx[i];
      for (int k = 0; k < m; k++) sv += Kt[k * n + i] * qsum[k] + Qxu[i * m + k] * dut[k];
      Wx[i] = sv;
    }
    __syncthreads();
    tc = 0;
    tc = wave_gemm<4>(MJPCX_LDS(tmp), n, MJPCX_LDS(Kt), n, true, MJPCX_LDS(KQ), n, n, n, m, MJPCX_LDS(Qxx), n, lane, wave, NW, tc);      // Qxx + K' Quu K
    __syncthreads();
    // Vxx = sym(Qxx + K'(Quu K) + Qxu K + (Qxu K)') (mju_symmetrize), written into the Qxx buffer, which then
    // becomes W for the next step (pointer swap)
    block_for_each(n, n, tid, NT, [&](int i, int j) {
      const double tij = tmp[i * n + j] + W[i * n + j] + W[j * n + i], tji = tmp[j * n + i] + W[j * n + i] + W[i * n + j];
      Qxx[i * n + j] = 0.5 * (tij + tji);
    });
    { double* sw = W; W = Qxx; Qxx = sw; }
    __syncthreads();
    MJPCX_STAMP(6);
    // ---- write step t
    for (int e = tid; e < nn; e += NT) a.Vxx[(size_t)t * nn + e] = W[e];
    for (int i = tid; i < n; i += NT) a.Vx[(size_t)t * n + i] = Wx[i];
    for (int e = tid; e < nm; e += NT) a.K[(size_t)t * nm + e] = Kt[e];
    for (int i = tid; i < m; i += NT) a.du[(size_t)t * m + i] = dut[i];
    MJPCX_STAMP(7);
  }
  __syncthreads();
  if (ok_s && T > 1) {  // backward_pass.cc:297-306: the last index repeats T-2
    for (int e = tid; e < nm; e += NT) a.K[(size_t)(T - 1) * nm + e] = a.K[(size_t)(T - 2) * nm + e];
    for (int i = tid; i < m; i += NT) a.du[(size_t)(T - 1) * m + i] = a.du[(size_t)(T - 2) * m + i];
  }
  if (tid == 0) { a.dV[0] = dV0; a.dV[1] = dV1; *a.status = ok_s; }
}

}  // namespace mjpcx
