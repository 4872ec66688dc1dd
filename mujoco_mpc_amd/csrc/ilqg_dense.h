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
__device__ __forceinline__ void wave_gemm(double* Cm, int ldc, const double* Am, int lda, bool transA, const double* Bm,
                                          int ldb, int M, int N, int K, const double* Dm, int ldd, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  for (int ti = 0; ti < M; ti += 16)
    for (int tj = 0; tj < N; tj += 16) {
      v4f64 acc = {0, 0, 0, 0};
      for (int k0 = 0; k0 < K; k0 += 4) {
        const int i = ti + li, k = k0 + lk, j = tj + li;
        const double av = (i < M && k < K) ? (transA ? Am[k * lda + i] : Am[i * lda + k]) : 0.0;
        const double bv = (k < K && j < N) ? Bm[k * ldb + j] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int i = ti + lk + 4 * rg, j = tj + li;
        if (i < M && j < N) Cm[i * ldc + j] = acc[rg] + (Dm ? Dm[i * ldd + j] : 0.0);
      }
    }
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

struct BackwardArgs {
  int n, m, T;
  double mu;
  int reg_type, use_limits;
  const double *A, *B, *cx, *cu, *cxx, *cxu, *cuu, *actions, *limits;
  double *Vx, *Vxx, *K, *du, *dV;
  int* status;  // 1 ok, 0 failed (Quu not PD at some step)
};

// ONE wavefront walks the horizon backwards. LDS carve (doubles), NP = n rounded up to 16 (row stride):
//   W[NP*NP] Wx[NP] At[NP*NP] Bt[NP*16] tmp[NP*NP] tmp2[16*NP] Qxx[NP*NP] Qxu[NP*16] Quu[256] Qxur[NP*16] Quur[256]
//   Qx[NP] Qu[16] Kt[16*NP] dut[16] + box-QP scratch
__global__ __launch_bounds__(64) void backward_pass_kernel(const BackwardArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = a.n, m = a.m, T = a.T, lane = threadIdx.x;
  const int NP = (n + 15) & ~15, MP = 16;
  double* W = reinterpret_cast<double*>(smem_raw);
  double* Wx = W + NP * NP; double* At = Wx + NP; double* Bt = At + NP * NP; double* tmp = Bt + NP * MP;
  double* tmp2 = tmp + NP * NP; double* Qxx = tmp2 + MP * NP; double* Qxu = Qxx + NP * NP; double* Quu = Qxu + NP * MP;
  double* Qxur = Quu + MP * MP; double* Quur = Qxur + NP * MP; double* Qx = Quur + MP * MP; double* Qu = Qx + NP;
  double* Kt = Qu + MP; double* dut = Kt + MP * NP; double* boxres = dut + MP; double* boxR = boxres + MP;
  double* lo = boxR + MP * (MP + 7); double* hi = lo + MP; double* scratch = hi + MP;  // 5*MP
  double* Hc = scratch + 5 * MP;  // compact m x m copy of Quu_reg
  double* KQ = Hc + MP * MP;      // MP x NP: Quu K
  __shared__ int index[16];
  __shared__ int nfree_s, ok_s;
  __shared__ double dV0, dV1;
  if (lane == 0) { dV0 = 0; dV1 = 0; ok_s = 1; }
  for (int i = lane; i < MP; i += 64) boxres[i] = 0;  // BoxQP::res warm start, reset per sweep
  // terminal condition: V = c at T-1
  for (int e = lane; e < n * n; e += 64) W[(e / n) * NP + e % n] = a.cxx[(size_t)(T - 1) * n * n + e];
  for (int i = lane; i < n; i += 64) Wx[i] = a.cx[(size_t)(T - 1) * n + i];
  for (int e = lane; e < n * n; e += 64) a.Vxx[(size_t)(T - 1) * n * n + e] = a.cxx[(size_t)(T - 1) * n * n + e];
  for (int i = lane; i < n; i += 64) a.Vx[(size_t)(T - 1) * n + i] = a.cx[(size_t)(T - 1) * n + i];
  __syncthreads();
  for (int t = T - 2; t >= 0; t--) {
    const double* Ag = a.A + (size_t)t * n * n; const double* Bg = a.B + (size_t)t * n * m;
    const double* cxg = a.cx + (size_t)t * n; const double* cug = a.cu + (size_t)t * m;
    const double* cxxg = a.cxx + (size_t)t * n * n; const double* cxug = a.cxu + (size_t)t * n * m;
    const double* cuug = a.cuu + (size_t)t * m * m;
    for (int e = lane; e < n * n; e += 64) At[(e / n) * NP + e % n] = Ag[e];
    for (int e = lane; e < n * m; e += 64) Bt[(e / m) * MP + e % m] = Bg[e];
    __syncthreads();
    // tmp = A' W ; Qxx = tmp A + cxx ; Qxu = tmp B + cxu ; tmp2 = B' W ; Quu = tmp2 B + cuu   (matrix cores)
    wave_gemm(tmp, NP, At, NP, true, W, NP, n, n, n, nullptr, 0, lane);
    __syncthreads();
    wave_gemm(Qxx, NP, tmp, NP, false, At, NP, n, n, n, nullptr, 0, lane);
    wave_gemm(Qxu, MP, tmp, NP, false, Bt, MP, n, m, n, nullptr, 0, lane);
    wave_gemm(tmp2, NP, Bt, MP, true, W, NP, m, n, n, nullptr, 0, lane);
    __syncthreads();
    wave_gemm(Quu, MP, tmp2, NP, false, Bt, MP, m, m, n, nullptr, 0, lane);
    __syncthreads();
    for (int e = lane; e < n * n; e += 64) Qxx[(e / n) * NP + e % n] += cxxg[e];
    for (int e = lane; e < n * m; e += 64) Qxu[(e / m) * MP + e % m] += cxug[e];
    for (int e = lane; e < m * m; e += 64) Quu[(e / m) * MP + e % m] += cuug[e];
    for (int i = lane; i < n; i += 64) { double s = cxg[i]; for (int k = 0; k < n; k++) s += At[k * NP + i] * Wx[k]; Qx[i] = s; }
    for (int i = lane; i < m; i += 64) { double s = cug[i]; for (int k = 0; k < n; k++) s += Bt[k * MP + i] * Wx[k]; Qu[i] = s; }
    __syncthreads();
    // ---- regularisation
    if (a.reg_type == 2) {  // value: recompute with W + mu I
      for (int i = lane; i < n; i += 64) W[i * NP + i] += a.mu;
      __syncthreads();
      wave_gemm(tmp, NP, At, NP, true, W, NP, n, n, n, nullptr, 0, lane);
      wave_gemm(tmp2, NP, Bt, MP, true, W, NP, m, n, n, nullptr, 0, lane);
      __syncthreads();
      wave_gemm(Qxur, MP, tmp, NP, false, Bt, MP, n, m, n, nullptr, 0, lane);
      wave_gemm(Quur, MP, tmp2, NP, false, Bt, MP, m, m, n, nullptr, 0, lane);
      __syncthreads();
      for (int e = lane; e < n * m; e += 64) Qxur[(e / m) * MP + e % m] += cxug[e];
      for (int e = lane; e < m * m; e += 64) Quur[(e / m) * MP + e % m] += cuug[e];
      for (int i = lane; i < n; i += 64) W[i * NP + i] -= a.mu;
    } else {
      for (int e = lane; e < n * m; e += 64) Qxur[(e / m) * MP + e % m] = Qxu[(e / m) * MP + e % m];
      for (int e = lane; e < m * m; e += 64) Quur[(e / m) * MP + e % m] = Quu[(e / m) * MP + e % m];
      __syncthreads();
      if (a.mu != 0 && a.reg_type == 0) {
        for (int i = lane; i < m; i += 64) Quur[i * MP + i] += a.mu;
      } else if (a.mu != 0 && a.reg_type == 1) {
        wave_gemm(tmp, MP, At, NP, true, Bt, MP, n, m, n, nullptr, 0, lane);   // A'B (n x m) into tmp (ld MP)
        wave_gemm(tmp2, MP, Bt, MP, true, Bt, MP, m, m, n, nullptr, 0, lane);  // B'B (m x m)
        __syncthreads();
        for (int e = lane; e < n * m; e += 64) Qxur[(e / m) * MP + e % m] += a.mu * tmp[(e / m) * MP + e % m];
        for (int e = lane; e < m * m; e += 64) Quur[(e / m) * MP + e % m] += a.mu * tmp2[(e / m) * MP + e % m];
      }
    }
    __syncthreads();
    // ---- du and K
    for (int e = lane; e < m * n; e += 64) Kt[(e / n) * NP + e % n] = 0;
    for (int e = lane; e < m * m; e += 64) Hc[e] = Quur[(e / m) * MP + e % m];
    __syncthreads();
    if (lane == 0) {
      int ok = 1;
      if (a.use_limits) {
        for (int i = 0; i < m; i++) { lo[i] = a.limits[2 * i] - a.actions[(size_t)t * m + i]; hi[i] = a.limits[2 * i + 1] - a.actions[(size_t)t * m + i]; }
        const int mf = boxqp_serial(boxres, boxR, index, Hc, Qu, m, lo, hi, scratch);
        if (mf < 0) ok = 0;
        nfree_s = mf < 0 ? 0 : mf;
        for (int i = 0; i < m; i++) dut[i] = boxres[i];
      } else {
        for (int e = 0; e < m * m; e++) boxR[e] = Hc[e];
        if (chol_factor_serial(boxR, m) < m) ok = 0;
        for (int i = 0; i < m; i++) index[i] = i;
        nfree_s = m;
        if (ok) { chol_solve_serial(dut, boxR, Qu, m); for (int i = 0; i < m; i++) dut[i] = -dut[i]; }
      }
      if (!ok) ok_s = 0;
    }
    __syncthreads();
    if (!ok_s) break;
    {  // K_free = -H_ff^-1 Qux_free: one lane per state column j
      const int mf = nfree_s;
      for (int j = lane; j < n; j += 64) {
        double rhs[16], sol[16];
        for (int i = 0; i < mf; i++) rhs[i] = Qxu[j * MP + index[i]];
        chol_solve_serial(sol, boxR, rhs, mf);
        for (int i = 0; i < mf; i++) Kt[index[i] * NP + j] = -sol[i];
      }
    }
    __syncthreads();
    // ---- cost-to-go update
    if (lane == 0) {
      double d0 = 0, d1 = 0;
      for (int i = 0; i < m; i++) {
        double s = 0;
        for (int k = 0; k < m; k++) s += Quu[i * MP + k] * dut[k];
        scratch[i] = s + Qu[i];  // Quu du + Qu
        d0 += dut[i] * Qu[i];
        d1 += 0.5 * dut[i] * s;
      }
      dV0 += d0; dV1 += d1;
    }
    wave_gemm(KQ, NP, Quu, MP, false, Kt, NP, m, n, m, nullptr, 0, lane);  // Quu K  (m x n)
    __syncthreads();
    // Vx = Qx + K'(Quu du + Qu) + Qxu du
    for (int i = lane; i < n; i += 64) {
      double s = Qx[i];
      for (int k = 0; k < m; k++) s += Kt[k * NP + i] * scratch[k] + Qxu[i * MP + k] * dut[k];
      Wx[i] = s;
    }
    // Vxx = Qxx + K'(Quu K) + Qxu K + (Qxu K)'
    wave_gemm(tmp, NP, Kt, NP, true, KQ, NP, n, n, m, Qxx, NP, lane);      // Qxx + K' Quu K
    wave_gemm(W, NP, Qxu, MP, false, Kt, NP, n, n, m, nullptr, 0, lane);   // Qxu K -> W (the old W is dead now)
    __syncthreads();
    for (int e = lane; e < n * n; e += 64) {
      const int i = e / n, j = e % n;
      tmp[i * NP + j] += W[i * NP + j] + W[j * NP + i];
    }
    __syncthreads();
    for (int e = lane; e < n * n; e += 64) {  // mju_symmetrize
      const int i = e / n, j = e % n;
      W[i * NP + j] = 0.5 * (tmp[i * NP + j] + tmp[j * NP + i]);
    }
    __syncthreads();
    // ---- write step t
    for (int e = lane; e < n * n; e += 64) a.Vxx[(size_t)t * n * n + e] = W[(e / n) * NP + e % n];
    for (int i = lane; i < n; i += 64) a.Vx[(size_t)t * n + i] = Wx[i];
    for (int e = lane; e < m * n; e += 64) a.K[(size_t)t * m * n + e] = Kt[(e / n) * NP + e % n];
    for (int i = lane; i < m; i += 64) a.du[(size_t)t * m + i] = dut[i];
    __syncthreads();
  }
  __syncthreads();
  if (ok_s && T > 1) {  // backward_pass.cc:297-306: the last index repeats T-2
    for (int e = lane; e < m * n; e += 64) a.K[(size_t)(T - 1) * m * n + e] = a.K[(size_t)(T - 2) * m * n + e];
    for (int i = lane; i < m; i += 64) a.du[(size_t)(T - 1) * m + i] = a.du[(size_t)(T - 2) * m + i];
  }
  if (lane == 0) { a.dV[0] = dV0; a.dV[1] = dV1; *a.status = ok_s; }
}

}  // namespace mjpcx
