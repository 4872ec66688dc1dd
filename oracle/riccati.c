/* riccati.c -- oracle restatement of the iLQG backward pass:
 *   iLQGBackwardPass::RiccatiStep / Riccati (mjpc/planners/ilqg/backward_pass.cc:65-250, 253-324)
 * and of the box-constrained QP it calls, `mju_boxQP` (MuJoCo @088079ef, engine_util_solve.c; absent
 * from the reference tree, restated from its documented algorithm: the projected-Newton box-QP of
 * Tassa, Mansard & Todorov, "Control-limited differential dynamic programming", ICRA 2014).
 * TEST INFRASTRUCTURE ONLY (see oracle.h). Pinned by the reference's golden vectors in
 * mjpc/test/ilqg_planner/backward_pass_test.cc:101-138 (tests/test_oracle_riccati.py). */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ---- small dense helpers (row-major) ---- */
static void mat_t_mat(double* r, const double* a, const double* b, int ra, int ca, int cb) { /* r = a' b, a: ra x ca */
  for (int i = 0; i < ca; i++)
    for (int j = 0; j < cb; j++) {
      double s = 0;
      for (int k = 0; k < ra; k++) s += a[k * ca + i] * b[k * cb + j];
      r[i * cb + j] = s;
    }
}
static void mat_mat(double* r, const double* a, const double* b, int ra, int ca, int cb) {
  for (int i = 0; i < ra; i++)
    for (int j = 0; j < cb; j++) {
      double s = 0;
      for (int k = 0; k < ca; k++) s += a[i * ca + k] * b[k * cb + j];
      r[i * cb + j] = s;
    }
}
static void mat_vec(double* r, const double* a, const double* v, int ra, int ca) {
  for (int i = 0; i < ra; i++) {
    double s = 0;
    for (int k = 0; k < ca; k++) s += a[i * ca + k] * v[k];
    r[i] = s;
  }
}
static void mat_t_vec(double* r, const double* a, const double* v, int ra, int ca) {
  for (int j = 0; j < ca; j++) {
    double s = 0;
    for (int k = 0; k < ra; k++) s += a[k * ca + j] * v[k];
    r[j] = s;
  }
}
/* in-place Cholesky (lower); returns rank */
static int chol_factor(double* a, int n) {
  int rank = n;
  for (int j = 0; j < n; j++) {
    double s = a[j * n + j];
    for (int k = 0; k < j; k++) s -= a[j * n + k] * a[j * n + k];
    if (s < 1e-15) { s = 1e-15; rank--; }
    s = sqrt(s);
    a[j * n + j] = s;
    for (int i = j + 1; i < n; i++) {
      double t = a[i * n + j];
      for (int k = 0; k < j; k++) t -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = t / s;
    }
  }
  return rank;
}
static void chol_solve(double* x, const double* L, const double* b, int n) {
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

/* min 1/2 x'Hx + x'g, lower <= x <= upper. res: in = warm start, out = solution. R: Cholesky factor of
 * the free sub-Hessian (nfree x nfree), index: free dimensions. Returns nfree, or -1 if H_free is not PD. */
int oboxqp(double* res, double* R, int* index, const double* H, const double* g, int n, const double* lower,
           const double* upper) {
  const int maxiter = 100;
  const double mingrad = 1e-16, backtrack = 0.5, minstep = 1e-22, armijo = 0.1;
  double* grad = (double*)malloc(sizeof(double) * (5 * n + 1));
  double *search = grad + n, *cand = search + n, *tmp = cand + n, *rhs = tmp + n;
  int nfree = 0;
  for (int i = 0; i < n; i++) res[i] = res[i] < lower[i] ? lower[i] : (res[i] > upper[i] ? upper[i] : res[i]);
  double oldvalue = 0;
  for (int iter = 0; iter < maxiter; iter++) {
    mat_vec(tmp, H, res, n, n);
    double value = 0;
    for (int i = 0; i < n; i++) { value += 0.5 * res[i] * tmp[i] + g[i] * res[i]; grad[i] = tmp[i] + g[i]; }
    if (iter > 0 && (oldvalue - value) < 1e-8 * fabs(oldvalue)) { /* minimum relative improvement (boxQP.m) */
      /* free set of the final point, for the caller's K computation */
      nfree = 0;
      for (int i = 0; i < n; i++)
        if (!((res[i] <= lower[i] && grad[i] > 0) || (res[i] >= upper[i] && grad[i] < 0))) index[nfree++] = i;
      for (int a = 0; a < nfree; a++)
        for (int b = 0; b < nfree; b++) R[a * nfree + b] = H[index[a] * n + index[b]];
      if (nfree && chol_factor(R, nfree) < nfree) { free(grad); return -1; }
      break;
    }
    oldvalue = value;
    nfree = 0;
    for (int i = 0; i < n; i++) {
      int clamped = (res[i] <= lower[i] && grad[i] > 0) || (res[i] >= upper[i] && grad[i] < 0);
      if (!clamped) index[nfree++] = i;
    }
    if (nfree == 0) break;
    for (int a = 0; a < nfree; a++)
      for (int b = 0; b < nfree; b++) R[a * nfree + b] = H[index[a] * n + index[b]];
    if (chol_factor(R, nfree) < nfree) { free(grad); return -1; }
    double gnorm = 0;
    for (int a = 0; a < nfree; a++) gnorm += grad[index[a]] * grad[index[a]];
    if (sqrt(gnorm) < mingrad) break;
    /* Newton step in the free subspace with the clamped coordinates held */
    for (int i = 0; i < n; i++) tmp[i] = 0;
    { int a = 0;
      for (int i = 0; i < n; i++) { if (a < nfree && index[a] == i) a++; else tmp[i] = res[i]; } }
    mat_vec(cand, H, tmp, n, n); /* H * x_clamped */
    for (int a = 0; a < nfree; a++) rhs[a] = -(g[index[a]] + cand[index[a]]);
    chol_solve(tmp, R, rhs, nfree);
    for (int i = 0; i < n; i++) search[i] = 0;
    for (int a = 0; a < nfree; a++) search[index[a]] = tmp[a] - res[index[a]];
    double sdotg = 0;
    for (int i = 0; i < n; i++) sdotg += search[i] * grad[i];
    if (sdotg >= 0) break;
    double step = 1;
    int ok = 0;
    while (step > minstep) {
      for (int i = 0; i < n; i++) {
        double c = res[i] + step * search[i];
        cand[i] = c < lower[i] ? lower[i] : (c > upper[i] ? upper[i] : c);
      }
      mat_vec(tmp, H, cand, n, n);
      double vc = 0;
      for (int i = 0; i < n; i++) vc += 0.5 * cand[i] * tmp[i] + g[i] * cand[i];
      if ((vc - value) / (step * sdotg) >= armijo) { ok = 1; break; }
      step *= backtrack;
    }
    if (!ok) break;
    memcpy(res, cand, sizeof(double) * n);
  }
  free(grad);
  return nfree;
}

/* one Riccati step; W* = cost-to-go at t+1 */
static int riccati_step(int n, int m, double mu, const double* Wx, const double* Wxx, const double* At,
                        const double* Bt, const double* cxt, const double* cut, const double* cxxt,
                        const double* cxut, const double* cuut, double* Vxt, double* Vxxt, double* dut, double* Kt,
                        double* dV, double* boxres, const double* action, const double* limits, int reg_type,
                        int use_limits) {
  int mmn = m > n ? m : n;
  double* w = (double*)calloc((size_t)(2 * n + 2 * m + 2 * n * n + 2 * n * m + 2 * m * m + 4 * mmn * mmn + m * (m + 7) + 2 * m), sizeof(double));
  double *Qx = w, *Qu = Qx + n, *Qxx = Qu + m, *Qxu = Qxx + n * n, *Quu = Qxu + n * m;
  double *Qxu_reg = Quu + m * m, *Quu_reg = Qxu_reg + n * m, *Vxx_reg = Quu_reg + m * m;
  double *tmp = Vxx_reg + n * n, *tmp2 = tmp + mmn * mmn, *tmp3 = tmp2 + mmn * mmn, *tmp4 = tmp3 + mmn * mmn;
  double *R = tmp4 + mmn * mmn, *lo = R + m * (m + 7), *hi = lo + m;
  int status = 1;
  mat_t_mat(tmp, At, Wxx, n, n, n);                 /* A' Wxx */
  mat_t_vec(Qx, At, Wx, n, n);
  for (int i = 0; i < n; i++) Qx[i] += cxt[i];
  mat_mat(Qxx, tmp, At, n, n, n);
  for (int i = 0; i < n * n; i++) Qxx[i] += cxxt[i];
  mat_t_vec(Qu, Bt, Wx, n, m);
  for (int i = 0; i < m; i++) Qu[i] += cut[i];
  mat_mat(Qxu, tmp, Bt, n, n, m);
  for (int i = 0; i < n * m; i++) Qxu[i] += cxut[i];
  mat_t_mat(tmp2, Bt, Wxx, n, m, n);
  mat_mat(Quu, tmp2, Bt, m, n, m);
  for (int i = 0; i < m * m; i++) Quu[i] += cuut[i];
  if (reg_type == 2) { /* value regularisation */
    memcpy(Vxx_reg, Wxx, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) Vxx_reg[n * i + i] += mu;
    mat_t_mat(tmp, At, Vxx_reg, n, n, n);
    mat_mat(Qxu_reg, tmp, Bt, n, n, m);
    for (int i = 0; i < n * m; i++) Qxu_reg[i] += cxut[i];
    mat_t_mat(tmp2, Bt, Vxx_reg, n, m, n);
    mat_mat(Quu_reg, tmp2, Bt, m, n, m);
    for (int i = 0; i < m * m; i++) Quu_reg[i] += cuut[i];
  } else {
    memcpy(Qxu_reg, Qxu, sizeof(double) * n * m);
    memcpy(Quu_reg, Quu, sizeof(double) * m * m);
  }
  if (mu != 0) {
    if (reg_type == 0) {
      for (int i = 0; i < m; i++) Quu_reg[i * m + i] += mu;
    } else if (reg_type == 1) {
      mat_t_mat(tmp, At, Bt, n, n, m);
      for (int i = 0; i < n * m; i++) Qxu_reg[i] += mu * tmp[i];
      mat_t_mat(tmp, Bt, Bt, n, m, m);
      for (int i = 0; i < m * m; i++) Quu_reg[i] += mu * tmp[i];
    }
  }
  memset(Kt, 0, sizeof(double) * n * m);
  if (use_limits) {
    int* index = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
    for (int i = 0; i < m; i++) { lo[i] = limits[2 * i] - action[i]; hi[i] = limits[2 * i + 1] - action[i]; }
    int mf = oboxqp(boxres, R, index, Quu_reg, Qu, m, lo, hi);
    if (mf < 0) { status = 0; }
    else {
      /* K_free = -H_ff^-1 Qux_free */
      for (int j = 0; j < n; j++) {
        for (int i = 0; i < mf; i++) tmp[i] = Qxu[m * j + index[i]];
        chol_solve(tmp2, R, tmp, mf);
        for (int i = 0; i < mf; i++) Kt[j + n * index[i]] = -tmp2[i];
      }
      memcpy(dut, boxres, sizeof(double) * m);
    }
    free(index);
  } else {
    memcpy(tmp3, Quu_reg, sizeof(double) * m * m);
    if (chol_factor(tmp3, m) < m) status = 0;
    else {
      for (int j = 0; j < n; j++) {
        for (int i = 0; i < m; i++) tmp[i] = Qxu[m * j + i];
        chol_solve(tmp2, tmp3, tmp, m);
        for (int i = 0; i < m; i++) Kt[j + n * i] = -tmp2[i];
      }
      chol_solve(dut, tmp3, Qu, m);
      for (int i = 0; i < m; i++) dut[i] = -dut[i];
    }
  }
  if (status) {
    memcpy(Vxt, Qx, sizeof(double) * n);
    memcpy(Vxxt, Qxx, sizeof(double) * n * n);
    mat_vec(tmp, Quu, dut, m, m);
    for (int i = 0; i < m; i++) { dV[0] += dut[i] * Qu[i]; dV[1] += 0.5 * dut[i] * tmp[i]; }
    for (int i = 0; i < m; i++) tmp2[i] = tmp[i] + Qu[i];
    mat_t_vec(tmp, Kt, tmp2, m, n);
    for (int i = 0; i < n; i++) Vxt[i] += tmp[i];
    mat_vec(tmp, Qxu, dut, n, m);
    for (int i = 0; i < n; i++) Vxt[i] += tmp[i];
    mat_mat(tmp4, Quu, Kt, m, m, n);
    mat_t_mat(tmp3, Kt, tmp4, m, n, n);          /* K' Quu K */
    mat_mat(tmp2, Qxu, Kt, n, m, n);             /* Qxu K */
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) Vxxt[i * n + j] += tmp3[i * n + j] + tmp2[i * n + j] + tmp2[j * n + i];
    for (int i = 0; i < n; i++) /* mju_symmetrize */
      for (int j = i + 1; j < n; j++) {
        double s = 0.5 * (Vxxt[i * n + j] + Vxxt[j * n + i]);
        Vxxt[i * n + j] = Vxxt[j * n + i] = s;
      }
  }
  free(w);
  return status;
}

/* one backward sweep at regularisation `mu` (the retry loop with ScaleRegularization is the caller's,
 * backward_pass.cc:271-321). Returns 1 on success, 0 if some step failed. */
int oriccati(int n, int m, int T, double mu, int reg_type, int use_limits, const double* A, const double* B,
             const double* cx, const double* cu, const double* cxx, const double* cxu, const double* cuu,
             const double* actions, const double* action_limits, double* Vx, double* Vxx, double* K, double* du,
             double* dV) {
  dV[0] = dV[1] = 0;
  memcpy(Vx + (T - 1) * n, cx + (T - 1) * n, sizeof(double) * n);
  memcpy(Vxx + (T - 1) * n * n, cxx + (T - 1) * n * n, sizeof(double) * n * n);
  double* boxres = (double*)calloc(m > 0 ? m : 1, sizeof(double)); /* BoxQP::res: warm start across steps */
  int ok = 1;
  for (int t = T - 1; t > 0 && ok; t--) {
    ok = riccati_step(n, m, mu, Vx + t * n, Vxx + t * n * n, A + (t - 1) * n * n, B + (t - 1) * n * m,
                      cx + (t - 1) * n, cu + (t - 1) * m, cxx + (t - 1) * n * n, cxu + (t - 1) * n * m,
                      cuu + (t - 1) * m * m, Vx + (t - 1) * n, Vxx + (t - 1) * n * n, du + (t - 1) * m,
                      K + (t - 1) * m * n, dV, boxres, actions + (t - 1) * m, action_limits, reg_type, use_limits);
  }
  if (ok && T > 1) {
    memcpy(K + (T - 1) * m * n, K + (T - 2) * m * n, sizeof(double) * m * n);
    memcpy(du + (T - 1) * m, du + (T - 2) * m, sizeof(double) * m);
  }
  free(boxres);
  return ok;
}
