/* norm_cost.c -- oracle restatement of mjpc::Norm (mjpc/norm.cc:50-210) and of
 * BaseResidualFn::CostTerms / CostValue (mjpc/task.cc:71-110).
 * TEST INFRASTRUCTURE ONLY (see oracle.h). Pinned by mjpc/test/agent/norm_test.cc
 * (finite-difference consistency) and mjpc/test/tasks/task_test.cc:57-95
 * (closed-form cost and risk transform), ported in tests/test_oracle_norm_cost.py. */
#include <math.h>
#include <string.h>
#include "oracle.h"

#define OMINVAL 1e-15

int onorm_parameter_dimension(int type) { /* norm.cc:25-47 */
  switch (type) {
    case MJPCX_NORM_L22: return 2;
    case MJPCX_NORM_L2: return 1;
    case MJPCX_NORM_COSH: return 1;
    case MJPCX_NORM_POWER_LOSS: return 1;
    case MJPCX_NORM_SMOOTH_ABS: return 1;
    case MJPCX_NORM_SMOOTH_ABS2: return 2;
    case MJPCX_NORM_RECTIFY: return 1;
    default: return 0;
  }
}

double onorm(double* g, double* H, const double* x, const double* params, int n, int type) {
  double y = 0;
  double p = params ? params[0] : 0, q = params ? params[1] : 0;
  if (H) memset(H, 0, sizeof(double) * n * n);
  switch (type) {
    case MJPCX_NORM_NULL:
      y = x[0];
      if (g) g[0] = 1.0;
      if (H) H[0] = 0.0;
      break;
    case MJPCX_NORM_QUADRATIC: /* 0.5 x'x */
      for (int i = 0; i < n; i++) y += x[i] * x[i];
      y *= 0.5;
      if (g) for (int i = 0; i < n; i++) g[i] = x[i];
      if (H) for (int i = 0; i < n; i++) H[i * n + i] = 1.0;
      break;
    case MJPCX_NORM_L22: { /* ((x'x)^(q/2) + p^q)^(1/q) - p */
      double c = 0;
      for (int i = 0; i < n; i++) c += x[i] * x[i];
      double a = pow(c, q / 2) + pow(p, q);
      double s = pow(a, 1 / q);
      y = s - p;
      double d = pow(c, q / 2 - 1);
      double b = s / a * d;
      if (g) for (int i = 0; i < n; i++) g[i] = b * x[i];
      if (H) {
        c = (1 - q) * d / a + (q - 2) / (c > OMINVAL ? c : OMINVAL);
        for (int i = 0; i < n; i++)
          for (int j = 0; j < n; j++) H[i + j * n] = b * ((i == j ? 1.0 : 0.0) + x[i] * x[j] * c);
      }
      break;
    }
    case MJPCX_NORM_L2: { /* sqrt(x'x + p^2) - p */
      double dot = 0;
      for (int i = 0; i < n; i++) dot += x[i] * x[i];
      double s = sqrt(dot + p * p);
      y = s - p;
      if (g) for (int i = 0; i < n; i++) g[i] = s ? x[i] * (1 / s) : 0;
      if (H && s)
        for (int i = 0; i < n; i++)
          for (int j = 0; j < n; j++) H[i + j * n] = ((i == j ? 1 : 0) - g[i] * g[j]) / s;
      break;
    }
    case MJPCX_NORM_COSH: /* p^2 (cosh(x/p) - 1) */
      for (int i = 0; i < n; i++) {
        y += p * p * (cosh(x[i] / p) - 1.0);
        if (g) g[i] = p * sinh(x[i] / p);
        if (H) H[i * n + i] = cosh(x[i] / p);
      }
      break;
    case MJPCX_NORM_POWER_LOSS: /* |x|^p */
      for (int i = 0; i < n; i++) {
        double s = fabs(x[i]);
        y += pow(s, p);
        if (g) g[i] = (x[i] > 0 ? 1 : (x[i] < 0 ? -1 : 0)) * p * pow(s, p - 1);
        if (H) H[i * n + i] = (p - 1) * p * pow(s, p - 2);
      }
      break;
    case MJPCX_NORM_SMOOTH_ABS: /* sqrt(x^2 + p^2) - p */
      for (int i = 0; i < n; i++) {
        double s = sqrt(x[i] * x[i] + p * p);
        y += s - p;
        if (g) g[i] = s ? x[i] / s : 0;
        if (H) H[n * i + i] = s ? (1 - g[i] * g[i]) / s : 0;
      }
      break;
    case MJPCX_NORM_SMOOTH_ABS2: /* (|x|^q + p^q)^(1/q) - p */
      for (int i = 0; i < n; i++) {
        double a = fabs(x[i]);
        double d = pow(a, q);
        double e = d + pow(p, q);
        double s = pow(e, 1 / q);
        y += s - p;
        double c = s * pow(a, q - 2) / e;
        if (g) g[i] = c * x[i];
        if (H) H[i * n + i] = c * (q - 1) * (1 - d / e);
      }
      break;
    case MJPCX_NORM_RECTIFY: /* p log(1 + exp(x/p)) */
      for (int i = 0; i < n; i++) {
        if (p > 0) {
          double s = exp(x[i] / p);
          y += p * log(1 + s);
          if (g) g[i] = s / (1 + s);
          if (H) H[i * n + i] = s / (p * (1 + s) * (1 + s));
        } else {
          y += x[i] > 0 ? x[i] : 0;
          if (g) g[i] = x[i] > 0 ? 1 : 0;
          if (H) H[i * n + i] = 0;
        }
      }
      break;
    default:
      return NAN;
  }
  return y;
}

void ocost_terms(const mjpcx_task* task, const double* residual, double* terms, int weighted) {
  int f_shift = 0, p_shift = 0;
  for (int k = 0; k < task->num_term; k++) {
    terms[k] = (weighted ? task->weight[k] : 1) *
               onorm(NULL, NULL, residual + f_shift, task->norm_parameter + p_shift,
                     task->dim_norm_residual[k], task->norm[k]);
    f_shift += task->dim_norm_residual[k];
    p_shift += task->num_norm_parameter[k];
  }
}

double ocost_value(const mjpcx_task* task, const double* residual) {
  double terms[MJPCX_MAX_COST_TERMS];
  ocost_terms(task, residual, terms, 1);
  double cost = 0.0;
  for (int i = 0; i < task->num_term; i++) cost += terms[i];
  if (fabs(task->risk) < 1.0e-6) return cost; /* kRiskNeutralTolerance */
  return (exp(task->risk * cost) - 1.0) / task->risk;
}
