#include "norm.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace mjpc {

int NormParameterDimension(int type) {
  switch (type) {
    case kL22: case kSmoothAbs2Loss: return 2;
    case kL2: case kCosh: case kPowerLoss: case kSmoothAbsLoss: case kRectifyLoss: return 1;
    default: return 0;
  }
}

namespace {
// separable norms: y = sum_i f(x_i), g_i = f'(x_i), H_ii = f''(x_i)
template <typename F>
double Separable(double* g, double* H, const double* x, int n, F f) {
  double y = 0;
  for (int i = 0; i < n; i++) {
    double d1 = 0, d2 = 0;
    y += f(x[i], d1, d2);
    if (g) g[i] = d1;
    if (H) H[i * n + i] = d2;
  }
  return y;
}
}  // namespace

double Norm(double* g, double* H, const double* x, const double* params, int n, NormType type) {
  if (H && !g) throw std::invalid_argument("Called Norm with H and no g");
  const double p = params ? params[0] : 0, q = params ? params[1] : 0;
  if (H) std::fill(H, H + n * n, 0.0);
  switch (type) {
    case kNull:
      if (g) g[0] = 1.0;
      return x[0];
    case kQuadratic:
      return Separable(g, H, x, n, [](double v, double& d1, double& d2) { d1 = v; d2 = 1.0; return 0.5 * v * v; });
    case kL22: {  // ((x'x)^(q/2) + p^q)^(1/q) - p
      double c = 0;
      for (int i = 0; i < n; i++) c += x[i] * x[i];
      const double a = std::pow(c, q / 2) + std::pow(p, q);
      const double s = std::pow(a, 1 / q);
      const double d = std::pow(c, q / 2 - 1);
      const double b = s / a * d;
      if (g) for (int i = 0; i < n; i++) g[i] = b * x[i];
      if (H) {
        const double e = (1 - q) * d / a + (q - 2) / std::max(c, 1e-15);
        for (int i = 0; i < n; i++)
          for (int j = 0; j < n; j++) H[i + j * n] = b * ((i == j ? 1.0 : 0.0) + x[i] * x[j] * e);
      }
      return s - p;
    }
    case kL2: {  // sqrt(x'x + p^2) - p
      double c = p * p;
      for (int i = 0; i < n; i++) c += x[i] * x[i];
      const double s = std::sqrt(c);
      if (g) for (int i = 0; i < n; i++) g[i] = s ? x[i] * (1 / s) : 0.0;
      if (H && s)
        for (int i = 0; i < n; i++)
          for (int j = 0; j < n; j++) H[i + j * n] = ((i == j ? 1 : 0) - g[i] * g[j]) / s;
      return s - p;
    }
    case kCosh:  // p^2 (cosh(x/p) - 1)
      return Separable(g, H, x, n, [p](double v, double& d1, double& d2) {
        d1 = p * std::sinh(v / p); d2 = std::cosh(v / p); return p * p * (std::cosh(v / p) - 1.0); });
    case kPowerLoss:  // |x|^p
      return Separable(g, H, x, n, [p](double v, double& d1, double& d2) {
        const double s = std::fabs(v);
        d1 = (v > 0 ? 1 : (v < 0 ? -1 : 0)) * p * std::pow(s, p - 1);
        d2 = (p - 1) * p * std::pow(s, p - 2);
        return std::pow(s, p); });
    case kSmoothAbsLoss:  // sqrt(x^2 + p^2) - p
      return Separable(g, H, x, n, [p](double v, double& d1, double& d2) {
        const double s = std::sqrt(v * v + p * p);
        d1 = s ? v / s : 0; d2 = s ? (1 - d1 * d1) / s : 0;
        return s - p; });
    case kSmoothAbs2Loss:  // (|x|^q + p^q)^(1/q) - p
      return Separable(g, H, x, n, [p, q](double v, double& d1, double& d2) {
        const double a = std::fabs(v), d = std::pow(a, q), e = d + std::pow(p, q), s = std::pow(e, 1 / q);
        const double c = s * std::pow(a, q - 2) / e;
        d1 = c * v; d2 = c * (q - 1) * (1 - d / e);
        return s - p; });
    case kRectifyLoss:  // p log(1 + exp(x/p))
      return Separable(g, H, x, n, [p](double v, double& d1, double& d2) {
        if (p > 0) {
          const double s = std::exp(v / p);
          d1 = s / (1 + s); d2 = s / (p * (1 + s) * (1 + s));
          return p * std::log(1 + s);
        }
        d1 = v > 0 ? 1 : 0; d2 = 0;
        return v > 0 ? v : 0.0; });
  }
  throw std::invalid_argument("mj_norm: unknown norm type");
}

}  // namespace mjpc
