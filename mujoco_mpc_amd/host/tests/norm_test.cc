// mjpc/test/agent/norm_test.cc: analytic gradient / Hessian of every norm vs centred finite differences.
#include "mjpc/norm.h"

#include <algorithm>
#include <cmath>

#include "check.h"
using namespace mjpc;

int main() {
  struct Case { NormType type; double params[2]; };
  const Case cases[] = {{kQuadratic, {0.1, 0}}, {kL22, {0.1, 2}}, {kL2, {0.1, 0}}, {kCosh, {0.1, 0}}, {kPowerLoss, {2, 0}},
                        {kSmoothAbsLoss, {0.1, 0}}, {kSmoothAbs2Loss, {0.1, 2}}, {kRectifyLoss, {0.1, 0}}};
  const double pts[5][2] = {{0, 0}, {1, 0}, {-1, 0}, {1, 1}, {-1, -1}};
  const double eps = 1e-4;
  for (const Case& c : cases)
    for (const auto& x0 : pts) {
      auto f = [&](double a, double b) { double x[2] = {a, b}; return Norm(nullptr, nullptr, x, c.params, 2, c.type); };
      double g[2], H[4], x[2] = {x0[0], x0[1]};
      Norm(g, H, x, c.params, 2, c.type);
      const double fdg[2] = {(f(x[0] + eps / 2, x[1]) - f(x[0] - eps / 2, x[1])) / eps,
                             (f(x[0], x[1] + eps / 2) - f(x[0], x[1] - eps / 2)) / eps};
      const double gmax = std::max(std::fabs(g[0]), std::fabs(g[1]));
      for (int i = 0; i < 2; i++) CHECK_NEAR(g[i], fdg[i], gmax * 1e-3 + 1e-15);
      const double f0 = f(x[0], x[1]);
      const double fdH[4] = {(f(x[0] + 2 * eps, x[1]) - 2 * f(x[0] + eps, x[1]) + f0) / (eps * eps),
                             (f(x[0] + eps, x[1] + eps) - f(x[0] + eps, x[1]) - f(x[0], x[1] + eps) + f0) / (eps * eps), 0,
                             (f(x[0], x[1] + 2 * eps) - 2 * f(x[0], x[1] + eps) + f0) / (eps * eps)};
      const double hmax = *std::max_element(H, H + 4, [](double a, double b) { return std::fabs(a) < std::fabs(b); });
      CHECK_NEAR(H[0], fdH[0], std::fabs(hmax) * 1e-2 + 1e-15);
      CHECK_NEAR(H[1], fdH[1], std::fabs(hmax) * 1e-2 + 1e-15);
      CHECK_NEAR(H[2], fdH[1], std::fabs(hmax) * 1e-2 + 1e-15);
      CHECK_NEAR(H[3], fdH[3], std::fabs(hmax) * 1e-2 + 1e-15);
    }
  CHECK(NormParameterDimension(kL22) == 2 && NormParameterDimension(kQuadratic) == 0 && NormParameterDimension(kNull) == 0);
  double x[1] = {0.5};
  CHECK_THROWS(Norm(nullptr, x, x, nullptr, 1, kQuadratic));  // "Called Norm with H and no g"
  TEST_MAIN_END();
}
