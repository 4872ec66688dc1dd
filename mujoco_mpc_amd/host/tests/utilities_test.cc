// Port of the helper cases of mjpc/test/agent/agent_utilities_test.cc (Clamp :208-221, FindInterval :237-261,
// LinearInterpolation :263-282, CustomNumeric :69-99) plus LogScale / the iLQG helpers built on them
// (GpuILQGPlanner::BestRollout = ilqg/planner.cc:727-740; the regularisation schedule = backward_pass.cc:327-356).
#include <cmath>
#include <vector>

#include "check.h"
#include "mjpc/planners/gpu_ilqg/planner.h"
#include "mjpc/utilities.h"

using namespace mjpc;

int main() {
  {  // Clamp
    double bounds[6] = {-1.0, 1.0, -1.0, 1.0, -1.0, 1.0};
    double x[3] = {-2.0, 3.0, 0.0};
    Clamp(x, bounds, 3);
    CHECK_NEAR(x[0], -1.0, 1e-5); CHECK_NEAR(x[1], 1.0, 1e-5); CHECK_NEAR(x[2], 0.0, 1e-5);
  }
  {  // FindInterval
    std::vector<double> sequence{-1.0, 0.0, 1.0, 2.0};
    int bounds[2];
    FindInterval(bounds, sequence.data(), 0.5, 4);
    CHECK(bounds[0] == 1 && bounds[1] == 2);
    FindInterval(bounds, sequence.data(), -2.0, 4);
    CHECK(bounds[0] == 0 && bounds[1] == 0);
    FindInterval(bounds, sequence.data(), 2.1, 4);
    CHECK(bounds[0] == 3 && bounds[1] == 3);
    FindInterval(bounds, sequence.data(), 1.0, 4);  // on a knot: upper_bound semantics
    CHECK(bounds[0] == 2 && bounds[1] == 3);
  }
  {  // LinearInterpolation / ZeroInterpolation
    std::vector<double> x{1.0, 2.0};
    double y[2] = {1.0, 2.0}, out;
    LinearInterpolation(&out, 1.5, x.data(), y, 1, 2); CHECK_NEAR(out, 1.5, 1e-5);
    LinearInterpolation(&out, 0.5, x.data(), y, 1, 2); CHECK_NEAR(out, 1.0, 1e-5);
    LinearInterpolation(&out, 2.5, x.data(), y, 1, 2); CHECK_NEAR(out, 2.0, 1e-5);
    ZeroInterpolation(&out, 1.9, x.data(), y, 1, 2); CHECK_NEAR(out, 1.0, 1e-12);
  }
  {  // CubicInterpolation (utilities.cc:397-422): Hermite with finite-difference slopes
    std::vector<double> x{0.0, 1.0, 2.0, 4.0};
    double out;
    double lin[4] = {1.0, 3.0, 5.0, 9.0};          // y = 1 + 2 x: every secant is 2, the cubic reproduces the line
    for (double q : {0.25, 1.5, 2.7, 3.9}) { CubicInterpolation(&out, q, x.data(), lin, 1, 4); CHECK_NEAR(out, 1.0 + 2.0 * q, 1e-12); }
    CubicInterpolation(&out, -1.0, x.data(), lin, 1, 4); CHECK_NEAR(out, 1.0, 1e-12);   // below the grid: first value
    CubicInterpolation(&out, 5.0, x.data(), lin, 1, 4); CHECK_NEAR(out, 9.0, 1e-12);    // above: last value
    double y[4] = {0.0, 1.0, 0.0, 2.0};
    // interval [1, 2]: m(1) = mean of the secants +1 and -1 = 0, m(2) = mean of +1 and -1 = 0: the middle value is 0.5 (1 + 0)
    CubicInterpolation(&out, 1.5, x.data(), y, 1, 4); CHECK_NEAR(out, 0.5, 1e-12);
    // first interval: m(0) = secant(1, 0) = 1 (one-sided), m(1) = 0: p(0.5) = 0.5 * 0 + 0.125 * 1 + 0.5 * 1 - 0.125 * 0 = 0.625
    CubicInterpolation(&out, 0.5, x.data(), y, 1, 4); CHECK_NEAR(out, 0.625, 1e-12);
    // last interval [2, 4]: m(2) = 0, m(4) = secant(3, 2) = 1 (one-sided): p(3) = 0.5 * 0 + 0 + 0.5 * 2 - 0.125 * 2 * 1 = 0.75
    CubicInterpolation(&out, 3.0, x.data(), y, 1, 4); CHECK_NEAR(out, 0.75, 1e-12);
    // two-point grid: the end slopes are zero except the first interval's one-sided secant
    std::vector<double> x2{0.0, 1.0};
    double y2[2] = {0.0, 1.0};
    CHECK_NEAR(FiniteDifferenceSlope(1.0, x2.data(), y2, 1, 2, 0), 0.0, 1e-15);
    CHECK_NEAR(FiniteDifferenceSlope(0.0, x2.data(), y2, 1, 2, 0), 1.0, 1e-15);
  }
  {  // LogScale: ascending from min to max, geometric
    double v[4];
    LogScale(v, 1.0, 1.0e-3, 4);
    CHECK_NEAR(v[0], 1e-3, 1e-15); CHECK_NEAR(v[1], 1e-2, 1e-14); CHECK_NEAR(v[2], 1e-1, 1e-13); CHECK_NEAR(v[3], 1.0, 1e-12);
  }
  {  // BestRollout: scans from the last index with a strict <, skipping failed rollouts
    CHECK(GpuILQGPlanner::BestRollout({3.0, 1.0, 1.0, 2.0}, {0, 0, 0, 0}) == 2);
    CHECK(GpuILQGPlanner::BestRollout({0.5, 1.0, 1.0, 2.0}, {1, 0, 0, 0}) == 2);
    CHECK(GpuILQGPlanner::BestRollout({0.5, 1.0}, {1, 1}) == -1);
  }
  {  // regularisation schedule
    GpuILQGPlanner p;
    p.ScaleRegularization(2.0, 1e-6, 1e6);
    CHECK_NEAR(p.regularization_rate, 2.0, 0); CHECK_NEAR(p.regularization, 2.0, 0);
    p.ScaleRegularization(2.0, 1e-6, 1e6);
    CHECK_NEAR(p.regularization_rate, 4.0, 0); CHECK_NEAR(p.regularization, 8.0, 0);
    p.ScaleRegularization(0.5, 1e-6, 1e6);  // min(4 * 0.5, 0.5) = 0.5
    CHECK_NEAR(p.regularization_rate, 0.5, 0); CHECK_NEAR(p.regularization, 4.0, 0);
    p.UpdateRegularization(1e-6, 1e6, /*z=*/0.6, /*s=*/0.1);  // good step: divide
    CHECK_NEAR(p.regularization_rate, 0.25, 0); CHECK_NEAR(p.regularization, 1.0, 0);
    p.UpdateRegularization(1e-6, 1e6, NAN, 0.1);              // bad: factor^2
    CHECK_NEAR(p.regularization_rate, 4.0, 0); CHECK_NEAR(p.regularization, 4.0, 0);
    p.UpdateRegularization(1e-6, 1e6, 0.3, 0.1);              // neither: unchanged
    CHECK_NEAR(p.regularization, 4.0, 0);
  }
  TEST_MAIN_END();
}
