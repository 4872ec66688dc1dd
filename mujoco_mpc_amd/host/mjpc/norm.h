// mjpc::Norm with the reference's interface (mjpc/norm.h).
#pragma once
namespace mjpc {
inline constexpr int kMaxNormParameters = 3;
enum NormType : int {
  kNull = -1, kQuadratic = 0, kL22 = 1, kL2 = 2, kCosh = 3,
  kPowerLoss = 5, kSmoothAbsLoss = 6, kSmoothAbs2Loss = 7, kRectifyLoss = 8,
};
int NormParameterDimension(int type);
// value of the norm of x[0:n]; optionally its gradient g[n] and Hessian H[n*n] (H requires g)
double Norm(double* g, double* H, const double* x, const double* params, int n, NormType type);
}  // namespace mjpc
