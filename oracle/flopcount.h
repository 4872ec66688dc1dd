#pragma once
// flopcount.h -- the oracle with a floating-point operation counter. TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// The oracle's C sources are compiled here, unchanged, as C++ with `double` replaced by a class whose arithmetic operators count
// what they execute: additions / subtractions, multiplications, divisions, square roots, other libm calls (sin, cos, pow, exp, log,
// atan2 ...: one each). Comparisons, negations, fabs, copies and conversions are free. The count is of the REFERENCE ALGORITHM as
// the oracle restates it (dense Jacobian rows, dense nv x nv Cholesky, mj_step's stages), not of what the device kernels execute:
// it is the numerator of bench.py's `roofline.fp64` (flop per candidate-step x candidate-steps / kernel time, against the
// 78.6 TFLOP/s FP64 vector peak of MI355X). Built on demand as liboracle_flops.so (make liboracle_flops.so; pyoracle.flops_build());
// never loaded by the parity tests' checker path and never timed.
//
// Used as a forced include (g++ -x c++ -include flopcount.h -c physics.c ...): every system header the C sources use is included
// here FIRST, so that the macros at the end only reach the oracle's own text.
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

namespace oflops {
// per-thread tallies (the oracle's fan-out runs rollouts on worker threads), summed into the process totals
extern std::atomic<unsigned long long> g_total[5];
struct Tally {
  unsigned long long add = 0, mul = 0, div = 0, sqrt_ = 0, libm = 0;
  void flush() {
    g_total[0] += add; g_total[1] += mul; g_total[2] += div; g_total[3] += sqrt_; g_total[4] += libm;
    add = mul = div = sqrt_ = libm = 0;
  }
  ~Tally() { flush(); }  // a worker thread of the oracle's fan-out folds its tally in when it exits (the batch entry points join their workers)
};
extern thread_local Tally t_tally;
inline void flush() { t_tally.flush(); }
typedef double real;
struct cdbl {
  real v;
  cdbl() = default;
  cdbl(real x) : v(x) {}
  cdbl(int x) : v(x) {}
  cdbl(unsigned x) : v(x) {}
  cdbl(long x) : v((real)x) {}
  cdbl(unsigned long x) : v((real)x) {}
  cdbl(long long x) : v((real)x) {}
  cdbl(unsigned long long x) : v((real)x) {}
  cdbl(float x) : v(x) {}
  explicit operator int() const { return (int)v; }
  explicit operator unsigned() const { return (unsigned)v; }
  explicit operator long() const { return (long)v; }
  explicit operator unsigned long() const { return (unsigned long)v; }
  explicit operator long long() const { return (long long)v; }
  explicit operator unsigned long long() const { return (unsigned long long)v; }
  explicit operator float() const { return (float)v; }
  explicit operator bool() const { return v != 0; }
  cdbl& operator+=(cdbl o) { t_tally.add++; v += o.v; return *this; }
  cdbl& operator-=(cdbl o) { t_tally.add++; v -= o.v; return *this; }
  cdbl& operator*=(cdbl o) { t_tally.mul++; v *= o.v; return *this; }
  cdbl& operator/=(cdbl o) { t_tally.div++; v /= o.v; return *this; }
  cdbl operator-() const { return cdbl(-v); }
  cdbl operator+() const { return *this; }
  bool operator!() const { return v == 0; }
};
#define OFL_BIN(op, field) \
  inline cdbl operator op(cdbl a, cdbl b) { t_tally.field++; return cdbl(a.v op b.v); } \
  inline cdbl operator op(cdbl a, real b) { t_tally.field++; return cdbl(a.v op b); } \
  inline cdbl operator op(real a, cdbl b) { t_tally.field++; return cdbl(a op b.v); } \
  inline cdbl operator op(cdbl a, int b) { t_tally.field++; return cdbl(a.v op b); } \
  inline cdbl operator op(int a, cdbl b) { t_tally.field++; return cdbl(a op b.v); }
OFL_BIN(+, add) OFL_BIN(-, add) OFL_BIN(*, mul) OFL_BIN(/, div)
#undef OFL_BIN
#define OFL_CMP(op) \
  inline bool operator op(cdbl a, cdbl b) { return a.v op b.v; } \
  inline bool operator op(cdbl a, real b) { return a.v op b; } \
  inline bool operator op(real a, cdbl b) { return a op b.v; } \
  inline bool operator op(cdbl a, int b) { return a.v op b; } \
  inline bool operator op(int a, cdbl b) { return a op b.v; }
OFL_CMP(<) OFL_CMP(<=) OFL_CMP(>) OFL_CMP(>=) OFL_CMP(==) OFL_CMP(!=)
#undef OFL_CMP
inline cdbl c_sqrt(cdbl x) { t_tally.sqrt_++; return cdbl(::sqrt(x.v)); }
inline cdbl c_fabs(cdbl x) { return cdbl(::fabs(x.v)); }
#define OFL_LIBM1(name) inline cdbl c_##name(cdbl x) { t_tally.libm++; return cdbl(::name(x.v)); }
OFL_LIBM1(sin) OFL_LIBM1(cos) OFL_LIBM1(tan) OFL_LIBM1(exp) OFL_LIBM1(log) OFL_LIBM1(acos) OFL_LIBM1(asin) OFL_LIBM1(atan) OFL_LIBM1(cosh) OFL_LIBM1(sinh)
OFL_LIBM1(tanh) OFL_LIBM1(floor) OFL_LIBM1(ceil) OFL_LIBM1(round) OFL_LIBM1(log1p) OFL_LIBM1(expm1)
#undef OFL_LIBM1
inline cdbl c_pow(cdbl x, cdbl y) { t_tally.libm++; return cdbl(::pow(x.v, y.v)); }
inline cdbl c_atan2(cdbl y, cdbl x) { t_tally.libm++; return cdbl(::atan2(y.v, x.v)); }
inline cdbl c_fmod(cdbl x, cdbl y) { t_tally.libm++; return cdbl(::fmod(x.v, y.v)); }
inline cdbl c_fmin(cdbl x, cdbl y) { return cdbl(::fmin(x.v, y.v)); }
inline cdbl c_fmax(cdbl x, cdbl y) { return cdbl(::fmax(x.v, y.v)); }
inline cdbl c_fma(cdbl a, cdbl b, cdbl c) { t_tally.add++; t_tally.mul++; return cdbl(::fma(a.v, b.v, c.v)); }
inline bool c_isnan(cdbl x) { return x.v != x.v; }
inline bool c_isfinite(cdbl x) { return ::isfinite(x.v); }
inline bool c_isinf(cdbl x) { return ::isinf(x.v); }
}  // namespace oflops
using oflops::cdbl;
static_assert(sizeof(cdbl) == sizeof(oflops::real), "the counting type has the layout of a double (the C API's arrays are shared with numpy)");

#define sqrt oflops::c_sqrt
#define fabs oflops::c_fabs
#define sin oflops::c_sin
#define cos oflops::c_cos
#define tan oflops::c_tan
#define exp oflops::c_exp
#define log oflops::c_log
#define acos oflops::c_acos
#define asin oflops::c_asin
#define atan oflops::c_atan
#define cosh oflops::c_cosh
#define sinh oflops::c_sinh
#define tanh oflops::c_tanh
#define floor oflops::c_floor
#define ceil oflops::c_ceil
#define round oflops::c_round
#define log1p oflops::c_log1p
#define expm1 oflops::c_expm1
#define pow oflops::c_pow
#define atan2 oflops::c_atan2
#define fmod oflops::c_fmod
#define fmin oflops::c_fmin
#define fmax oflops::c_fmax
#define fma oflops::c_fma
#undef isnan
#undef isfinite
#undef isinf
#define isnan oflops::c_isnan
#define isfinite oflops::c_isfinite
#define isinf oflops::c_isinf
#define double cdbl

