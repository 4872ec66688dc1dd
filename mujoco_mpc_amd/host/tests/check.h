// Tiny assertion helpers for the C++ host tests (googletest is not available offline).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static int g_failures = 0;
#define CHECK(cond)                                                              \
  do {                                                                           \
    if (!(cond)) { std::printf("%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); g_failures++; } \
  } while (0)
#define CHECK_NEAR(a, b, tol)                                                    \
  do {                                                                           \
    const double a__ = (a), b__ = (b);                                           \
    if (!(std::fabs(a__ - b__) <= (tol))) {                                      \
      std::printf("%s:%d: CHECK_NEAR failed: %s = %.17g vs %s = %.17g (tol %g)\n", __FILE__, __LINE__, #a, a__, #b, b__, (double)(tol)); \
      g_failures++;                                                              \
    }                                                                            \
  } while (0)
#define CHECK_THROWS(stmt)                                                       \
  do {                                                                           \
    bool thrown__ = false;                                                       \
    try { stmt; } catch (...) { thrown__ = true; }                               \
    if (!thrown__) { std::printf("%s:%d: expected an exception: %s\n", __FILE__, __LINE__, #stmt); g_failures++; } \
  } while (0)
inline bool Eq(const std::vector<double>& a, std::initializer_list<double> b) {
  return a == std::vector<double>(b);
}
#define TEST_MAIN_END()                                                          \
  do {                                                                           \
    if (g_failures) { std::printf("FAILED: %d check(s)\n", g_failures); return 1; } \
    std::printf("OK\n");                                                         \
    return 0;                                                                    \
  } while (0)
