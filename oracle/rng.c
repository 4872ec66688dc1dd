/* rng.c -- oracle restatement of the counter-based candidate noise.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference draws noise from a function-local, non-deterministically seeded
 * absl::BitGen inside each pool task (mjpc/planners/sampling/planner.cc:331,
 * cross_entropy/planner.cc:360), so "identical seeds" cannot be reproduced
 * against it. The build replaces it by Philox4x32-10 + Box-Muller
 * as specified in include/mjpcx.h; this file is the CPU statement of that spec.
 * Philox is pinned by the Random123 known-answer vectors
 * (tests/test_oracle_rng.py). */
#include <math.h>
#include "oracle.h"

static inline void mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
  uint64_t p = (uint64_t)a * b;
  *hi = (uint32_t)(p >> 32);
  *lo = (uint32_t)p;
}

void ophilox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(0xD2511F53u, c0, &hi0, &lo0);
    mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform in (0,1) from two words */
static double u53(uint32_t hi, uint32_t lo) {
  uint64_t k = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)k + 0.5) * (1.0 / 9007199254740992.0);
}

static void uniforms(uint64_t seed, uint32_t candidate, uint32_t c, uint32_t iteration,
                     uint32_t stream, double u[2]) {
  uint32_t ctr[4] = {candidate, c, iteration, stream};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t o[4];
  ophilox4x32_10(ctr, key, o);
  u[0] = u53(o[0], o[1]);
  u[1] = u53(o[2], o[3]);
}

void ogaussian_pair(uint64_t seed, uint32_t candidate, uint32_t pair, uint32_t iteration, double z[2]) {
  double u[2];
  uniforms(seed, candidate, pair, iteration, 0, u);
  double r = sqrt(-2.0 * log(u[0]));
  double th = 6.283185307179586476925286766559 * u[1];
  z[0] = r * cos(th);
  z[1] = r * sin(th);
}

void onoise_candidate(const mjpcx_model* m, const mjpcx_noise_spec* ns, int num_nodes,
                      const double* nominal, int gi, double* out) {
  int nu = m->nu;
  int np = num_nodes * nu;
  for (int j = 0; j < np; j++) out[j] = nominal[j];
  if (gi == ns->nominal_candidate) return; /* sampling/planner.cc:374: `if (i != 0)` */
  double std = ns->std0;
  if (ns->mode == MJPCX_NOISE_SAMPLING && ns->std1 > 0) { /* planner.cc:334-338 */
    double u[2];
    uniforms(ns->seed, (uint32_t)gi, 0, ns->iteration, 1, u);
    if (u[0] < 0.2) std = ns->std1;
  }
  for (int j = 0; j < np; j++) {
    double z[2];
    ogaussian_pair(ns->seed, (uint32_t)gi, (uint32_t)(j / 2), ns->iteration, z);
    int k = j % nu;
    double sigma;
    if (ns->mode == MJPCX_NOISE_SAMPLING) {
      sigma = 0.5 * (m->actuator_ctrlrange[2 * k + 1] - m->actuator_ctrlrange[2 * k]) * std;
    } else { /* cross_entropy/planner.cc:367-370, 399-405 */
      double floor_ = gi < ns->explore_count ? ns->std0 : ns->std1;
      double s = sqrt(ns->param_variance[j]);
      sigma = s > floor_ ? s : floor_;
    }
    double v = out[j] + sigma * z[j & 1];
    double lo = m->actuator_ctrlrange[2 * k], hi = m->actuator_ctrlrange[2 * k + 1];
    out[j] = v < lo ? lo : (v > hi ? hi : v); /* Clamp, utilities.cc:112-116 */
  }
}
