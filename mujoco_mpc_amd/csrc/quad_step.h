// quad_step.h -- the step function of the QUAD kernel family: Trajectory::Rollout (mjpc/trajectory.cc:100-210) of ONE candidate
// by FOUR lanes, one per leg of a floating-base quadruped (quad_model.h). Written as a SIMT program: every variable is the
// calling lane's own; `leg` is the lane's index in its quad. The only cross-lane operations are the quad primitives
//     qd_sum(x)        sum over the four lanes of the quad, bit-identical in all four
//     qd_bcast<k>(x)   lane k's value
//     qd_or(i)         bitwise or over the quad
//     qd_rotv(x, d)    lane (l + d) mod 4's value, d at run time (quad-uniform)
// which the including translation unit provides (DPP quad permutes on gfx950: quad_kernel.h; a four-thread lock-step
// emulator on the CPU: tests/quademu, test infrastructure). Control flow around a primitive is quad-uniform by
// construction: every loop bound and branch condition on such a path derives from quad-summed or trunk (replicated) values.
//
// The physics is the oracle's (oracle/physics.c, contact.inc, quadruped.inc -- restating mj_step / mj_forward and
// QuadrupedFlat::ResidualFn::Residual), re-derived for the arrowhead structure of a legged tree:
//   * everything of the trunk (pose, cdof, 6 x 6 block T of M) is computed redundantly in the four lanes;
//   * a leg's links, its 3 x 3 block L of M, the 3 x 6 coupling B, its contacts, friction-loss and limit rows live in its lane;
//   * sums over the legs (centre of mass, composite inertia and bias force of the trunk, Schur complements, line-search
//     derivatives, costs) are quad sums.
// No Jacobian is formed: a contact is its frame F and offset from the centre of mass; J v = A V_b with V_b the spatial velocity
// of its body, J' f a spatial force, and J' D J = S' X S with X = A' (d2s) A a 6 x 6 (as csrc/wave_tree.h does per wavefront).
//
// A candidate the quad form does not cover at some step (a contact between two moving geoms, more than kQMaxCon contacts in
// one lane, an indefinite Hessian, a non-finite state, both limits of a joint) is FLAGGED and handed to the
// wavefront-per-candidate kernel, which rolls it out from the start (quad_kernel.h): results never depend on which kernel ran.
#pragma once
#include <stdint.h>

#include "quad_model.h"
#include "solid_pairs.h"

#ifndef QPROF
#define QPROF(pf, idx)
#define QPROF_COUNT(pf, idx, n)
#define QPROF_WAVE_HIST(pf, idx, v)
#define QWAVE_TIMES(a, iters, general, ncon)
#endif
#ifndef QCLASS_NOW
#define QCLASS_NOW(a) 0ll
#define QCLASS_ADD(a, base, have_rel, pmask, ncon, t0)
#endif
#ifndef QUNROLL
#define QUNROLL
#endif
#ifndef QNOUNROLL
#define QNOUNROLL   // (the device build keeps a loop so marked ROLLED: code size against the 64 KB instruction cache)
#endif
// An out-of-line function receives the model image and its caller's locals through plain references; the device build says where they
// live (LDS / the private segment) so that they are read with ds_read / scratch_load instead of FLAT loads (quad_kernel.h).
#ifndef QREBIND_LDS
#define QREBIND_LDS(T, ref) (ref)
#define QREBIND_PRIVATE(T, ptr) (ptr)
#endif
#ifndef QNOINLINE
#define QNOINLINE QD
#endif
#ifndef QD
#error "define QD (function qualifiers) and the quad primitives before including quad_step.h"
#endif

namespace mjpcx { namespace quad {

constexpr double kQMinVal = 1e-15, kQMaxVal = 1e10, kQPi = 3.14159265358979323846;
constexpr double kQLsTol = 0.01;

enum { kFlagOverflow = 1, kFlagPair = 2, kFlagNotPD = 4, kFlagBad = 8, kFlagLimits = 16, kFlagPairTrunk = 32, kFlagRange = 64 };

// ---------------------------------------------------------------- small algebra
// square root with its reciprocal, and a reciprocal, for the line search's inner loop: the includer may supply faster ones (the device
// build refines v_rsq_f64 / v_rcp_f64 by Newton steps: an ulp or two from the correctly rounded values, a third of the instructions)
#ifndef QFAST_MATH
QD void q_sqrt_rsqrt(double x, double& s, double& r) { s = sqrt(x); r = 1.0 / s; }
QD double q_rcp(double x) { return 1.0 / x; }
#endif
QD bool qbad(double x) { return !(x <= kQMaxVal && x >= -kQMaxVal); }
QD void q_mul(double* r, const double* a, const double* b) {
  const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
QD void q2mat(double* m, const double* q) {
  const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  const double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
QD void q_norm(double* q) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < kQMinVal) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { const double s = 1.0 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
QD void mv3(double* r, const double* m, const double* v) {
  const double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
               z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
QD void q_rot(double* r, const double* v, const double* q) { double m[9]; q2mat(m, q); mv3(r, m, v); }
QD void cr3(double* r, const double* a, const double* b) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
QD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
QD double dot6(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }
// rotated inertia R diag(I) R' (6 unique: xx yy zz xy xz yz) -- the part of inert_com that does not need the reference point
QD void rot_inertia(double* res, const double* inert, const double* mat) {
  double tmp[9];
  QUNROLL for (int c = 0; c < 3; c++) { tmp[c] = inert[0] * mat[3 * c]; tmp[3 + c] = inert[1] * mat[3 * c + 1]; tmp[6 + c] = inert[2] * mat[3 * c + 2]; }
  res[0] = mat[0] * tmp[0] + mat[1] * tmp[3] + mat[2] * tmp[6];
  res[1] = mat[3] * tmp[1] + mat[4] * tmp[4] + mat[5] * tmp[7];
  res[2] = mat[6] * tmp[2] + mat[7] * tmp[5] + mat[8] * tmp[8];
  res[3] = mat[0] * tmp[1] + mat[1] * tmp[4] + mat[2] * tmp[7];
  res[4] = mat[0] * tmp[2] + mat[1] * tmp[5] + mat[2] * tmp[8];
  res[5] = mat[3] * tmp[2] + mat[4] * tmp[5] + mat[5] * tmp[8];
}
// spatial inertia about the reference point: rotated inertia + parallel-axis terms of the offset `dif` (oracle inert_com)
QD void inert_shift(double* res, const double* irot, const double* dif, double mass) {
  res[0] = irot[0] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] = irot[1] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] = irot[2] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] = irot[3] - mass * dif[0] * dif[1];
  res[4] = irot[4] - mass * dif[0] * dif[2];
  res[5] = irot[5] - mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2];
  res[9] = mass;
}
QD void mul_inert(double* res, const double* i, const double* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
QD void cross_motion(double* res, const double* vel, const double* v) {
  double a[3], b[3], c[3];
  cr3(a, vel, v); cr3(b, vel, v + 3); cr3(c, vel + 3, v);
  res[0] = a[0]; res[1] = a[1]; res[2] = a[2];
  res[3] = b[0] + c[0]; res[4] = b[1] + c[1]; res[5] = b[2] + c[2];
}
QD void cross_force(double* res, const double* vel, const double* f) {
  double a[3], b[3], c[3];
  cr3(a, vel, f); cr3(b, vel + 3, f + 3); cr3(c, vel, f + 3);
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  res[3] = c[0]; res[4] = c[1]; res[5] = c[2];
}
QD double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// solimp (digested: quad_model.h) -> impedance at violation `dist` (oracle impedance())
QD double impedance(const double* d, double dist) {
  const double dmin = d[0], dmax = d[1], width = d[2], mid = d[3], power = d[4];
  if (dmin == dmax || width <= kQMinVal) return 0.5 * (dmin + dmax);
  const double x = fabs(dist) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  double y;
  if (power == 1) y = x;
  else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
  else y = x <= mid ? pow(x, power) / pow(mid, power - 1) : 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}
QD void make_frame(double* frame) {
  double* x = frame; double* y = frame + 3; double* z = frame + 6;
  if (x[1] < 0.5 && x[1] > -0.5) { y[0] = 0; y[1] = 1; y[2] = 0; }
  else { y[0] = 0; y[1] = 0; y[2] = 1; }
  const double dt = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
  QUNROLL for (int k = 0; k < 3; k++) y[k] -= dt * x[k];
  const double n = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
  QUNROLL for (int k = 0; k < 3; k++) y[k] /= n;
  cr3(z, x, y);
}
QD void qd_sum_n(double* v, int n) { for (int i = 0; i < n; i++) v[i] = qd_sum(v[i]); }

// packed lower triangle of a symmetric 6 x 6: (r >= c) at r (r + 1) / 2 + c
QD constexpr int tri(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }

// ---------------------------------------------------------------- arrowhead matrices
// SUPER-LEGS. A contact between two legs does not involve the trunk (its motion cancels in the relative velocity), so it adds to the
// Hessian the two legs' own blocks and ONE cross block between them: legs A < B paired by `pmode` (B = A xor pmode) form a 6-dof
// super-leg whose cross block ab (rows: A's dofs, columns: B's) lives in A's lane. arrow_factor(a, leg, pmode) eliminates A first
// (Y = Haa^-1 Hab replaces ab), hands B the Schur updates of its block and of its trunk coupling (24 numbers through qd_partner), and
// carries on as for independent legs; arrow_solve mirrors it (3 numbers each way). pmode = 0: no pairing (M, the integrator's matrix).
//
// Symmetric positive-definite matrix of the legged tree: per lane the leg block (3 x 3, packed lower triangle l[tri]), the coupling
// b (3 leg dofs x 6 trunk dofs); replicated in the four lanes the trunk block t (6 x 6, packed). arrow_factor turns it IN PLACE into
// its factor: the leg block as unit-lower L D L' (l[1] = l10, l[3] = l20, l[4] = l21, reciprocal pivots in l[0], l[2], l[5]),
// b := Z = Lblock^-1 B, t := the L D L' factor of the Schur complement S = T - sum_legs B' Z (unit-lower entries below the diagonal,
// RECIPROCAL pivots on it).
struct Arrow { double l[6], b[3][6], t[21]; double ab[3][3]; };

// unit-lower L D L' of a packed 3 x 3 block in place (l[1] = l10, l[3] = l20, l[4] = l21, reciprocal pivots in l[0], l[2], l[5])
QD bool leg_ldl(double* l) {
  bool ok = true;
  const double d0 = l[0];
  ok &= d0 > kQMinVal;
  const double i0 = 1.0 / d0, l10 = l[1] * i0, l20 = l[3] * i0;
  const double d1 = l[2] - l10 * l10 * d0;
  ok &= d1 > kQMinVal;
  const double i1 = 1.0 / d1, l21 = (l[4] - l20 * l10 * d0) * i1;
  const double d2 = l[5] - l20 * l20 * d0 - l21 * l21 * d1;
  ok &= d2 > kQMinVal;
  l[0] = i0; l[1] = l10; l[2] = i1; l[3] = l20; l[4] = l21; l[5] = 1.0 / d2;
  return ok;
}
QD void leg_solve_l(const double* l, double* x) {  // x := block^-1 x, block factored by leg_ldl
  x[1] -= l[1] * x[0];
  x[2] -= l[3] * x[0] + l[4] * x[1];
  x[0] *= l[0]; x[1] *= l[2]; x[2] *= l[5];
  x[1] -= l[4] * x[2];
  x[0] -= l[1] * x[1] + l[3] * x[2];
}
// returns false (quad-uniform) if a pivot is not positive
QD bool arrow_factor(Arrow& a, int leg, int pmode) {
  bool ok = true;
  if (pmode != 0) {  // (quad-uniform)
    // super-legs: the lower leg of each pair eliminates itself first; its partner receives the Schur updates
    const bool isA = leg < (leg ^ pmode);
    double lf[6];
    QUNROLL for (int i = 0; i < 6; i++) lf[i] = a.l[i];
    (void)leg_ldl(lf);  // (A's pivots are checked by the common pass below, which factors the unchanged block again)
    double G[6], W[3][6];
    {
      double Y[3][3];  // Haa^-1 Hab, column by column
      QUNROLL for (int c = 0; c < 3; c++) {
        double col[3] = {a.ab[0][c], a.ab[1][c], a.ab[2][c]};
        leg_solve_l(lf, col);
        Y[0][c] = col[0]; Y[1][c] = col[1]; Y[2][c] = col[2];
      }
      QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c <= r; c++) G[tri(r, c)] = a.ab[0][r] * Y[0][c] + a.ab[1][r] * Y[1][c] + a.ab[2][r] * Y[2][c];
      QUNROLL for (int k = 0; k < 6; k++) {
        double col[3] = {a.b[0][k], a.b[1][k], a.b[2][k]};
        leg_solve_l(lf, col);
        QUNROLL for (int r = 0; r < 3; r++) W[r][k] = a.ab[0][r] * col[0] + a.ab[1][r] * col[1] + a.ab[2][r] * col[2];
      }
      QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c < 3; c++) a.ab[r][c] = Y[r][c];
    }
    QUNROLL for (int i = 0; i < 6; i++) { const double g = qd_partner(G[i], pmode); if (!isA) a.l[i] -= g; }
    QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int k = 0; k < 6; k++) { const double w = qd_partner(W[r][k], pmode); if (!isA) a.b[r][k] -= w; }
  }
  ok &= leg_ldl(a.l);
  const double i0 = a.l[0], l10 = a.l[1], i1 = a.l[2], l20 = a.l[3], l21 = a.l[4], i2 = a.l[5];
  // Y = L^-1 B in place (forward substitution only); the Schur complement needs Y' D^-1 Y
  QUNROLL for (int k = 0; k < 6; k++) { a.b[1][k] -= l10 * a.b[0][k]; a.b[2][k] -= l20 * a.b[0][k] + l21 * a.b[1][k]; }
  QUNROLL for (int r = 0; r < 6; r++)
    QUNROLL for (int c = 0; c <= r; c++)
      a.t[tri(r, c)] -= qd_sum(a.b[0][r] * a.b[0][c] * i0 + a.b[1][r] * a.b[1][c] * i1 + a.b[2][r] * a.b[2][c] * i2);
  // Z = L^-T D^-1 Y in place
  QUNROLL for (int k = 0; k < 6; k++) {
    a.b[2][k] *= i2;
    a.b[1][k] = a.b[1][k] * i1 - l21 * a.b[2][k];
    a.b[0][k] = a.b[0][k] * i0 - l10 * a.b[1][k] - l20 * a.b[2][k];
  }
  // L D L' of S (replicated)
  double d[6];
  QUNROLL for (int j = 0; j < 6; j++) {
    double dj = a.t[tri(j, j)];
    QUNROLL for (int k = 0; k < j; k++) dj -= a.t[tri(j, k)] * a.t[tri(j, k)] * d[k];
    ok &= dj > kQMinVal;
    d[j] = dj;
    const double inv = 1.0 / dj;
    QUNROLL for (int i = j + 1; i < 6; i++) {
      double v = a.t[tri(i, j)];
      QUNROLL for (int k = 0; k < j; k++) v -= a.t[tri(i, k)] * a.t[tri(j, k)] * d[k];
      a.t[tri(i, j)] = v * inv;
    }
    a.t[tri(j, j)] = inv;
  }
  return qd_or(ok ? 0 : 1) == 0;
}
// x := A^-1 x  (xl: the lane's three leg entries, xt: the six trunk entries, replicated); f factored with the same leg / pmode
QD void arrow_solve(const Arrow& f, double* xl, double* xt, int leg, int pmode) {
  const bool isA = pmode != 0 && leg < (leg ^ pmode);
  double ya[3] = {0, 0, 0};
  if (pmode != 0) {
    // B's right-hand side loses Hba Haa^-1 g_A = Y' g_A
    double v[3];
    QUNROLL for (int c = 0; c < 3; c++) v[c] = f.ab[0][c] * xl[0] + f.ab[1][c] * xl[1] + f.ab[2][c] * xl[2];
    QUNROLL for (int c = 0; c < 3; c++) { const double w = qd_partner(v[c], pmode); if (!isA) xl[c] -= w; }
  }
  QUNROLL for (int k = 0; k < 6; k++) xt[k] -= qd_sum(f.b[0][k] * xl[0] + f.b[1][k] * xl[1] + f.b[2][k] * xl[2]);
  QUNROLL for (int i = 1; i < 6; i++) QUNROLL for (int k = 0; k < i; k++) xt[i] -= f.t[tri(i, k)] * xt[k];
  QUNROLL for (int i = 0; i < 6; i++) xt[i] *= f.t[tri(i, i)];
  QUNROLL for (int i = 4; i >= 0; i--) QUNROLL for (int k = i + 1; k < 6; k++) xt[i] -= f.t[tri(k, i)] * xt[k];
  leg_solve_l(f.l, xl);
  QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int k = 0; k < 6; k++) xl[j] -= f.b[j][k] * xt[k];
  if (pmode != 0) {
    // A: x_A -= Y x_B
    QUNROLL for (int c = 0; c < 3; c++) ya[c] = qd_partner(xl[c], pmode);
    if (isA) { QUNROLL for (int r = 0; r < 3; r++) xl[r] -= f.ab[r][0] * ya[0] + f.ab[r][1] * ya[1] + f.ab[r][2] * ya[2]; }
  }
}
// ---- self-collision, a leg in contact with TWO OR THREE others: leaf elimination.
// The legs' contact graph (an edge per pair of legs with a contact between them) of a tangled robot is nearly always a FOREST: a chain
// A-B-C or a star. A leaf -- a leg with one partner -- is eliminated into that partner exactly as the lower leg of a pair is (Y = Haa^-1 Hab
// replaces its cross block; the partner's block and trunk coupling receive the Schur updates), whether or not the partner has other partners,
// and no fill appears between legs; what remains after at most three such slots is the arrowhead of independent legs. Every step is the pair
// step at one xor value, so the state is an Arrow as for pairs -- until round 5 these candidates took a dense block elimination with fill
// (ArrowG: 81 + 96 doubles in flight, 434 spill reloads in the solver's loop) and, rare as they are (17 of 16384), set the time of the
// launch through the wavefronts that held them (57.9 -> 49.5 ms with them taken out). A graph with a CYCLE (three legs touching one another)
// is not eliminated here: the candidate is handed to the wavefront-per-candidate kernel (kFlagPair), as anything else the quad form does not
// cover -- a call to a dense elimination inside the solver's loop, however rarely taken, costs every step of every candidate (the values
// live across the call leave the registers: 145 -> 1103 spill reloads in the loop).
struct QPlan {
  int nslots, x[3];  // slot s eliminates the leaves whose edge is the xor value x[s]
  int eslot[4];      // the slot in which leg k is eliminated into leg k ^ x[eslot], or -1
  bool cyclic;
};
QD int q_popc3(int m) { return ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1); }
// mymask: bit x (1..3) set if the lane's leg touches leg ^ x. Computed identically in the four lanes from the four masks.
QD QPlan make_plan(int mymask, int leg) {
  int mk[4];
  QUNROLL for (int k = 0; k < 4; k++) mk[k] = qd_or(leg == k ? (mymask & 14) : 0);
  QPlan p;
  p.nslots = 0; p.cyclic = false;
  QUNROLL for (int k = 0; k < 4; k++) p.eslot[k] = -1;
  QUNROLL for (int s = 0; s < 3; s++) {
    p.x[s] = 0;
    if ((mk[0] | mk[1] | mk[2] | mk[3]) != 0 && !p.cyclic) {
      int x = 0;
      QUNROLL for (int k = 3; k >= 0; k--) if (q_popc3(mk[k]) == 1) x = __builtin_ctz(mk[k]);  // the lowest leaf's edge
      if (x == 0) p.cyclic = true;
      else {
        bool el[4];
        QUNROLL for (int k = 0; k < 4; k++) el[k] = q_popc3(mk[k]) == 1 && __builtin_ctz(mk[k]) == x;
        bool el2[4];  // (an isolated pair: both are leaves of each other -- the lower leg eliminates)
        QUNROLL for (int k = 0; k < 4; k++) el2[k] = el[k] && !(el[k ^ x] && (k ^ x) < k);
        p.x[s] = x;
        p.nslots = s + 1;
        QUNROLL for (int k = 0; k < 4; k++) if (el2[k]) p.eslot[k] = s;
        QUNROLL for (int k = 0; k < 4; k++) if (el2[k] || el2[k ^ x]) mk[k] &= ~(1 << x);
      }
    }
  }
  if ((mk[0] | mk[1] | mk[2] | mk[3]) != 0) p.cyclic = true;
  return p;
}
QD int plan_eslot(const QPlan& p, int k) { return k == 0 ? p.eslot[0] : (k == 1 ? p.eslot[1] : (k == 2 ? p.eslot[2] : p.eslot[3])); }
// the xor value of the lane's own elimination edge, 0 if it is never a leaf
QD int plan_my_x(const QPlan& p, int leg) {
  const int s = plan_eslot(p, leg);
  return s == 0 ? p.x[0] : (s == 1 ? p.x[1] : (s == 2 ? p.x[2] : 0));
}
// the leaves' eliminations, slot by slot (a.ab: the lane's cross block to the leg it is eliminated into -- rows: own dofs); then the
// arrowhead of independent legs. Returns false (quad-uniform) if a pivot is not positive.
QD bool arrow_factor_plan(Arrow& a, int leg, const QPlan& p) {
  const int my_slot = plan_eslot(p, leg);
  QUNROLL for (int s = 0; s < 3; s++) {
    if (s >= p.nslots) continue;  // (quad-uniform)
    const int x = p.x[s];
    const bool elim = my_slot == s, recv = plan_eslot(p, leg ^ x) == s;
    double lf[6];
    QUNROLL for (int i = 0; i < 6; i++) lf[i] = a.l[i];
    (void)leg_ldl(lf);  // (the pivots are checked by the common pass below, which factors the block again)
    double G[6], W[3][6], Y[3][3];
    QUNROLL for (int c = 0; c < 3; c++) {
      double col[3] = {a.ab[0][c], a.ab[1][c], a.ab[2][c]};
      leg_solve_l(lf, col);
      Y[0][c] = col[0]; Y[1][c] = col[1]; Y[2][c] = col[2];
    }
    QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c <= r; c++) G[tri(r, c)] = a.ab[0][r] * Y[0][c] + a.ab[1][r] * Y[1][c] + a.ab[2][r] * Y[2][c];
    QUNROLL for (int k = 0; k < 6; k++) {
      double col[3] = {a.b[0][k], a.b[1][k], a.b[2][k]};
      leg_solve_l(lf, col);
      QUNROLL for (int r = 0; r < 3; r++) W[r][k] = a.ab[0][r] * col[0] + a.ab[1][r] * col[1] + a.ab[2][r] * col[2];
    }
    if (elim) { QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c < 3; c++) a.ab[r][c] = Y[r][c]; }
    QUNROLL for (int i = 0; i < 6; i++) { const double g = qd_partner(G[i], x); if (recv) a.l[i] -= g; }
    QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int k = 0; k < 6; k++) { const double w = qd_partner(W[r][k], x); if (recv) a.b[r][k] -= w; }
  }
  return arrow_factor(a, leg, 0);
}
QD void arrow_solve_plan(const Arrow& f, double* xl, double* xt, int leg, const QPlan& p) {
  const int my_slot = plan_eslot(p, leg);
  QUNROLL for (int s = 0; s < 3; s++) {  // forward: the partner's right-hand side loses Y' g of the leaf
    if (s >= p.nslots) continue;
    const int x = p.x[s];
    const bool recv = plan_eslot(p, leg ^ x) == s;
    QUNROLL for (int c = 0; c < 3; c++) {
      const double w = qd_partner(f.ab[0][c] * xl[0] + f.ab[1][c] * xl[1] + f.ab[2][c] * xl[2], x);
      if (recv) xl[c] -= w;
    }
  }
  arrow_solve(f, xl, xt, leg, 0);
  QUNROLL for (int s = 2; s >= 0; s--) {  // backward: the leaf's solution loses Y x of its partner
    if (s >= p.nslots) continue;
    const int x = p.x[s];
    const bool elim = my_slot == s;
    double ya[3];
    QUNROLL for (int c = 0; c < 3; c++) ya[c] = qd_partner(xl[c], x);
    if (elim) { QUNROLL for (int r = 0; r < 3; r++) xl[r] -= f.ab[r][0] * ya[0] + f.ab[r][1] * ya[1] + f.ab[r][2] * ya[2]; }
  }
}

// y = A x; yt needs the quad sum of the coupling term
QD void arrow_mul(const Arrow& a, const double* xl, const double* xt, double* yl, double* yt) {
  QUNROLL for (int j = 0; j < 3; j++) {
    double s = 0;
    QUNROLL for (int i = 0; i < 3; i++) s += a.l[tri(j, i)] * xl[i];
    QUNROLL for (int k = 0; k < 6; k++) s += a.b[j][k] * xt[k];
    yl[j] = s;
  }
  QUNROLL for (int k = 0; k < 6; k++) {
    double s = qd_sum(a.b[0][k] * xl[0] + a.b[1][k] * xl[1] + a.b[2][k] * xl[2]);
    QUNROLL for (int i = 0; i < 6; i++) s += a.t[tri(k, i)] * xt[i];
    yt[k] = s;
  }
}
// x' y over all 18 dofs (trunk part counted once)
QD double arrow_dot(const double* xl, const double* xt, const double* yl, const double* yt) {
  const double s = qd_sum(xl[0] * yl[0] + xl[1] * yl[1] + xl[2] * yl[2]);
  return s + (xt[0] * yt[0] + xt[1] * yt[1] + xt[2] * yt[2] + xt[3] * yt[3] + xt[4] * yt[4] + xt[5] * yt[5]);
}

// ---------------------------------------------------------------- per-candidate data
struct QState {
  double tq[7], tv[6], lq[3], lv[3];  // trunk qpos (position, quaternion) / qvel, the leg's joint positions / velocities
  double wt[6], wl[3];                // previous step's qacc (solver warm start)
  double time;
};

// A contact in WORLD axes at the contact point. The rows of MuJoCo's contact frame (normal, two tangents; then torsion and two
// rolling axes) only ever enter the elliptic-cone penalty through the normal component and the LENGTH of the tangential / rolling
// parts -- both tangents share one friction coefficient, both rolling axes another -- so the tangent axes are never formed: the
// "row space" of a contact is the 6-vector [angular; linear] of relative velocity / acceleration at the point, and
//     jar = A V_body - aref,   A [w; v] = [w; v + w x off]   (off = point - centre of mass, V_body about the centre of mass).
// condim 1 / 3 / 4 are the same formulas with the friction coefficients of the missing rows set to zero.
struct QContact {
  double n[3], off[3];
  double D0;      // D of the normal row
  double jar[6];  // [angular; linear]; holds -aref from the contact's creation until the solver's first pass adds J qacc_smooth
  int depth;      // leg dofs on the body's chain (0: trunk)
  int fid;        // friction set (QuadModel::fric): regularised mu, tangential / torsional / rolling friction (0: row absent)
  // A contact between two MOVING geoms (self-collision): the trunk's motion cancels in the relative velocity, so its rows act on the
  // own leg's dofs below the body (depth) and, for a leg-leg contact, on the partner leg's (pd > 0; pd = 0: the other geom is the
  // trunk's). Both legs' lanes hold a copy of a leg-leg contact (bit-identical jar; each counts half its cost and applies its own
  // side of the force). sgn = +1 if the own body carries geom2 (J = jac(body2) - jac(body1)).
  int rel, sgn, pd;
  int px;  // leg-leg contact: own leg index xor the partner's (1..3); 0 otherwise
  // A contact between two bodies of the SAME leg (a calf or foot on the leg's own hip; self = 1, px = 0): pd is the depth of the shallower
  // body, the rows act on the leg's dofs pd <= j < depth alone, and no other lane holds a copy.
  int self;
};
// whether leg dof j is in the rows of the self-collision contact c
QD bool rel_dof(const QContact& c, int j) { return j < c.depth && !(c.self != 0 && j < c.pd); }
constexpr int kQConRec = 14;  // doubles per stored contact (quad_kernel.h / the emulator provide the store: qcs_load, qcs_store, qcs_store_jar)

// world poses of the static geoms, computed once per rollout (mocap bodies do not move during a rollout)
struct QStaticPose { double pos[3], mat[9]; };

// world pose of the lane's sphere | capsule geoms that take part in moving-geom pairs (self-collision test)
struct QPairGeoms { double c[kQPairGeom][3], a[kQPairGeom][3]; };

// the position-dependent part of a forward pass that the constraint solve needs
struct QKin {
  double cdof[3][6];             // the leg's three hinge dofs
  double ca[3][3], cl[3][3];     // the trunk's rotational dofs: angular = body axes, linear = axis x (com - trunk origin)
};

// spatial velocity [angular; linear about the centre of mass] of the bodies on the lane's chain for the dof vector (xl, xt):
// Vp[0] trunk, Vp[d] = + leg dofs < d
QD void chain_velocity(const QKin& k, const double* xl, const double* xt, double Vp[4][6]) {
  QUNROLL for (int c = 0; c < 3; c++) {
    Vp[0][c] = k.ca[0][c] * xt[3] + k.ca[1][c] * xt[4] + k.ca[2][c] * xt[5];
    Vp[0][3 + c] = xt[c] + k.cl[0][c] * xt[3] + k.cl[1][c] * xt[4] + k.cl[2][c] * xt[5];
  }
  QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int c = 0; c < 6; c++) Vp[j + 1][c] = Vp[j][c] + k.cdof[j][c] * xl[j];
}
// the contact's point-space 6-vector of body velocity V: [w; v + w x off]. The body's velocity is picked arithmetically (the last link's
// minus the dofs below the body, 0 / 1 weights): selecting among the four chain velocities element by element compiles to a cascade of
// exec-mask branches that costs more than the contact's arithmetic.
QD void point_vel(const QContact& c, const double Vp[4][6], double* out) {
  const double w2 = c.depth < 3 ? 1.0 : 0.0, w1 = c.depth < 2 ? 1.0 : 0.0, w0 = c.depth < 1 ? 1.0 : 0.0;
  double V[6];
  QUNROLL for (int k = 0; k < 6; k++) V[k] = Vp[3][k] - w2 * (Vp[3][k] - Vp[2][k]) - w1 * (Vp[2][k] - Vp[1][k]) - w0 * (Vp[1][k] - Vp[0][k]);
  double w[3];
  cr3(w, V, c.off);
  QUNROLL for (int k = 0; k < 3; k++) { out[k] = V[k]; out[3 + k] = V[3 + k] + w[k]; }
}
// what a lane's pass over its rows needs of the PARTNER leg for self-collision contacts: the partner's chain velocities relative to
// the trunk, dq[d - 1] = Vp[d] - Vp[0] of the partner's dof vector (exchanged once per pass through qd_partner)
struct QRel { double dq[3][6]; };
// A lane's pass over its contacts runs once per xor value x present in pmask (bits 1..3), exchanging with the partner A xor x each time
// and taking the leg-leg contacts of that partner; everything else belongs to the first pass, which always runs. next_x: the smallest
// value above `after`, 4 when there is none.
QD int next_x(int pmask, int after) {
  const int mk = pmask & ~((2 << after) - 1) & 14;
  return mk ? __builtin_ctz(mk) : 4;
}
QD bool in_pass(const QContact& c, int pass, int x) { return (c.rel && c.px != 0) ? c.px == x : pass == 0; }
QD void rel_exchange(const double Vp[4][6], int pmode, QRel& q) {
  QUNROLL for (int d = 0; d < 3; d++) QUNROLL for (int k = 0; k < 6; k++) q.dq[d][k] = qd_partner(Vp[d + 1][k] - Vp[0][k], pmode);
}
// the point-space relative velocity of a self-collision contact: sgn ((V_own_body - V_trunk) - (V_partner_body - V_trunk))
QD void point_vel_rel(const QContact& c, const double Vp[4][6], const QRel& q, double* out) {
  const double w2 = c.depth < 3 ? 1.0 : 0.0, w1 = c.depth < 2 ? 1.0 : 0.0;
  const double ws = c.self != 0 ? 1.0 : 0.0, wp = 1.0 - ws;  // (the other body: on the own chain | the partner's)
  const double u3 = c.pd == 3 ? wp : 0.0, u2 = c.pd == 2 ? wp : 0.0, u1 = c.pd == 1 ? wp : 0.0, sg = c.sgn;
  const double s2 = c.pd == 2 ? ws : 0.0, s1 = c.pd == 1 ? ws : 0.0;
  double V[6];
  QUNROLL for (int k = 0; k < 6; k++) {
    const double own = (Vp[3][k] - Vp[0][k]) - w2 * (Vp[3][k] - Vp[2][k]) - w1 * (Vp[2][k] - Vp[1][k]);  // (depth >= 1 for a moving geom of a leg)
    double other = s2 * (Vp[2][k] - Vp[0][k]) + s1 * (Vp[1][k] - Vp[0][k]);
    if (c.self == 0) other = u3 * q.dq[2][k] + u2 * q.dq[1][k] + u1 * q.dq[0][k];
    V[k] = sg * (own - other);
  }
  double w[3];
  cr3(w, V, c.off);
  QUNROLL for (int k = 0; k < 3; k++) { out[k] = V[k]; out[3 + k] = V[3 + k] + w[k]; }
}
// trunk dof k of the spatial force Fs: cdofT_k . Fs
QD double trunk_dot(const QKin& k, int dof, const double* Fs) {
  if (dof < 3) return Fs[3 + dof];
  return dot3(k.ca[dof - 3], Fs) + dot3(k.cl[dof - 3], Fs + 3);
}
// X (packed symmetric 6 x 6 over [angular; linear]) += w v v'
QD void add_outer(double* X, double w, const double* v) {
  QUNROLL for (int p = 0; p < 6; p++) { const double wp = w * v[p]; QUNROLL for (int q = 0; q <= p; q++) X[tri(p, q)] += wp * v[q]; }
}
// X += w Qt, Qt = A' blkdiag(f3^2 nn' + f4^2 P, f1^2 P) A about the centre of mass (P = I - nn', g = off x n):
//   ll = f1^2 P;  al = f1^2 ([off]x - g n');  aa = f3^2 nn' + f4^2 P + f1^2 (|off|^2 I - off off' - g g')
QD void add_Qt(double* X, double w, const QContact& c, const double* fr, const double* g) {
  const double f1s = w * fr[1] * fr[1], f3s = w * fr[2] * fr[2], f4s = w * fr[3] * fr[3];
  const double* n = c.n; const double* o = c.off;
  const double o2 = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
  QUNROLL for (int p = 0; p < 3; p++)
    QUNROLL for (int q = 0; q <= p; q++) {
      const double nn = n[p] * n[q], id = p == q ? 1.0 : 0.0;
      X[tri(p, q)] += f3s * nn + f4s * (id - nn) + f1s * (o2 * id - o[p] * o[q] - g[p] * g[q]);
      X[tri(3 + p, 3 + q)] += f1s * (id - nn);
    }
  // rows linear (3 + i), columns angular (j): al[j][i] = f1^2 ([off]x[j][i] - g[j] n[i]);  [off]x = [[0,-o2,o1],[o2,0,-o0],[-o1,o0,0]]
  const double ox[3][3] = {{0, -o[2], o[1]}, {o[2], 0, -o[0]}, {-o[1], o[0], 0}};
  QUNROLL for (int i = 0; i < 3; i++) QUNROLL for (int j = 0; j < 3; j++) X[tri(3 + i, j)] += f1s * (ox[j][i] - g[j] * n[i]);
}
// Penalty of one contact at c.jar: returns the cost, adds the spatial force about the centre of mass (J' force for the chain) to Fs
// and, if X, the Hessian block A' (d2s / djar2) A to X. zone: 0 top (nothing), 1 middle, 2 bottom (oracle constraint_cost /
// constraint_hessian, in point space).
QD double contact_eval(const QContact& c, const double* fr, double* Fs, double* X, int& zone) {
  const double* n = c.n;
  const double jn = dot3(n, c.jar + 3), an = dot3(n, c.jar);
  double tl[3], ar[3];
  QUNROLL for (int k = 0; k < 3; k++) { tl[k] = c.jar[3 + k] - jn * n[k]; ar[k] = c.jar[k] - an * n[k]; }
  const double mu = fr[0], f1s = fr[1] * fr[1], f3s = fr[2] * fr[2], f4s = fr[3] * fr[3];
  const double T2 = f1s * dot3(tl, tl) + f3s * an * an + f4s * dot3(ar, ar), N = mu * jn;
  double T = 0, iTq = 0;  // (below 1e-200 the cone's axis: the zones are told apart by the sign of N alone, as in line_eval)
  if (T2 > 1e-200) q_sqrt_rsqrt(T2, T, iTq);
  if (N >= mu * T || (T <= 0 && N >= 0)) { zone = 0; return 0; }
  double ra[3], rl[3], g[3], Fl[3], Fa[3], cost;
  QUNROLL for (int k = 0; k < 3; k++) { ra[k] = f3s * an * n[k] + f4s * ar[k]; rl[k] = f1s * tl[k]; }
  cr3(g, c.off, n);
  const double et[6] = {g[0], g[1], g[2], n[0], n[1], n[2]};  // the normal row about the centre of mass
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    const double Dq = c.D0 * fr[4];
    cost = 0.5 * c.D0 * jn * jn + 0.5 * Dq * T2;
    QUNROLL for (int k = 0; k < 3; k++) { Fl[k] = -c.D0 * jn * n[k] - Dq * rl[k]; Fa[k] = -Dq * ra[k]; }
    if (X) { add_outer(X, c.D0, et); add_Qt(X, Dq, c, fr, g); }
    zone = 2;
  } else {
    const double Dm = c.D0 * fr[5], NT = N - mu * T, s = Dm * NT * mu, iT = iTq, sT = s * iT;
    cost = 0.5 * Dm * NT * NT;
    QUNROLL for (int k = 0; k < 3; k++) { Fl[k] = -s * n[k] + sT * rl[k]; Fa[k] = sT * ra[k]; }
    if (X) {
      // X = Dm a a' - (s / T) (Qt - r r'),  r = A' (Q jar) / T,  a = mu (et - r)
      double rt[6], at[6], w[3];
      cr3(w, c.off, rl);
      QUNROLL for (int k = 0; k < 3; k++) { rt[k] = (ra[k] + w[k]) * iT; rt[3 + k] = rl[k] * iT; }
      QUNROLL for (int k = 0; k < 6; k++) at[k] = mu * (et[k] - rt[k]);
      add_outer(X, Dm, at);
      add_Qt(X, -sT, c, fr, g);
      add_outer(X, sT, rt);
    }
    zone = 1;
  }
  double w[3];
  cr3(w, c.off, Fl);
  QUNROLL for (int k = 0; k < 3; k++) { Fs[k] += w[k] + Fa[k]; Fs[3 + k] += Fl[k]; }
  return cost;
}
// The contact's Hessian block X = A' (d2s / djar2) A APPLIED to a spatial vector, without forming its 21 entries: what the rare passes over
// single contacts need (self-collision, shallow contacts) -- a dense block per contact there lives in memory, not in registers.
//   bottom zone:  X v = D0 et (et . v) + Dq Qt v
//   middle zone:  X v = Dm at (at . v) - sT Qt v + sT rt (rt . v)
// with Qt v = A' Qd A v evaluated through the point: u = [w; vl + w x off], Qd u = [f3^2 n (n . w) + f4^2 (w - n (n . w)); f1^2 (ul - n (n . ul))],
// A' [qa; ql] = [qa + off x ql; ql].
struct QHessOp { int zone; double et[6], rt[6], at[6], ca, cq, cr, f1s, f3s, f4s; };
QD void contact_hess_prepare(const QContact& c, const double* fr, QHessOp& h) {
  const double* n = c.n;
  const double jn = dot3(n, c.jar + 3), an = dot3(n, c.jar);
  double tl[3], ar[3];
  QUNROLL for (int k = 0; k < 3; k++) { tl[k] = c.jar[3 + k] - jn * n[k]; ar[k] = c.jar[k] - an * n[k]; }
  const double mu = fr[0];
  h.f1s = fr[1] * fr[1]; h.f3s = fr[2] * fr[2]; h.f4s = fr[3] * fr[3];
  const double T2 = h.f1s * dot3(tl, tl) + h.f3s * an * an + h.f4s * dot3(ar, ar), N = mu * jn;
  double T = 0, iT = 0;
  if (T2 > 1e-200) q_sqrt_rsqrt(T2, T, iT);
  QUNROLL for (int k = 0; k < 6; k++) { h.et[k] = 0; h.rt[k] = 0; h.at[k] = 0; }
  h.ca = 0; h.cq = 0; h.cr = 0;
  if (N >= mu * T || (T <= 0 && N >= 0)) { h.zone = 0; return; }
  double g[3];
  cr3(g, c.off, n);
  QUNROLL for (int k = 0; k < 3; k++) { h.et[k] = g[k]; h.et[3 + k] = n[k]; }
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    h.zone = 2; h.ca = c.D0; h.cq = c.D0 * fr[4];
    QUNROLL for (int k = 0; k < 6; k++) h.at[k] = h.et[k];
    return;
  }
  const double Dm = c.D0 * fr[5], NT = N - mu * T, sT = Dm * NT * mu * iT;
  double ra[3], rl[3], w[3];
  QUNROLL for (int k = 0; k < 3; k++) { ra[k] = h.f3s * an * n[k] + h.f4s * ar[k]; rl[k] = h.f1s * tl[k]; }
  cr3(w, c.off, rl);
  QUNROLL for (int k = 0; k < 3; k++) { h.rt[k] = (ra[k] + w[k]) * iT; h.rt[3 + k] = rl[k] * iT; }
  QUNROLL for (int k = 0; k < 6; k++) h.at[k] = mu * (h.et[k] - h.rt[k]);
  h.zone = 1; h.ca = Dm; h.cq = -sT; h.cr = sT;
}
QD void contact_hess_apply(const QContact& c, const QHessOp& h, const double* v, double* Y) {
  const double* n = c.n;
  // Qt v through the point
  double wxo[3], ul[3];
  cr3(wxo, v, c.off);
  QUNROLL for (int k = 0; k < 3; k++) ul[k] = v[3 + k] + wxo[k];
  const double nw = dot3(n, v), nu = dot3(n, ul);
  double qa[3], ql[3], oxq[3];
  QUNROLL for (int k = 0; k < 3; k++) { qa[k] = h.f3s * n[k] * nw + h.f4s * (v[k] - n[k] * nw); ql[k] = h.f1s * (ul[k] - n[k] * nu); }
  cr3(oxq, c.off, ql);
  const double av = h.ca * dot6(h.at, v), rv = h.cr * dot6(h.rt, v);
  QUNROLL for (int k = 0; k < 3; k++) {
    Y[k] = av * h.at[k] + h.cq * (qa[k] + oxq[k]) + rv * h.rt[k];
    Y[3 + k] = av * h.at[3 + k] + h.cq * ql[k] + rv * h.rt[3 + k];
  }
}
// First and second derivative of a contact's penalty along a fixed search direction (oracle constraint_line, in point space): alpha enters a contact's penalty only through
// jn = jn0 + alpha vn and the quadratic T^2 = A + 2 B alpha + C alpha^2 (B = UV at 0, C = VV, which does not depend on alpha), so a
// trial costs a dozen flops per contact instead of a pass over its record. D0 carries the half weight of a leg-leg contact.
#ifndef QEXP_LINE_SLOTS
#define QEXP_LINE_SLOTS 4
#endif
constexpr int kQLineSlots = QEXP_LINE_SLOTS;
#ifndef QREC
#define QREC(dst, v) (dst) = (v)   // a trajectory-buffer store (the device build streams them past the caches: quad_kernel.h)
#endif
#ifndef QUNIFORM_TIME
#define QUNIFORM_TIME(t) (t)       // the rollout's time is the same in every lane (the device build says so to the compiler)
#endif
#ifndef QGENERAL_FROM
#define QGENERAL_FROM 2  // (tests build the emulator with 1: single pairs through the dense elimination as well)
#endif
constexpr int kQGeneralFrom = QGENERAL_FROM;
// (what a trial needs of a contact, products of the direction-only factors taken once per line search: D0vn = D0 vn, muvn = mu vn,
// hb = D0 vn^2 + Dq C, the bottom zone's second derivative, which does not depend on alpha)
struct QLine { double jn0, vn, A, B, C, mu, Dq, Dm, D0vn, muvn, hb; };
QD void line_empty(QLine& q) { q.jn0 = 1; q.vn = 0; q.A = 0; q.B = 0; q.C = 0; q.mu = 1; q.Dq = 0; q.Dm = 0; q.D0vn = 0; q.muvn = 0; q.hb = 0; }  // (top zone at every alpha)
QD void line_coeffs(const QContact& c, const double* fr, const double* jv, double w, QLine& q) {
  const double* n = c.n;
  const double jn = dot3(n, c.jar + 3), an = dot3(n, c.jar), vn = dot3(n, jv + 3), wn = dot3(n, jv);
  double tl[3], ar[3];
  QUNROLL for (int k = 0; k < 3; k++) { tl[k] = c.jar[3 + k] - jn * n[k]; ar[k] = c.jar[k] - an * n[k]; }
  const double f1s = fr[1] * fr[1], f3s = fr[2] * fr[2], f4s = fr[3] * fr[3];
  q.jn0 = jn; q.vn = vn;
  q.A = f1s * dot3(tl, tl) + f3s * an * an + f4s * dot3(ar, ar);
  q.B = f1s * dot3(tl, jv + 3) + f3s * an * wn + f4s * dot3(ar, jv);
  q.C = f1s * (dot3(jv + 3, jv + 3) - vn * vn) + f3s * wn * wn + f4s * (dot3(jv, jv) - wn * wn);
  const double D0 = w * c.D0;
  q.mu = fr[0]; q.Dq = D0 * fr[4]; q.Dm = D0 * fr[5];
  q.D0vn = D0 * vn; q.muvn = q.mu * vn; q.hb = q.D0vn * vn + q.Dq * q.C;
}
// One trial of one slot: about 35 fp64 operations and a dozen selects (this loop is two thirds of the kernel's instructions: a wavefront
// runs ~12 trials in each of ~10 Newton iterations per step). No branches: the slots of a lane are independent chains the scheduler can
// interleave, and the three zones differ by a handful of flops. Where T^2 is not positive (the cone's axis) the root and its reciprocal
// are garbage and are replaced by zeros, which makes the zone tests those of the sign of N alone.
QD void line_eval(const QLine& q, double alpha, double& g, double& h) {
  const double jn = fma(alpha, q.vn, q.jn0), UV = fma(alpha, q.C, q.B), T2 = fma(alpha, q.B + UV, q.A);  // T^2 = A + 2 B alpha + C alpha^2
  const bool pos = T2 > 1e-200;
  double Ts, iTs;
  q_sqrt_rsqrt(T2, Ts, iTs);
  const double T = pos ? Ts : 0.0, iT = pos ? iTs : 0.0, N = q.mu * jn;
  const bool top = N >= q.mu * T, bottom = fma(q.mu, N, T) <= 0;
  const double gb = fma(q.D0vn, jn, q.Dq * UV);
  const double NT = fma(-q.mu, T, N), u = UV * iT, dNT = fma(-q.mu, u, q.muvn);
  const double d2NT = -(q.mu * iT) * fma(-u, u, q.C);  // -mu (C / T - UV^2 / T^3)
  const double gm = q.Dm * NT * dNT, hm = q.Dm * fma(NT, d2NT, dNT * dNT);
  g += top ? 0.0 : (bottom ? gb : gm);
  h += top ? 0.0 : (bottom ? q.hb : hm);
}

// ---------------------------------------------------------------- constraint solve (oracle o_constraint_newton)
struct QRows {
  // friction loss (one row per leg dof with frictionloss > 0) and the active joint limit of each joint (side 0: none); the jar entries
  // hold -aref until the solver's first pass
  double fl_jar[3];
  double lm_D[3], lm_jar[3];
  int lm_side[3];
};
enum { kEvalKeep = 0, kEvalStep = 2 };
// One pass over the lane's rows. what = kEvalStep: jar += alpha J x for the dof vector x (xl: its leg part, Vp: chain_velocity of x);
// kEvalKeep: jar as it is. (The rows are created holding -aref, so the solver's first pass -- a step of 1 along qacc_smooth -- makes
// jar = J qacc_smooth - aref, and moving to the warm start is a step along their difference: aref itself is never stored.) Then the penalty at jar: returns the cost of ALL rows of the candidate (quad sum), J' force in jl (lane's
// leg dofs) / jt (trunk dofs, replicated), and in X the sum of the lane's contacts' Hessian blocks (packed 6 x 6; NOT yet quad-summed).
// nshallow counts the lane's contacts in a penalty zone whose body is not the last link (their blocks need the correction of
// hessian_blocks).
template <bool MULTI, class CS>
QD double rows_eval(const QuadModel& m, const QuadLeg& L, const QKin& kin, QRows& R, CS& cs, int ncon, int what, const double* xl, const double Vp[4][6], double alpha,
                    double* jl, double* jt, int pmask) {
  double cost = 0;
  double Fown[6];
  QUNROLL for (int c = 0; c < 6; c++) Fown[c] = 0;
  QUNROLL for (int j = 0; j < 3; j++) {
    if (what == kEvalStep) { R.fl_jar[j] += alpha * xl[j]; R.lm_jar[j] += alpha * (-R.lm_side[j] * xl[j]); }
    double f = 0;
    if (L.floss[j] > 0) {
      const double x = R.fl_jar[j], fl = L.floss[j], Rr = L.floss_R[j];
      if (x <= -Rr * fl) { cost += -0.5 * Rr * fl * fl - fl * x; f += fl; }
      else if (x >= Rr * fl) { cost += -0.5 * Rr * fl * fl + fl * x; f += -fl; }
      else { cost += 0.5 * L.floss_D[j] * x * x; f += -L.floss_D[j] * x; }
    }
    if (R.lm_side[j] != 0 && R.lm_jar[j] < 0) {
      cost += 0.5 * R.lm_D[j] * R.lm_jar[j] * R.lm_jar[j];
      f += -R.lm_side[j] * (-R.lm_D[j] * R.lm_jar[j]);  // J' force, J = -side
    }
    jl[j] = f;
  }
  QRel rq;
  int x = next_x(pmask, 0);
  for (int pass = 0; pass == 0 || (MULTI && x < 4); pass++) {  // (quad-uniform; one pass unless MULTI)
    if (x < 4 && what != kEvalKeep) rel_exchange(Vp, x, rq);
    for (int i = 0; i < ncon; i++) {
      QContact c;
      qcs_load(cs, i, c);
      if (MULTI && !in_pass(c, pass, x)) continue;
      if (what != kEvalKeep) {
        double pv[6];
        if (c.rel) point_vel_rel(c, Vp, rq, pv); else point_vel(c, Vp, pv);
        QUNROLL for (int k = 0; k < 6; k++) c.jar[k] += alpha * pv[k];
        qcs_store_jar(cs, i, c);
      }
      double Fs[6] = {0, 0, 0, 0, 0, 0};
      int zone;
      // (ONE instance of the penalty for both kinds of contact: lanes of a wavefront that hold different kinds at the same index would
      // otherwise run two copies of it one after the other)
      const double cc = contact_eval(c, m.fric[c.fid], Fs, nullptr, zone);
      cost += (c.rel && c.px != 0) ? 0.5 * cc : cc;  // (a leg-leg contact is counted half here, half in its partner's lane)
      if (zone == 0) continue;
      if (c.rel) {  // self-collision: own leg's dofs only, the own side of the force
        QUNROLL for (int j = 0; j < 3; j++) if (rel_dof(c, j)) jl[j] += c.sgn * dot6(kin.cdof[j], Fs);
        continue;
      }
      QUNROLL for (int k = 0; k < 6; k++) Fown[k] += Fs[k];
      if (c.depth < 3) {  // rare: a contact on the trunk (this lane's share), the hip or the thigh link does not act on the dofs below it
        QUNROLL for (int j = 0; j < 3; j++) if (j >= c.depth) jl[j] -= dot6(kin.cdof[j], Fs);
      }
    }
    if (MULTI) x = next_x(pmask, x);
  }
  QUNROLL for (int j = 0; j < 3; j++) jl[j] += dot6(kin.cdof[j], Fown);
  qd_sum_n(Fown, 6);
  QUNROLL for (int k = 0; k < 6; k++) jt[k] = trunk_dot(kin, k, Fown);
  return qd_sum(cost);
}
// The sum of the lane's contacts' Hessian blocks at the current jar (packed 6 x 6 over [angular; linear]; NOT yet quad-summed), for the
// iteration that is about to factor: a pass of its own, so that the cost / force passes (two per step more than there are
// factorisations, and the line search between them) do not carry 21 accumulators. nshallow counts the lane's contacts in a penalty
// zone whose body is not the last link (their blocks need the correction of hessian_common).
template <class CS>
QD void rows_X(const QuadModel& m, CS& cs, int nstat, double* X, int& nshallow) {
  QUNROLL for (int e = 0; e < 21; e++) X[e] = 0;
  nshallow = 0;
  for (int i = 0; i < nstat; i++) {  // (the contacts with static geoms: the self-collision contacts behind them are hessian_rel's)
    QContact c;
    qcs_load(cs, i, c);
    double Fs[6] = {0, 0, 0, 0, 0, 0};
    int zone;
    (void)contact_eval(c, m.fric[c.fid], Fs, X, zone);
    if (zone != 0 && c.depth < 3) nshallow++;
  }
}
// jar += alpha J x and nothing else (the way back from a rejected warm start)
template <bool MULTI, class CS>
QD void rows_step_only(const QuadLeg& L, QRows& R, CS& cs, int ncon, const double* xl, const double Vp[4][6], double alpha, int pmask) {
  QUNROLL for (int j = 0; j < 3; j++) { R.fl_jar[j] += alpha * xl[j]; R.lm_jar[j] += alpha * (-R.lm_side[j] * xl[j]); }
  QRel rq;
  int x = next_x(pmask, 0);
  for (int pass = 0; pass == 0 || (MULTI && x < 4); pass++) {
    if (x < 4) rel_exchange(Vp, x, rq);
    for (int i = 0; i < ncon; i++) {
      QContact c;
      qcs_load(cs, i, c);
      if (MULTI && !in_pass(c, pass, x)) continue;
      double pv[6];
      if (c.rel) point_vel_rel(c, Vp, rq, pv); else point_vel(c, Vp, pv);
      QUNROLL for (int k = 0; k < 6; k++) c.jar[k] += alpha * pv[k];
      qcs_store_jar(cs, i, c);
    }
    if (MULTI) x = next_x(pmask, x);
  }
}
// derivatives of the row penalties along the search direction at step alpha (quad sums)
// (xl: the search direction's leg part, Vp: its chain_velocity; J search is recomputed per row: 9 flops per contact)
// qx (BEYOND): the coefficients of the lane's contacts beyond the slots, in an array that lives in memory (written once per line search,
// read by every trial: the trial loop touches no other memory, so these reads stay in the vector L1) -- until round 5 a trial walked those
// contacts' RECORDS (point velocity, square root, division per contact and trial): wavefront-steps with such a lane, a fifth of the bench's
// gait, took 1.54 M cycles against 1.11 M
template <bool MULTI, bool BEYOND, class CS>
QD void rows_line_prepare(const QuadModel& m, CS& cs, int ncon, const double Vp[4][6], int pmask, bool beyond_slots, QLine* ql, QLine* qx) {
  QUNROLL for (int i = 0; i < kQLineSlots; i++) line_empty(ql[i]);
  int x = next_x(pmask, 0);
  for (int pass = 0; pass == 0 || (MULTI && x < 4); pass++) {
    QRel rx;
    if (x < 4) rel_exchange(Vp, x, rx);
    QUNROLL for (int i = 0; i < kQLineSlots; i++) {
      if (i >= ncon) continue;
      QContact c;
      qcs_load(cs, i, c);
      if (MULTI && !in_pass(c, pass, x)) continue;
      double jv[6];
      if (c.rel) point_vel_rel(c, Vp, rx, jv); else point_vel(c, Vp, jv);
      line_coeffs(c, m.fric[c.fid], jv, c.rel && c.px != 0 ? 0.5 : 1.0, ql[i]);
    }
    if (BEYOND && beyond_slots) {  // (quad-uniform)
      for (int i = kQLineSlots; i < ncon; i++) {
        QContact c;
        qcs_load(cs, i, c);
        if (MULTI && !in_pass(c, pass, x)) continue;
        double jv[6];
        if (c.rel) point_vel_rel(c, Vp, rx, jv); else point_vel(c, Vp, jv);
        QLine t;
        line_coeffs(c, m.fric[c.fid], jv, c.rel && c.px != 0 ? 0.5 : 1.0, t);
        qx[i - kQLineSlots] = t;
      }
    }
    if (MULTI) x = next_x(pmask, x);
  }
}
// the lane's diagonal rows along the search direction, prepared once per line search: friction loss (x0 = jar, jv = the dof's component of
// the direction; Rfl = R fl, fljv = fl jv, Djv = D jv, Djv2 = D jv^2; a dof without friction loss: Rfl = inf, the rest 0) and the active
// joint limit (lx0, ljv = -side component; lDjv, lDjv2 zero when no limit is active)
struct QDiag { double x0[3], jv[3], Rfl[3], fljv[3], Djv[3], Djv2[3], lx0[3], ljv[3], lDjv[3], lDjv2[3]; };
QD void diag_prepare(const QuadLeg& L, const QRows& R, const double* xl, QDiag& dg) {
  QUNROLL for (int j = 0; j < 3; j++) {
    const double fl = L.floss[j], jv = xl[j];
    const bool on = fl > 0;
    dg.x0[j] = R.fl_jar[j]; dg.jv[j] = jv;
    dg.Rfl[j] = on ? L.floss_R[j] * fl : 1e300;
    dg.fljv[j] = on ? fl * jv : 0.0;
    dg.Djv[j] = on ? L.floss_D[j] * jv : 0.0;
    dg.Djv2[j] = on ? L.floss_D[j] * jv * jv : 0.0;
    const double lj = -R.lm_side[j] * xl[j];
    dg.lx0[j] = R.lm_jar[j]; dg.ljv[j] = lj;
    dg.lDjv[j] = R.lm_side[j] != 0 ? R.lm_D[j] * lj : 0.0;
    dg.lDjv2[j] = R.lm_side[j] != 0 ? R.lm_D[j] * lj * lj : 0.0;
  }
}
template <bool BEYOND>
QD void rows_line(const QDiag& dg, bool any_limit, int ncon, double alpha, bool beyond_slots, int nslot_wave, const QLine* ql, const QLine* qx, double& d1, double& d2) {
  double g = 0, h = 0;
  QUNROLL for (int j = 0; j < 3; j++) {
    const double x = fma(alpha, dg.jv[j], dg.x0[j]);
    const bool low = x <= -dg.Rfl[j], high = x >= dg.Rfl[j];
    g += low ? -dg.fljv[j] : (high ? dg.fljv[j] : dg.Djv[j] * x);
    h += (low || high) ? 0.0 : dg.Djv2[j];
  }
  if (any_limit) {  // (wavefront-uniform: joint limits are rarely active)
    QUNROLL for (int j = 0; j < 3; j++) {
      const double x = fma(alpha, dg.ljv[j], dg.lx0[j]);
      const bool on = x < 0;
      g += on ? dg.lDjv[j] * x : 0.0; h += on ? dg.lDjv2[j] : 0.0;
    }
  }
  // (slots no lane of the wavefront fills are skipped by a scalar branch: an empty slot adds exactly zero)
  QUNROLL for (int i = 0; i < kQLineSlots; i++) if (i < nslot_wave) line_eval(ql[i], alpha, g, h);
  if (BEYOND && beyond_slots) {  // (quad-uniform: a lane of the quad holds more contacts than slots)
    for (int i = kQLineSlots; i < ncon; i++) { const QLine t = qx[i - kQLineSlots]; line_eval(t, alpha, g, h); }
  }
  d1 = qd_sum(g); d2 = qd_sum(h);
}
// The exact line search of one Newton iteration (oracle constraint_newton's inner loop: Newton on the derivative, bracketed, with the
// rtsafe safeguard). Out of line on the device, like the solver itself: the loop's working set (the contacts' coefficients, the diagonal
// rows) then has the register file to itself instead of competing with everything the iteration keeps alive around it.
// BEYOND = false: no lane of the wavefront holds more contacts than slots (the common case) -- the trial loop then lives on the slots'
// coefficients and the diagonal rows alone; the records, the chain velocities and the partner exchange are dead after the preparation.
// (inlined into the Newton iteration since the split: the common variant is small, and the call -- arguments through the stack, the
// chain velocities re-read from memory -- cost 7 % of the launch)
#if defined(QEXP_LS_NOINLINE)
#define QLS_ATTR QNOINLINE
#else
#define QLS_ATTR QD
#endif
template <bool MULTI, bool BEYOND, class CS, class QProfT>
QLS_ATTR double line_search(const QuadModel& m_in, int leg, const QRows R, CS cs, int ncon, double hl0, double hl1, double hl2, const double (*Vs_in)[6], int pmask,
                             double q1, double q2, double gtol, QProfT& pf) {
  // (everything small arrives by value: an argument passed by reference pins the caller's copy in memory for the whole iteration)
  const QuadModel& m = QREBIND_LDS(QuadModel, m_in);
  const QuadLeg& L = m.leg[leg];
  const double hl[3] = {hl0, hl1, hl2};
  double Vs[4][6];
  {
    const auto* vp = QREBIND_PRIVATE(double, &Vs_in[0][0]);
    QUNROLL for (int d = 0; d < 4; d++) QUNROLL for (int k = 0; k < 6; k++) Vs[d][k] = vp[6 * d + k];
  }
  QDiag dg;
  diag_prepare(L, R, hl, dg);
  const bool any_limit = qw_any(R.lm_side[0] != 0 || R.lm_side[1] != 0 || R.lm_side[2] != 0);
  QLine ql[kQLineSlots];
  QLine qx[BEYOND ? kQMaxCon - kQLineSlots : 1];
  QPROF(pf, 40);
  const bool beyond = BEYOND && qd_or(ncon > kQLineSlots ? 1 : 0) != 0;
  rows_line_prepare<MULTI, BEYOND>(m, cs, ncon, Vs, pmask, beyond, ql, qx);
  const int nslot_wave = qw_max(ncon < kQLineSlots ? ncon : kQLineSlots);
  QPROF(pf, 41);
  double lo = 0, hi = -1, alpha = 0, d1, d2;
  rows_line<BEYOND>(dg, any_limit, ncon, 0.0, beyond, nslot_wave, ql, qx, d1, d2);
  d1 += q1; d2 += q2;
  const double d10 = fabs(d1);
  double step1 = 1e300, step2 = 1e300;  // the last step and the one before (rtsafe safeguard, oracle/contact.inc)
  int trials = 0;
#ifdef QEXP_MAXLS
  for (int ls = 0; ls < QEXP_MAXLS && d10 >= gtol; ls++) {
#else
  for (int ls = 0; ls < 50 && d10 >= gtol; ls++) {
#endif
    double an = alpha - d1 * q_rcp(d2);
    if (!(an > lo) || (hi >= 0 && !(an < hi))) an = hi >= 0 ? 0.5 * (lo + hi) : 2 * alpha + 1;
    else if (hi >= 0 && fabs(an - alpha) > 0.5 * step2) an = 0.5 * (lo + hi);
    if (an == alpha) break;
    step2 = step1; step1 = fabs(an - alpha);
    alpha = an;
    rows_line<BEYOND>(dg, any_limit, ncon, alpha, beyond, nslot_wave, ql, qx, d1, d2);
    d1 += q1 + alpha * q2; d2 += q2;
    if (fabs(d1) < gtol) break;
    if (d1 < 0) lo = alpha; else hi = alpha;
    trials++;
  }
  QPROF_COUNT(pf, 17, trials);
  QPROF(pf, 42);
  return alpha;
}
#ifdef QEXP_LS_SLOW_NOINLINE
template <bool MULTI, class CS, class QProfT>
QNOINLINE double line_search_beyond(const QuadModel& m_in, int leg, const QRows R, CS cs, int ncon, double hl0, double hl1, double hl2, const double (*Vs_in)[6], int pmask,
                                    double q1, double q2, double gtol, QProfT& pf) {
  return line_search<MULTI, true>(m_in, leg, R, cs, ncon, hl0, hl1, hl2, Vs_in, pmask, q1, q2, gtol, pf);
}
#else
template <bool MULTI, class CS, class QProfT>
QD double line_search_beyond(const QuadModel& m_in, int leg, const QRows R, CS cs, int ncon, double hl0, double hl1, double hl2, const double (*Vs_in)[6], int pmask,
                             double q1, double q2, double gtol, QProfT& pf) {
  return line_search<MULTI, true>(m_in, leg, R, cs, ncon, hl0, hl1, hl2, Vs_in, pmask, q1, q2, gtol, pf);
}
#endif
// H += J' (d2s) J of the contacts: the lane's leg block and coupling from its own contacts' blocks (X, from rows_eval), the trunk block
// from the quad sum of X. Contacts whose body is not the last link were counted for dofs below their body: taken out again.
template <class CS>
QD void hessian_common(const QuadModel& m, const QKin& kin, CS& cs, int nstat, double* X, int nshallow, Arrow& H);
// Self-collision contacts: the own leg's block (Hl, packed 3 x 3) and, for the lower leg of a pair, the cross block (Hab; rows: own dofs,
// columns: the partner's). Called BEFORE the iteration's X and H exist (QEXP_REL_LATE: after, as until round 4), so that their 75
// numbers are not alive around this loop.
template <class CS>
QD void hessian_rel(const QuadModel& m, const QKin& kin, CS& cs, int nstat, int ncon, int leg, int pmode, double* Hl, double (*Hab)[3]) {
  double cq[3][6];  // the partner leg's dof axes
  QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int k = 0; k < 6; k++) cq[j][k] = qd_partner(kin.cdof[j][k], pmode);
  const bool isA = leg < (leg ^ pmode);
  for (int i = nstat; i < ncon; i++) {  // (the self-collision contacts are the last of the lane's list)
    QContact c;
    qcs_load(cs, i, c);
    QHessOp h;
    contact_hess_prepare(c, m.fric[c.fid], h);
    if (h.zone == 0) continue;
    QUNROLL for (int j = 0; j < 3; j++) {
      if (!rel_dof(c, j)) continue;
      double Y[6];
      contact_hess_apply(c, h, kin.cdof[j], Y);
      QUNROLL for (int ii = 0; ii <= j; ii++) if (rel_dof(c, ii)) Hl[tri(j, ii)] += dot6(kin.cdof[ii], Y);
      if (isA && c.self == 0) { QUNROLL for (int ii = 0; ii < 3; ii++) if (ii < c.pd) Hab[j][ii] -= dot6(cq[ii], Y); }
    }
  }
}
// The same for a leg with several partners (leaf elimination, QPlan): one pass per xor value present; the lane accumulates its own block
// from all its self-collision contacts and the cross block of the ONE edge it is eliminated along (rows: own dofs, columns: that partner's).
template <class CS>
QD void hessian_rel_plan(const QuadModel& m, const QKin& kin, CS& cs, int nstat, int ncon, int leg, int pmask, const QPlan& plan, double* Hl, double (*Hab)[3]) {
  const int myx = plan_my_x(plan, leg);
  int x = next_x(pmask, 0);
  for (int pass = 0; pass == 0 || x < 4; pass++) {  // (quad-uniform)
    double cq[3][6];  // the dof axes of the leg x lanes across
    QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int k = 0; k < 6; k++) cq[j][k] = qd_partner(kin.cdof[j][k], x & 3);
    const bool holder = x < 4 && x == myx;
    for (int i = nstat; i < ncon; i++) {
      QContact c;
      qcs_load(cs, i, c);
      if (!in_pass(c, pass, x)) continue;
      QHessOp h;
      contact_hess_prepare(c, m.fric[c.fid], h);
      if (h.zone == 0) continue;
      QUNROLL for (int j = 0; j < 3; j++) {
        if (!rel_dof(c, j)) continue;
        double Y[6];
        contact_hess_apply(c, h, kin.cdof[j], Y);
        QUNROLL for (int ii = 0; ii <= j; ii++) if (rel_dof(c, ii)) Hl[tri(j, ii)] += dot6(kin.cdof[ii], Y);
        if (holder && c.px != 0) { QUNROLL for (int ii = 0; ii < 3; ii++) if (ii < c.pd) Hab[j][ii] -= dot6(cq[ii], Y); }
      }
    }
    x = next_x(pmask, x);
  }
}
template <class CS>
QD void hessian_common(const QuadModel& m, const QKin& kin, CS& cs, int nstat, double* X, int nshallow, Arrow& H) {
  QUNROLL for (int j = 0; j < 3; j++) {
    double Y[6];
    QUNROLL for (int p = 0; p < 6; p++) { double v = 0; QUNROLL for (int q = 0; q < 6; q++) v += X[tri(p, q)] * kin.cdof[j][q]; Y[p] = v; }
    QUNROLL for (int i = 0; i <= j; i++) H.l[tri(j, i)] += dot6(kin.cdof[i], Y);
    QUNROLL for (int k = 0; k < 6; k++) H.b[j][k] += trunk_dot(kin, k, Y);
  }
  if (nshallow > 0) {
    for (int i = 0; i < nstat; i++) {
      QContact c;
      qcs_load(cs, i, c);
      if (c.depth >= 3) continue;
      QHessOp h;
      contact_hess_prepare(c, m.fric[c.fid], h);
      if (h.zone == 0) continue;
      QUNROLL for (int j = 0; j < 3; j++) {
        if (j < c.depth) continue;
        double Y[6];
        contact_hess_apply(c, h, kin.cdof[j], Y);
        QUNROLL for (int ii = 0; ii <= j; ii++) H.l[tri(j, ii)] -= dot6(kin.cdof[ii], Y);
        QUNROLL for (int k = 0; k < 6; k++) H.b[j][k] -= trunk_dot(kin, k, Y);
      }
    }
  }
  qd_sum_n(X, 21);
  QUNROLL for (int k = 0; k < 6; k++) {
    double Y[6];  // X cdofT_k
    QUNROLL for (int p = 0; p < 6; p++) {
      if (k < 3) Y[p] = X[tri(p, 3 + k)];
      else { double v = 0; QUNROLL for (int q = 0; q < 3; q++) v += X[tri(p, q)] * kin.ca[k - 3][q] + X[tri(p, 3 + q)] * kin.cl[k - 3][q]; Y[p] = v; }
    }
    QUNROLL for (int i = 0; i <= k; i++) H.t[tri(k, i)] += trunk_dot(kin, i, Y);
  }
}

// M lives in the includer's store while the solver runs (LDS on the device: the leg block and the coupling per lane, the trunk block
// once per quad): element accessors qms_l / qms_b / qms_t, setters qms_set_*.
template <class MS>
QD void store_arrow(MS& ms, const Arrow& M) {
  QUNROLL for (int i = 0; i < 6; i++) qms_set_l(ms, i, M.l[i]);
  QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int k = 0; k < 6; k++) qms_set_b(ms, j, k, M.b[j][k]);
  QUNROLL for (int i = 0; i < 21; i++) qms_set_t(ms, i, M.t[i]);
}
template <class MS>
QD void load_arrow(const MS& ms, Arrow& M) {
  QUNROLL for (int i = 0; i < 6; i++) M.l[i] = qms_l(ms, i);
  QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int k = 0; k < 6; k++) M.b[j][k] = qms_b(ms, j, k);
  QUNROLL for (int i = 0; i < 21; i++) M.t[i] = qms_t(ms, i);
  QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c < 3; c++) M.ab[r][c] = 0;
}
template <class MS>
QD void arrow_mul_s(const MS& ms, const double* xl, const double* xt, double* yl, double* yt) {
  double c[6];
  QUNROLL for (int k = 0; k < 6; k++) c[k] = 0;
  QUNROLL for (int j = 0; j < 3; j++) {
    double s = 0;
    QUNROLL for (int i = 0; i < 3; i++) s += qms_l(ms, tri(j, i)) * xl[i];
    QUNROLL for (int k = 0; k < 6; k++) { const double b = qms_b(ms, j, k); s += b * xt[k]; c[k] += b * xl[j]; }
    yl[j] = s;
  }
  QUNROLL for (int k = 0; k < 6; k++) {
    double s = qd_sum(c[k]);
    QUNROLL for (int i = 0; i < 6; i++) s += qms_t(ms, tri(k, i)) * xt[i];
    yt[k] = s;
  }
}

// Newton solver. (sl, st) = qacc_smooth, (wl, wt) = warm start, M in the store `ms`; leaves qacc in (al, at) and J' force in
// (fc_l, fc_t). Returns the flag bits (quad-uniform).
template <bool GENERAL, class CS, class MS, class QProfT>
QD int newton_body(const QuadModel& m, const QuadLeg& L, const QKin& kin, const MS& ms, QRows& R, CS& cs, int ncon, int nrel, int leg, int pmask, int mymask, bool have_rel,
                   const double* sl, const double* st, const double* wl, const double* wt, bool have_warm,
                   double* al, double* at, double* fc_l, double* fc_t, int& iters, QProfT& pf) {
  iters = 0;
  const int nstat = ncon - nrel;  // the lane's contacts with static geoms come first, its nrel self-collision contacts after them
  QPlan plan;
  if constexpr (GENERAL) {
    plan = make_plan(mymask, leg);
    if (plan.cyclic) return kFlagPair;  // (quad-uniform) three legs touching one another: handed to the wavefront-per-candidate kernel
  }
  // !GENERAL: one pair pattern (or none), the super-leg factorisation; GENERAL: leaf elimination over several patterns (QPlan)
  const int pmode = GENERAL ? 0 : (next_x(pmask, 0) & 3);
  QUNROLL for (int j = 0; j < 3; j++) al[j] = sl[j];
  QUNROLL for (int k = 0; k < 6; k++) at[k] = st[k];
  double Mal[3] = {0, 0, 0}, Mat[6] = {0, 0, 0, 0, 0, 0};  // M (qacc - qacc_smooth), carried through the iterations
  double cost;
  QPROF(pf, 13);
  {
    double Vp[4][6];
    chain_velocity(kin, al, at, Vp);
    cost = rows_eval<GENERAL>(m, L, kin, R, cs, ncon, kEvalStep, al, Vp, 1.0, fc_l, fc_t, pmask);  // jar = J qacc_smooth - aref; the Gauss term is zero here
    QPROF(pf, 14);
    if (have_warm) {
      double dl[3], dt[6], Ml[3], Mt[6];
      QUNROLL for (int j = 0; j < 3; j++) dl[j] = wl[j] - sl[j];
      QUNROLL for (int k = 0; k < 6; k++) dt[k] = wt[k] - st[k];
      arrow_mul_s(ms, dl, dt, Ml, Mt);
      const double gauss = 0.5 * arrow_dot(dl, dt, Ml, Mt);
      double jl[3], jt[6];
      chain_velocity(kin, dl, dt, Vp);  // the rows move from qacc_smooth to the warm start along their difference
      const double cw = gauss + rows_eval<GENERAL>(m, L, kin, R, cs, ncon, kEvalStep, dl, Vp, 1.0, jl, jt, pmask);
      if (cw < cost) {
        cost = cw;
        QUNROLL for (int j = 0; j < 3; j++) { al[j] = wl[j]; fc_l[j] = jl[j]; Mal[j] = Ml[j]; }
        QUNROLL for (int k = 0; k < 6; k++) { at[k] = wt[k]; fc_t[k] = jt[k]; Mat[k] = Mt[k]; }
      } else {
        rows_step_only<GENERAL>(L, R, cs, ncon, dl, Vp, -1.0, pmask);  // and back: the first pass's cost and forces are still held
      }
    }
  }
  QPROF(pf, 15);
  const double scale = 1.0 / (m.meaninertia * 18.0);
  double improvement = 0;
#ifdef QEXP_MAXITER
  for (int iter = 0; iter < QEXP_MAXITER; iter++) {
#else
  for (int iter = 0; iter < m.iterations; iter++) {
#endif
    // gradient = M (qacc - qacc_smooth) - J' force
    double hl[3], ht[6];
    QUNROLL for (int j = 0; j < 3; j++) hl[j] = Mal[j] - fc_l[j];
    QUNROLL for (int k = 0; k < 6; k++) ht[k] = Mat[k] - fc_t[k];
    const double gnorm = sqrt(arrow_dot(hl, ht, hl, ht));
    if (gnorm == 0) break;
    if (iter > 0 && (scale * improvement < m.tolerance || scale * gnorm < m.tolerance)) break;
    QPROF(pf, 8);
    double X[21];
    int nshallow;
    {
      // H = M + J' (d2s) J: the self-collision contacts' blocks first (into small accumulators, nothing else of the Hessian alive yet),
      // then the diagonal rows and the other contacts through their 6 x 6 spatial blocks; factored in place
      double Hl_rel[6] = {0, 0, 0, 0, 0, 0}, Hab_rel[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      if (have_rel) {  // (quad-uniform)
        if constexpr (GENERAL) hessian_rel_plan(m, kin, cs, nstat, ncon, leg, pmask, plan, Hl_rel, Hab_rel);
        else hessian_rel(m, kin, cs, nstat, ncon, leg, pmode, Hl_rel, Hab_rel);
      }
      rows_X(m, cs, nstat, X, nshallow);
      Arrow H;
      load_arrow(ms, H);

      QUNROLL for (int i = 0; i < 6; i++) H.l[i] += Hl_rel[i];
      QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c < 3; c++) H.ab[r][c] = Hab_rel[r][c];
      QUNROLL for (int j = 0; j < 3; j++) {
        if (L.floss[j] > 0) { const double x = R.fl_jar[j]; if (x > -L.floss_R[j] * L.floss[j] && x < L.floss_R[j] * L.floss[j]) H.l[tri(j, j)] += L.floss_D[j]; }
        if (R.lm_side[j] != 0 && R.lm_jar[j] < 0) H.l[tri(j, j)] += R.lm_D[j];
      }
      hessian_common(m, kin, cs, nstat, X, nshallow, H);
      QPROF(pf, 9);
      bool pd;
      if constexpr (GENERAL) pd = arrow_factor_plan(H, leg, plan); else pd = arrow_factor(H, leg, pmode);
      if (!pd) return kFlagNotPD;
      QUNROLL for (int j = 0; j < 3; j++) hl[j] = -hl[j];  // search direction
      QUNROLL for (int k = 0; k < 6; k++) ht[k] = -ht[k];
      if constexpr (GENERAL) arrow_solve_plan(H, hl, ht, leg, plan); else arrow_solve(H, hl, ht, leg, pmode);
    }
    QPROF(pf, 10);
    double q1, q2, snorm;
    {
      double Msl[3], Mst[6];
      arrow_mul_s(ms, hl, ht, Msl, Mst);
      q1 = arrow_dot(hl, ht, Mal, Mat); q2 = arrow_dot(hl, ht, Msl, Mst); snorm = arrow_dot(hl, ht, hl, ht);
      // M (qacc - qacc_smooth) moves along M search: Mal += alpha Msl after the line search
      double Vs[4][6];
      chain_velocity(kin, hl, ht, Vs);
      const double gtol = m.tolerance * kQLsTol * sqrt(snorm) / scale;
      const double alpha = qw_any(ncon > kQLineSlots) ? line_search_beyond<GENERAL>(m, leg, R, cs, ncon, hl[0], hl[1], hl[2], Vs, pmask, q1, q2, gtol, pf)
                                                      : line_search<GENERAL, false>(m, leg, R, cs, ncon, hl[0], hl[1], hl[2], Vs, pmask, q1, q2, gtol, pf);
      QPROF(pf, 11);
      QUNROLL for (int j = 0; j < 3; j++) al[j] += alpha * hl[j];
      QUNROLL for (int k = 0; k < 6; k++) at[k] += alpha * ht[k];
      // the Gauss term at the new point, as the oracle evaluates it: from M (qacc - qacc_smooth) recomputed
      double dl[3], dt[6];
      QUNROLL for (int j = 0; j < 3; j++) dl[j] = al[j] - sl[j];
      QUNROLL for (int k = 0; k < 6; k++) dt[k] = at[k] - st[k];
      arrow_mul_s(ms, dl, dt, Mal, Mat);
      const double gauss = 0.5 * arrow_dot(dl, dt, Mal, Mat);
#ifdef QEXP_VE
      double Ve[4][6];  // (the chain velocities again)
      chain_velocity(kin, hl, ht, Ve);
      const double newcost = gauss + rows_eval<GENERAL>(m, L, kin, R, cs, ncon, kEvalStep, hl, Ve, alpha, fc_l, fc_t, pmask);
#else
      const double newcost = gauss + rows_eval<GENERAL>(m, L, kin, R, cs, ncon, kEvalStep, hl, Vs, alpha, fc_l, fc_t, pmask);
#endif
      improvement = cost - newcost;
      cost = newcost;
    }
    iters = iter + 1;
    QPROF(pf, 12);
  }
  return 0;
}

#ifdef QEXP_NEWTON_INLINE
#define QNEWTON_ATTR QD
#else
#define QNEWTON_ATTR QNOINLINE
#endif
template <bool GENERAL, class CS, class MS, class QProfT>
QNEWTON_ATTR int constraint_newton(const QuadModel& m_in, const QKin& kin_in, const MS& ms_in, QRows& R_in, CS& cs_in, int ncon, int nrel, int leg, int pmask, int mymask, bool have_rel,
                         const double* sl_in, const double* st_in, const double* wl_in, const double* wt_in, bool have_warm,
                         double* al_out, double* at_out, double* fc_l_out, double* fc_t_out, int& iters_out, QProfT& pf_in) {
  // (an out-of-line function on the device: its register allocation starts afresh, so the rollout's state is parked once per step instead
  // of being spilled around inside the iteration. Everything that arrives by reference is copied to locals first: a by-reference operand
  // is re-read from the caller's stack at every use. The model image is in LDS, the caller's locals in its private segment.)
  const QuadModel& m = QREBIND_LDS(QuadModel, m_in);
  const QuadLeg& L = m.leg[leg];
  QProfT pf = pf_in;  // (a local copy: through the reference every stamp would start with a load of pf.buf from the caller's stack and wait for it)
  QKin kin;
  {
    const auto* kp = QREBIND_PRIVATE(double, &kin_in.cdof[0][0]);
    static_assert(sizeof(QKin) == 36 * sizeof(double), "QKin is 36 doubles: cdof, ca, cl");
    QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int k = 0; k < 6; k++) kin.cdof[j][k] = kp[6 * j + k];
    QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int k = 0; k < 3; k++) { kin.ca[j][k] = kp[18 + 3 * j + k]; kin.cl[j][k] = kp[27 + 3 * j + k]; }
  }
  const MS ms = ms_in;
  QRows R = R_in;
  CS cs = cs_in;
  double sl[3], st[6], wl[3], wt[6], al[3], at[6], fc_l[3], fc_t[6];
  {
    const auto* slp = QREBIND_PRIVATE(double, sl_in); const auto* stp = QREBIND_PRIVATE(double, st_in);
    const auto* wlp = QREBIND_PRIVATE(double, wl_in); const auto* wtp = QREBIND_PRIVATE(double, wt_in);
    QUNROLL for (int j = 0; j < 3; j++) { sl[j] = slp[j]; wl[j] = wlp[j]; }
    QUNROLL for (int k = 0; k < 6; k++) { st[k] = stp[k]; wt[k] = wtp[k]; }
  }
  int iters = 0;
  const int rc = newton_body<GENERAL>(m, L, kin, ms, R, cs, ncon, nrel, leg, pmask, mymask, have_rel, sl, st, wl, wt, have_warm, al, at, fc_l, fc_t, iters, pf);
  {
    auto* alp = QREBIND_PRIVATE(double, al_out); auto* atp = QREBIND_PRIVATE(double, at_out);
    auto* flp = QREBIND_PRIVATE(double, fc_l_out); auto* ftp = QREBIND_PRIVATE(double, fc_t_out);
    QUNROLL for (int j = 0; j < 3; j++) { alp[j] = al[j]; flp[j] = fc_l[j]; }
    QUNROLL for (int k = 0; k < 6; k++) { atp[k] = at[k]; ftp[k] = fc_t[k]; }
  }
  iters_out = iters;
  pf_in = pf;
  return rc;
}

// ---------------------------------------------------------------- collision of the lane's geoms with the static geoms
// a contact found: its record (mj_instantiateContact + mj_makeImpedance for its rows, in point space) goes to the lane's store
template <class PAIR, class CS>
QD void add_contact(const PAIR& p, const double* com, const double* cvel, int depth, double dist, const double* pos, const double* normal,
                    CS& cs, int& ncon, int& flags, int rel = 0, int sgn = 1, int pd = 0, int px = 0, int self = 0) {
  if (!(dist < p.margin)) return;
  if (ncon >= kQMaxCon) { flags |= kFlagOverflow; return; }
  QContact c;
  c.depth = depth; c.rel = rel; c.sgn = sgn; c.pd = pd; c.px = px; c.self = self;
  QUNROLL for (int k = 0; k < 3; k++) { c.n[k] = normal[k]; c.off[k] = pos[k] - com[k]; }
  c.fid = p.fid;
  const double x = dist - p.includemargin;
  const double imp = impedance(p.imp, x);
  double R0 = (1 - imp) / imp * p.diag;
  if (R0 < kQMinVal) R0 = kQMinVal;
  c.D0 = 1.0 / R0;
  // jar starts as -aref, aref = -b (J qvel) - k imp x on the normal row; J qvel in point space is the body's velocity at the point
  double w[3];
  cr3(w, cvel, c.off);
  const double kx = p.k * imp * x;
  QUNROLL for (int k = 0; k < 3; k++) {
    c.jar[k] = p.dim >= 4 ? p.b * cvel[k] : 0.0;
    c.jar[3 + k] = p.b * (cvel[3 + k] + w[k]) + kx * c.n[k];
  }
  qcs_store(cs, ncon, c);
  ncon++;
}
#ifdef QEXP_SPAIR_GLOBAL
#define QSPAIR_T QuadPair
#define QSPAIR(m, g, s) pairs[(s) * pair_stride]
#else
#define QSPAIR_T QuadSPair
#define QSPAIR(m, g, s) (m).spair[((g).spair >> (8 * (s))) & 255]
#endif
// one moving geom (world pose gp / gR) against every static geom; oracle o_collision's pair table.
// COMPACT on purpose (round 5): a ROLLED loop over the static geoms that first collects the pair's candidate points -- at most four: a sphere's
// one, a capsule's two ends, a box's first four corners within the margin, a cylinder's four rim points -- and then creates the contacts in ONE
// instance of add_contact. With the static loop unrolled and add_contact inlined at every shape's site the body of the caller's geom loop was
// 12.8 k instructions (77 KB against 64 KB of instruction cache): the stage took 79 k cycles per step for ~2 k executed instructions.
// `near`: bit s set if static geom s can be within the margin of anything `reach` from `origin` (static_near_mask, once per step and leg):
// the other static geoms of a scene -- props metres away -- then cost a register test per geom instead of a chain of LDS reads
QD int static_near_mask(const QuadModel& m, const QStaticPose* sp, const double* origin, double reach) {
  int near = 0;
  QNOUNROLL for (int s = 0; s < m.nstatic; s++) {
    const QuadStatic& S = m.stat[s];
    if (S.type < 0) continue;
    const double* p1 = sp[s].pos; const double* R1 = sp[s].mat;
    const double rel[3] = {origin[0] - p1[0], origin[1] - p1[1], origin[2] - p1[2]};
    double clear;  // a lower bound of the distance between the static geom and anything within `reach` of the origin
    if (S.type == MJPCX_GEOM_PLANE) clear = rel[0] * R1[2] + rel[1] * R1[5] + rel[2] * R1[8] - reach;
    else clear = sqrt(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]) - (S.type == MJPCX_GEOM_SPHERE ? S.size[0] : S.bound) - reach;
    if (clear < 0.01) near |= 1 << s;  // (the slack of collide_geom's own bounding tests; contact margins are below 9 mm: quad_build)
  }
  return near;
}
template <class CS>
QD void collide_geom(const QuadModel& m, const QStaticPose* sp, const QuadGeom& g, int near, const QuadPair* pairs /* [kQStatic] stride: QEXP_SPAIR_GLOBAL only */, int pair_stride, const double* com, const double* cvel, int depth, const double* gp, const double* gR, CS& cs, int& ncon, int& flags) {
  QNOUNROLL for (int s = 0; s < m.nstatic; s++) {
    if (((near >> s) & 1) == 0) continue;
    const QuadStatic& S = m.stat[s];
    if (S.type < 0) continue;
    const double* p1 = sp[s].pos; const double* R1 = sp[s].mat;
    double cp[4][3], cd[4], cn[3] = {0, 0, 0};  // the pair's candidate points: contact position, distance; one normal
    QUNROLL for (int k = 0; k < 4; k++) { cd[k] = 0; QUNROLL for (int c = 0; c < 3; c++) cp[k][c] = 0; }
    int nc = 0;
    if (S.type == MJPCX_GEOM_PLANE) {
      const double n[3] = {R1[2], R1[5], R1[8]};
      // bounding-sphere rejection: nothing of the geom within the margin of the plane (margins are far below this slack)
      const double cdist = (gp[0] - p1[0]) * n[0] + (gp[1] - p1[1]) * n[1] + (gp[2] - p1[2]) * n[2];
      if (cdist - g.bound >= 0.01) continue;
      if (!((g.static_mask >> s) & 1)) continue;
      QUNROLL for (int k = 0; k < 3; k++) cn[k] = n[k];
      if (g.type == MJPCX_GEOM_SPHERE || g.type == MJPCX_GEOM_CAPSULE) {
        // (a capsule: its two end spheres; a sphere: one of zero half length -- the second point is then not counted)
        const double half = g.type == MJPCX_GEOM_CAPSULE ? g.size[1] : 0.0, r = g.size[0];
        QUNROLL for (int e = 0; e < 2; e++) {
          const double sg = e == 0 ? -1.0 : 1.0;
          double c[3];
          QUNROLL for (int k = 0; k < 3; k++) c[k] = gp[k] + sg * half * gR[3 * k + 2];
          const double dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2] - r;
          QUNROLL for (int k = 0; k < 3; k++) cp[e][k] = c[k] - n[k] * (r + 0.5 * dist);
          cd[e] = dist;
        }
        nc = g.type == MJPCX_GEOM_CAPSULE ? 2 : 1;
        if (g.type == MJPCX_GEOM_SPHERE) { QUNROLL for (int k = 0; k < 3; k++) cp[0][k] = gp[k] - n[k] * (r + 0.5 * cd[0]); }  // (exactly the sphere's own formula: half = 0 adds -0.0 * axis)
      } else if (g.type == MJPCX_GEOM_BOX) {
        const QSPAIR_T& pb = QSPAIR(m, g, s);
        QNOUNROLL for (int i = 0; i < 8; i++) {
          const double loc[3] = {(i & 1 ? g.size[0] : -g.size[0]), (i & 2 ? g.size[1] : -g.size[1]), (i & 4 ? g.size[2] : -g.size[2])};
          double c[3];
          mv3(c, gR, loc);
          QUNROLL for (int k = 0; k < 3; k++) c[k] += gp[k];
          const double dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2];
          const bool hit = dist < pb.margin && nc < 4;
          QUNROLL for (int q4 = 0; q4 < 4; q4++) {
            const bool put = hit && nc == q4;
            cd[q4] = put ? dist : cd[q4];
            QUNROLL for (int k = 0; k < 3; k++) cp[q4][k] = put ? c[k] - 0.5 * dist * n[k] : cp[q4][k];
          }
          nc += hit ? 1 : 0;
        }
      } else if (g.type == MJPCX_GEOM_CYLINDER) {
        const double a[3] = {gR[2], gR[5], gR[8]};
        const double pa = n[0] * a[0] + n[1] * a[1] + n[2] * a[2];
        const double sgn = pa > 0 ? -1.0 : 1.0;
        double v[3], vn = 0;
        QUNROLL for (int k = 0; k < 3; k++) { v[k] = -(n[k] - pa * a[k]); vn += v[k] * v[k]; }
        vn = sqrt(vn);
        if (vn < 1e-10) { v[0] = gR[0]; v[1] = gR[3]; v[2] = gR[6]; vn = 1; }
        QUNROLL for (int k = 0; k < 3; k++) v[k] /= vn;
        double w[3];
        cr3(w, a, v);
        const double cs3[3] = {1.0, -0.5, -0.5}, sn3[3] = {0.0, 0.8660254037844386, -0.8660254037844386};
        QUNROLL for (int i = 0; i < 4; i++) {
          double c[3];
          const double side = i < 3 ? sgn : -sgn, cc = i < 3 ? cs3[i] : 1.0, ss = i < 3 ? sn3[i] : 0.0;
          QUNROLL for (int k = 0; k < 3; k++) c[k] = gp[k] + side * g.size[1] * a[k] + g.size[0] * (cc * v[k] + ss * w[k]);
          const double dist = (c[0] - p1[0]) * n[0] + (c[1] - p1[1]) * n[1] + (c[2] - p1[2]) * n[2];
          QUNROLL for (int k = 0; k < 3; k++) cp[i][k] = c[k] - 0.5 * dist * n[k];
          cd[i] = dist;
        }
        nc = 4;
      }
    } else if (g.type != MJPCX_GEOM_SPHERE) {
      continue;  // static spheres and boxes collide with moving spheres only
    } else if (S.type == MJPCX_GEOM_SPHERE) {
      double n[3], len = 0;
      QUNROLL for (int k = 0; k < 3; k++) { n[k] = gp[k] - p1[k]; len += n[k] * n[k]; }
      const double reach = S.size[0] + g.size[0] + 0.01;
      if (len >= reach * reach) continue;
      if (!((g.static_mask >> s) & 1)) continue;
      len = sqrt(len);
      const double r1 = S.size[0], dist = len - r1 - g.size[0];
      if (len < kQMinVal) { n[0] = 1; n[1] = n[2] = 0; } else QUNROLL for (int k = 0; k < 3; k++) n[k] /= len;
      QUNROLL for (int k = 0; k < 3; k++) { cn[k] = n[k]; cp[0][k] = p1[k] + n[k] * (r1 + 0.5 * dist); }
      cd[0] = dist; nc = 1;
    } else if (S.type == MJPCX_GEOM_BOX) {
      const double* s1 = S.size;
      double rel[3], loc[3], clamped[3];
      QUNROLL for (int k = 0; k < 3; k++) rel[k] = gp[k] - p1[k];
      const double br = S.bound + g.size[0] + 0.01;
      if (rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2] >= br * br) continue;
      if (!((g.static_mask >> s) & 1)) continue;
      QUNROLL for (int k = 0; k < 3; k++) loc[k] = R1[k] * rel[0] + R1[3 + k] * rel[1] + R1[6 + k] * rel[2];
      bool inside = true;
      QUNROLL for (int k = 0; k < 3; k++) {
        clamped[k] = loc[k] < -s1[k] ? -s1[k] : (loc[k] > s1[k] ? s1[k] : loc[k]);
        if (clamped[k] != loc[k]) inside = false;
      }
      double nl[3] = {0, 0, 0}, dist;
      if (!inside) {
        double len = 0;
        QUNROLL for (int k = 0; k < 3; k++) { nl[k] = loc[k] - clamped[k]; len += nl[k] * nl[k]; }
        len = sqrt(len);
        QUNROLL for (int k = 0; k < 3; k++) nl[k] /= len;
        dist = len - g.size[0];
      } else {
        int best = 0; double bd = 1e300;
        QUNROLL for (int k = 0; k < 3; k++) { const double dd = s1[k] - fabs(loc[k]); if (dd < bd) { bd = dd; best = k; } }
        QUNROLL for (int k = 0; k < 3; k++) if (k == best) { nl[k] = loc[k] >= 0 ? 1 : -1; clamped[k] = nl[k] * s1[k]; }
        dist = -bd - g.size[0];
      }
      double n[3], surf[3];
      mv3(n, R1, nl);
      mv3(surf, R1, clamped);
      QUNROLL for (int k = 0; k < 3; k++) { cn[k] = n[k]; cp[0][k] = p1[k] + surf[k] + 0.5 * dist * n[k]; }
      cd[0] = dist; nc = 1;
    }
    if (nc == 0) continue;
    // the pair's contacts, in the candidates' order: one instance of the contact's creation, the candidates shifted through slot 0
    const QSPAIR_T& p = QSPAIR(m, g, s);
    QNOUNROLL for (int k = 0; k < nc; k++) {
      add_contact(p, com, cvel, depth, cd[0], cp[0], cn, cs, ncon, flags);
      QUNROLL for (int j = 0; j < 3; j++) { cd[j] = cd[j + 1]; QUNROLL for (int c = 0; c < 3; c++) cp[j][c] = cp[j + 1][c]; }
    }
  }
}

// (sphere | capsule, cylinder): solid_pairs.h in a frame built on the cylinder's axis (the solid is one of revolution: any frame about its
// axis gives the same contact; oracle pair_thin_solid uses the geom's own). geo: thin geom's centre, axis; cylinder's centre, axis; the thin
// geom's half length, radius. Returns the distance; res: world normal (thin geom -> cylinder), contact position.
QNOINLINE double cylinder_contact(const double* geo_in, double r2, double h2, double* res_out) {
  const auto* geo = QREBIND_PRIVATE(double, geo_in);
  auto* res = QREBIND_PRIVATE(double, res_out);
  const double p1[3] = {geo[0], geo[1], geo[2]}, a1[3] = {geo[3], geo[4], geo[5]}, p2[3] = {geo[6], geo[7], geo[8]}, a2[3] = {geo[9], geo[10], geo[11]};
  const double h1 = geo[12], r1 = geo[13];
  double e1[3], e2[3];
  const bool yy = a2[1] < 0.5 && a2[1] > -0.5;
  e1[0] = 0; e1[1] = yy ? 1.0 : 0.0; e1[2] = yy ? 0.0 : 1.0;
  const double dt = a2[0] * e1[0] + a2[1] * e1[1] + a2[2] * e1[2];
  for (int k = 0; k < 3; k++) e1[k] -= dt * a2[k];
  const double nn = 1.0 / sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
  for (int k = 0; k < 3; k++) e1[k] *= nn;
  cr3(e2, a2, e1);
  const double rel[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const double pl[3] = {rel[0] * e1[0] + rel[1] * e1[1] + rel[2] * e1[2], rel[0] * e2[0] + rel[1] * e2[1] + rel[2] * e2[2], rel[0] * a2[0] + rel[1] * a2[1] + rel[2] * a2[2]};
  const double al[3] = {a1[0] * e1[0] + a1[1] * e1[1] + a1[2] * e1[2], a1[0] * e2[0] + a1[1] * e2[1] + a1[2] * e2[2], a1[0] * a2[0] + a1[1] * a2[1] + a1[2] * a2[2]};
  const double size[3] = {r2, h2, 0.0};
  double nl[3], cl[3];
  const double dist = solid::thin_vs_solid<double>(solid::kSolidCylinder, size, pl, al, h1, r1, nl, cl);
  const double s = r1 + 0.5 * dist;
  for (int k = 0; k < 3; k++) {
    res[k] = nl[0] * e1[k] + nl[1] * e2[k] + nl[2] * a2[k];
    res[3 + k] = p2[k] + (cl[0] + nl[0] * s) * e1[k] + (cl[1] + nl[1] * s) * e2[k] + (cl[2] + nl[2] * s) * a2[k];
  }
  return dist;
}

// Self-collision (oracle pair_collide over the baked moving-geom pairs): the cross product of the own leg's pair geoms with the trunk's
// and with every other leg's (their centres / axes / link velocities arrive through quad rotations). Bounding spheres first, the exact
// nearest points for the pairs that pass; a pair within its margin becomes a RELATIVE contact (QContact::rel) in the own lane's list --
// and, for a leg-leg pair, identically in the partner's lane, which walks the same pair from its side with the same arithmetic (the two
// geoms are always taken in MuJoCo's order: geom1 first). pmask collects 1 << (own leg xor partner leg) of the leg-leg contacts.
// what the tests need of the lane's state: handed over in memory (the function is out of line: below)
struct QPairArgs {
  QPairGeoms pg;
  double txpos[3], txm[9], com[3], cvel[3][6], cvelT[6];
  int need;                      // bit 0: the trunk's pair geoms, bit 1: the own leg's pairs, bit 1 + d: the leg d lanes on (quad-uniform)
  int ncon, flags, pmask, nrel;  // in / out
};
// Out of line and COMPACT: one instance of the bounding-sphere test, of the exact tests and of the contact's creation, in rolled loops over
// arrays that live in memory and are indexed at run time. (Inlined into the forward pass and unrolled over the 8 x 8 grid of a partner the
// stage was 30 % of the launch -- 52.7 -> 37.1 ms with the stage skipped at run time, same box -- although the north-star batch never
// has a pair within reach: tens of kilobytes of cold straight-line code per step.) Called only at the steps at which the leg-level cull
// (pair_contacts) leaves something to test for some candidate of the wavefront.
template <class CS, class QProfT>
QNOINLINE void pair_contacts_tests(const QuadModel& m_in, const QuadTables& tab, int leg, QPairArgs* args_in, CS cs, QProfT& pf_in) {
  const QuadModel& m = QREBIND_LDS(QuadModel, m_in);
  auto* args = QREBIND_PRIVATE(QPairArgs, args_in);
  QProfT pf = pf_in;
#ifdef QEXP_NEED_MASK
  const int need = args->need & (QEXP_NEED_MASK);  // (tuning: which of the sources below the time goes to)
#else
  const int need = args->need;
#endif
  int ncon = args->ncon, flags = args->flags, pmask = args->pmask, nrel = args->nrel;
  const QuadLeg& L = m.leg[leg];
  const double mg = m.pair_margin;
#ifdef QEXP_PAIRS_ENTRY
  if (mg > -1.0 && need >= 0) { pf_in = pf; return; }  // (tuning: the cost of the call and of handing the arguments over, without any test)
#endif
  const double com[3] = {args->com[0], args->com[1], args->com[2]};
  // The pretest of a pair: bounding spheres AND the boxes of the two geoms in the trunk's axes (a capsule's: half length along the axis plus
  // the radius). In a gait the legs work close to each other and the bounding spheres of their long capsules overlap all the time -- 1.7
  // exact tests per lane and step on the bench's batch, run in lock-step by the wavefront: that was the 14 ms of this stage -- while the
  // boxes of two near-vertical capsules a hand apart do not. The own geoms' centres and half extents (trunk axes) in registers
  // (compile-time indices only); everything indexed at run time lives in memory: the other side of the source being walked
  // (world centres, axes, link velocities), fetched when a pair passes the pretest.
  double xs[kQPairGeom][3], es[kQPairGeom][3], own_reach[kQPairGeom];
  QUNROLL for (int i = 0; i < kQPairGeom; i++) {
    const double rel[3] = {args->pg.c[i][0] - args->txpos[0], args->pg.c[i][1] - args->txpos[1], args->pg.c[i][2] - args->txpos[2]};
    QUNROLL for (int k = 0; k < 3; k++) {
      xs[i][k] = args->txm[k] * rel[0] + args->txm[3 + k] * rel[1] + args->txm[6 + k] * rel[2];  // (txm' rel)
      es[i][k] = L.pg_half[i] * fabs(args->txm[k] * args->pg.a[i][0] + args->txm[3 + k] * args->pg.a[i][1] + args->txm[6 + k] * args->pg.a[i][2]) + L.pg_rad[i];
    }
    own_reach[i] = L.pg_reach[i] + mg;
  }
  const double bmg = mg + 1e-9;
  double oc[kQPairGeom][3], oa[kQPairGeom][3], ov[3][6];
  for (int src = 0; src < 5; src++) {  // 0: the trunk's pair geoms, 1: the own leg (its cylinders), 2..4: the leg d = src - 1 lanes on
    const bool mine = ((need >> src) & 1) != 0;
    if (qd_or(mine ? 1 : 0) == 0) continue;  // (quad-uniform; the bits of the other legs are quad-uniform themselves)
    const int d = src - 1, o = src == 0 ? kQLegs : (src == 1 ? leg : ((leg + d) & 3));
    const int on = !mine ? 0 : (src == 0 ? m.ntpg : m.leg[o & 3].npg);
    // the pretest: a ROLLED loop over the other side's geoms (centre and half extents of geom j arrive -- through the quad rotation for
    // another leg -- and meet the eight own geoms in registers): a hundred instructions instead of the 8 x 8 grid unrolled
    unsigned long long mask = 0;
    for (int j = 0; j < (src == 0 ? kQTrunkPairGeom : kQPairGeom); j++) {
      double xj[3], ej[3];
      if (src == 0) {  // (a trunk geom: constant in the trunk's axes)
        const QuadGeom& g = m.trunk_geom[m.tpg_slot[j < m.ntpg ? j : 0]];
        const double half = g.type == MJPCX_GEOM_CAPSULE ? g.size[1] : 0.0, rad = g.type == MJPCX_GEOM_CAPSULE ? g.size[0] : g.bound;
        QUNROLL for (int k = 0; k < 3; k++) { xj[k] = g.pos[k]; ej[k] = half * fabs(g.rot[3 * k + 2]) + rad; }
      } else {
        QUNROLL for (int k = 0; k < 3; k++) {  // (the own geom j by 0 / 1 weights: a run-time index would put the arrays in memory)
          double vx = 0, ve = 0;
          QUNROLL for (int q = 0; q < kQPairGeom; q++) { const double w = j == q ? 1.0 : 0.0; vx += w * xs[q][k]; ve += w * es[q][k]; }
          xj[k] = src == 1 ? vx : qd_rotv(vx, d);
          ej[k] = src == 1 ? ve : qd_rotv(ve, d);
        }
      }
      if (j >= on) continue;
      const double oreach = src == 0 ? m.tpg_reach[j] : m.leg[o & 3].pg_reach[j];
      QUNROLL for (int i = 0; i < kQPairGeom; i++) {
        const double reach = own_reach[i] + oreach;
        const double dx = xs[i][0] - xj[0], dy = xs[i][1] - xj[1], dz = xs[i][2] - xj[2];
        const bool boxes = fabs(dx) < es[i][0] + ej[0] + bmg && fabs(dy) < es[i][1] + ej[1] + bmg && fabs(dz) < es[i][2] + ej[2] + bmg;
        if (i < L.npg && boxes && dx * dx + dy * dy + dz * dz < reach * reach) mask |= 1ull << (8 * i + j);
      }
    }
    mask &= L.pg_active[o];
    if (src >= 2) {
      if (qd_or(mask != 0 ? 1 : 0) == 0) continue;  // (quad-uniform: the axes and velocities are only fetched for a partner that is near)
      for (int j = 0; j < kQPairGeom; j++) for (int k = 0; k < 3; k++) { oc[j][k] = qd_rotv(args->pg.c[j][k], d); oa[j][k] = qd_rotv(args->pg.a[j][k], d); }
      for (int j = 0; j < 3; j++) for (int k = 0; k < 6; k++) ov[j][k] = qd_rotv(args->cvel[j][k], d);
      QPROF_COUNT(pf, 45, 1);
    } else if (mask != 0) {
      if (src == 0) {
        for (int j = 0; j < m.ntpg; j++) {
          const QuadGeom& g = m.trunk_geom[m.tpg_slot[j]];
          for (int k = 0; k < 3; k++) {
            oc[j][k] = args->txm[3 * k] * g.pos[0] + args->txm[3 * k + 1] * g.pos[1] + args->txm[3 * k + 2] * g.pos[2] + args->txpos[k];
            oa[j][k] = args->txm[3 * k] * g.rot[2] + args->txm[3 * k + 1] * g.rot[5] + args->txm[3 * k + 2] * g.rot[8];
          }
        }
        for (int j = 0; j < 3; j++) for (int k = 0; k < 6; k++) ov[j][k] = args->cvelT[k];
      } else {
        for (int j = 0; j < kQPairGeom; j++) for (int k = 0; k < 3; k++) { oc[j][k] = args->pg.c[j][k]; oa[j][k] = args->pg.a[j][k]; }
        for (int j = 0; j < 3; j++) for (int k = 0; k < 6; k++) ov[j][k] = args->cvel[j][k];
      }
    }
    while (mask) {  // (ascending bits: own geom first, then the other's -- the order contacts are created in)
      const int bit = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int i = bit >> 3, j = bit & 7;
      {
        const QuadGeom& g = L.geom[L.pg_slot[i]];
        const double ci[3] = {args->pg.c[i][0], args->pg.c[i][1], args->pg.c[i][2]}, ai[3] = {args->pg.a[i][0], args->pg.a[i][1], args->pg.a[i][2]};
        const double r0 = g.size[0], h0 = g.type != MJPCX_GEOM_SPHERE ? g.size[1] : 0.0;  // (capsule, cylinder: half length)
        const QuadGeom& g2 = o == kQLegs ? m.trunk_geom[m.tpg_slot[j]] : m.leg[o & 3].geom[m.leg[o & 3].pg_slot[j]];
        QPROF_COUNT(pf, src == 0 ? 43 : 44, 1);
        // ---- the exact test of one pair: own geom (index i of the leg's pair geoms) against (other leg o or kQLegs = trunk, index j).
        // (Nothing of the pair's table record -- global memory -- is touched before a distance is below the largest margin: the order of the
        // two geoms comes from the leg's bit mask, the record is read when a contact is created.)
        const int otype = g2.type, olink = o == kQLegs ? 0 : g2.link, odepth = o == kQLegs ? 0 : g2.link + 1;
        const double orad = g2.size[0], ohalf = g2.type != MJPCX_GEOM_SPHERE ? g2.size[1] : 0.0;
        const bool own_first = ((L.pg_first[o] >> (8 * i + j)) & 1) != 0;
        // geom1 / geom2 in MuJoCo's order
        double p1[3], p2[3], a1[3], a2[3];
        for (int k = 0; k < 3; k++) { p1[k] = own_first ? ci[k] : oc[j][k]; p2[k] = own_first ? oc[j][k] : ci[k]; a1[k] = own_first ? ai[k] : oa[j][k]; a2[k] = own_first ? oa[j][k] : ai[k]; }
        const int t1 = own_first ? g.type : otype, t2 = own_first ? otype : g.type;
        const double r1 = own_first ? r0 : orad, r2 = own_first ? orad : r0, h1 = own_first ? h0 : ohalf, h2 = own_first ? ohalf : h0;
        const int sgn = own_first ? -1 : 1, depth = g.link + 1;
        // up to four candidate point pairs (centres of the two balls the contact reduces to), of which at most `limit` become contacts
        double q1[4][3], q2[4][3], qn[3] = {0, 0, 0}, qpos[3] = {0, 0, 0}, qdist = 0;
        int nq = 0, limit = 1;
        bool direct = false;  // (a cylinder: the contact comes whole from solid_pairs.h)
        auto seg = [](const double* p, const double* a, double h, const double* c) {
          const double x = (c[0] - p[0]) * a[0] + (c[1] - p[1]) * a[1] + (c[2] - p[2]) * a[2];
          return x < -h ? -h : (x > h ? h : x);
        };
        if (t2 == MJPCX_GEOM_CYLINDER) {
          // (sphere | capsule, cylinder) -- a calf or foot against a hip (two solids: proven apart at bake time or reported, never walked)
          if (t1 == MJPCX_GEOM_CYLINDER) continue;
          double geo[14] = {p1[0], p1[1], p1[2], a1[0], a1[1], a1[2], p2[0], p2[1], p2[2], a2[0], a2[1], a2[2], t1 == MJPCX_GEOM_CAPSULE ? h1 : 0.0, r1};
          double res[6];
          qdist = cylinder_contact(geo, r2, h2, res);
          for (int k = 0; k < 3; k++) { qn[k] = res[k]; qpos[k] = res[3 + k]; }
          direct = true; nq = 1;
        } else if (t1 == MJPCX_GEOM_SPHERE && t2 == MJPCX_GEOM_SPHERE) {
          for (int k = 0; k < 3; k++) { q1[0][k] = p1[k]; q2[0][k] = p2[k]; }
          nq = 1;
        } else if (t1 == MJPCX_GEOM_SPHERE) {  // (sphere, capsule): spheres come first in MuJoCo's order
          const double x = seg(p2, a2, h2, p1);
          for (int k = 0; k < 3; k++) { q1[0][k] = p1[k]; q2[0][k] = p2[k] + x * a2[k]; }
          nq = 1;
        } else {
          const double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
          const double mb = -(a1[0] * a2[0] + a1[1] * a2[1] + a1[2] * a2[2]);
          const double u = -(a1[0] * dif[0] + a1[1] * dif[1] + a1[2] * dif[2]);
          const double v = a2[0] * dif[0] + a2[1] * dif[1] + a2[2] * dif[2];
          const double det = 1.0 - mb * mb;
          if (fabs(det) >= kQMinVal) {
            double x1 = (u - mb * v) / det, x2 = (v - mb * u) / det;
            if (x1 > h1) { x1 = h1; x2 = v - mb * x1; } else if (x1 < -h1) { x1 = -h1; x2 = v - mb * x1; }
            if (x2 > h2) { x2 = h2; x1 = u - mb * x2; x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1); }
            else if (x2 < -h2) { x2 = -h2; x1 = u - mb * x2; x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1); }
            for (int k = 0; k < 3; k++) { q1[0][k] = p1[k] + x1 * a1[k]; q2[0][k] = p2[k] + x2 * a2[k]; }
            nq = 1;
          } else {  // parallel: the ends of capsule 1 against axis 2, then the ends of capsule 2 against axis 1, two contacts at most
            for (int e = 0; e < 4; e++) {
              const double sg = (e & 1) ? -1.0 : 1.0;
              if (e < 2) {
                for (int k = 0; k < 3; k++) q1[e][k] = p1[k] + sg * h1 * a1[k];
                const double x2 = seg(p2, a2, h2, q1[e]);
                for (int k = 0; k < 3; k++) q2[e][k] = p2[k] + x2 * a2[k];
              } else {
                for (int k = 0; k < 3; k++) q2[e][k] = p2[k] + sg * h2 * a2[k];
                const double x1 = seg(p1, a1, h1, q2[e]);
                for (int k = 0; k < 3; k++) q1[e][k] = p1[k] + x1 * a1[k];
              }
            }
            nq = 4; limit = 2;
          }
        }
        int added = 0;
        for (int e = 0; e < nq && added < limit; e++) {  // oracle sphere_vs_sphere -> add_contact
          double n[3], pos[3], dist;
          double len = 0;
          if (direct) { dist = qdist; for (int k = 0; k < 3; k++) { n[k] = qn[k]; pos[k] = qpos[k]; } }
          else {
            for (int k = 0; k < 3; k++) { n[k] = q2[e][k] - q1[e][k]; len += n[k] * n[k]; }
            len = sqrt(len);
            dist = len - r1 - r2;
          }
          if (!(dist < mg)) continue;
          const QuadPair& P = tab.mm[leg][i][o][j];
          if (!P.collide || !(dist < P.margin)) continue;
          if (!direct) {
            if (len < kQMinVal) { n[0] = 1; n[1] = n[2] = 0; } else { for (int k = 0; k < 3; k++) n[k] /= len; }
            for (int k = 0; k < 3; k++) pos[k] = q1[e][k] + n[k] * (r1 + 0.5 * dist);
          }
          double vrel[6];  // J qvel of J = jac(body2) - jac(body1), about the centre of mass
          for (int k = 0; k < 6; k++) {
            const double vo = args->cvel[g.link][k], vp = ov[olink][k];
            vrel[k] = own_first ? vp - vo : vo - vp;
          }
          const int before = ncon;
#ifdef QEXP_PAIRS_DRY
          if (dist > -1e30) continue;
#endif
          add_contact(P, com, vrel, depth, dist, pos, n, cs, ncon, flags, 1, sgn, odepth, o < kQLegs ? (leg ^ o) : 0, o == leg ? 1 : 0);
          if (ncon > before) { nrel++; if (o < kQLegs && o != leg) pmask |= 1 << (leg ^ o); }
          added++;
        }
      }
    }
  }
  args->ncon = ncon; args->flags = flags; args->pmask = pmask; args->nrel = nrel;
  pf_in = pf;
}

// The leg-level cull, exact: the box of the leg's pair geoms in the trunk's axes -- of their end spheres (a capsule's or cylinder's two ends,
// the radius about each; a cylinder's rim lies within its bounding sphere's radius of the axis ends' midpoint, so it takes the bounding
// sphere). Two geoms whose bounding volumes come within the margin have boxes that overlap (by more than -margin) on every axis, so a
// partner (another leg; the trunk's pair geoms: a constant box; the own leg: a joint box, pair_cull.h) whose box is clear on some axis
// needs no test. In a gait that is every partner at nearly every step; the tests themselves are out of line (pair_contacts_tests).
// The boxes of the cull, grown geom by geom inside the collision loop (QPairBoxes; until round 5 the loop kept the 48 doubles of the pair geoms'
// poses alive -- selects into compile-time slots, 96 v_cndmask per geom -- and spilled the link frames around them: 9 scratch round trips per
// geom, 76 k cycles per step). The poses themselves go straight into the argument block of the tests, which lives in memory anyway.
struct QPairBoxes { double blo[3], bhi[3], tlo[3], thi[3]; };  // the box of all the leg's pair geoms; of those that pair with a trunk geom (not the hip's)
QD void pair_boxes_init(QPairBoxes& b) { QUNROLL for (int k = 0; k < 3; k++) { b.blo[k] = b.tlo[k] = 1e30; b.bhi[k] = b.thi[k] = -1e30; } }
QD void pair_boxes_add(const QuadLeg& L, int i, const double* gp, const double* ga, const double* txpos, const double* txm, QPairBoxes& b) {
  const double rel[3] = {gp[0] - txpos[0], gp[1] - txpos[1], gp[2] - txpos[2]};
  const double outT = ((L.pg_active[kQLegs] >> (8 * i)) & 0xffull) != 0 ? 0.0 : 1e30;
  const double half = L.pg_half[i], rad = L.pg_rad[i];
  QUNROLL for (int k = 0; k < 3; k++) {
    const double x = txm[k] * rel[0] + txm[3 + k] * rel[1] + txm[6 + k] * rel[2];        // (txm' rel)
    const double ax = txm[k] * ga[0] + txm[3 + k] * ga[1] + txm[6 + k] * ga[2];
    const double ext = half * fabs(ax) + rad;                                            // half length along the axis, then the end sphere
    b.blo[k] = fmin(b.blo[k], x - ext); b.bhi[k] = fmax(b.bhi[k], x + ext);
    b.tlo[k] = fmin(b.tlo[k], x - ext + outT); b.thi[k] = fmax(b.thi[k], x + ext - outT);
  }
}
template <class CS, class QProfT>
QD void pair_contacts(const QuadModel& m, const QuadTables& tab, int leg, QPairArgs& args, const QPairBoxes& bx, const double* txpos, const double* txm, const double* com,
                      const double cvel[3][6], const double* cvelT, bool self_walk, CS& cs, int& ncon, int& flags, int& pmask, int& nrel, QProfT& pf) {
  const QuadLeg& L = m.leg[leg];
  const double mg = m.pair_margin;
#ifdef QEXP_PAIRS_SKIP
  if (mg > -1.0) return;  // (tuning: what a perfect cull of the self-collision stage would save; the solver keeps its self-collision paths)
#endif
  const double* blo = bx.blo; const double* bhi = bx.bhi; const double* tlo = bx.tlo; const double* thi = bx.thi;
  const double bmg = mg + 1e-9;
  bool trunk_near = L.pg_active[kQLegs] != 0;
  QUNROLL for (int k = 0; k < 3; k++) trunk_near = trunk_near && !(tlo[k] > m.tpg_box[1][k] + bmg) && !(m.tpg_box[0][k] > thi[k] + bmg);
  int need = (trunk_near ? 1 : 0) | ((L.pg_active[leg] != 0 && self_walk) ? 2 : 0);
  QUNROLL for (int d = 1; d <= 3; d++) {
    bool apart = false;
    QUNROLL for (int k = 0; k < 3; k++) {
      const double olo = qd_rotv(blo[k], d), ohi = qd_rotv(bhi[k], d);
      apart = apart || blo[k] > ohi + bmg || olo > bhi[k] + bmg;
    }
    need |= qd_or(apart ? 0 : 1) << (1 + d);  // (quad-uniform: clear only if all four (leg, leg + d) pairs are)
  }
  if (qd_or(need) == 0) return;
#ifdef QEXP_PAIRS_SKIP2
  if (mg > -1.0) return;  // (tuning: the leg-level cull runs, the tests never do)
#endif
  QUNROLL for (int k = 0; k < 3; k++) { args.txpos[k] = txpos[k]; args.com[k] = com[k]; }
  QUNROLL for (int k = 0; k < 9; k++) args.txm[k] = txm[k];
  QUNROLL for (int j = 0; j < 3; j++) QUNROLL for (int k = 0; k < 6; k++) args.cvel[j][k] = cvel[j][k];
  QUNROLL for (int k = 0; k < 6; k++) args.cvelT[k] = cvelT[k];
  args.need = need; args.ncon = ncon; args.flags = flags; args.pmask = pmask; args.nrel = nrel;
  pair_contacts_tests(m, tab, leg, &args, cs, pf);
  ncon = args.ncon; flags = args.flags; pmask = args.pmask; nrel = args.nrel;
}

// ---------------------------------------------------------------- mj_forward (oracle o_forward) for the lane's share of one candidate
// what the sensor stage (residual, traces) reads: nothing of it depends on the constraint solve
struct QSense {
  double txm[9], txq[4], txipos[3], com[3], comvel[3], head[3], foot[3];
  double trace[kQMaxTrace][3];
  double act_force[3];
};
// what the constraint solve and the integrator read
struct QDyn {
  QKin kin;
  QRows R;
  double sl[3], st[6];      // qacc_smooth
  double fs_l[3], fs_t[6];  // qfrc_smooth
  int ncon, nrel;           // the lane's contacts; the last nrel of them are between two moving geoms
  int pmask, have_rel;      // self-collision: bit x set if some leg A touches leg A xor x; whether the candidate has such contacts at all (quad-uniform)
  int mymask;               // bit x set if THIS leg touches leg ^ x
};
// Position and velocity stages, collision, smooth dynamics, constraint rows: everything of mj_forward before the constraint solve.
// ctrl: the leg's three controls. Returns flag bits (quad-uniform; 0: fine).
template <class CS, class MS, class QProfT>
QD int forward_smooth(const QuadModel& m, const QuadTables& tab, const QStaticPose* sp, int leg, const QState& S, const double* ctrl, CS& cs, MS& ms,
                      QDyn& D, QSense& out, QProfT& pf) {
  const QuadLeg& L = m.leg[leg];
  QKin& kin = D.kin;
  int flags = 0;
  // ================= kinematics (o_kinematics): trunk, then the leg's chain
  double txpos[3] = {S.tq[0], S.tq[1], S.tq[2]};
  double txq[4] = {S.tq[3], S.tq[4], S.tq[5], S.tq[6]};
  q_norm(txq); q_norm(txq);  // (the oracle normalises when it reads qpos and again at the end of the body loop)
  double txm[9];
  q2mat(txm, txq);
  double tmp3[3], tmpq[4];
  mv3(tmp3, txm, m.trunk_ipos);
  double txipos[3] = {txpos[0] + tmp3[0], txpos[1] + tmp3[1], txpos[2] + tmp3[2]};
  double tirot[6];
  { double timat[9]; q_mul(tmpq, txq, m.trunk_iquat); q2mat(timat, tmpq); rot_inertia(tirot, m.trunk_inertia, timat); }
  double xpos[3][3], xmat[3][9], xipos[3][3], irot[3][6], anchor[3][3], axis[3][3];
  {
    double ppos[3] = {txpos[0], txpos[1], txpos[2]}, pquat[4] = {txq[0], txq[1], txq[2], txq[3]}, pmat[9];
    QUNROLL for (int k = 0; k < 9; k++) pmat[k] = txm[k];
    QUNROLL for (int j = 0; j < 3; j++) {
      double xp[3], xquat[4];
      mv3(xp, pmat, L.body_pos[j]);
      QUNROLL for (int k = 0; k < 3; k++) xp[k] += ppos[k];
      q_mul(xquat, pquat, L.body_quat[j]);
      q_rot(anchor[j], L.jnt_pos[j], xquat);
      QUNROLL for (int k = 0; k < 3; k++) anchor[j][k] += xp[k];
      q_rot(axis[j], L.jnt_axis[j], xquat);
      const double angle = S.lq[j] - L.qpos0[j];
      double qloc[4] = {1, 0, 0, 0};
      if (angle != 0) {
        double sn, cs_;
        sincos(0.5 * angle, &sn, &cs_);
        qloc[0] = cs_; qloc[1] = L.jnt_axis[j][0] * sn; qloc[2] = L.jnt_axis[j][1] * sn; qloc[3] = L.jnt_axis[j][2] * sn;
      }
      q_mul(xquat, xquat, qloc);
      double vec[3];
      q_rot(vec, L.jnt_pos[j], xquat);
      QUNROLL for (int k = 0; k < 3; k++) xp[k] = anchor[j][k] - vec[k];
      q_norm(xquat);
      q2mat(pmat, xquat);
      QUNROLL for (int k = 0; k < 3; k++) { ppos[k] = xp[k]; xpos[j][k] = xp[k]; }
      QUNROLL for (int k = 0; k < 4; k++) pquat[k] = xquat[k];
      QUNROLL for (int k = 0; k < 9; k++) xmat[j][k] = pmat[k];
      mv3(tmp3, pmat, L.body_ipos[j]);
      QUNROLL for (int k = 0; k < 3; k++) xipos[j][k] = xp[k] + tmp3[k];
      double imat[9];
      q_mul(tmpq, xquat, L.body_iquat[j]);
      q2mat(imat, tmpq);
      rot_inertia(irot[j], L.body_inertia[j], imat);
    }
  }
  // ================= centre of mass, spatial inertias, dof axes (o_compos)
  double com[3];
  QUNROLL for (int k = 0; k < 3; k++) {
    const double s = L.body_mass[0] * xipos[0][k] + L.body_mass[1] * xipos[1][k] + L.body_mass[2] * xipos[2][k];
    com[k] = (qd_sum(s) + m.trunk_mass * txipos[k]) / m.total_mass;
  }
  {
    const double off[3] = {com[0] - txpos[0], com[1] - txpos[1], com[2] - txpos[2]};
    QUNROLL for (int k = 0; k < 3; k++) {
      kin.ca[k][0] = txm[k]; kin.ca[k][1] = txm[3 + k]; kin.ca[k][2] = txm[6 + k];
      cr3(kin.cl[k], kin.ca[k], off);
    }
    QUNROLL for (int j = 0; j < 3; j++) {
      const double o[3] = {com[0] - anchor[j][0], com[1] - anchor[j][1], com[2] - anchor[j][2]};
      QUNROLL for (int k = 0; k < 3; k++) kin.cdof[j][k] = axis[j][k];
      cr3(kin.cdof[j] + 3, axis[j], o);
    }
  }
  // (from here on the stages are ordered so that the big per-link arrays die early: velocities and the bias-acceleration recursion, the
  // sensor values, inertias -> bias forces -> M (the last readers of the inertias and bias accelerations), then collision, which only
  // needs the link frames and velocities: measured 58.4 -> 57.9 ms against collision first)
  // ================= velocities (o_comvel) and the acceleration recursion of o_rne (cdof_dot q-dot terms), link by link
  double cvel[3][6], cvelT[6], cacc[3][6], caccT[6];
  {
    double cv[6] = {0, 0, 0, S.tv[0], S.tv[1], S.tv[2]};
    double ca_[6] = {0, 0, 0, -m.gravity[0], -m.gravity[1], -m.gravity[2]};
    QUNROLL for (int k = 0; k < 3; k++) {  // the free joint's rotational dofs: cdof_dot = cvel (after the translations) x cdof
      const double cd[6] = {kin.ca[k][0], kin.ca[k][1], kin.ca[k][2], kin.cl[k][0], kin.cl[k][1], kin.cl[k][2]};
      double dd[6];
      cross_motion(dd, cv, cd);
      QUNROLL for (int c = 0; c < 6; c++) ca_[c] += dd[c] * S.tv[3 + k];
    }
    QUNROLL for (int k = 0; k < 3; k++) QUNROLL for (int c = 0; c < 3; c++) { cv[c] += kin.ca[k][c] * S.tv[3 + k]; cv[3 + c] += kin.cl[k][c] * S.tv[3 + k]; }
    QUNROLL for (int c = 0; c < 6; c++) { cvelT[c] = cv[c]; caccT[c] = ca_[c]; }
    QUNROLL for (int j = 0; j < 3; j++) {
      double dd[6];
      cross_motion(dd, cv, kin.cdof[j]);
      QUNROLL for (int c = 0; c < 6; c++) { ca_[c] += dd[c] * S.lv[j]; cacc[j][c] = ca_[c]; cv[c] += kin.cdof[j][c] * S.lv[j]; cvel[j][c] = cv[c]; }
    }
  }
  // ================= what the sensor stage reads
  QUNROLL for (int k = 0; k < 9; k++) out.txm[k] = txm[k];
  QUNROLL for (int k = 0; k < 4; k++) out.txq[k] = txq[k];
  QUNROLL for (int k = 0; k < 3; k++) { out.txipos[k] = txipos[k]; out.com[k] = com[k]; }
  mv3(tmp3, txm, m.head_pos);
  QUNROLL for (int k = 0; k < 3; k++) out.head[k] = txpos[k] + tmp3[k];
  QUNROLL for (int t = 0; t < kQMaxTrace; t++) { mv3(tmp3, txm, m.trace_pos[t]); QUNROLL for (int k = 0; k < 3; k++) out.trace[t][k] = txpos[k] + tmp3[k]; }  // (a fixed trip count: a run-time one would index out.trace at run time and put QSense in scratch)
  {  // subtree linear velocity of the trunk (o_subtree_linvel)
    double mom[3] = {0, 0, 0};
    QUNROLL for (int j = 0; j < 3; j++) {
      const double off[3] = {xipos[j][0] - com[0], xipos[j][1] - com[1], xipos[j][2] - com[2]};
      double lin[3];
      cr3(lin, cvel[j], off);
      QUNROLL for (int k = 0; k < 3; k++) mom[k] += L.body_mass[j] * (cvel[j][3 + k] + lin[k]);
    }
    const double off[3] = {txipos[0] - com[0], txipos[1] - com[1], txipos[2] - com[2]};
    double lin[3];
    cr3(lin, cvelT, off);
    QUNROLL for (int k = 0; k < 3; k++) out.comvel[k] = (qd_sum(mom[k]) + m.trunk_mass * (cvelT[3 + k] + lin[k])) / m.total_mass;
  }
  QPROF(pf, 1);
  // ================= spatial inertias about the centre of mass (o_compos); bias forces (o_rne), passive, actuation -> qfrc_smooth
  double cin[3][10], cinT[10];
  QUNROLL for (int j = 0; j < 3; j++) {
    const double dif[3] = {xipos[j][0] - com[0], xipos[j][1] - com[1], xipos[j][2] - com[2]};
    inert_shift(cin[j], irot[j], dif, L.body_mass[j]);
  }
  { const double dif[3] = {txipos[0] - com[0], txipos[1] - com[1], txipos[2] - com[2]}; inert_shift(cinT, tirot, dif, m.trunk_mass); }
  {
    double cfrc[3][6], cfrcT[6], t1[6], t2[6], t3[6];
    mul_inert(t1, cinT, caccT); mul_inert(t2, cinT, cvelT); cross_force(t3, cvelT, t2);
    QUNROLL for (int c = 0; c < 6; c++) cfrcT[c] = t1[c] + t3[c];
    QUNROLL for (int j = 0; j < 3; j++) {
      mul_inert(t1, cin[j], cacc[j]); mul_inert(t2, cin[j], cvel[j]); cross_force(t3, cvel[j], t2);
      QUNROLL for (int c = 0; c < 6; c++) cfrc[j][c] = t1[c] + t3[c];
    }
    QUNROLL for (int c = 0; c < 6; c++) { cfrc[1][c] += cfrc[2][c]; cfrc[0][c] += cfrc[1][c]; cfrcT[c] += qd_sum(cfrc[0][c]); }
    QUNROLL for (int j = 0; j < 3; j++) {
      const double bias = dot6(kin.cdof[j], cfrc[j]);
      double passive = -L.damping[j] * S.lv[j];
      if (L.stiffness[j] != 0) passive -= L.stiffness[j] * (S.lq[j] - L.qpos_spring[j]);
      double u = ctrl[j];
      if (L.ctrllimited[j]) u = clampd(u, L.ctrlrange[j][0], L.ctrlrange[j][1]);
      double force = L.act_gain[j] * u;
      if (L.act_biastype[j] == 1) force += L.act_bias[j][0] + L.act_bias[j][1] * L.act_gear[j] * S.lq[j] + L.act_bias[j][2] * L.act_gear[j] * S.lv[j];
      if (L.forcelimited[j]) force = clampd(force, L.forcerange[j][0], L.forcerange[j][1]);
      out.act_force[j] = force;
      D.fs_l[j] = passive - bias + L.act_gear[j] * force;
    }
    QUNROLL for (int k = 0; k < 6; k++) D.fs_t[k] = -trunk_dot(kin, k, cfrcT);
  }
  // ================= composite inertia (in place) -> M (o_crb), in arrowhead form; M goes to the store (the solver and the integrator read
  // it there); its factor gives qacc_smooth and is dropped
  {
    Arrow M;
    QUNROLL for (int e = 0; e < 10; e++) { cin[1][e] += cin[2][e]; cin[0][e] += cin[1][e]; }
    QUNROLL for (int e = 0; e < 10; e++) cinT[e] += qd_sum(cin[0][e]);
    QUNROLL for (int j = 0; j < 3; j++) {
      double buf[6];
      mul_inert(buf, cin[j], kin.cdof[j]);
      M.l[tri(j, j)] = L.armature[j] + dot6(kin.cdof[j], buf);
      QUNROLL for (int i = 0; i < j; i++) M.l[tri(j, i)] = dot6(kin.cdof[i], buf);
      QUNROLL for (int k = 0; k < 6; k++) M.b[j][k] = trunk_dot(kin, k, buf);
    }
    QUNROLL for (int k = 0; k < 6; k++) {
      double cd[6], buf[6];
      if (k < 3) { QUNROLL for (int c = 0; c < 6; c++) cd[c] = 0; cd[3 + k] = 1; }
      else { QUNROLL for (int c = 0; c < 3; c++) { cd[c] = kin.ca[k - 3][c]; cd[3 + c] = kin.cl[k - 3][c]; } }
      mul_inert(buf, cinT, cd);
      QUNROLL for (int i = 0; i <= k; i++) M.t[tri(k, i)] = trunk_dot(kin, i, buf);
    }
    QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c < 3; c++) M.ab[r][c] = 0;
    store_arrow(ms, M);
    if (!arrow_factor(M, leg, 0)) flags |= kFlagNotPD;
    QUNROLL for (int j = 0; j < 3; j++) D.sl[j] = D.fs_l[j];
    QUNROLL for (int k = 0; k < 6; k++) D.st[k] = D.fs_t[k];
    arrow_solve(M, D.sl, D.st, leg, 0);
  }
  // ================= collision (o_collision): the leg's geoms and the lane's share of the trunk geoms; the self-collision test
  int ncon = 0;
  {
    const int near_leg = static_near_mask(m, sp, xpos[0], L.reach), near_trunk = static_near_mask(m, sp, txpos, m.trunk_reach);
    QPairArgs pargs;  // (the self-collision tests' argument block: in memory -- they are out of line; the loop below writes the pair geoms' poses into it)
    QPairBoxes pbox;
    pair_boxes_init(pbox);
    for (int i = 0; i < kQPairGeom; i++) for (int k = 0; k < 3; k++) { pargs.pg.c[i][k] = 0; pargs.pg.a[i][k] = 0; }
    for (int gi = 0; gi < L.ngeom; gi++) {
      const QuadGeom& g = L.geom[gi];
      double gp[3], gR[9];
      const int lk = g.link;
      double bm[9], bp[3], bv[6];
      QUNROLL for (int k = 0; k < 9; k++) bm[k] = lk == 0 ? xmat[0][k] : (lk == 1 ? xmat[1][k] : xmat[2][k]);
      QUNROLL for (int k = 0; k < 3; k++) bp[k] = lk == 0 ? xpos[0][k] : (lk == 1 ? xpos[1][k] : xpos[2][k]);
      QUNROLL for (int k = 0; k < 6; k++) bv[k] = lk == 0 ? cvel[0][k] : (lk == 1 ? cvel[1][k] : cvel[2][k]);
      mv3(gp, bm, g.pos);
      QUNROLL for (int k = 0; k < 3; k++) gp[k] += bp[k];
      QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c < 3; c++) gR[3 * r + c] = bm[3 * r] * g.rot[c] + bm[3 * r + 1] * g.rot[3 + c] + bm[3 * r + 2] * g.rot[6 + c];
      if (gi == L.foot_slot) { QUNROLL for (int k = 0; k < 3; k++) out.foot[k] = gp[k]; }
      if (g.pgi >= 0) {  // a pair geom: its pose to the tests' argument block (memory, run-time slot), its box into the cull's
        const double ga[3] = {gR[2], gR[5], gR[8]};
        for (int k = 0; k < 3; k++) { pargs.pg.c[g.pgi][k] = gp[k]; pargs.pg.a[g.pgi][k] = ga[k]; }
        pair_boxes_add(L, g.pgi, gp, ga, txpos, txm, pbox);
      }
      collide_geom(m, sp, g, near_leg, &tab.leg[leg][0][gi], kQLegGeom, com, bv, lk + 1, gp, gR, cs, ncon, flags);
    }
    for (int gi = leg; gi < m.ntrunk_geom; gi += kQLegs) {
      const QuadGeom& g = m.trunk_geom[gi];
      double gp[3], gR[9];
      mv3(gp, txm, g.pos);
      QUNROLL for (int k = 0; k < 3; k++) gp[k] += txpos[k];
      QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c < 3; c++) gR[3 * r + c] = txm[3 * r] * g.rot[c] + txm[3 * r + 1] * g.rot[3 + c] + txm[3 * r + 2] * g.rot[6 + c];
      collide_geom(m, sp, g, near_trunk, &tab.trunk[0][gi], kQTrunkGeom, com, cvelT, 0, gp, gR, cs, ncon, flags);
    }
    QPROF(pf, 2);
    int pmask = 0, nrel = 0;
#ifndef QEXP_NOPAIRS
    // (the leg's own pairs -- a calf or foot on its hip -- are proven apart inside a joint box: pair_cull.h)
    const bool self_walk = !(S.lq[0] >= L.self_box[0][0] && S.lq[0] <= L.self_box[0][1] && S.lq[1] >= L.self_box[1][0] && S.lq[1] <= L.self_box[1][1] &&
                             S.lq[2] >= L.self_box[2][0] && S.lq[2] <= L.self_box[2][1]);
    pair_contacts(m, tab, leg, pargs, pbox, txpos, txm, com, cvel, cvelT, self_walk, cs, ncon, flags, pmask, nrel, pf);
#endif
    D.ncon = ncon; D.nrel = nrel;
    // bit x of pmask: some leg A touches leg A xor x. One bit set (the common case of self-collision) means disjoint pairs, which the
    // arrowhead factorisation takes as super-legs; two or three (a leg touching two others) go through the dense elimination of the
    // leg blocks (newton_direction_general)
    D.pmask = qd_or(pmask);
    D.mymask = pmask;
    D.have_rel = qd_or(nrel > 0 ? 1 : 0);
    QPROF(pf, 3);
  }
  // ================= constraint rows of the lane: friction loss, joint limits (o_make_constraint_full); contacts are in the store
  QRows& R = D.R;
  QUNROLL for (int j = 0; j < 3; j++) {
    R.fl_jar[j] = L.floss_b[j] * S.lv[j];  // -aref
    R.lm_side[j] = 0; R.lm_D[j] = 0; R.lm_jar[j] = 0;
    // (geom pairs this kernel does not walk were proven apart for joints inside `guard`: beyond it the candidate goes to the kernel that walks them)
    if (S.lq[j] < L.guard[j][0] || S.lq[j] > L.guard[j][1]) flags |= kFlagRange;
    if (L.limited[j]) {
      const double dlo = S.lq[j] - L.range[j][0], dhi = L.range[j][1] - S.lq[j];
      int side = 0; double dist = 0;
      if (dlo < L.margin[j]) { side = -1; dist = dlo; }
      if (dhi < L.margin[j]) { if (side != 0) flags |= kFlagLimits; side = 1; dist = dhi; }
      if (side != 0) {
        const double pos = dist - L.margin[j], imp = impedance(L.lim_imp[j], pos), vel = -side * S.lv[j];
        double Rr = (1 - imp) / imp * L.lim_diag[j];
        if (Rr < kQMinVal) Rr = kQMinVal;
        R.lm_side[j] = side; R.lm_D[j] = 1.0 / Rr; R.lm_jar[j] = L.lim_b[j] * vel + L.lim_k[j] * imp * pos;  // -aref
      }
    }
  }
  QPROF(pf, 4);
  return qd_or(flags);
}

// ---------------------------------------------------------------- mj_Euler with implicit joint damping + mj_advance (oracle o_euler)
// (al, at) = the solver's qacc, (fc_l, fc_t) = qfrc_constraint; M from the store
template <class MS>
QD void euler(const QuadModel& m, int leg, QState& S, const QDyn& D, const MS& ms, const double* al, const double* at, const double* fc_l, const double* fc_t) {
  const QuadLeg& L = m.leg[leg];
  const double h = m.timestep;
  double ql[3], qt[6];
  QUNROLL for (int j = 0; j < 3; j++) ql[j] = al[j];
  QUNROLL for (int k = 0; k < 6; k++) qt[k] = at[k];
  const bool damped = qd_or((L.damping[0] > 0 || L.damping[1] > 0 || L.damping[2] > 0) ? 1 : 0) != 0;
  if (damped) {
    Arrow A;
    load_arrow(ms, A);
    QUNROLL for (int j = 0; j < 3; j++) A.l[tri(j, j)] += h * L.damping[j];
    if (arrow_factor(A, leg, 0)) {
      QUNROLL for (int j = 0; j < 3; j++) ql[j] = D.fs_l[j] + fc_l[j];
      QUNROLL for (int k = 0; k < 6; k++) qt[k] = D.fs_t[k] + fc_t[k];
      arrow_solve(A, ql, qt, leg, 0);
    }
  }
  QUNROLL for (int j = 0; j < 3; j++) { S.wl[j] = al[j]; S.lv[j] += h * ql[j]; S.lq[j] += h * S.lv[j]; }
  QUNROLL for (int k = 0; k < 6; k++) { S.wt[k] = at[k]; S.tv[k] += h * qt[k]; }
  QUNROLL for (int k = 0; k < 3; k++) S.tq[k] += h * S.tv[k];
  {
    double ax[3] = {S.tv[3], S.tv[4], S.tv[5]};
    const double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (n < kQMinVal) { ax[0] = 1; ax[1] = ax[2] = 0; } else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
    const double angle = h * n;
    double qrot[4] = {1, 0, 0, 0};
    if (angle != 0) { double sn, cs_; sincos(0.5 * angle, &sn, &cs_); qrot[0] = cs_; qrot[1] = ax[0] * sn; qrot[2] = ax[1] * sn; qrot[3] = ax[2] * sn; }
    q_norm(S.tq + 3);
    q_mul(S.tq + 3, S.tq + 3, qrot);
  }
  S.time += h;
}

// ---------------------------------------------------------------- QuadrupedFlat residual (oracle/quadruped.inc) + cost (task.cc:71-110)
// mj_ray straight down against the static geoms of group 0 (oracle ray_down)
QD double ray_down(const QuadModel& m, const QStaticPose* sp, const double* from) {
  double best = -1;
  for (int s = 0; s < m.nstatic; s++) {
    const QuadStatic& S = m.stat[s];
    if (!S.ray) continue;
    const int type = S.type < 0 ? -1 - S.type : S.type;
    const double* p = sp[s].pos; const double* R = sp[s].mat; const double* sz = S.size;
    double x = -1;
    if (type == MJPCX_GEOM_PLANE) {
      const double n[3] = {R[2], R[5], R[8]};
      const double denom = -n[2];
      if (fabs(denom) >= kQMinVal) {
        const double t = -((from[0] - p[0]) * n[0] + (from[1] - p[1]) * n[1] + (from[2] - p[2]) * n[2]) / denom;
        if (t >= 0) {
          const double hit[3] = {from[0] - p[0], from[1] - p[1], from[2] - t - p[2]};
          const double lx = R[0] * hit[0] + R[3] * hit[1] + R[6] * hit[2], ly = R[1] * hit[0] + R[4] * hit[1] + R[7] * hit[2];
          if ((sz[0] <= 0 || fabs(lx) <= sz[0]) && (sz[1] <= 0 || fabs(ly) <= sz[1])) x = t;
        }
      }
    } else if (type == MJPCX_GEOM_SPHERE) {
      const double o[3] = {from[0] - p[0], from[1] - p[1], from[2] - p[2]};
      const double b = -o[2], c = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - sz[0] * sz[0];
      const double disc = b * b - c;
      if (disc >= 0) { const double sq = sqrt(disc), t0 = -b - sq, t1 = -b + sq; x = t0 >= 0 ? t0 : (t1 >= 0 ? t1 : -1); }
    } else if (type == MJPCX_GEOM_BOX) {
      double o[3], dl[3];
      const double rel[3] = {from[0] - p[0], from[1] - p[1], from[2] - p[2]};
      QUNROLL for (int k = 0; k < 3; k++) { o[k] = R[k] * rel[0] + R[3 + k] * rel[1] + R[6 + k] * rel[2]; dl[k] = -R[6 + k]; }
      double tmin = -1e300, tmax = 1e300;
      bool miss = false;
      QUNROLL for (int k = 0; k < 3; k++) {
        if (fabs(dl[k]) < kQMinVal) { if (fabs(o[k]) > sz[k]) miss = true; continue; }
        double ta = (-sz[k] - o[k]) / dl[k], tb = (sz[k] - o[k]) / dl[k];
        if (ta > tb) { const double tt = ta; ta = tb; tb = tt; }
        if (ta > tmin) tmin = ta;
        if (tb < tmax) tmax = tb;
      }
      if (!(miss || tmin > tmax || tmax < 0)) x = tmin >= 0 ? tmin : tmax;
    }
    if (x >= 0 && (best < 0 || x < best)) best = x;
  }
  return best;
}
QD double step_height(double time, double footphase, double duty_ratio) {
  double angle = fmod(time + kQPi - footphase, 2 * kQPi) - kQPi;
  double value = 0;
  if (duty_ratio < 1) {
    angle *= 0.5 / (1 - duty_ratio);
    value = cos(clampd(angle, -kQPi / 2, kQPi / 2));
  }
  return fabs(value) < 1e-6 ? 0.0 : value;
}
QD void flip_quat(const double* re, int flip_dir, double* quat, double time) {
  const double jump_time = re[22], flight_time = re[18], land_time = re[24], crouch_time = re[20];
  double angle = 0;
  if (time >= jump_time + flight_time + land_time) angle = 2 * kQPi;
  else if (time >= crouch_time && time < jump_time) { time -= crouch_time; angle = 0.5 * re[28] * time * time + re[27] * time; }
  else if (time >= jump_time && time < jump_time + flight_time) { time -= jump_time; angle = kQPi / 2 + re[26] * time; }
  else if (time >= jump_time + flight_time) { time -= jump_time + flight_time; angle = 1.75 * kQPi + re[26] * time - 0.5 * re[29] * time * time; }
  double q[4] = {1, 0, 0, 0};
  if (angle != 0) { double s, c; sincos(0.5 * angle, &s, &c); q[0] = c; q[1] = 0; q[2] = (flip_dir ? 1.0 : -1.0) * s; q[3] = 0; }
  q_mul(quat, re + 9, q);
}
QD double flip_height(const double* re, double time) {
  const double jump_time = re[22], flight_time = re[18], land_time = re[24], ground = re[8];
  if (time >= jump_time + flight_time + land_time) return 0.25 + ground;
  double h = 0;
  if (time < jump_time) h = 0.25 + time * re[23] + 0.5 * time * time * re[19];
  else if (time >= jump_time && time < jump_time + flight_time) { time -= jump_time; h = 0.5 + re[17] * time - 0.5 * 9.81 * time * time; }
  else if (time >= jump_time + flight_time) { time -= jump_time + flight_time; h = 0.5 - re[17] * time + 0.5 * re[25] * time * time; }
  return h + ground;
}
QD void sub_quat(double* res, const double* qa, const double* qb) {
  const double qn[4] = {qb[0], -qb[1], -qb[2], -qb[3]};
  double qdif[4];
  q_mul(qdif, qn, qa);
  double axis[3] = {qdif[1], qdif[2], qdif[3]};
  const double sin_a_2 = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  if (sin_a_2 > kQMinVal) QUNROLL for (int k = 0; k < 3; k++) axis[k] /= sin_a_2;
  double speed = 2 * atan2(sin_a_2, qdif[0]);
  if (speed > kQPi) speed -= 2 * kQPi;
  QUNROLL for (int k = 0; k < 3; k++) res[k] = axis[k] * speed;
}
QD double norm_elem(double x, int type, double p, double q) {
  switch (type) {
    case -1: return x;
    case 0: case 1: case 2: return x * x;
    case 3: return p * p * (cosh(x / p) - 1.0);
    case 5: return pow(fabs(x), p);
    case 6: return sqrt(x * x + p * p) - p;
    case 7: return pow(pow(fabs(x), q) + pow(p, q), 1 / q) - p;
    case 8: return p > 0 ? p * log(1 + exp(x / p)) : (x > 0 ? x : 0.0);
    default: return 0;
  }
}
QD double norm_finish(double c, int type, double p, double q) {
  switch (type) {
    case 0: return c * 0.5;
    case 1: return pow(pow(c, q / 2) + pow(p, q), 1 / q) - p;
    case 2: return sqrt(c + p * p) - p;
    default: return c;
  }
}

// per-plan task values (the blob WaveHost::fill_blob stages: wave_model.h)
struct QTask {
  const double *mocap, *weight, *norm_p, *norm_q, *param, *re;  // mocap[7 nmocap] ...
  const int* ri;
  double risk;
};

// The residual entries are dealt over the quad: `shared` = the 18 entries of Upright Height Position Balance Yaw Angmom + the four
// Gait entries (every lane computes the shared ones identically, lane `foot_index` owns Gait entry foot_index), `own` = the
// lane's three Effort and three Posture entries. Returns the cost (task.cc:71-110 + the risk transform), replicated.
struct QResidual { double shared[18]; double gait; double effort[3], posture[3]; };
QD double residual_cost(const QuadModel& m, const QTask& tk, const QStaticPose* sp, int leg, const QState& S, const QSense& f, QResidual& r) {
  const QuadLeg& L = m.leg[leg];
  const int* ri = tk.ri; const double* re = tk.re; const double* par = tk.param;
  const int mode = ri[0], handstand = ri[10], fi = L.foot_index;
  const double kGaitPhase[5][4] = {{0, 0, 0, 0}, {0, 0.75, 0.5, 0.25}, {0, 0.5, 0.5, 0}, {0, 0.33, 0.33, 0.66}, {0, 0.4, 0.05, 0.35}};
  // foot positions of the four legs in the reference's order FL HL FR HR
  double fp[4][3];
  {
    double mine[4][3];
    QUNROLL for (int q = 0; q < 4; q++) QUNROLL for (int k = 0; k < 3; k++) mine[q][k] = q == fi ? f.foot[k] : 0.0;
    QUNROLL for (int q = 0; q < 4; q++) QUNROLL for (int k = 0; k < 3; k++) fp[q][k] = qd_sum(mine[q][k]);
  }
  const bool is_biped = mode == 1;
  double avg[3];
  if (is_biped) { const int a = handstand ? 0 : 1, b = handstand ? 2 : 3; QUNROLL for (int k = 0; k < 3; k++) avg[k] = 0.5 * (fp[a][k] + fp[b][k]); }
  else QUNROLL for (int k = 0; k < 3; k++) avg[k] = 0.25 * (fp[1][k] + fp[3][k] + fp[0][k] + fp[2][k]);
  const double* goal = tk.mocap + 7 * m.goal_mocap;
  double* R = r.shared;  // [0..2] Upright [3] Height [4..6] Position [7..8] Balance [9..10] Yaw [11..13] Angmom
  // Upright
  if (mode != 4) {
    R[0] = is_biped ? f.txm[6] - (handstand ? -1 : 1) : f.txm[8] - 1;
    R[1] = 0; R[2] = 0;
  } else {
    double quat[4], up[3];
    flip_quat(re, ri[9], quat, S.time - re[0]);
    sub_quat(up, f.txq, quat);
    R[0] = up[0]; R[1] = up[1]; R[2] = up[2];
  }
  // Height
  const double height_goal = is_biped ? 0.6 : 0.25;
  if (mode == 3) R[3] = 0;
  else if (mode == 4) R[3] = f.txipos[2] - flip_height(re, S.time - re[0]);
  else R[3] = (f.txipos[2] - avg[2]) - height_goal;
  // Position
  double target[3];
  if (mode == 2) {
    const double t = S.time - re[0];
    const double* position = re + 1; const double* heading = re + 4;
    const double speed = re[6], angvel = re[7];
    if (fabs(angvel) < 0.01) {
      double fwd[2] = {heading[0], heading[1]};
      const double nn = sqrt(fwd[0] * fwd[0] + fwd[1] * fwd[1]);
      if (nn > kQMinVal) { fwd[0] /= nn; fwd[1] /= nn; } else { fwd[0] = 1; fwd[1] = 0; }
      target[0] = position[0] + heading[0] + t * speed * fwd[0];
      target[1] = position[1] + heading[1] + t * speed * fwd[1];
    } else {
      const double angle = t * angvel, c = cos(angle), s = sin(angle);
      target[0] = c * heading[0] - s * heading[1] + position[0];
      target[1] = s * heading[0] + c * heading[1] + position[1];
    }
    target[2] = 0;
  } else { target[0] = goal[0]; target[1] = goal[1]; target[2] = goal[2]; }
  R[4] = f.head[0] - target[0];
  R[5] = f.head[1] - target[1];
  R[6] = mode == 3 ? 2 * (f.head[2] - target[2]) : 0;
  // Gait: the lane's own foot
  {
    const int gait = is_biped ? 2 : ri[8];
    const double phase = re[13] + (S.time - re[14]) * re[15];
    const double amplitude = par[ri[11]], duty_ratio = par[ri[12]];
    const double step = amplitude * step_height(phase, 2 * kQPi * kGaitPhase[gait][fi], duty_ratio);
    const bool front_hand = is_biped && !handstand && (fi == 0 || fi == 2), back_hand = is_biped && handstand && (fi == 1 || fi == 3);
    if (front_hand || back_hand) r.gait = 0;
    else {
      double query[3] = {f.foot[0], f.foot[1], f.foot[2]};
      if (mode == 3) {
        double v[3] = {goal[0] - f.foot[0], goal[1] - f.foot[1], 0};
        const double nn = sqrt(v[0] * v[0] + v[1] * v[1]);
        if (nn > kQMinVal) { v[0] /= nn; v[1] /= nn; } else { v[0] = 1; v[1] = 0; }
        QUNROLL for (int k = 0; k < 3; k++) query[k] += 0.15 * v[k];
      }
      const double q3[3] = {query[0], query[1], query[2] + 0.5};
      const double ground = query[2] + 0.5 - ray_down(m, sp, q3);
      double hd = f.foot[2] - (ground + 0.02 + step);
      if (mode == 3) hd = hd < 0 ? hd : 0;
      r.gait = step ? hd : 0;
    }
  }
  // Balance
  const double fall_time = sqrt(2 * height_goal / 9.81);
  R[7] = f.com[0] + f.comvel[0] * fall_time - avg[0];
  R[8] = f.com[1] + f.comvel[1] * fall_time - avg[1];
  // Effort, Posture: the lane's joints
  QUNROLL for (int j = 0; j < 3; j++) {
    r.effort[j] = 2e-2 * f.act_force[j];
    double p = S.lq[j] - L.key_q[ri[15]][j];
    if (mode == 4) {
      const double ft = S.time - re[0];
      if (ft < re[20]) p = S.lq[j] - L.key_q[ri[16]][j];
      else if (ft >= re[20] && ft < re[22] + re[18]) p = 0;
    }
    p *= j == 0 ? 2.0 : 1.0;
    // actuators (= legs) 0..5 / 6..11 are the reference's "arm" halves of the Posture term
    if (is_biped) { const int base = handstand ? 6 : 0; const int u = 3 * leg + j; if (u >= base && u < base + 6) p *= par[ri[13]]; }
    r.posture[j] = p;
  }
  // Yaw
  double th[2] = {f.txm[0], f.txm[3]};
  if (is_biped) { const int hs = handstand ? 1 : -1; th[0] = hs * f.txm[2]; th[1] = hs * f.txm[5]; }
  const double nn = sqrt(th[0] * th[0] + th[1] * th[1]);
  if (nn < kQMinVal) { th[0] = 1; th[1] = 0; } else { th[0] /= nn; th[1] /= nn; }
  const double heading_goal = par[ri[14]];
  R[9] = th[0] - cos(heading_goal);
  R[10] = th[1] - sin(heading_goal);
  QUNROLL for (int k = 0; k < 3; k++) R[11 + k] = f.comvel[k];
  // ---- cost: terms Upright(3) Height(1) Position(3) Gait(4) Balance(2) Effort(12) Posture(12) Yaw(2) Angmom(3)
  // (the entries by value, at most three per term: a pointer + run-time count would index the residual at run time and put it in scratch)
  auto term_sum = [&](int term, int cnt, double x0, double x1, double x2) {
    double c = norm_elem(x0, m.term_norm[term], tk.norm_p[term], tk.norm_q[term]);
    if (cnt > 1) c += norm_elem(x1, m.term_norm[term], tk.norm_p[term], tk.norm_q[term]);
    if (cnt > 2) c += norm_elem(x2, m.term_norm[term], tk.norm_p[term], tk.norm_q[term]);
    return c;
  };
  auto term_shared = [&](int term, int cnt, double x0, double x1, double x2) {
    return tk.weight[term] * norm_finish(term_sum(term, cnt, x0, x1, x2), m.term_norm[term], tk.norm_p[term], tk.norm_q[term]);
  };
  auto term_dealt = [&](int term, int cnt, double x0, double x1, double x2) {
    return tk.weight[term] * norm_finish(qd_sum(term_sum(term, cnt, x0, x1, x2)), m.term_norm[term], tk.norm_p[term], tk.norm_q[term]);
  };
  double cost = 0;
  cost += term_shared(0, 3, R[0], R[1], R[2]);
  cost += term_shared(1, 1, R[3], 0, 0);
  cost += term_shared(2, 3, R[4], R[5], R[6]);
  cost += term_dealt(3, 1, r.gait, 0, 0);
  cost += term_shared(4, 2, R[7], R[8], 0);
  cost += term_dealt(5, 3, r.effort[0], r.effort[1], r.effort[2]);
  cost += term_dealt(6, 3, r.posture[0], r.posture[1], r.posture[2]);
  cost += term_shared(7, 2, R[9], R[10], 0);
  cost += term_shared(8, 3, R[11], R[12], R[13]);
  if (!(fabs(tk.risk) < 1.0e-6)) cost = (exp(tk.risk * cost) - 1.0) / tk.risk;
  return cost;
}


// ---------------------------------------------------------------- candidate generation + Trajectory::Rollout
// Philox4x32-10 + Box-Muller exactly as include/mjpcx.h specifies (device_common.h gaussian_pair; oracle/rng.c)
QD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  QUNROLL for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
QD double u53(uint32_t hi, uint32_t lo) {
  const uint64_t k = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)k + 0.5) * (1.0 / 9007199254740992.0);
}
QD void gaussian_pair(uint64_t seed, uint32_t cand, uint32_t pair, uint32_t iter, double* z) {
  uint32_t o[4];
  philox4x32_10(cand, pair, iter, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  const double u1 = u53(o[0], o[1]), u2 = u53(o[2], o[3]);
  const double r = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
  z[0] = r * cs; z[1] = r * sn;
}
QD double bernoulli_uniform(uint64_t seed, uint32_t cand, uint32_t iter) {
  uint32_t o[4];
  philox4x32_10(cand, 0u, iter, 1u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return u53(o[0], o[1]);
}

// ---------------------------------------------------------------- the iLQG feedback policy in quad form
// FindInterval / the interpolation weights of Zero / Linear / CubicInterpolation (mjpc/utilities.h:124-144, utilities.cc:304-422) on a
// quad-uniform query: the arithmetic of ilqg_kernels.h interp_weights (the wavefront-per-candidate kernels' and the oracle's)
struct QInterp { int i[4]; double w[4]; };
QD void q_find_interval(const double* xs, double value, int length, int& b0, int& b1) {
  int up = 0;
  while (up < length && xs[up] <= value) up++;
  const int lo = up - 1;
  if (lo < 0) { b0 = b1 = 0; }
  else if (lo > length - 1) { b0 = b1 = length - 1; }
  else { b0 = lo; b1 = up < length - 1 ? up : length - 1; }
}
QD QInterp q_interp_weights(const double* xs, double value, int length, int representation) {
  QInterp r;
  int b0, b1;
  q_find_interval(xs, value, length, b0, b1);
  const int last = length - 1;
  r.i[0] = b0 > 0 ? b0 - 1 : 0; r.i[1] = b0; r.i[2] = b0 < last ? b0 + 1 : last; r.i[3] = b0 + 2 <= last ? b0 + 2 : last;
  r.w[0] = r.w[2] = r.w[3] = 0; r.w[1] = 1;
  if (b0 == b1 || representation == 0) return r;
  const double span = xs[b1] - xs[b0], t = (value - xs[b0]) / span;
  if (representation != 2) { r.w[1] = 1.0 - t; r.w[2] = t; return r; }
  const double t2 = t * t, t3 = t2 * t;
  const double c0 = 2.0 * t3 - 3.0 * t2 + 1.0, c1 = (t3 - 2.0 * t2 + t) * span, c2 = -2.0 * t3 + 3.0 * t2, c3 = (t3 - t2) * span;
  double m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};
  const double is1 = 1.0 / span;
  if (b0 == 0) { m0[1] = -is1; m0[2] = is1; }
  else { const double isl = 1.0 / (xs[b0] - xs[b0 - 1]); m0[0] = -0.5 * isl; m0[1] = 0.5 * isl - 0.5 * is1; m0[2] = 0.5 * is1; }
  if (b1 == last) { if (length > 2) { m1[1] = -is1; m1[2] = is1; } }
  else { const double isr = 1.0 / (xs[b1 + 1] - xs[b1]); m1[1] = -0.5 * is1; m1[2] = 0.5 * is1 - 0.5 * isr; m1[3] = 0.5 * isr; }
  r.w[0] = c1 * m0[0];
  r.w[1] = c0 + c1 * m0[1] + c3 * m1[1];
  r.w[2] = c2 + c1 * m0[2] + c3 * m1[2];
  r.w[3] = c3 * m1[3];
  return r;
}
// StateDiff(ref, x) in the tangent space (mj_differentiatePos, utilities.cc:543-553; wave_ilqg.h w_state_diff): all 36 entries in every
// lane -- [trunk position 3, rotation 3, the legs' joints 12 | trunk velocity 6, the legs' 12] -- the trunk's computed redundantly, a
// leg's six broadcast from its lane. The reference's quaternion is used as it is (the caller normalises an interpolated one).
// the entries of a state row a lane reads: the trunk's 13 and its own leg's 6
struct QRefState { double tpos[3], tquat[4], tvel[6], lq[3], lv[3]; };
QD void q_state_diff(const QRefState& ref, const QState& S, double* dx) {
  QUNROLL for (int k = 0; k < 3; k++) dx[k] = S.tq[k] - ref.tpos[k];
  sub_quat(dx + 3, S.tq + 3, ref.tquat);
  QUNROLL for (int k = 0; k < 6; k++) dx[18 + k] = S.tv[k] - ref.tvel[k];
  double mine[6];
  QUNROLL for (int j = 0; j < 3; j++) { mine[j] = S.lq[j] - ref.lq[j]; mine[3 + j] = S.lv[j] - ref.lv[j]; }
  QUNROLL for (int j = 0; j < 3; j++) {
    dx[6 + j] = qd_bcast<0>(mine[j]); dx[9 + j] = qd_bcast<1>(mine[j]); dx[12 + j] = qd_bcast<2>(mine[j]); dx[15 + j] = qd_bcast<3>(mine[j]);
    dx[24 + j] = qd_bcast<0>(mine[3 + j]); dx[27 + j] = qd_bcast<1>(mine[3 + j]); dx[30 + j] = qd_bcast<2>(mine[3 + j]); dx[33 + j] = qd_bcast<3>(mine[3 + j]);
  }
}
// the lane's three controls under the feedback policy at step t / time S.time (before the clamp)
QD void feedback_ctrl(const QFeedback& fb, double alpha, int t, const QState& S, int leg, double* u) {
  constexpr int nu = kQLegs * kQLinks, ds = 37, ndx = 36;
  double dx[ndx];
  // (every loop over the 36 gains of a row is unrolled: its loads are then all in flight together -- one memory latency per row instead of
  // one per entry on a path that is nothing but latency)
  if (fb.mode == 0) {  // index policy: u = actions[t] + alpha improvement[t] + K[t] StateDiff(states[t], x)
    const int tt = t < fb.Tn ? t : fb.Tn - 1;
    const double* row = fb.states + (size_t)tt * ds;
    QRefState ref;
    QUNROLL for (int k = 0; k < 3; k++) ref.tpos[k] = row[k];
    QUNROLL for (int k = 0; k < 4; k++) ref.tquat[k] = row[3 + k];
    QUNROLL for (int k = 0; k < 6; k++) ref.tvel[k] = row[19 + k];
    QUNROLL for (int j = 0; j < 3; j++) { ref.lq[j] = row[7 + 3 * leg + j]; ref.lv[j] = row[25 + 3 * leg + j]; }
    q_state_diff(ref, S, dx);
    QUNROLL for (int e = 0; e < 3; e++) {
      const int r = tt * nu + 3 * leg + e;
      const double* K = fb.gains + (size_t)r * ndx;
      double s = 0;
      QUNROLL for (int j = 0; j < ndx; j++) s += K[j] * dx[j];
      u[e] = fb.actions[r] + alpha * fb.improvement[r] + s;
    }
    return;
  }
  // iLQGPolicy::Action at the rollout's time: actions / gains over the Tn - 1 entries the reference passes, states over Tn
  const double now = QUNIFORM_TIME(S.time);
  int b0, b1;
  q_find_interval(fb.times, now, fb.Tn, b0, b1);
  const int rep = (b0 == b1) ? 0 : fb.representation;
  const QInterp wa = q_interp_weights(fb.times, now, fb.Tn - 1, rep);
  QUNROLL for (int e = 0; e < 3; e++) {
    double v = 0;
    QUNROLL for (int p = 0; p < 4; p++) v += wa.w[p] * fb.actions[(size_t)wa.i[p] * nu + 3 * leg + e];
    u[e] = v;
  }
  if (!fb.use_state) return;
  const QInterp ws = q_interp_weights(fb.times, now, fb.Tn, rep);
  auto interp_state = [&](int i) {
    double v = 0;
    QUNROLL for (int p = 0; p < 4; p++) v += ws.w[p] * fb.states[(size_t)ws.i[p] * ds + i];
    return v;
  };
  QRefState ref;
  QUNROLL for (int k = 0; k < 3; k++) ref.tpos[k] = interp_state(k);
  QUNROLL for (int k = 0; k < 4; k++) ref.tquat[k] = interp_state(3 + k);
  QUNROLL for (int k = 0; k < 6; k++) ref.tvel[k] = interp_state(19 + k);
  QUNROLL for (int j = 0; j < 3; j++) { ref.lq[j] = interp_state(7 + 3 * leg + j); ref.lv[j] = interp_state(25 + 3 * leg + j); }
  q_norm(ref.tquat);  // (policy.cc:118-125: interpolated quaternions are renormalised)
  q_state_diff(ref, S, dx);
  QUNROLL for (int e = 0; e < 3; e++) {
    double s = 0;
    QUNROLL for (int p = 0; p < 4; p++) {
      if (wa.w[p] == 0) continue;  // (quad-uniform: zero-order and linear policies read one or two gain rows, not four)
      const double* K = fb.gains + ((size_t)wa.i[p] * nu + 3 * leg + e) * ndx;
      double sp = 0;
      QUNROLL for (int j = 0; j < ndx; j++) sp += K[j] * dx[j];
      s += wa.w[p] * sp;
    }
    u[e] += alpha * s;
  }
}

// One lane's share of one candidate's rollout. `state0` = qpos[19] qvel[18] of the plan (Planner::SetState), `con` the lane's contact
// list storage (kQMaxCon records). Returns the flag bits (0: rolled out; otherwise failure[cand] carries kQFallback and the
// wavefront-per-candidate kernel takes the candidate over).
// FEEDBACK: the candidate follows the iLQG feedback policy `fb` (Trajectory::RolloutDiscrete / Rollout with iLQGPolicy::Action) instead of a spline
template <bool FEEDBACK, class CS, class MS, class QProfT>
QD int rollout(const QuadModel& m, const QuadTables& tab, const QStaticPose* sp, const QTask& tk, const double* state0, double time0, const QArgs& a,
               const QFeedback& fb, int cand, int leg, CS& cs, MS& ms, QProfT& pf) {
  const QuadLeg& L = m.leg[leg];
  const int nu = kQLegs * kQLinks, P = a.P, H = a.H, nr = m.nr;
  const size_t N = (size_t)a.N;
  const double fb_alpha = FEEDBACK ? fb.alpha[cand] : 0.0;
  // ---- the candidate's spline nodes of this leg's three actuators (AddNoiseToPolicy)
  if (!FEEDBACK && a.noise_mode >= 0) {
    const int gi = a.candidate_offset + cand;
    double std = a.std0;
    if (a.noise_mode == 0 && a.std1 > 0) { if (bernoulli_uniform(a.seed, (uint32_t)gi, a.iteration) < 0.2) std = a.std1; }
    const bool noised = gi != a.nominal_candidate;
    for (int p = 0; p < P; p++)
      QUNROLL for (int e = 0; e < 3; e++) {
        const int k = 3 * leg + e, j = p * nu + k;
        double v = a.nominal[j];
        if (noised) {
          double z[2];
          gaussian_pair(a.seed, (uint32_t)gi, (uint32_t)(j >> 1), a.iteration, z);
          const double lo = L.ctrlrange[e][0], hi = L.ctrlrange[e][1];
          double sigma;
          if (a.noise_mode == 0) sigma = 0.5 * (hi - lo) * std;
          else {
            const double fl = gi < a.explore_count ? a.std0 : a.std1;
            const double sd = sqrt(a.param_variance[j]);
            sigma = sd > fl ? sd : fl;
          }
          v = clampd(v + sigma * ((j & 1) ? z[1] : z[0]), lo, hi);
        }
        a.nodes[(size_t)j * N + cand] = v;
      }
  }
#define QNODE(p, e) a.nodes[(size_t)((p) * nu + 3 * leg + (e)) * N + cand]
  QState S;
  QUNROLL for (int k = 0; k < 7; k++) S.tq[k] = state0[k];
  QUNROLL for (int j = 0; j < 3; j++) { S.lq[j] = state0[7 + 3 * leg + j]; S.lv[j] = state0[19 + 6 + 3 * leg + j]; S.wl[j] = 0; }
  QUNROLL for (int k = 0; k < 6; k++) { S.tv[k] = state0[19 + k]; S.wt[k] = 0; }
  S.time = time0;
  const size_t ds = 37;
  double total = 0;
  double ctrl[3] = {0, 0, 0};
  int flags = 0, flag_step = 0;
  for (int t = 0; t < H; t++) {
    flag_step = t;
    const long long step_t0 = QCLASS_NOW(a);
    const bool last = t == H - 1;
    bool bad = false;
    if (!last) {
      double ufb[3] = {0, 0, 0};
      if (FEEDBACK) feedback_ctrl(fb, fb_alpha, t, S, leg, ufb);
      // policy: TimeSpline::Sample + Clamp (SamplingPolicy::Action)
      int up = 0;
      const double now = QUNIFORM_TIME(S.time);
      while (!FEEDBACK && up < P && a.node_times[up] <= now) up++;
      QUNROLL for (int e = 0; e < 3; e++) {
        double u;
        if (FEEDBACK) u = ufb[e];
        else if (up == P || up == 0) u = QNODE(up == 0 ? 0 : P - 1, e);
        else {
          const int lo = up - 1;
          const double tl = a.node_times[lo], tu = a.node_times[up];
          const double p0 = QNODE(lo, e), p1 = QNODE(up, e);
          if (a.interp == 0) u = p0;
          else {
            const double s = (S.time - tl) / (tu - tl);
            if (a.interp == 1) u = p0 * (1 - s) + p1 * s;
            else {
              const double dt_mid = tu - tl, fwd = (p1 - p0) / dt_mid;
              double m0, m1;
              if (lo == 0) m0 = fwd;
              else m0 = 0.5 * (p1 - p0) / dt_mid + 0.5 * (p0 - QNODE(lo - 1, e)) / (tl - a.node_times[lo - 1]);
              if (up == P - 1) m1 = fwd;
              else m1 = 0.5 * (QNODE(up + 1, e) - p1) / (a.node_times[up + 1] - tu) + 0.5 * (p1 - p0) / dt_mid;
              const double s2 = s * s, s3 = s * s * s;
              const double c0 = 2 * s3 - 3 * s2 + 1, c1 = (s3 - 2 * s2 + s) * (tu - tl), c2 = -2 * s3 + 3 * s2, c3 = (s3 - s2) * (tu - tl);
              u = c0 * p0 + c1 * m0 + c2 * p1 + c3 * m1;
            }
          }
        }
        bad |= qbad(u);
        ctrl[e] = clampd(u, L.ctrlrange[e][0], L.ctrlrange[e][1]);
      }
      QUNROLL for (int k = 0; k < 7; k++) bad |= qbad(S.tq[k]);
      QUNROLL for (int k = 0; k < 6; k++) bad |= qbad(S.tv[k]);
      QUNROLL for (int j = 0; j < 3; j++) bad |= qbad(S.lq[j]) || qbad(S.lv[j]);
    }
    if (qd_or(bad ? 1 : 0)) { flags = kFlagBad; break; }
    QPROF(pf, 0);
    QDyn D;
    QSense f;
    flags = forward_smooth(m, tab, sp, leg, S, ctrl, cs, ms, D, f, pf);
    if (flags) break;
    // the sensor stage (residual, cost, traces) does not depend on the constraint solve: evaluated and recorded first, so that
    // nothing of it is live during the solve
    QResidual r;
    const double cost = residual_cost(m, tk, sp, leg, S, f, r);
    // ---- record step t: the quad's four lanes share the row
    {
      double* st = a.states + ((size_t)cand * H + t) * ds;
      if (leg == 0) QUNROLL for (int k = 0; k < 7; k++) QREC(st[k], S.tq[k]);
      if (leg == 1) QUNROLL for (int k = 0; k < 6; k++) QREC(st[19 + k], S.tv[k]);
      QUNROLL for (int j = 0; j < 3; j++) { QREC(st[7 + 3 * leg + j], S.lq[j]); QREC(st[25 + 3 * leg + j], S.lv[j]); }
      double* ac = a.actions + ((size_t)cand * H + t) * nu;
      QUNROLL for (int j = 0; j < 3; j++) QREC(ac[3 * leg + j], ctrl[j]);
      double* rs = a.residual + ((size_t)cand * H + t) * nr;
      if (leg == 2) { QUNROLL for (int i = 0; i < 7; i++) QREC(rs[i], r.shared[i]); }
      if (leg == 3) { QREC(rs[11], r.shared[7]); QREC(rs[12], r.shared[8]); QUNROLL for (int i = 0; i < 5; i++) QREC(rs[37 + i], r.shared[9 + i]); }
      QREC(rs[7 + L.foot_index], r.gait);
      QUNROLL for (int j = 0; j < 3; j++) { QREC(rs[13 + 3 * leg + j], r.effort[j]); QREC(rs[25 + 3 * leg + j], r.posture[j]); }
      if (leg == 0) { QREC(a.times[(size_t)cand * H + t], S.time); QREC(a.costs[(size_t)cand * H + t], cost); }
      if (leg == 1) { QUNROLL for (int q = 0; q < kQMaxTrace; q++) if (q < m.ntrace) QUNROLL for (int k = 0; k < 3; k++) QREC(a.trace[((size_t)cand * H + t) * 3 * m.ntrace + 3 * q + k], f.trace[q][k]); }
    }
    total += cost;
    if (a.con_cap > 0 && qd_or(D.ncon > a.con_cap ? 1 : 0)) { flags = kFlagOverflow; break; }
    QPROF(pf, 5);
    if (last) break;  // (the last step's mj_forward only feeds the sensor stage)
    double al[3], at[6], fc_l[3], fc_t[6];
    int iters;
    // one pair pattern of legs in contact (or none): the super-leg solver; a leg touching two others (rare): the general one -- for
    // every candidate of the wavefront then (it covers the other cases too, and the wavefront runs one solver instead of both in turn)
    const long long solve_t0 = QCLASS_NOW(a);
#ifdef QEXP_NO_GENERAL
    // (tuning: what the launch would take if no candidate ever needed the general solver -- such candidates stop here, flagged)
    if (((D.pmask >> 1) & 1) + ((D.pmask >> 2) & 1) + ((D.pmask >> 3) & 1) >= kQGeneralFrom) { flags = kFlagPair; break; }
#endif
    const bool wave_general = qw_any(((D.pmask >> 1) & 1) + ((D.pmask >> 2) & 1) + ((D.pmask >> 3) & 1) >= kQGeneralFrom);
    // (the solver's inputs are handed over as pointers into S and D, which pins those two structs in memory -- measured the better
    // trade: copying them into a block of their own so that S and D stay in registers costs 2 ms of 57 in register pressure)
    if (wave_general)
      flags = constraint_newton<true>(m, D.kin, ms, D.R, cs, D.ncon, D.nrel, leg, D.pmask, D.mymask, D.have_rel != 0, D.sl, D.st, S.wl, S.wt, t > 0, al, at, fc_l, fc_t, iters, pf);
    else
      flags = constraint_newton<false>(m, D.kin, ms, D.R, cs, D.ncon, D.nrel, leg, D.pmask, D.mymask, D.have_rel != 0, D.sl, D.st, S.wl, S.wt, t > 0, al, at, fc_l, fc_t, iters, pf);
    if (flags) break;
    QCLASS_ADD(a, 0, D.have_rel, D.pmask, D.ncon, solve_t0);
    QWAVE_TIMES(a, iters, wave_general, D.ncon);
    QPROF(pf, 6);
    QPROF_COUNT(pf, 16, iters);
    QPROF_WAVE_HIST(pf, 18, D.ncon);
    QPROF_WAVE_HIST(pf, 28, iters);
    QUNROLL for (int j = 0; j < 3; j++) bad |= qbad(al[j]);
    QUNROLL for (int k = 0; k < 6; k++) bad |= qbad(at[k]);
    if (qd_or(bad ? 1 : 0)) { flags = kFlagBad; break; }
    euler(m, leg, S, D, ms, al, at, fc_l, fc_t);
    QCLASS_ADD(a, 32, D.have_rel, D.pmask, D.ncon, step_t0);
    QPROF(pf, 7);
  }
#undef QNODE
  if (leg == 0) {
    a.total_return[cand] = flags ? 1.0e6 : total / (double)(H > 1 ? H : 1);
    a.failure[cand] = flags ? (kQFallback | flags | (flag_step << 8)) : 0;  // (the step it was handed on at: diagnostics)
  }
  return flags;
}

// world poses of the static geoms for this rollout (mocap poses from the plan blob)
QD void static_pose(const QuadModel& m, const double* mocap, int s, QStaticPose& out) {
  const QuadStatic& S = m.stat[s];
  if (S.mocap < 0) {
    QUNROLL for (int k = 0; k < 3; k++) out.pos[k] = S.pos[k];
    QUNROLL for (int k = 0; k < 9; k++) out.mat[k] = S.rot[k];
    return;
  }
  const double* mp = mocap + 7 * S.mocap;
  double q[4] = {mp[3], mp[4], mp[5], mp[6]}, bm[9], v[3];
  q_norm(q); q_norm(q);
  q2mat(bm, q);
  mv3(v, bm, S.pos);
  QUNROLL for (int k = 0; k < 3; k++) out.pos[k] = mp[k] + v[k];
  QUNROLL for (int r = 0; r < 3; r++) QUNROLL for (int c = 0; c < 3; c++) out.mat[3 * r + c] = bm[3 * r] * S.rot[c] + bm[3 * r + 1] * S.rot[3 + c] + bm[3 * r + 2] * S.rot[6 + c];
}

} }  // namespace mjpcx::quad
