// limb_step.h -- the step function of the LIMB kernel family: Trajectory::Rollout (mjpc/trajectory.cc:100-210) of ONE candidate of the
// Humanoid tracking task (mjpc/tasks/humanoid/tracking/tracking.cc:94-216) by FOUR lanes, one per limb (limb_model.h). Written as a SIMT
// program like quad_step.h: every variable is the calling lane's own, `lane` its index in the quad; the cross-lane operations are
//     qd_sum(x)  qd_bcast<k>(x)  qd_bcasti<k>(i)  qd_or(i)      over the four lanes of the quad (bit-identical results in the four)
//     qw_any(p)                                                   over the wavefront (scalar branches around rare work)
//     ld_sync()                                                   orders the quad's accesses to its SHARED block (geom poses, the
//                                                                 contacts between moving geoms)
// which the including translation unit provides (DPP quad permutes and an LDS fence on gfx950: limb_kernel.h; a four-thread lock-step
// emulator on the CPU: tests/limbemu, test infrastructure). Control flow around a primitive is quad-uniform by construction.
//
// The physics is the oracle's (oracle/physics.c, contact.inc, humanoid.inc -- restating mj_step / mj_forward), re-derived for the
// arrowhead structure of a trunk chain with four limbs:
//   * the trunk chain (poses, the 9 dof axes, the 9 x 9 block T of M) is computed redundantly in the four lanes;
//   * a limb's bodies, its 6 x 6 block of M, the 6 x 9 coupling, its joint limits, its hamstring tendon and its contacts with the floor
//     live in its lane (the arms run the same code with their unused slots switched off);
//   * sums over the limbs (centre of mass, composite inertias and bias forces of the trunk, Schur complements, line-search
//     derivatives, costs) are quad sums;
//   * a contact between two MOVING geoms (a hand on a thigh: frictionless, one row) couples two limbs and would break the arrowhead:
//     its row u enters the Newton Hessian H = A + sum D u u' through the Woodbury identity -- one more arrowhead solve per such row,
//     no fill, any pattern of limbs (at most kMaxX rows per candidate and step; beyond that the candidate is handed on).
// Every constraint row of this model class is a scalar unilateral row (joint and tendon limits, pyramid edges, frictionless normals):
// cost 1/2 D x^2 for x < 0 -- the exact line search is Newton's method on a piecewise-linear derivative.
//
// A candidate the limb form does not cover at some step (more floor contacts in a lane than kMaxPC, more moving-geom contacts than
// kMaxX, the trunk on the floor, both limits of a joint, an indefinite matrix, a non-finite state) is FLAGGED and rolled out from the
// start by rollout_tree_kernel<Humanoid>: results never depend on which kernel ran.
#pragma once
#include <stdint.h>

#include "limb_abi.h"
#include "limb_model.h"

#ifndef LD
#error "define LD (function qualifiers) and the quad primitives before including limb_step.h"
#endif
#ifndef LUNROLL
#define LUNROLL
#endif
#ifndef LREC
#define LREC(dst, v) (dst) = (v)
#endif
#ifndef LNOINLINE
#define LNOINLINE LD
#endif
// An out-of-line function of the step receives the model image and its caller's locals through plain references / pointers; the device build
// says where they live (LDS / the private segment) so that they are read with ds_read / scratch_load instead of FLAT accesses, and copies
// the locals in and out once per call (limb_kernel.h).
#ifndef LREBIND_LDS
#define LREBIND_LDS(T, ref) (ref)
#endif
#ifndef LREBIND_PRV
#define LREBIND_PRV(T, ptr) (ptr)   // a pointer to a caller's local: the device build says it is private memory (scratch_ instead of FLAT accesses)
#endif
#ifndef LPRV_LOAD
#define LPRV_LOAD(dst, src) (dst) = *(src)
#define LPRV_STORE(dst, src) *(dst) = (src)
#define LPRV_LOADN(dst, src, n) do { for (int i_ = 0; i_ < (n); i_++) (dst)[i_] = (src)[i_]; } while (0)
#define LPRV_STOREN(dst, src, n) do { for (int i_ = 0; i_ < (n); i_++) (dst)[i_] = (src)[i_]; } while (0)
#endif
#ifndef LEXP_COPY_MASK
#define LEXP_COPY_MASK 16
// Which out-of-line stages copy what they get by pointer into locals (bits: 1 forward, 2 newton: everything, 4 euler, 8 residual, 16 newton: only
// what it WRITES -- the rows' jar, the iterate, J' force: a read-modify-write through the caller's memory waits for its own store every time,
// while what is only read (dof axes, qacc_smooth) is served by the vector L1). Measured on MI355X, Humanoid 8192 x 64 fp32, same box: no
// copies 23.7 ms; 16: 23.1; 2 | 4 | 8: 44.2 (164 spilled registers instead of 38). The forward stage always works on the caller's structs:
// with copies its DEVICE build produced wrong states (the emulator's did not; not understood), and they would save 300 accesses in 30 k
// instructions.
#endif
#ifndef LUNIFORM
#define LUNIFORM(i) (i)   // (the device build: an int known to be the same in every lane, moved to a scalar register)
#endif
#ifndef LPOISON
#define LPOISON(x)   // (the emulator fills fresh locals with NaN patterns: a read before a write shows)
#endif
#ifndef LEXP_GFLOOR
#define LEXP_GFLOOR 16   // (multiples of the working precision's epsilon: see the Newton loop's floors)
#define LEXP_CFLOOR 8
#endif
#ifndef LEXP_LFLOOR
#define LEXP_LFLOOR 16
#endif
#ifdef LEXP_XSTAMPS
#define LXPROF(a, last, idx) LPROF(a, last, idx)
#else
#define LXPROF(a, last, idx)
#endif
#ifndef LPROF
#define LPROF(a, last, idx)   // phase cycle stamps of wavefront 0 (the device build: limb_kernel.h)
#define LPROF_COUNT(a, idx)
#endif

namespace mjpcx { namespace limb {

enum { kFlagOverflow = 1, kFlagCross = 2, kFlagNotPD = 4, kFlagBad = 8, kFlagLimits = 16, kFlagTrunkFloor = 32 };
constexpr int kLsTolInv = 100;  // mjOption.ls_tolerance = 0.01

// ---------------------------------------------------------------- small algebra
LD void lsincos(double x, double& s, double& c) { sincos(x, &s, &c); }   // (one range reduction for the pair)
LD void lsincos(float x, float& s, float& c) { sincosf(x, &s, &c); }
template <typename R> LD bool lbad(R x) { return !(x <= R(1e10) && x >= R(-1e10)); }
template <typename R> LD void q_mul(R* r, const R* a, const R* b) {
  const R w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const R y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
template <typename R> LD void q2mat(R* m, const R* q) {
  const R q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  const R q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
template <typename R> LD void q_norm(R* q) {
  const R n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < R(1e-15)) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { const R s = R(1) / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
template <typename R> LD void mv3(R* r, const R* m, const R* v) {
  const R x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename R> LD void q_rot(R* r, const R* v, const R* q) { R m[9]; q2mat(m, q); mv3(r, m, v); }
template <typename R> LD void cr3(R* r, const R* a, const R* b) {
  const R x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename R> LD R dot3(const R* a, const R* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// (generic over anything indexable: the dof axes live in the includer's store -- LDS on the device, read element by element through a proxy)
template <class A, class B> LD auto dot6(const A& a, const B& b) -> decltype(a[0] * b[0]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }
// R I R' of a symmetric I (xx yy zz xy xz yz) -> the same six
template <typename R> LD void rot_sym(R* res, const R* I, const R* m) {
  R t[9];  // t = m I
  LUNROLL for (int r = 0; r < 3; r++) {
    t[3 * r] = m[3 * r] * I[0] + m[3 * r + 1] * I[3] + m[3 * r + 2] * I[4];
    t[3 * r + 1] = m[3 * r] * I[3] + m[3 * r + 1] * I[1] + m[3 * r + 2] * I[5];
    t[3 * r + 2] = m[3 * r] * I[4] + m[3 * r + 1] * I[5] + m[3 * r + 2] * I[2];
  }
  res[0] = t[0] * m[0] + t[1] * m[1] + t[2] * m[2];
  res[1] = t[3] * m[3] + t[4] * m[4] + t[5] * m[5];
  res[2] = t[6] * m[6] + t[7] * m[7] + t[8] * m[8];
  res[3] = t[0] * m[3] + t[1] * m[4] + t[2] * m[5];
  res[4] = t[0] * m[6] + t[1] * m[7] + t[2] * m[8];
  res[5] = t[3] * m[6] + t[4] * m[7] + t[5] * m[8];
}
// spatial inertia about the reference point (oracle inert_com): 10 numbers
template <typename R> LD void inert_shift(R* res, const R* irot, const R* dif, R mass) {
  res[0] = irot[0] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] = irot[1] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] = irot[2] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] = irot[3] - mass * dif[0] * dif[1];
  res[4] = irot[4] - mass * dif[0] * dif[2];
  res[5] = irot[5] - mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2];
  res[9] = mass;
}
template <typename R> LD void mul_inert(R* res, const R* i, const R* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
template <typename R> LD void cross_motion(R* res, const R* vel, const R* v) {
  R a[3], b[3], c[3];
  cr3(a, vel, v); cr3(b, vel, v + 3); cr3(c, vel + 3, v);
  res[0] = a[0]; res[1] = a[1]; res[2] = a[2];
  res[3] = b[0] + c[0]; res[4] = b[1] + c[1]; res[5] = b[2] + c[2];
}
template <typename R> LD void cross_force(R* res, const R* vel, const R* f) {
  R a[3], b[3], c[3];
  cr3(a, vel, f); cr3(b, vel + 3, f + 3); cr3(c, vel, f + 3);
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  res[3] = c[0]; res[4] = c[1]; res[5] = c[2];
}
template <typename R> LD R clampr(R x, R lo, R hi) { return x < lo ? lo : (x > hi ? hi : x); }
// solimp (digested: limb_model.h) -> impedance at violation `dist` (oracle impedance())
// (the general exponent needs pow(): ONE out-of-line instance for all the call sites -- inlined it was four pow() bodies per site, and the
// step function's code has to stay near the 64 KB of the instruction cache)
template <typename R> LNOINLINE R impedance_pow(R x, R mid, R power) {
  return x <= mid ? R(pow(x, power) / pow(mid, power - 1)) : R(1 - pow(1 - x, power) / pow(1 - mid, power - 1));
}
template <typename R> LD R impedance(const R* d, R dist) {
  const R dmin = d[0], dmax = d[1], width = d[2], mid = d[3], power = d[4];
  if (dmin == dmax || width <= R(1e-15)) return R(0.5) * (dmin + dmax);
  const R x = fabs(dist) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  R y;
  if (power == 1) y = x;
  else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
  else y = impedance_pow(x, mid, power);
  return dmin + y * (dmax - dmin);
}
template <typename R> LD constexpr R kEps() { return sizeof(R) == 4 ? R(1.1920929e-7) : R(2.220446049250313e-16); }
LD constexpr int tri(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
template <typename R> LD void sym6_add_outer(R* X, R w, const R* v) {
  LUNROLL for (int p = 0; p < 6; p++) { const R wp = w * v[p]; LUNROLL for (int q = 0; q <= p; q++) X[tri(p, q)] += wp * v[q]; }
}
template <typename R, class V> LD void sym6_mul(R* y, const R* X, const V& vv) {
  R v[6];
  LUNROLL for (int q = 0; q < 6; q++) v[q] = vv[q];
  LUNROLL for (int p = 0; p < 6; p++) { R s = 0; LUNROLL for (int q = 0; q < 6; q++) s += X[tri(p, q)] * v[q]; y[p] = s; }
}

// ---------------------------------------------------------------- arrowhead matrices
// Symmetric positive-definite matrix of the tree: per lane the limb block l (6 x 6, packed lower triangle) and the coupling b (6 limb
// dofs x 9 trunk dofs); replicated in the four lanes the trunk block t (9 x 9, packed). arrow_factor turns it IN PLACE into its factor:
// l := unit-lower L D L' (entries below the diagonal, RECIPROCAL pivots on it), b := Z = block^-1 B, t := the L D L' factor of the Schur
// complement T - sum_limbs B' Z.
template <typename R> struct Arrow { R l[21], b[kLD][kTD], t[45]; };

template <int N, typename R> LD bool ldl_packed(R* a) {
  bool ok = true;
  R d[N];
  LUNROLL for (int j = 0; j < N; j++) {
    R dj = a[tri(j, j)];
    LUNROLL for (int k = 0; k < j; k++) dj -= a[tri(j, k)] * a[tri(j, k)] * d[k];
    ok &= dj > R(1e-15);
    d[j] = dj;
    const R inv = R(1) / dj;
    LUNROLL for (int i = j + 1; i < N; i++) {
      R v = a[tri(i, j)];
      LUNROLL for (int k = 0; k < j; k++) v -= a[tri(i, k)] * a[tri(j, k)] * d[k];
      a[tri(i, j)] = v * inv;
    }
    a[tri(j, j)] = inv;
  }
  return ok;
}
template <int N, typename R> LD void ldl_solve(const R* a, R* x) {
  LUNROLL for (int i = 1; i < N; i++) LUNROLL for (int k = 0; k < i; k++) x[i] -= a[tri(i, k)] * x[k];
  LUNROLL for (int i = 0; i < N; i++) x[i] *= a[tri(i, i)];
  LUNROLL for (int i = N - 2; i >= 0; i--) LUNROLL for (int k = i + 1; k < N; k++) x[i] -= a[tri(k, i)] * x[k];
}
// returns false (quad-uniform) if a pivot is not positive
template <typename R> LD bool arrow_factor(Arrow<R>& a) {
  bool ok = ldl_packed<kLD>(a.l);
  // Y = L^-1 B in place
  LUNROLL for (int k = 0; k < kTD; k++) LUNROLL for (int i = 1; i < kLD; i++) LUNROLL for (int j = 0; j < i; j++) a.b[i][k] -= a.l[tri(i, j)] * a.b[j][k];
  LUNROLL for (int r = 0; r < kTD; r++)
    LUNROLL for (int c = 0; c <= r; c++) {
      R s = 0;
      LUNROLL for (int i = 0; i < kLD; i++) s += a.b[i][r] * a.b[i][c] * a.l[tri(i, i)];
      a.t[tri(r, c)] -= qd_sum(s);
    }
  // Z = L^-T D^-1 Y in place
  LUNROLL for (int k = 0; k < kTD; k++)
    LUNROLL for (int i = kLD - 1; i >= 0; i--) {
      R v = a.b[i][k] * a.l[tri(i, i)];
      LUNROLL for (int j = i + 1; j < kLD; j++) v -= a.l[tri(j, i)] * a.b[j][k];
      a.b[i][k] = v;
    }
  ok &= ldl_packed<kTD>(a.t);
  return qd_or(ok ? 0 : 1) == 0;
}
// x := A^-1 x (xl: the lane's six limb entries, xt: the nine trunk entries, replicated)
template <typename R> LD void arrow_solve(const Arrow<R>& f, R* xl, R* xt) {
  LUNROLL for (int k = 0; k < kTD; k++) {
    R s = 0;
    LUNROLL for (int i = 0; i < kLD; i++) s += f.b[i][k] * xl[i];
    xt[k] -= qd_sum(s);
  }
  ldl_solve<kTD>(f.t, xt);
  ldl_solve<kLD>(f.l, xl);
  LUNROLL for (int i = 0; i < kLD; i++) LUNROLL for (int k = 0; k < kTD; k++) xl[i] -= f.b[i][k] * xt[k];
}
// x' y over all dofs (trunk part counted once)
template <typename R> LD R arrow_dot(const R* xl, const R* xt, const R* yl, const R* yt) {
  R s = 0, t = 0;
  LUNROLL for (int i = 0; i < kLD; i++) s += xl[i] * yl[i];
  LUNROLL for (int k = 0; k < kTD; k++) t += xt[k] * yt[k];
  return qd_sum(s) + t;
}
// M lives in the includer's store while the solver runs: accessors lms_l / lms_b / lms_t, setters lms_set_*
template <typename R, class MS> LD void store_arrow(MS& ms, const Arrow<R>& M) {
  LUNROLL for (int i = 0; i < 21; i++) lms_set_l(ms, i, M.l[i]);
  LUNROLL for (int j = 0; j < kLD; j++) LUNROLL for (int k = 0; k < kTD; k++) lms_set_b(ms, j, k, M.b[j][k]);
  LUNROLL for (int i = 0; i < 45; i++) lms_set_t(ms, i, M.t[i]);
}
template <typename R, class MS> LD void load_arrow(const MS& ms, Arrow<R>& M) {
  LUNROLL for (int i = 0; i < 21; i++) M.l[i] = lms_l(ms, i);
  LUNROLL for (int j = 0; j < kLD; j++) LUNROLL for (int k = 0; k < kTD; k++) M.b[j][k] = lms_b(ms, j, k);
  LUNROLL for (int i = 0; i < 45; i++) M.t[i] = lms_t(ms, i);
}
template <typename R, class MS> LD void arrow_mul_s(const MS& ms, const R* xl, const R* xt, R* yl, R* yt) {
  R c[kTD];
  LUNROLL for (int k = 0; k < kTD; k++) c[k] = 0;
  LUNROLL for (int j = 0; j < kLD; j++) {
    R s = 0;
    LUNROLL for (int i = 0; i < kLD; i++) s += lms_l(ms, tri(j, i)) * xl[i];
    LUNROLL for (int k = 0; k < kTD; k++) { const R b = lms_b(ms, j, k); s += b * xt[k]; c[k] += b * xl[j]; }
    yl[j] = s;
  }
  LUNROLL for (int k = 0; k < kTD; k++) {
    R s = qd_sum(c[k]);
    LUNROLL for (int i = 0; i < kTD; i++) s += lms_t(ms, tri(k, i)) * xt[i];
    yt[k] = s;
  }
}

// ---------------------------------------------------------------- per-candidate data
template <typename R> struct LState {
  R tq[10], tv[kTD];   // trunk: position, quaternion, the three hinge angles; velocities
  R lq[kLD], lv[kLD];
  R wt[kTD], wl[kLD];  // previous step's qacc (solver warm start)
  R time;
};
// the position-dependent part of a forward pass that the constraint solve needs: dof axes about the centre of mass
template <typename R> struct LKin { R cdof[kLD][6], cdofT[kTD][6]; };
// The free joint's three translations are unit vectors in world axes -- cdofT[k] = e_(3+k) for k < 3 --, which the solver, reading the axes from
// the includer's store, cannot know: products with them are picked, not computed (k is a compile-time index wherever these are used)
template <typename R, class KIN> LD R trunk_dot(const KIN& kin, int k, const R* F) { return k < 3 ? F[3 + k] : dot6(kin.cdofT[k], F); }
template <typename R, class KIN> LD void trunk_sym6_mul(R* y, const R* X, const KIN& kin, int k) {
  if (k < 3) { LUNROLL for (int p = 0; p < 6; p++) y[p] = p >= 3 + k ? X[tri(p, 3 + k)] : X[tri(3 + k, p)]; }
  else sym6_mul(y, X, kin.cdofT[k]);
}

// spatial velocities [angular; linear about the centre of mass] of the bodies for the dof vector (xl, xt): VT the trunk bodies, VL the
// limb's, VR the limb's relative to the trunk body it hangs on (what a contact between two moving geoms needs: the common motion never
// enters, so nothing cancels)
template <typename R, class KIN> LD void chain_velocity(const KIN& k, int attach, const R* xl, const R* xt, R VT[kTB][6], R VL[kLB][6], R VR[kLB][6]) {
  LUNROLL for (int c = 0; c < 6; c++) {
    R v = 0;
    if (c >= 3) v = xt[c - 3];  // (the translations: see trunk_dot)
    LUNROLL for (int d = 3; d < 6; d++) v += k.cdofT[d][c] * xt[d];
    VT[0][c] = v;
    VT[1][c] = v + k.cdofT[6][c] * xt[6] + k.cdofT[7][c] * xt[7];
    VT[2][c] = VT[1][c] + k.cdofT[8][c] * xt[8];
    const R base = attach == 0 ? VT[0][c] : (attach == 1 ? VT[1][c] : VT[2][c]);
    VR[0][c] = k.cdof[0][c] * xl[0] + k.cdof[1][c] * xl[1] + k.cdof[2][c] * xl[2];
    VR[1][c] = VR[0][c] + k.cdof[3][c] * xl[3];
    VR[2][c] = VR[1][c] + k.cdof[4][c] * xl[4] + k.cdof[5][c] * xl[5];
    LUNROLL for (int b = 0; b < kLB; b++) VL[b][c] = base + VR[b][c];
  }
}
// one of six bodies' values (the limb's three, the trunk's three) by 0 / 1 weights: a five-deep chain of ?: on a lane's own index is compiled
// into BRANCHES (the four sites and two traces of a step were 970 of them, 22 k cycles), the weights into a multiply-add per source
template <typename R> LD void pick6_weights(int d, R* w) { LUNROLL for (int i = 0; i < 6; i++) w[i] = d == i ? R(1) : R(0); }
template <typename R> LD R pick6(const R* w, R a0, R a1, R a2, R a3, R a4, R a5) { return w[0] * a0 + w[1] * a1 + w[2] * a2 + w[3] * a3 + w[4] * a4 + w[5] * a5; }
template <typename R> LD void pick3(const R V[3][6], int d, R* out) {  // V[d] by 0 / 1 weights (no run-time index into registers)
  const R w0 = d == 0 ? R(1) : R(0), w1 = d == 1 ? R(1) : R(0), w2 = d == 2 ? R(1) : R(0);
  LUNROLL for (int c = 0; c < 6; c++) out[c] = w0 * V[0][c] + w1 * V[1][c] + w2 * V[2][c];
}

// A contact with the floor, in WORLD axes at the contact point: offset from the centre of mass, the D of its rows (all edges of a pyramid
// share one), jar of the rows -- condim 3: the four edges normal +- mu t1, normal +- mu t2; condim 1: the normal -- holding -aref from
// the contact's creation until the solver's first pass adds J qacc_smooth. The frame is the floor's (LimbModelT::plane_*).
template <typename R> struct LContact { R off[3], D, mu, jar[4]; int body, nrow; };
constexpr int kLConRec = 10;  // reals per stored contact: off 0-2, D 3, mu 4, jar 5-8, (body, nrow) 9
// A contact between two moving geoms, in the candidate's SHARED block: the row's generator et = [off x n; n] about the centre of mass,
// what the row's instantiation needs, and which chains it acts on (body 1: minus, body 2: plus)
template <typename R> struct LCross { R et[6], D, b, kimpx; int la, sa, lb, sb; };
constexpr int kLCrossRec = 12;  // et 0-5, D 6, b 7, kimpx 8, meta 9; the solver's: jar 10, its rate along the search direction 11 (lsh_xget / lsh_xset)

template <typename R> LD R xsel(const R* v, int r) {  // v[r], r at run time, without indexing registers
  R o = 0;
  LUNROLL for (int i = 0; i < kMaxX; i++) o = i == r ? v[i] : o;
  return o;
}
// the lane's diagonal rows: the active limit of each joint (side 0: none; J = -side on the dof) and of the limb's tendon; the trunk's three
// hinges' (replicated)
template <typename R> struct LRows {
  R lm_D[kLD], lm_jar[kLD]; int lm_side[kLD];
  R tn_D, tn_jar; int tn_side;
  R tl_D[3], tl_jar[3]; int tl_side[3];
};  // (the rows of the contacts between moving geoms live in the quad's shared block: LCross, lsh_xget / lsh_xset)

// what the sensor stage reads
template <typename R> struct LSense { R spos[kLS][3], svel[kLS][3], trace[kMaxTrace][3]; };
template <typename R> struct LDyn {   // (the dof axes go to the includer's store: lkin_store)
  LRows<R> rows;
  R sl[kLD], st[kTD];        // qacc_smooth
  R fs_l[kLD], fs_t[kTD];    // qfrc_smooth
  R com[3];
  int ncon, nx;
};
// per-plan task data (WaveTaskT's blob: weights and norm parameters may change between plan steps)
template <typename R> struct LTask { const R *mocap, *weight, *norm_p, *norm_q, *re; const int* ri; R risk; const R* key_mpos; };

// one scalar unilateral row
template <typename R> LD void row_pen(R x, R D, R& cost, R& force) { const bool on = x < 0; cost += on ? R(0.5) * D * x * x : R(0); force = on ? -D * x : R(0); }
// row limit instantiation (oracle o_make_constraint_full + the impedance pass): dist < margin -> D and -aref
template <typename R> LD void limit_row(R dist, R margin, R vel, R invw, R k, R b, const R* imp, R& D, R& jar) {
  const R pos = dist - margin, im = impedance(imp, pos);
  R Rr = (1 - im) / im * invw;
  if (Rr < R(1e-15)) Rr = R(1e-15);
  D = R(1) / Rr;
  jar = b * vel + k * im * pos;  // -aref
}

// ---------------------------------------------------------------- mj_forward before the constraint solve
template <typename R, class CS, class MS, class SH, class KS>
LD int forward_smooth_body(const LimbModelT<R>& m, int lane, const LState<R>& S, const R* ctrl, const R* tctrl, CS& cs, MS& ms, SH& sh, KS& ks, LDyn<R>& D, LSense<R>& out, long long* stamps) {
  const LimbT<R>& L = m.limb[lane];
  struct { long long* stamps; } pa{stamps};
  long long prof_last = 0;
  LPROF(pa, prof_last, -1);
  LKin<R> kin;
  int flags = 0;
  // ================= kinematics (o_kinematics): the trunk chain, then the limb's chain
  R tpos[kTB][3], tmat[kTB][9], tipos[kTB][3], tanchor[3][3], taxis[3][3];
  R tquat[4] = {S.tq[3], S.tq[4], S.tq[5], S.tq[6]};
  {
    q_norm(tquat); q_norm(tquat);  // (the oracle normalises when it reads qpos and again at the end of the body loop)
    LUNROLL for (int k = 0; k < 3; k++) tpos[0][k] = S.tq[k];
    q2mat(tmat[0], tquat);
  }
  R pquat[4] = {tquat[0], tquat[1], tquat[2], tquat[3]};
  R aquat[kTB][4];  // the trunk bodies' orientations (a limb starts from the one it hangs on)
  LUNROLL for (int k = 0; k < 4; k++) aquat[0][k] = tquat[k];
  // a body with hinges (oracle's joint loop): from the parent's pose to the body's, joint anchors and axes on the way
  auto place = [&](const LBodyT<R>& B, const R* ppos, const R* pmat, const R* pq, R* xpos, R* xquat) {
    R v[3];
    mv3(v, pmat, B.pos);
    LUNROLL for (int k = 0; k < 3; k++) xpos[k] = ppos[k] + v[k];
    q_mul(xquat, pq, B.quat);
  };
  auto hinge = [&](const LJointT<R>& J, R q, R* xpos, R* xquat, R* anchor, R* axis) {
    R v[3];
    q_rot(v, J.pos, xquat);
    LUNROLL for (int k = 0; k < 3; k++) anchor[k] = xpos[k] + v[k];
    q_rot(axis, J.axis, xquat);
    const R ang = q - J.qpos0;
    R ql[4] = {1, 0, 0, 0};
    if (ang != 0) { R s, c; lsincos(R(0.5) * ang, s, c); ql[0] = c; ql[1] = J.axis[0] * s; ql[2] = J.axis[1] * s; ql[3] = J.axis[2] * s; }
    R nq[4];
    q_mul(nq, xquat, ql);
    LUNROLL for (int k = 0; k < 4; k++) xquat[k] = nq[k];
    q_rot(v, J.pos, xquat);
    LUNROLL for (int k = 0; k < 3; k++) xpos[k] = anchor[k] - v[k];
  };
  LUNROLL for (int i = 1; i < kTB; i++) {
    R xq[4];
    place(m.tbody[i], tpos[i - 1], tmat[i - 1], pquat, tpos[i], xq);
    if (i == 1) {
      if (m.tjnt[0].on) hinge(m.tjnt[0], S.tq[7], tpos[1], xq, tanchor[0], taxis[0]);
      if (m.tjnt[1].on) hinge(m.tjnt[1], S.tq[8], tpos[1], xq, tanchor[1], taxis[1]);
    } else if (m.tjnt[2].on) hinge(m.tjnt[2], S.tq[9], tpos[2], xq, tanchor[2], taxis[2]);
    q_norm(xq);
    q2mat(tmat[i], xq);
    LUNROLL for (int k = 0; k < 4; k++) { pquat[k] = xq[k]; aquat[i][k] = xq[k]; }
  }
  LUNROLL for (int i = 0; i < kTB; i++) { R v[3]; mv3(v, tmat[i], m.tbody[i].ipos); LUNROLL for (int k = 0; k < 3; k++) tipos[i][k] = tpos[i][k] + v[k]; }
  R xpos[kLB][3], xmat[kLB][9], xipos[kLB][3], anchor[kLD][3], axis[kLD][3];
  {
    const int at = L.attach;
    R ppos[3], pmat[9], pq[4];
    LUNROLL for (int k = 0; k < 3; k++) ppos[k] = at == 0 ? tpos[0][k] : (at == 1 ? tpos[1][k] : tpos[2][k]);
    LUNROLL for (int k = 0; k < 9; k++) pmat[k] = at == 0 ? tmat[0][k] : (at == 1 ? tmat[1][k] : tmat[2][k]);
    LUNROLL for (int k = 0; k < 4; k++) pq[k] = at == 0 ? aquat[0][k] : (at == 1 ? aquat[1][k] : aquat[2][k]);
    LUNROLL for (int b = 0; b < kLB; b++) {
      R xq[4];
      place(L.body[b], ppos, pmat, pq, xpos[b], xq);
      LUNROLL for (int j = 0; j < kLD; j++) {
        if (slot_body(j) != b) continue;
        if (L.jnt[j].on) hinge(L.jnt[j], S.lq[j], xpos[b], xq, anchor[j], axis[j]);
        else { LUNROLL for (int k = 0; k < 3; k++) { anchor[j][k] = xpos[b][k]; axis[j][k] = 0; } }
      }
      q_norm(xq);
      q2mat(xmat[b], xq);
      R v[3];
      mv3(v, xmat[b], L.body[b].ipos);
      LUNROLL for (int k = 0; k < 3; k++) { xipos[b][k] = xpos[b][k] + v[k]; ppos[k] = xpos[b][k]; }
      LUNROLL for (int k = 0; k < 9; k++) pmat[k] = xmat[b][k];
      LUNROLL for (int k = 0; k < 4; k++) pq[k] = xq[k];
    }
  }
  // ================= centre of mass (o_compos), spatial inertias and dof axes about it
  R com[3];
  {
    R mass = 0;
    LUNROLL for (int k = 0; k < 3; k++) com[k] = 0;
    LUNROLL for (int b = 0; b < kLB; b++) { mass += L.body[b].mass; LUNROLL for (int k = 0; k < 3; k++) com[k] += L.body[b].mass * xipos[b][k]; }
    mass = qd_sum(mass);
    LUNROLL for (int k = 0; k < 3; k++) com[k] = qd_sum(com[k]);
    LUNROLL for (int i = 0; i < kTB; i++) { mass += m.tbody[i].mass; LUNROLL for (int k = 0; k < 3; k++) com[k] += m.tbody[i].mass * tipos[i][k]; }
    LUNROLL for (int k = 0; k < 3; k++) { com[k] /= mass; D.com[k] = com[k]; }
  }
  R cin[kLB][10], cinT[kTB][10];
  LUNROLL for (int b = 0; b < kLB; b++) {
    R ir[6], dif[3] = {xipos[b][0] - com[0], xipos[b][1] - com[1], xipos[b][2] - com[2]};
    rot_sym(ir, L.body[b].inertia, xmat[b]);
    inert_shift(cin[b], ir, dif, L.body[b].mass);
  }
  LUNROLL for (int i = 0; i < kTB; i++) {
    R ir[6], dif[3] = {tipos[i][0] - com[0], tipos[i][1] - com[1], tipos[i][2] - com[2]};
    rot_sym(ir, m.tbody[i].inertia, tmat[i]);
    inert_shift(cinT[i], ir, dif, m.tbody[i].mass);
  }
  LUNROLL for (int j = 0; j < kLD; j++) {
    R off[3] = {com[0] - anchor[j][0], com[1] - anchor[j][1], com[2] - anchor[j][2]}, w[3];
    cr3(w, axis[j], off);
    LUNROLL for (int k = 0; k < 3; k++) { kin.cdof[j][k] = axis[j][k]; kin.cdof[j][3 + k] = L.jnt[j].on ? w[k] : R(0); }
  }
  {
    const R off[3] = {com[0] - tpos[0][0], com[1] - tpos[0][1], com[2] - tpos[0][2]};
    LUNROLL for (int k = 0; k < 3; k++) {
      LUNROLL for (int c = 0; c < 6; c++) kin.cdofT[k][c] = c == 3 + k ? R(1) : R(0);
      const R ax[3] = {tmat[0][k], tmat[0][3 + k], tmat[0][6 + k]};
      R w[3];
      cr3(w, ax, off);
      LUNROLL for (int c = 0; c < 3; c++) { kin.cdofT[3 + k][c] = ax[c]; kin.cdofT[3 + k][3 + c] = w[c]; }
    }
    LUNROLL for (int h = 0; h < 3; h++) {
      if (m.tjnt[h].on) {
        R o2[3] = {com[0] - tanchor[h][0], com[1] - tanchor[h][1], com[2] - tanchor[h][2]}, w[3];
        cr3(w, taxis[h], o2);
        LUNROLL for (int c = 0; c < 3; c++) { kin.cdofT[6 + h][c] = taxis[h][c]; kin.cdofT[6 + h][3 + c] = w[c]; }
      } else { LUNROLL for (int c = 0; c < 6; c++) kin.cdofT[6 + h][c] = 0; }
    }
  }
  LPROF(pa, prof_last, 20);
  lkin_store(ks, kin);
  // ================= composite inertias -> M as an arrowhead (o_crb)
  {
    Arrow<R> M;
    R crb[kLB][10], crbT[kTB][10];
    LUNROLL for (int e = 0; e < 10; e++) { crb[2][e] = cin[2][e]; crb[1][e] = cin[1][e] + crb[2][e]; crb[0][e] = cin[0][e] + crb[1][e]; }
    LUNROLL for (int i = kTB - 1; i >= 0; i--)
      LUNROLL for (int e = 0; e < 10; e++) {
        R v = cinT[i][e] + (i < kTB - 1 ? crbT[i + 1][e] : R(0));
        if (m.nattach[i] > 0) v += qd_sum(L.attach == i ? crb[0][e] : R(0));
        crbT[i][e] = v;
      }
    LUNROLL for (int i = 0; i < kLD; i++) {
      R buf[6];
      mul_inert(buf, crb[slot_body(i)], kin.cdof[i]);
      M.l[tri(i, i)] = L.jnt[i].armature + dot6(kin.cdof[i], buf);
      LUNROLL for (int j = 0; j < i; j++) M.l[tri(i, j)] = dot6(kin.cdof[j], buf);
      LUNROLL for (int k = 0; k < kTD; k++) M.b[i][k] = k < L.nanc ? dot6(kin.cdofT[k], buf) : R(0);
    }
    LUNROLL for (int k = 0; k < kTD; k++) {
      R buf[6];
      mul_inert(buf, crbT[trunk_dof_body(k)], kin.cdofT[k]);
      const R arm = k < 6 ? m.tarm[k] : m.tjnt[k - 6].armature;
      M.t[tri(k, k)] = arm + dot6(kin.cdofT[k], buf);
      LUNROLL for (int l = 0; l < k; l++) M.t[tri(k, l)] = dot6(kin.cdofT[l], buf);
    }
    store_arrow(ms, M);
  }
  LPROF(pa, prof_last, 21);
  // ================= velocity stage (o_comvel): body velocities, cdof_dot; bias forces (o_rne); passive and actuator forces
  R fs_l[kLD], fs_t[kTD];
  R VTq[kTB][6], VLq[kLB][6];
  {
    R cddT[kTD][6], cdd[kLD][6], cv[6];
    LUNROLL for (int c = 0; c < 6; c++) cv[c] = 0;
    LUNROLL for (int k = 0; k < 3; k++) { LUNROLL for (int c = 0; c < 6; c++) { cddT[k][c] = 0; cv[c] += kin.cdofT[k][c] * S.tv[k]; } }
    LUNROLL for (int k = 3; k < 6; k++) cross_motion(cddT[k], cv, kin.cdofT[k]);
    LUNROLL for (int k = 3; k < 6; k++) LUNROLL for (int c = 0; c < 6; c++) cv[c] += kin.cdofT[k][c] * S.tv[k];
    LUNROLL for (int c = 0; c < 6; c++) VTq[0][c] = cv[c];
    LUNROLL for (int k = 6; k < kTD; k++) {
      cross_motion(cddT[k], cv, kin.cdofT[k]);
      LUNROLL for (int c = 0; c < 6; c++) cv[c] += kin.cdofT[k][c] * S.tv[k];
      if (k == 7) { LUNROLL for (int c = 0; c < 6; c++) VTq[1][c] = cv[c]; }
      if (k == 8) { LUNROLL for (int c = 0; c < 6; c++) VTq[2][c] = cv[c]; }
    }
    LUNROLL for (int c = 0; c < 6; c++) cv[c] = L.attach == 0 ? VTq[0][c] : (L.attach == 1 ? VTq[1][c] : VTq[2][c]);
    LUNROLL for (int j = 0; j < kLD; j++) {
      cross_motion(cdd[j], cv, kin.cdof[j]);
      LUNROLL for (int c = 0; c < 6; c++) cv[c] += kin.cdof[j][c] * S.lv[j];
      if (j == 2) { LUNROLL for (int c = 0; c < 6; c++) VLq[0][c] = cv[c]; }
      if (j == 3) { LUNROLL for (int c = 0; c < 6; c++) VLq[1][c] = cv[c]; }
      if (j == 5) { LUNROLL for (int c = 0; c < 6; c++) VLq[2][c] = cv[c]; }
    }
    // RNE with zero acceleration
    R caT[kTB][6], cfT[kTB][6], ca[6], cf[kLB][6];
    LUNROLL for (int c = 0; c < 6; c++) ca[c] = c < 3 ? R(0) : -m.gravity[c - 3];
    LUNROLL for (int i = 0; i < kTB; i++) {
      LUNROLL for (int k = 0; k < kTD; k++) if (trunk_dof_body(k) == i) { LUNROLL for (int c = 0; c < 6; c++) ca[c] += cddT[k][c] * S.tv[k]; }
      LUNROLL for (int c = 0; c < 6; c++) caT[i][c] = ca[c];
      R t1[6], t2[6], t3[6];
      mul_inert(t1, cinT[i], ca); mul_inert(t2, cinT[i], VTq[i]); cross_force(t3, VTq[i], t2);
      LUNROLL for (int c = 0; c < 6; c++) cfT[i][c] = t1[c] + t3[c];
    }
    LUNROLL for (int c = 0; c < 6; c++) ca[c] = L.attach == 0 ? caT[0][c] : (L.attach == 1 ? caT[1][c] : caT[2][c]);
    LUNROLL for (int b = 0; b < kLB; b++) {
      LUNROLL for (int j = 0; j < kLD; j++) if (slot_body(j) == b) { LUNROLL for (int c = 0; c < 6; c++) ca[c] += cdd[j][c] * S.lv[j]; }
      R t1[6], t2[6], t3[6];
      mul_inert(t1, cin[b], ca); mul_inert(t2, cin[b], VLq[b]); cross_force(t3, VLq[b], t2);
      LUNROLL for (int c = 0; c < 6; c++) cf[b][c] = t1[c] + t3[c];
    }
    LUNROLL for (int c = 0; c < 6; c++) { cf[1][c] += cf[2][c]; cf[0][c] += cf[1][c]; }
    LUNROLL for (int i = kTB - 1; i >= 0; i--)
      LUNROLL for (int c = 0; c < 6; c++) {
        if (m.nattach[i] > 0) cfT[i][c] += qd_sum(L.attach == i ? cf[0][c] : R(0));
        if (i < kTB - 1) cfT[i][c] += cfT[i + 1][c];
      }
    LUNROLL for (int j = 0; j < kLD; j++) {
      const LJointT<R>& J = L.jnt[j];
      const R bias = dot6(kin.cdof[j], cf[slot_body(j)]);
      const R passive = -J.stiffness * (S.lq[j] - J.qspring) - J.damping * S.lv[j];
      fs_l[j] = J.on ? passive - bias + J.gear_gain * ctrl[j] : R(0);
    }
    LUNROLL for (int k = 0; k < kTD; k++) {
      const R bias = dot6(kin.cdofT[k], cfT[trunk_dof_body(k)]);
      if (k < 6) fs_t[k] = -m.tdamp[k] * S.tv[k] - bias;
      else {
        const LJointT<R>& J = m.tjnt[k - 6];
        fs_t[k] = J.on ? -J.stiffness * (S.tq[1 + k] - J.qspring) - J.damping * S.tv[k] - bias + J.gear_gain * tctrl[k - 6] : R(0);
      }
    }
  }
  LPROF(pa, prof_last, 22);
  LUNROLL for (int j = 0; j < kLD; j++) { D.fs_l[j] = fs_l[j]; D.sl[j] = fs_l[j]; }
  LUNROLL for (int k = 0; k < kTD; k++) { D.fs_t[k] = fs_t[k]; D.st[k] = fs_t[k]; }
  // ================= sites (tracking markers), traces
  LUNROLL for (int s = 0; s < kLS; s++) {
    const LSiteT<R>& St = L.site[s];
    const int b = St.body;
    R bp[3], bm[9], cv[6];
    R pw[6];
    pick6_weights(b, pw);
    LUNROLL for (int k = 0; k < 3; k++) bp[k] = pick6(pw, xpos[0][k], xpos[1][k], xpos[2][k], tpos[0][k], tpos[1][k], tpos[2][k]);
    LUNROLL for (int k = 0; k < 9; k++) bm[k] = pick6(pw, xmat[0][k], xmat[1][k], xmat[2][k], tmat[0][k], tmat[1][k], tmat[2][k]);
    LUNROLL for (int k = 0; k < 6; k++) cv[k] = pick6(pw, VLq[0][k], VLq[1][k], VLq[2][k], VTq[0][k], VTq[1][k], VTq[2][k]);
    R v[3], off[3], lin[3], sp[3];
    mv3(v, bm, St.pos);
    LUNROLL for (int k = 0; k < 3; k++) { sp[k] = bp[k] + v[k]; off[k] = sp[k] - com[k]; out.spos[s][k] = sp[k]; }
    cr3(lin, cv, off);
    LUNROLL for (int k = 0; k < 3; k++) out.svel[s][k] = cv[3 + k] + lin[k];
  }
  LXPROF(pa, prof_last, 35);
  LUNROLL for (int q = 0; q < kMaxTrace; q++) {
    const LTraceT<R>& T = m.trace[q];
    const int b = T.lane == 4 ? 3 + T.body : T.body;
    R bp[3], bm[9], v[3];
    R pw[6];
    pick6_weights(b, pw);
    LUNROLL for (int k = 0; k < 3; k++) bp[k] = pick6(pw, xpos[0][k], xpos[1][k], xpos[2][k], tpos[0][k], tpos[1][k], tpos[2][k]);
    LUNROLL for (int k = 0; k < 9; k++) bm[k] = pick6(pw, xmat[0][k], xmat[1][k], xmat[2][k], tmat[0][k], tmat[1][k], tmat[2][k]);
    mv3(v, bm, T.pos);
    LUNROLL for (int k = 0; k < 3; k++) out.trace[q][k] = bp[k] + v[k];
  }
  LPROF(pa, prof_last, 24);
  // ================= constraint rows: limits of the lane's joints, of its tendon, of the trunk's hinges
  LRows<R>& Rw = D.rows;
  LUNROLL for (int j = 0; j < kLD; j++) {
    const LJointT<R>& J = L.jnt[j];
    Rw.lm_side[j] = 0; Rw.lm_D[j] = 0; Rw.lm_jar[j] = 0;
    if (!J.on || !J.limited) continue;
    const R value = S.lq[j], lo = value - J.range[0], hi = J.range[1] - value;
    if (lo < J.margin && hi < J.margin) flags |= kFlagLimits;
    if (lo < J.margin) { Rw.lm_side[j] = -1; limit_row(lo, J.margin, S.lv[j], J.invw, J.lim_k, J.lim_b, J.lim_imp, Rw.lm_D[j], Rw.lm_jar[j]); }
    else if (hi < J.margin) { Rw.lm_side[j] = 1; limit_row(hi, J.margin, -S.lv[j], J.invw, J.lim_k, J.lim_b, J.lim_imp, Rw.lm_D[j], Rw.lm_jar[j]); }
  }
  LUNROLL for (int h = 0; h < 3; h++) {
    const LJointT<R>& J = m.tjnt[h];
    Rw.tl_side[h] = 0; Rw.tl_D[h] = 0; Rw.tl_jar[h] = 0;
    if (!J.on || !J.limited) continue;
    const R value = S.tq[7 + h], lo = value - J.range[0], hi = J.range[1] - value;
    if (lo < J.margin && hi < J.margin) flags |= kFlagLimits;
    if (lo < J.margin) { Rw.tl_side[h] = -1; limit_row(lo, J.margin, S.tv[6 + h], J.invw, J.lim_k, J.lim_b, J.lim_imp, Rw.tl_D[h], Rw.tl_jar[h]); }
    else if (hi < J.margin) { Rw.tl_side[h] = 1; limit_row(hi, J.margin, -S.tv[6 + h], J.invw, J.lim_k, J.lim_b, J.lim_imp, Rw.tl_D[h], Rw.tl_jar[h]); }
  }
  Rw.tn_side = 0; Rw.tn_D = 0; Rw.tn_jar = 0;
  if (L.tendon.on) {
    const LTendonT<R>& T = L.tendon;
    R value = 0, vel = 0;
    LUNROLL for (int j = 0; j < kLD; j++) {
      const R c = (j == T.slot[0] ? T.coef[0] : R(0)) + (j == T.slot[1] ? T.coef[1] : R(0));
      value += c * S.lq[j]; vel += c * S.lv[j];
    }
    const R lo = value - T.range[0], hi = T.range[1] - value;
    if (lo < T.margin && hi < T.margin) flags |= kFlagLimits;
    if (lo < T.margin) { Rw.tn_side = -1; limit_row(lo, T.margin, vel, T.invw, T.k, T.b, T.imp, Rw.tn_D, Rw.tn_jar); }
    else if (hi < T.margin) { Rw.tn_side = 1; limit_row(hi, T.margin, -vel, T.invw, T.k, T.b, T.imp, Rw.tn_D, Rw.tn_jar); }
  }
  LPROF(pa, prof_last, 25);
  // ================= collision: the lane's geoms against the floor; their poses into the quad's shared table
  int ncon = 0;
  auto floor_test = [&](const LGeomT<R>& G, bool trunk_geom, const R* bp, const R* bm, const R* bvel) {
    R v[3], gp[3], ga[3];
    mv3(v, bm, G.pos);
    mv3(ga, bm, G.axis);
    LUNROLL for (int k = 0; k < 3; k++) gp[k] = bp[k] + v[k];
    lsh_set_geom(sh, G.gslot, gp, ga);
    if (G.pdim == 0) return;
    const LPSetT<R>& P = m.pset[G.pset];
    LUNROLL for (int e = 0; e < 2; e++) {
      if (e == 1 && G.half == 0) continue;
      const R sg = G.half == 0 ? R(0) : (e == 0 ? R(-1) : R(1));
      R c[3];
      LUNROLL for (int k = 0; k < 3; k++) c[k] = gp[k] + sg * G.half * ga[k];
      const R dist = (c[0] - m.plane_pos[0]) * m.plane_n[0] + (c[1] - m.plane_pos[1]) * m.plane_n[1] + (c[2] - m.plane_pos[2]) * m.plane_n[2] - G.radius;
      if (!(dist < P.margin)) continue;
      if (trunk_geom) { flags |= kFlagTrunkFloor; continue; }
      if (ncon >= kMaxPC) { flags |= kFlagOverflow; continue; }
      LContact<R> C;
      LUNROLL for (int k = 0; k < 3; k++) C.off[k] = c[k] - m.plane_n[k] * (G.radius + R(0.5) * dist) - com[k];
      C.body = G.body; C.mu = G.pmu; C.nrow = G.pdim == 1 ? 1 : 4;
      const R x = dist - P.includemargin, im = impedance(P.imp, x);
      // mj_makeImpedance: a pyramid edge's R from diag (1 + mu^2), then every edge gets 2 mu^2 R of the first
      R R0 = (1 - im) / im * G.pdiag * (G.pdim == 1 ? R(1) : 1 + G.pmu * G.pmu);
      if (R0 < R(1e-15)) R0 = R(1e-15);
      if (G.pdim != 1) { R0 = 2 * G.pmu * G.pmu * R0; if (R0 < R(1e-15)) R0 = R(1e-15); }
      C.D = R(1) / R0;
      R w[3], pv[3];
      cr3(w, bvel, C.off);
      LUNROLL for (int k = 0; k < 3; k++) pv[k] = bvel[3 + k] + w[k];
      const R vn = dot3(m.plane_n, pv), v1 = dot3(m.plane_t1, pv), v2 = dot3(m.plane_t2, pv), kx = P.k * im * x;
      if (G.pdim == 1) { C.jar[0] = P.b * vn + kx; C.jar[1] = C.jar[2] = C.jar[3] = 0; }
      else {
        C.jar[0] = P.b * (vn + G.pmu * v1) + kx; C.jar[1] = P.b * (vn - G.pmu * v1) + kx;
        C.jar[2] = P.b * (vn + G.pmu * v2) + kx; C.jar[3] = P.b * (vn - G.pmu * v2) + kx;
      }
      lcs_store(cs, ncon, C);
      ncon++;
    }
  };
  LUNROLL for (int g = 0; g < kLG; g++) {
    const LGeomT<R>& G = L.geom[g];
    if (!G.on) continue;
    const int b = G.body;
    R bp[3], bm[9], bv[6];
    LUNROLL for (int k = 0; k < 3; k++) bp[k] = b == 0 ? xpos[0][k] : (b == 1 ? xpos[1][k] : xpos[2][k]);
    LUNROLL for (int k = 0; k < 9; k++) bm[k] = b == 0 ? xmat[0][k] : (b == 1 ? xmat[1][k] : xmat[2][k]);
    LUNROLL for (int k = 0; k < 6; k++) bv[k] = b == 0 ? VLq[0][k] : (b == 1 ? VLq[1][k] : VLq[2][k]);
    floor_test(G, false, bp, bm, bv);
  }
  LUNROLL for (int g = 0; g < kTGL; g++) {
    const LGeomT<R>& G = L.tgeom[g];
    if (!G.on) continue;
    const int b = G.body;
    R bp[3], bm[9], bv[6];
    LUNROLL for (int k = 0; k < 3; k++) bp[k] = b == 0 ? tpos[0][k] : (b == 1 ? tpos[1][k] : tpos[2][k]);
    LUNROLL for (int k = 0; k < 9; k++) bm[k] = b == 0 ? tmat[0][k] : (b == 1 ? tmat[1][k] : tmat[2][k]);
    LUNROLL for (int k = 0; k < 6; k++) bv[k] = b == 0 ? VTq[0][k] : (b == 1 ? VTq[1][k] : VTq[2][k]);
    floor_test(G, true, bp, bm, bv);
  }
  D.ncon = ncon;
  ld_sync();
  LPROF(pa, prof_last, 26);
  // ================= self-collision (oracle pair_collide over the baked moving-geom pairs): the lane's share of the pairs
  int nmine = 0;
  R xb_n[kMaxX][3], xb_p[kMaxX][3], xb_d[kMaxX]; int xb_i[kMaxX];
  LUNROLL for (int i = 0; i < kMaxX; i++) { xb_d[i] = 0; xb_i[i] = 0; LUNROLL for (int k = 0; k < 3; k++) { xb_n[i][k] = 0; xb_p[i][k] = 0; } }
  auto ball_pair = [&](int pi, const R* c1, R r1, const R* c2, R r2, R margin) {  // oracle sphere_vs_sphere + add_contact
    R n[3], len = 0;
    LUNROLL for (int k = 0; k < 3; k++) { n[k] = c2[k] - c1[k]; len += n[k] * n[k]; }
    len = sqrt(len);
    if (len < R(1e-15)) { n[0] = 1; n[1] = n[2] = 0; } else { LUNROLL for (int k = 0; k < 3; k++) n[k] /= len; }
    const R dist = len - r1 - r2;
    if (!(dist < margin)) return;
    if (nmine >= kMaxX) { flags |= kFlagCross; return; }
    LUNROLL for (int i = 0; i < kMaxX; i++) if (i == nmine) {
      xb_d[i] = dist; xb_i[i] = pi;
      LUNROLL for (int k = 0; k < 3; k++) { xb_n[i][k] = n[k]; xb_p[i][k] = c1[k] + n[k] * (r1 + R(0.5) * dist); }
    }
    nmine++;
  };
  // (the bounding-sphere filter of mj_collideGeoms for FOUR pairs at a time -- their centres' loads in flight together; one pair per pass had
  // every pass wait for its own LDS round trip, 940 cycles a pair -- then the narrow phase, one instance, for the pairs that passed)
  for (int q0 = 0; q0 < kPairsPerLane; q0 += 4) {
    if (q0 >= L.npair) break;
    int hits = 0;
    LUNROLL for (int u = 0; u < 4; u++) {
      const int qq = q0 + u < L.npair ? q0 + u : L.npair - 1;
      const LPairT<R>& Pq = m.pair[L.pair0 + qq];
      R c1[3], c2[3], ax[3];
      lsh_get_geom(sh, Pq.ga, c1, ax);
      lsh_get_geom(sh, Pq.gb, c2, ax);
      const R ex = c1[0] - c2[0], ey = c1[1] - c2[1], ez = c1[2] - c2[2];
      if (q0 + u < L.npair && !(ex * ex + ey * ey + ez * ez > Pq.reach * Pq.reach)) hits |= 1 << u;
    }
  while (hits) {
    const int u = __builtin_ctz(hits);
    hits &= hits - 1;
    const int pi = L.pair0 + q0 + u;
    const LPairT<R>& P = m.pair[pi];
    R p1[3], a1[3], p2[3], a2[3];
    lsh_get_geom(sh, P.ga, p1, a1);
    lsh_get_geom(sh, P.gb, p2, a2);
    const R dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
    const R r1 = m.grad[P.ga], r2 = m.grad[P.gb], h1 = m.ghalf[P.ga], h2 = m.ghalf[P.gb], margin = m.pset[P.pset].margin;
    if (h1 == 0 && h2 == 0) ball_pair(pi, p1, r1, p2, r2, margin);
    else if (h1 == 0) {
      R x = (p1[0] - p2[0]) * a2[0] + (p1[1] - p2[1]) * a2[1] + (p1[2] - p2[2]) * a2[2];
      x = x < -h2 ? -h2 : (x > h2 ? h2 : x);
      const R c2[3] = {p2[0] + x * a2[0], p2[1] + x * a2[1], p2[2] + x * a2[2]};
      ball_pair(pi, p1, r1, c2, r2, margin);
    } else {
      const R dif[3] = {dx, dy, dz};
      const R mb = -dot3(a1, a2), u = -dot3(a1, dif), v = dot3(a2, dif), det = 1 - mb * mb;
      R c1[3], c2[3];
      if (fabs(det) >= R(1e-15)) {
        R x1 = (u - mb * v) / det, x2 = (v - mb * u) / det;
        if (x1 > h1) { x1 = h1; x2 = v - mb * x1; } else if (x1 < -h1) { x1 = -h1; x2 = v - mb * x1; }
        if (x2 > h2) { x2 = h2; x1 = u - mb * x2; x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1); }
        else if (x2 < -h2) { x2 = -h2; x1 = u - mb * x2; x1 = x1 > h1 ? h1 : (x1 < -h1 ? -h1 : x1); }
        LUNROLL for (int k = 0; k < 3; k++) { c1[k] = p1[k] + x1 * a1[k]; c2[k] = p2[k] + x2 * a2[k]; }
        ball_pair(pi, c1, r1, c2, r2, margin);
      } else {  // parallel axes: the ends of capsule 1 against axis 2, then the ends of capsule 2 against axis 1, two contacts at most
        const int before = nmine;
        for (int e = 0; e < 4 && nmine - before < 2; e++) {
          const R sgn = (e & 1) ? R(-1) : R(1);
          if (e < 2) {
            LUNROLL for (int k = 0; k < 3; k++) c1[k] = p1[k] + sgn * h1 * a1[k];
            R x2 = (c1[0] - p2[0]) * a2[0] + (c1[1] - p2[1]) * a2[1] + (c1[2] - p2[2]) * a2[2];
            x2 = x2 < -h2 ? -h2 : (x2 > h2 ? h2 : x2);
            LUNROLL for (int k = 0; k < 3; k++) c2[k] = p2[k] + x2 * a2[k];
          } else {
            LUNROLL for (int k = 0; k < 3; k++) c2[k] = p2[k] + sgn * h2 * a2[k];
            R x1 = (c2[0] - p1[0]) * a1[0] + (c2[1] - p1[1]) * a1[1] + (c2[2] - p1[2]) * a1[2];
            x1 = x1 < -h1 ? -h1 : (x1 > h1 ? h1 : x1);
            LUNROLL for (int k = 0; k < 3; k++) c1[k] = p1[k] + x1 * a1[k];
          }
          ball_pair(pi, c1, r1, c2, r2, margin);
        }
      }
    }
  }
  }
  LPROF(pa, prof_last, 27);
  // the quad's list: lane by lane, each lane's contacts in its pairs' order
  int nx = 0;
  {
    const int c0 = qd_bcasti<0>(nmine), c1 = qd_bcasti<1>(nmine), c2 = qd_bcasti<2>(nmine), c3 = qd_bcasti<3>(nmine);
    nx = c0 + c1 + c2 + c3;
    if (nx > kMaxX) flags |= kFlagCross;
    const int base = lane == 0 ? 0 : (lane == 1 ? c0 : (lane == 2 ? c0 + c1 : c0 + c1 + c2));
    if (qw_any(nx > 0)) {
      if (nx <= kMaxX) {
        LUNROLL for (int i = 0; i < kMaxX; i++) {
          if (i >= nmine) continue;
          const LPairT<R>& P = m.pair[xb_i[i]];
          const LPSetT<R>& PS = m.pset[P.pset];
          LCross<R> C;
          const R off[3] = {xb_p[i][0] - com[0], xb_p[i][1] - com[1], xb_p[i][2] - com[2]};
          cr3(C.et, off, xb_n[i]);
          LUNROLL for (int k = 0; k < 3; k++) C.et[3 + k] = xb_n[i][k];
          const R x = xb_d[i] - PS.includemargin, im = impedance(PS.imp, x);
          R R0 = (1 - im) / im * P.diag;
          if (R0 < R(1e-15)) R0 = R(1e-15);
          C.D = R(1) / R0; C.b = PS.b; C.kimpx = PS.k * im * x;
          C.la = m.glane[P.ga]; C.sa = m.gbody[P.ga]; C.lb = m.glane[P.gb]; C.sb = m.gbody[P.gb];
          lsh_set_cross(sh, base + i, C);
        }
      }
      ld_sync();
    }
  }
  LPROF(pa, prof_last, 28);
  if (nx > kMaxX) nx = 0;
  D.nx = nx;
  // ================= qacc_smooth = M^-1 qfrc_smooth: LAST -- the factor's hundred and twenty numbers push whatever else is alive out of the
  // registers, and the sites, rows and collision stages above all read the body frames (with the factor in its place after the bias forces
  // the four sites alone took 22 k cycles a step re-fetching them)
  {
    Arrow<R> F;
    load_arrow(ms, F);
    if (!arrow_factor(F)) flags |= kFlagNotPD;
    arrow_solve(F, D.sl, D.st);
  }
  LPROF(pa, prof_last, 23);
  flags = qd_or(flags);
  return flags;
}

template <typename R, class CS, class MS, class SH, class KS>
LNOINLINE int forward_smooth(const LimbModelT<R>& m_in, int lane, const LState<R>* S_in, const R* ctrl_in, const R* tctrl_in, CS cs, MS ms, SH sh, KS ks,
                             LDyn<R>* D_out, LSense<R>* out_out, long long* stamps) {
  const LimbModelT<R>& m = LREBIND_LDS(LimbModelT<R>, m_in);
  if (!(LEXP_COPY_MASK & 1))
    return forward_smooth_body(m, lane, *LREBIND_PRV(const LState<R>, S_in), LREBIND_PRV(const R, ctrl_in), LREBIND_PRV(const R, tctrl_in), cs, ms, sh, ks,
                               *LREBIND_PRV(LDyn<R>, D_out), *LREBIND_PRV(LSense<R>, out_out), stamps);
  LState<R> S;
  LPRV_LOAD(S, S_in);
  R ctrl[kLD], tctrl[3];
  LPRV_LOADN(ctrl, ctrl_in, kLD); LPRV_LOADN(tctrl, tctrl_in, 3);
  LDyn<R> D;
  LSense<R> out;
  LPOISON(D); LPOISON(out);
  const int flags = forward_smooth_body(m, lane, S, ctrl, tctrl, cs, ms, sh, ks, D, out, stamps);
  LPRV_STORE(D_out, D); LPRV_STORE(out_out, out);
  return flags;
}

// the value J_r x of the contact between moving geoms `C` for a dof vector with relative chain velocities VR and trunk part xt
template <typename R, class KIN> LD R cross_value(const LimbModelT<R>& m, const KIN& kin, const LCross<R>& C, int lane, const R VR[kLB][6], const R* xt) {
  R own = 0, v[6];
  if (C.la == lane) { pick3(VR, C.sa, v); own -= dot6(C.et, v); }
  if (C.lb == lane) { pick3(VR, C.sb, v); own += dot6(C.et, v); }
  const int na = C.la < 4 ? m.limb[C.la & 3].nanc : (C.sa == 0 ? 6 : (C.sa == 1 ? 8 : 9)), nb = C.lb < 4 ? m.limb[C.lb & 3].nanc : (C.sb == 0 ? 6 : (C.sb == 1 ? 8 : 9));
  R tr = 0;
  LUNROLL for (int k = 6; k < kTD; k++) { const int coef = (k < nb ? 1 : 0) - (k < na ? 1 : 0); if (coef != 0) tr += R(coef) * dot6(C.et, kin.cdofT[k]) * xt[k]; }
  return qd_sum(own) + tr;
}
// the row as a distributed vector u (ul: the lane's limb part, ut: the trunk part, replicated)
template <typename R, class KIN> LD void cross_vector(const LimbModelT<R>& m, const KIN& kin, const LCross<R>& C, int lane, R* ul, R* ut) {
  LUNROLL for (int j = 0; j < kLD; j++) {
    R s = 0;
    if (C.la == lane && slot_body(j) <= C.sa) s -= dot6(C.et, kin.cdof[j]);
    if (C.lb == lane && slot_body(j) <= C.sb) s += dot6(C.et, kin.cdof[j]);
    ul[j] = s;
  }
  const int na = C.la < 4 ? m.limb[C.la & 3].nanc : (C.sa == 0 ? 6 : (C.sa == 1 ? 8 : 9)), nb = C.lb < 4 ? m.limb[C.lb & 3].nanc : (C.sb == 0 ? 6 : (C.sb == 1 ? 8 : 9));
  LUNROLL for (int k = 0; k < kTD; k++) { const int coef = k < 6 ? 0 : (k < nb ? 1 : 0) - (k < na ? 1 : 0); ut[k] = coef != 0 ? R(coef) * dot6(C.et, kin.cdofT[k]) : R(0); }
}

// ---------------------------------------------------------------- constraint solve (oracle o_constraint_newton)
// One pass over the candidate's rows: jar += alpha J x (x = (xl, xt)), then the penalty at jar: returns the cost of ALL rows (the same in
// the four lanes), J' force in jl (the lane's limb dofs) / jt (trunk dofs, replicated).
template <typename R, class CS, class SH, class KIN>
LD R rows_eval(const LimbModelT<R>& m, const LimbT<R>& L, int lane, const KIN& kin, LRows<R>& Rw, CS& cs, int ncon, SH& sh, int nx, bool step,
               const R* xl, const R* xt, R alpha, R* jl, R* jt, const R (*VLin)[6] = nullptr, const R (*VRin)[6] = nullptr) {
  // (VLin / VRin: the bodies' velocities along (xl, xt) where the caller has them already -- the line search's, for the step that follows it)
  R VT[kTB][6], VL[kLB][6], VR[kLB][6];
  if (VLin) { LUNROLL for (int b = 0; b < kLB; b++) LUNROLL for (int c = 0; c < 6; c++) { VL[b][c] = VLin[b][c]; VR[b][c] = VRin[b][c]; } }
  else if (step) chain_velocity(kin, L.attach, xl, xt, VT, VL, VR);
  R cost = 0, costT = 0;
  R Fb[kLB][6], Fp[6];
  LUNROLL for (int b = 0; b < kLB; b++) LUNROLL for (int c = 0; c < 6; c++) Fb[b][c] = 0;
  LUNROLL for (int c = 0; c < 6; c++) Fp[c] = 0;
  LUNROLL for (int j = 0; j < kLD; j++) {
    if (step) Rw.lm_jar[j] += alpha * (-Rw.lm_side[j]) * xl[j];
    R f = 0;
    if (Rw.lm_side[j] != 0) row_pen(Rw.lm_jar[j], Rw.lm_D[j], cost, f);
    jl[j] = -Rw.lm_side[j] * f;
  }
  if (L.tendon.on) {
    const LTendonT<R>& T = L.tendon;
    R jv = 0;
    LUNROLL for (int j = 0; j < kLD; j++) jv += ((j == T.slot[0] ? T.coef[0] : R(0)) + (j == T.slot[1] ? T.coef[1] : R(0))) * (step ? xl[j] : R(0));
    if (step) Rw.tn_jar += alpha * (-Rw.tn_side) * jv;
    R f = 0;
    if (Rw.tn_side != 0) row_pen(Rw.tn_jar, Rw.tn_D, cost, f);
    LUNROLL for (int j = 0; j < kLD; j++) jl[j] += -Rw.tn_side * f * ((j == T.slot[0] ? T.coef[0] : R(0)) + (j == T.slot[1] ? T.coef[1] : R(0)));
  }
  LUNROLL for (int k = 0; k < kTD; k++) jt[k] = 0;
  LUNROLL for (int h = 0; h < 3; h++) {
    if (step) Rw.tl_jar[h] += alpha * (-Rw.tl_side[h]) * xt[6 + h];
    R f = 0;
    if (Rw.tl_side[h] != 0) row_pen(Rw.tl_jar[h], Rw.tl_D[h], costT, f);
    jt[6 + h] = -Rw.tl_side[h] * f;
  }
  for (int i = 0; i < ncon; i++) {
    LContact<R> C;
    lcs_load(cs, i, C);
    if (step) {
      R V[6], w[3], pv[3];
      pick3(VL, C.body, V);
      cr3(w, V, C.off);
      LUNROLL for (int k = 0; k < 3; k++) pv[k] = V[3 + k] + w[k];
      const R vn = dot3(m.plane_n, pv), v1 = C.mu * dot3(m.plane_t1, pv), v2 = C.mu * dot3(m.plane_t2, pv);
      if (C.nrow == 1) C.jar[0] += alpha * vn;
      else { C.jar[0] += alpha * (vn + v1); C.jar[1] += alpha * (vn - v1); C.jar[2] += alpha * (vn + v2); C.jar[3] += alpha * (vn - v2); }
      lcs_store_jar(cs, i, C);
    }
    R f[4] = {0, 0, 0, 0};
    LUNROLL for (int e = 0; e < 4; e++) if (e < C.nrow) row_pen(C.jar[e], C.D, cost, f[e]);
    // J' force: edge e pulls along n +- mu t
    const R fn = f[0] + f[1] + f[2] + f[3], f1 = C.mu * (f[0] - f[1]), f2 = C.mu * (f[2] - f[3]);
    R Fl[3], Fa[3];
    LUNROLL for (int k = 0; k < 3; k++) Fl[k] = fn * m.plane_n[k] + (C.nrow == 1 ? R(0) : f1 * m.plane_t1[k] + f2 * m.plane_t2[k]);
    cr3(Fa, C.off, Fl);
    LUNROLL for (int b = 0; b < kLB; b++) {
      const R w = C.body == b ? R(1) : R(0);
      LUNROLL for (int k = 0; k < 3; k++) { Fb[b][k] += w * Fa[k]; Fb[b][3 + k] += w * Fl[k]; }
    }
    LUNROLL for (int k = 0; k < 3; k++) { Fp[k] += Fa[k]; Fp[3 + k] += Fl[k]; }
  }
  for (int r = 0; r < nx; r++) {  // (quad-uniform trip count; a rolled loop: one instance of the row's code for the eight slots)
    LCross<R> C;
    lsh_get_cross(sh, r, C);
    R jar = lsh_xget(sh, r, 10);
    if (step) { jar += alpha * cross_value(m, kin, C, lane, VR, xt); lsh_xset(sh, r, 10, jar); }
    R f = 0;
    row_pen(jar, C.D, costT, f);
    const int na = C.la < 4 ? m.limb[C.la & 3].nanc : (C.sa == 0 ? 6 : (C.sa == 1 ? 8 : 9)), nb = C.lb < 4 ? m.limb[C.lb & 3].nanc : (C.sb == 0 ? 6 : (C.sb == 1 ? 8 : 9));
    LUNROLL for (int k = 6; k < kTD; k++) { const int coef = (k < nb ? 1 : 0) - (k < na ? 1 : 0); jt[k] += R(coef) * f * dot6(C.et, kin.cdofT[k]); }
    LUNROLL for (int b = 0; b < kLB; b++) {
      const R w = (C.lb == lane && C.sb == b ? f : R(0)) - (C.la == lane && C.sa == b ? f : R(0));
      LUNROLL for (int c = 0; c < 6; c++) Fb[b][c] += w * C.et[c];
    }
  }
  // limb dofs: the forces on the bodies at or below the dof's
  LUNROLL for (int c = 0; c < 6; c++) { Fb[1][c] += Fb[2][c]; Fb[0][c] += Fb[1][c]; }
  LUNROLL for (int j = 0; j < kLD; j++) jl[j] += dot6(kin.cdof[j], Fb[slot_body(j)]);
  // trunk dofs: the floor's forces on the limbs, through the bodies they hang on
  if (qw_any(ncon > 0)) {
    R FT[kTB][6];
    LUNROLL for (int i = kTB - 1; i >= 0; i--)
      LUNROLL for (int c = 0; c < 6; c++) {
        R v = i < kTB - 1 ? FT[i + 1][c] : R(0);
        if (m.nattach[i] > 0) v += qd_sum(L.attach == i ? Fp[c] : R(0));
        FT[i][c] = v;
      }
    LUNROLL for (int k = 0; k < kTD; k++) jt[k] += trunk_dot(kin, k, FT[trunk_dof_body(k)]);
  }
  return qd_sum(cost) + costT;
}

// the exact line search's view of a floor contact: the rows' values at the start and their rates along the direction
template <typename R> struct LLine { R x0[4], v[4], D; };

// Newton solver. (sl, st) = qacc_smooth, (wl, wt) = warm start, M in the store `ms`; leaves qacc in (al, at) and J' force in (fc_l, fc_t).
// Returns the flag bits (quad-uniform).
template <typename R, class CS, class MS, class SH, class KIN>
LD int newton_body(const LimbModelT<R>& m, int lane, const KIN& kin, const MS& ms, LRows<R>& Rw, CS& cs, int ncon, SH& sh, int nx,
              const R* sl, const R* st, const R* wl, const R* wt, bool have_warm, const R* qvl, const R* qvt,
              R* al, R* at, R* fc_l, R* fc_t, int& iters, long long* stamps) {
  const LimbT<R>& L = m.limb[lane];
  iters = 0;
  struct { long long* stamps; } pa{stamps};
  long long prof_last = 0;
  LPROF(pa, prof_last, -1);
  LUNROLL for (int j = 0; j < kLD; j++) al[j] = sl[j];
  LUNROLL for (int k = 0; k < kTD; k++) at[k] = st[k];
  // the contacts between moving geoms: D and -aref of their rows (aref needs J qvel: a quad sum per row)
  if (nx > 0) {
    R VT[kTB][6], VL[kLB][6], VR[kLB][6];
    chain_velocity(kin, L.attach, qvl, qvt, VT, VL, VR);
    for (int r = 0; r < nx; r++) {
      LCross<R> C;
      lsh_get_cross(sh, r, C);
      lsh_xset(sh, r, 10, C.b * cross_value(m, kin, C, lane, VR, qvt) + C.kimpx);
    }
  }
  R Mal[kLD], Mat[kTD];
  LUNROLL for (int j = 0; j < kLD; j++) Mal[j] = 0;
  LUNROLL for (int k = 0; k < kTD; k++) Mat[k] = 0;
  R cost = rows_eval(m, L, lane, kin, Rw, cs, ncon, sh, nx, true, al, at, R(1), fc_l, fc_t);  // jar = J qacc_smooth - aref; the Gauss term is zero here
  if (have_warm) {
    R dl[kLD], dt[kTD], Ml[kLD], Mt[kTD], jl[kLD], jt[kTD];
    LUNROLL for (int j = 0; j < kLD; j++) dl[j] = wl[j] - sl[j];
    LUNROLL for (int k = 0; k < kTD; k++) dt[k] = wt[k] - st[k];
    arrow_mul_s(ms, dl, dt, Ml, Mt);
    const R gauss = R(0.5) * arrow_dot(dl, dt, Ml, Mt);
    const R cw = gauss + rows_eval(m, L, lane, kin, Rw, cs, ncon, sh, nx, true, dl, dt, R(1), jl, jt);
    if (cw < cost) {
      cost = cw;
      LUNROLL for (int j = 0; j < kLD; j++) { al[j] = wl[j]; fc_l[j] = jl[j]; Mal[j] = Ml[j]; }
      LUNROLL for (int k = 0; k < kTD; k++) { at[k] = wt[k]; fc_t[k] = jt[k]; Mat[k] = Mt[k]; }
    } else {
      (void)rows_eval(m, L, lane, kin, Rw, cs, ncon, sh, nx, true, dl, dt, R(-1), jl, jt);  // and back (the first pass's cost and forces are still held)
    }
  }
  const R scale = R(1) / (m.meaninertia * R(m.nv > 1 ? m.nv : 1));
  R improvement = 0;
  LPROF(pa, prof_last, 8);
  for (int iter = 0; iter < m.iterations; iter++) {
    R hl[kLD], ht[kTD];
    LUNROLL for (int j = 0; j < kLD; j++) hl[j] = Mal[j] - fc_l[j];
    LUNROLL for (int k = 0; k < kTD; k++) ht[k] = Mat[k] - fc_t[k];
    const R gnorm = sqrt(arrow_dot(hl, ht, hl, ht));
    if (gnorm == 0) break;
    if (iter > 0 && (scale * improvement < m.tolerance || scale * gnorm < m.tolerance)) break;
    // The working precision's floor under the two tests: a gradient that is the rounding residue of its two parts (M (a - a_smooth) and
    // J' force cancel at the minimum), or a cost that no longer moves by more than its own rounding, cannot be improved on. In double both
    // floors lie far below the tolerance and never fire first (the iterates are the oracle's); in float the tolerance of 1e-8 is below the
    // noise, and without the floors every solve would run a confirming iteration or two on noise.
    if (iter > 0) {
      const R gref = sqrt(arrow_dot(Mal, Mat, Mal, Mat) + arrow_dot(fc_l, fc_t, fc_l, fc_t));
      if (gnorm <= LEXP_GFLOOR * kEps<R>() * gref || improvement <= LEXP_CFLOOR * kEps<R>() * fabs(cost)) break;
    }
    {
      // A = M + J' D J of the rows that keep the arrowhead: limits, the tendon, the floor's contacts through 6 x 6 blocks on the limb's
      // bodies (the composite-rigid-body recursion with the contacts' curvature in place of inertias)
      Arrow<R> H;
      load_arrow(ms, H);
      LUNROLL for (int j = 0; j < kLD; j++) if (Rw.lm_side[j] != 0 && Rw.lm_jar[j] < 0) H.l[tri(j, j)] += Rw.lm_D[j];
      if (L.tendon.on && Rw.tn_side != 0 && Rw.tn_jar < 0) {
        const LTendonT<R>& T = L.tendon;
        LUNROLL for (int i = 0; i < kLD; i++) LUNROLL for (int j = 0; j <= i; j++) {
          const R ci = (i == T.slot[0] ? T.coef[0] : R(0)) + (i == T.slot[1] ? T.coef[1] : R(0)), cj = (j == T.slot[0] ? T.coef[0] : R(0)) + (j == T.slot[1] ? T.coef[1] : R(0));
          H.l[tri(i, j)] += Rw.tn_D * ci * cj;
        }
      }
      LXPROF(pa, prof_last, 32);
      if (qw_any(ncon > 0)) {
        // the contacts' 6 x 6 blocks by body: nearly every contact is on the limb's LAST body (a foot, a hand), so one block is accumulated in
        // the pass over the contacts and the other two bodies get a pass of their own only when some lane of the wavefront holds such a contact
        R X[kLB][21];
        LUNROLL for (int b = 0; b < kLB; b++) LUNROLL for (int e = 0; e < 21; e++) X[b][e] = 0;
        bool shallow = false;
        auto contact_block = [&](const LContact<R>& C, R* Xc) {
          // (no branch per row: the wavefront runs the union of its lanes' rows anyway, and an inactive row enters with weight zero)
          LUNROLL for (int e = 0; e < 4; e++) {
            const R d = (e < C.nrow && C.jar[e] < 0) ? C.D : R(0);
            const R s1 = C.nrow == 1 ? R(0) : (e == 0 ? C.mu : (e == 1 ? -C.mu : R(0))), s2 = C.nrow == 1 ? R(0) : (e == 2 ? C.mu : (e == 3 ? -C.mu : R(0)));
            R a[6];
            LUNROLL for (int k = 0; k < 3; k++) a[3 + k] = m.plane_n[k] + s1 * m.plane_t1[k] + s2 * m.plane_t2[k];
            cr3(a, C.off, a + 3);
            sym6_add_outer(Xc, d, a);
          }
        };
        for (int i = 0; i < ncon; i++) {
          LContact<R> C;
          lcs_load(cs, i, C);
          if (C.body != kLB - 1) { shallow = true; continue; }
          contact_block(C, X[kLB - 1]);
        }
        if (qw_any(shallow)) {
          for (int i = 0; i < ncon; i++) {
            LContact<R> C;
            lcs_load(cs, i, C);
            if (C.body == kLB - 1) continue;
            R Xc[21];
            LUNROLL for (int e = 0; e < 21; e++) Xc[e] = 0;
            contact_block(C, Xc);
            LUNROLL for (int b = 0; b < kLB - 1; b++) { const R w = C.body == b ? R(1) : R(0); LUNROLL for (int e = 0; e < 21; e++) X[b][e] += w * Xc[e]; }
          }
          LUNROLL for (int e = 0; e < 21; e++) { X[1][e] += X[2][e]; X[0][e] += X[1][e]; }
        } else {
          LUNROLL for (int e = 0; e < 21; e++) { X[1][e] = X[2][e]; X[0][e] = X[2][e]; }
        }
        LXPROF(pa, prof_last, 33);
        // (the trunk dofs below the limb's attachment do not move it: weight zero instead of a branch per entry; the free joint's six always do)
        R wanc[kTD];
        LUNROLL for (int k = 0; k < kTD; k++) wanc[k] = (k < 6 || k < L.nanc) ? R(1) : R(0);
        LUNROLL for (int i = 0; i < kLD; i++) {
          R Y[6];
          sym6_mul(Y, X[slot_body(i)], kin.cdof[i]);
          LUNROLL for (int j = 0; j <= i; j++) H.l[tri(i, j)] += dot6(kin.cdof[j], Y);
          LUNROLL for (int k = 0; k < kTD; k++) H.b[i][k] += wanc[k] * trunk_dot(kin, k, Y);
        }
        LXPROF(pa, prof_last, 34);
        R XT[21];
        LUNROLL for (int e = 0; e < 21; e++) XT[e] = 0;
        LUNROLL for (int i = kTB - 1; i >= 0; i--) {
          if (m.nattach[i] > 0) { LUNROLL for (int e = 0; e < 21; e++) XT[e] += qd_sum(L.attach == i ? X[0][e] : R(0)); }
          LUNROLL for (int k = 0; k < kTD; k++) {
            if (trunk_dof_body(k) != i) continue;
            R Y[6];
            trunk_sym6_mul(Y, XT, kin, k);
            LUNROLL for (int l = 0; l <= k; l++) H.t[tri(k, l)] += trunk_dot(kin, l, Y);
          }
        }
      }
      LUNROLL for (int h = 0; h < 3; h++) if (Rw.tl_side[h] != 0 && Rw.tl_jar[h] < 0) H.t[tri(6 + h, 6 + h)] += Rw.tl_D[h];
      LPROF(pa, prof_last, 9);
      if (!arrow_factor(H)) return kFlagNotPD;
      LUNROLL for (int j = 0; j < kLD; j++) hl[j] = -hl[j];
      LUNROLL for (int k = 0; k < kTD; k++) ht[k] = -ht[k];
      arrow_solve(H, hl, ht);
      LPROF(pa, prof_last, 10);
      // the contacts between moving geoms: H = A + sum_r D_r u_r u_r' -> (Woodbury) s = y - Z (D^-1 + U' Z)^-1 U' y, Z = A^-1 U
      if (nx > 0) {
        // (rolled loops over the rows, their vectors in arrays indexed at run time -- private memory: touched only by candidates that have such
        // contacts, one wavefront-step in two; unrolled over the eight slots this block was a quarter of the solver's code and 200 registers)
        R zl[kMaxX][kLD], zt[kMaxX][kTD], Sm[kMaxX][kMaxX], rhs[kMaxX];
        int act = 0;
        for (int r = 0; r < nx; r++) {
          R ul[kLD], ut[kTD];
          LUNROLL for (int j = 0; j < kLD; j++) ul[j] = 0;
          LUNROLL for (int k = 0; k < kTD; k++) ut[k] = 0;
          if (lsh_xget(sh, r, 10) < 0) {  // (quad-uniform: the row's jar is the quad's)
            act |= 1 << r;
            LCross<R> C;
            lsh_get_cross(sh, r, C);
            cross_vector(m, kin, C, lane, ul, ut);
            rhs[r] = arrow_dot(ul, ut, hl, ht);
            // row r of U' Z needs the earlier rows' z: S_rs = u_r . z_s (s < r), S_sr by symmetry; the diagonal after the solve
            for (int q = 0; q < r; q++) { const R v = (act >> q) & 1 ? arrow_dot(ul, ut, zl[q], zt[q]) : R(0); Sm[r][q] = v; Sm[q][r] = v; }
            R dl[kLD], dt[kTD];
            LUNROLL for (int j = 0; j < kLD; j++) dl[j] = ul[j];
            LUNROLL for (int k = 0; k < kTD; k++) dt[k] = ut[k];
            arrow_solve(H, dl, dt);
            Sm[r][r] = arrow_dot(ul, ut, dl, dt) + R(1) / C.D;
            LUNROLL for (int j = 0; j < kLD; j++) zl[r][j] = dl[j];
            LUNROLL for (int k = 0; k < kTD; k++) zt[r][k] = dt[k];
          } else {
            rhs[r] = 0;
            for (int q = 0; q < r; q++) { Sm[r][q] = 0; Sm[q][r] = 0; }
            Sm[r][r] = 1;
            LUNROLL for (int j = 0; j < kLD; j++) zl[r][j] = 0;
            LUNROLL for (int k = 0; k < kTD; k++) zt[r][k] = 0;
          }
        }
        // the small symmetric positive-definite system (inactive rows: identity), Gaussian elimination in place
        for (int j = 0; j < nx; j++) {
          const R inv = R(1) / Sm[j][j];
          for (int i = j + 1; i < nx; i++) {
            const R f = Sm[i][j] * inv;
            for (int k = j; k < nx; k++) Sm[i][k] -= f * Sm[j][k];
            rhs[i] -= f * rhs[j];
          }
        }
        for (int i = nx - 1; i >= 0; i--) {
          R v = rhs[i];
          for (int k = i + 1; k < nx; k++) v -= Sm[i][k] * rhs[k];
          rhs[i] = v / Sm[i][i];
        }
        for (int r = 0; r < nx; r++) {
          const R c = rhs[r];
          LUNROLL for (int j = 0; j < kLD; j++) hl[j] -= c * zl[r][j];
          LUNROLL for (int k = 0; k < kTD; k++) ht[k] -= c * zt[r][k];
        }
      }
    }
    LPROF(pa, prof_last, 11);
    R q1, q2, snorm, Msl[kLD], Mst[kTD];
    arrow_mul_s(ms, hl, ht, Msl, Mst);
    q1 = arrow_dot(hl, ht, Mal, Mat); q2 = arrow_dot(hl, ht, Msl, Mst); snorm = arrow_dot(hl, ht, hl, ht);
    const R gtol = m.tolerance * R(0.01) * sqrt(snorm) / scale;
    LPROF(pa, prof_last, 12);
    // ---- exact line search (oracle: Newton on the derivative in a bracket, rtsafe safeguard)
    R alpha = 0;
    R VT[kTB][6], VL[kLB][6], VR[kLB][6];  // the bodies' velocities along the direction: the search's and the step's after it
    chain_velocity(kin, L.attach, hl, ht, VT, VL, VR);
    {
      // the rows' values and rates: the lane's diagonal rows, its first four floor contacts in registers (further ones from the store)
      R lx0[kLD], lv[kLD], lD[kLD];
      LUNROLL for (int j = 0; j < kLD; j++) { lx0[j] = Rw.lm_jar[j]; lv[j] = -Rw.lm_side[j] * hl[j]; lD[j] = Rw.lm_side[j] != 0 ? Rw.lm_D[j] : R(0); }
      R tnx0 = Rw.tn_jar, tnv = 0, tnD = Rw.tn_side != 0 ? Rw.tn_D : R(0);
      if (L.tendon.on) { LUNROLL for (int j = 0; j < kLD; j++) tnv += ((j == L.tendon.slot[0] ? L.tendon.coef[0] : R(0)) + (j == L.tendon.slot[1] ? L.tendon.coef[1] : R(0))) * hl[j]; tnv *= -Rw.tn_side; }
      R tx0[3], tv[3], tD[3];
      LUNROLL for (int h = 0; h < 3; h++) { tx0[h] = Rw.tl_jar[h]; tv[h] = -Rw.tl_side[h] * ht[6 + h]; tD[h] = Rw.tl_side[h] != 0 ? Rw.tl_D[h] : R(0); }
      R xx0[kMaxX], xv[kMaxX], xDv[kMaxX];
      LUNROLL for (int r = 0; r < kMaxX; r++) { xx0[r] = 0; xv[r] = 0; xDv[r] = 0; }
      for (int r = 0; r < nx; r++) {  // (rolled: the rate of each row along the direction goes through the shared block)
        LCross<R> C;
        lsh_get_cross(sh, r, C);
        lsh_xset(sh, r, 11, cross_value(m, kin, C, lane, VR, ht));
      }
      if (nx > 0) { LUNROLL for (int r = 0; r < kMaxX; r++) if (r < nx) { xx0[r] = lsh_xget(sh, r, 10); xv[r] = lsh_xget(sh, r, 11); xDv[r] = lsh_xget(sh, r, 6); } }
      auto contact_rates = [&](const LContact<R>& C, LLine<R>& q) {
        R V[6], w[3], pv[3];
        pick3(VL, C.body, V);
        cr3(w, V, C.off);
        LUNROLL for (int k = 0; k < 3; k++) pv[k] = V[3 + k] + w[k];
        const R vn = dot3(m.plane_n, pv), v1 = C.mu * dot3(m.plane_t1, pv), v2 = C.mu * dot3(m.plane_t2, pv);
        q.D = C.D;
        LUNROLL for (int e = 0; e < 4; e++) q.x0[e] = e < C.nrow ? C.jar[e] : R(1);
        q.v[0] = C.nrow == 1 ? vn : vn + v1; q.v[1] = C.nrow == 1 ? R(0) : vn - v1; q.v[2] = C.nrow == 1 ? R(0) : vn + v2; q.v[3] = C.nrow == 1 ? R(0) : vn - v2;
      };
      LLine<R> ql[4];
      LUNROLL for (int i = 0; i < 4; i++) {
        ql[i].D = 0;
        LUNROLL for (int e = 0; e < 4; e++) { ql[i].x0[e] = 1; ql[i].v[e] = 0; }
        if (i < ncon) { LContact<R> C; lcs_load(cs, i, C); contact_rates(C, ql[i]); }
      }
      const bool beyond = qw_any(ncon > 4);
      auto derivs = [&](R a, R& d1, R& d2) {
        R g = 0, h = 0, gT = 0, hT = 0;
        LUNROLL for (int j = 0; j < kLD; j++) { const R x = lx0[j] + a * lv[j]; const bool on = x < 0; g += on ? lD[j] * x * lv[j] : R(0); h += on ? lD[j] * lv[j] * lv[j] : R(0); }
        { const R x = tnx0 + a * tnv; const bool on = x < 0; g += on ? tnD * x * tnv : R(0); h += on ? tnD * tnv * tnv : R(0); }
        LUNROLL for (int i = 0; i < 4; i++)
          LUNROLL for (int e = 0; e < 4; e++) { const R x = ql[i].x0[e] + a * ql[i].v[e]; const bool on = x < 0; g += on ? ql[i].D * x * ql[i].v[e] : R(0); h += on ? ql[i].D * ql[i].v[e] * ql[i].v[e] : R(0); }
        if (beyond) {
          for (int i = 4; i < ncon; i++) {
            LContact<R> C; lcs_load(cs, i, C);
            LLine<R> q; contact_rates(C, q);
            LUNROLL for (int e = 0; e < 4; e++) { const R x = q.x0[e] + a * q.v[e]; const bool on = x < 0; g += on ? q.D * x * q.v[e] : R(0); h += on ? q.D * q.v[e] * q.v[e] : R(0); }
          }
        }
        LUNROLL for (int hh = 0; hh < 3; hh++) { const R x = tx0[hh] + a * tv[hh]; const bool on = x < 0; gT += on ? tD[hh] * x * tv[hh] : R(0); hT += on ? tD[hh] * tv[hh] * tv[hh] : R(0); }
        LUNROLL for (int r = 0; r < kMaxX; r++) { const R x = xx0[r] + a * xv[r]; const bool on = x < 0; gT += on ? xDv[r] * x * xv[r] : R(0); hT += on ? xDv[r] * xv[r] * xv[r] : R(0); }
        d1 = qd_sum(g) + gT; d2 = qd_sum(h) + hT;
      };
      // (one call site of the derivative evaluation: the first pass is the evaluation at alpha = 0.) The loop runs until no candidate of the
      // WAVEFRONT is still searching -- one uniform branch per pass; what a candidate's own exits were (its step stopped moving alpha, the
      // derivative below the tolerance or its floor) freezes its alpha through selects. The wavefront ran the longest search before, too.
      R lo = 0, hi = -1, d1 = 0, d2 = 1, d10 = 0;
      R step1 = R(1e30), step2 = R(1e30);
      bool done = false;
      for (int ls = -1; ls < 50; ls++) {
        if (!qw_any(!done)) break;
        if (ls >= 0) {
          R an = alpha - d1 / d2;
          const bool outside = !(an > lo) || (hi >= 0 && !(an < hi));
          const R mid = R(0.5) * (lo + hi);
          an = outside ? (hi >= 0 ? mid : 2 * alpha + 1) : ((hi >= 0 && fabs(an - alpha) > R(0.5) * step2) ? mid : an);
          done = done || an == alpha;
          step2 = done ? step2 : step1; step1 = done ? step1 : fabs(an - alpha);
          alpha = done ? alpha : an;
        }
        R e1, e2;
        derivs(alpha, e1, e2);
        if (pa.stamps && !done) LPROF_COUNT(pa, 16);
        e1 += q1 + alpha * q2; e2 += q2;
        if (ls < 0) { d1 = e1; d2 = e2; d10 = fabs(d1); done = !(d10 >= gtol); continue; }
        // (the tolerance, or the working precision's floor under it: a derivative that is the rounding residue of its start value -- its terms
        // cancel at the minimum -- cannot be reduced further. In float the tolerance lies below that floor and the search would otherwise run
        // until its steps stop moving alpha: nine evaluations per search instead of three)
        const bool live = !done;
        d1 = live ? e1 : d1; d2 = live ? e2 : d2;
        const bool conv = fabs(e1) < gtol || fabs(e1) <= LEXP_LFLOOR * kEps<R>() * d10;
        lo = (live && !conv && e1 < 0) ? alpha : lo;
        hi = (live && !conv && !(e1 < 0)) ? alpha : hi;
        done = done || conv;
      }
    }
    LPROF(pa, prof_last, 13);
    LUNROLL for (int j = 0; j < kLD; j++) al[j] += alpha * hl[j];
    LUNROLL for (int k = 0; k < kTD; k++) at[k] += alpha * ht[k];
    R dl[kLD], dt[kTD];
    LUNROLL for (int j = 0; j < kLD; j++) dl[j] = al[j] - sl[j];
    LUNROLL for (int k = 0; k < kTD; k++) dt[k] = at[k] - st[k];
    arrow_mul_s(ms, dl, dt, Mal, Mat);
    const R gauss = R(0.5) * arrow_dot(dl, dt, Mal, Mat);
    const R newcost = gauss + rows_eval(m, L, lane, kin, Rw, cs, ncon, sh, nx, true, hl, ht, alpha, fc_l, fc_t, VL, VR);
    improvement = cost - newcost;
    cost = newcost;
    iters = iter + 1;
    LPROF(pa, prof_last, 14);
    if (pa.stamps) LPROF_COUNT(pa, 15);
  }
  return 0;
}

// what the solver leaves (what it reads -- qacc_smooth, the warm start, qvel -- it takes from the step's own structs: no copies through
// the private segment)
template <typename R> struct LNewtonOut { R al[kLD], at[kTD], fc_l[kLD], fc_t[kTD]; int iters; };
#ifndef LNEWTON_ATTR
#define LNEWTON_ATTR LNOINLINE
#define LEULER_ATTR LNOINLINE
#define LRESID_ATTR LNOINLINE
#endif
template <typename R, class CS, class MS, class SH, class KS>
LNEWTON_ATTR int newton(const LimbModelT<R>& m_in, int lane, KS ks, MS ms, const LDyn<R>* D_in, const LState<R>* S_in, CS cs, SH sh, bool have_warm,
                     LNewtonOut<R>* out_ptr, long long* stamps) {
  const LimbModelT<R>& m = LREBIND_LDS(LimbModelT<R>, m_in);
  // what the solver WRITES lives in registers for the call (the rows' jar, the iterate, J' force: a read-modify-write through the caller's
  // memory waits for its own store every time); what it only reads (qacc_smooth, the warm start) stays where it is; the dof axes are in the
  // includer's store (LDS)
  LRows<R> Rw;
  LPRV_LOAD(Rw, &D_in->rows);
  R al[kLD], at[kTD], fc_l[kLD], fc_t[kTD];
  int iters = 0;
  const int rc = newton_body(m, lane, ks, ms, Rw, cs, D_in->ncon, sh, D_in->nx, D_in->sl, D_in->st, S_in->wl, S_in->wt, have_warm, S_in->lv, S_in->tv, al, at, fc_l, fc_t, iters, stamps);
  LPRV_STOREN(out_ptr->al, al, kLD); LPRV_STOREN(out_ptr->at, at, kTD); LPRV_STOREN(out_ptr->fc_l, fc_l, kLD); LPRV_STOREN(out_ptr->fc_t, fc_t, kTD);
  out_ptr->iters = iters;
  return rc;
}

// ---------------------------------------------------------------- mj_Euler with implicit joint damping, then mj_advance (oracle o_euler)
template <typename R, class MS>
LEULER_ATTR void euler(const LimbModelT<R>& m_in, int lane, LState<R>* S_io, const LDyn<R>* D_in, const LNewtonOut<R>* in_ptr, MS ms) {
  const LimbModelT<R>& m = LREBIND_LDS(LimbModelT<R>, m_in);
  const LimbT<R>& L = m.limb[lane];
  LState<R> S;
  LPRV_LOAD(S, S_io);
  LNewtonOut<R> in;
  LPRV_LOAD(in, in_ptr);
  const R* al = in.al; const R* at = in.at;
  const R h = m.timestep;
  R ql[kLD], qt[kTD];
  LUNROLL for (int j = 0; j < kLD; j++) ql[j] = D_in->fs_l[j] + in.fc_l[j];
  LUNROLL for (int k = 0; k < kTD; k++) qt[k] = D_in->fs_t[k] + in.fc_t[k];
  Arrow<R> A;
  load_arrow(ms, A);
  LUNROLL for (int j = 0; j < kLD; j++) A.l[tri(j, j)] += h * L.jnt[j].damping;
  LUNROLL for (int k = 0; k < kTD; k++) A.t[tri(k, k)] += h * (k < 6 ? m.tdamp[k] : m.tjnt[k - 6].damping);
  if (arrow_factor(A)) arrow_solve(A, ql, qt);
  else { LUNROLL for (int j = 0; j < kLD; j++) ql[j] = al[j]; LUNROLL for (int k = 0; k < kTD; k++) qt[k] = at[k]; }
  LUNROLL for (int j = 0; j < kLD; j++) { S.wl[j] = al[j]; S.lv[j] += h * ql[j]; S.lq[j] += h * S.lv[j]; }
  LUNROLL for (int k = 0; k < kTD; k++) { S.wt[k] = at[k]; S.tv[k] += h * qt[k]; }
  LUNROLL for (int k = 0; k < 3; k++) S.tq[k] += h * S.tv[k];
  {  // mj_integratePos of the free joint's quaternion
    R ax[3] = {S.tv[3], S.tv[4], S.tv[5]};
    const R n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (n < R(1e-15)) { ax[0] = 1; ax[1] = ax[2] = 0; } else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
    const R ang = h * n;
    R qr[4] = {1, 0, 0, 0};
    if (ang != 0) { const R s = sin(R(0.5) * ang), c = cos(R(0.5) * ang); qr[0] = c; qr[1] = ax[0] * s; qr[2] = ax[1] * s; qr[3] = ax[2] * s; }
    R q[4] = {S.tq[3], S.tq[4], S.tq[5], S.tq[6]}, r[4];
    q_norm(q);
    q_mul(r, q, qr);
    LUNROLL for (int k = 0; k < 4; k++) S.tq[3 + k] = r[k];
  }
  LUNROLL for (int hh = 0; hh < 3; hh++) S.tq[7 + hh] += h * S.tv[6 + hh];
  S.time += h;
  LPRV_STORE(S_io, S);
}

// ---------------------------------------------------------------- the tracking residual and its cost (oracle humanoid_track_residual, ocost_value)
// mjpc::Norm value per entry (every norm of this task class is a sum over entries: limb_model.h checks): quadratic, cosh, power loss,
// smooth abs, smooth abs 2, rectify
#ifndef LNORM_ATTR
#define LNORM_ATTR LNOINLINE
#endif
template <typename R> LNORM_ATTR R norm_entry(R x, int type, R p, R q) {
  switch (type) {
    case 0: return R(0.5) * x * x;
    case 3: return p * p * (cosh(x / p) - R(1));
    case 5: return R(pow(fabs(x), p));
    case 6: return sqrt(x * x + p * p) - p;
    case 7:  // (q = 2 and q = 4 -- the tracking task's -- by products and square roots: the same function without three pow() calls)
      if (q == 2) return sqrt(x * x + p * p) - p;
      if (q == 4) { const R x2 = x * x, p2 = p * p; return sqrt(sqrt(x2 * x2 + p2 * p2)) - p; }
      return R(pow(pow(fabs(x), q) + pow(p, q), 1 / q)) - p;
    case 8: return p > 0 ? p * R(log(1 + exp(x / p))) : (x > 0 ? x : R(0));
    default: return 0;
  }
}
// writes the lane's entries of residual row `rs` (nullptr: cost only) and returns the step's cost (the same in the four lanes)
// (one term's parameters, read once for all the entries of the term that a lane holds: weight, norm type, its two parameters)
// kind: 0 quadratic, 1 the smooth-abs family -- sqrt(x^2 + p^2) - p, or (x^4 + p^4)^(1/4) - p (`four`) --, 2 anything else. The first two are
// evaluated in line when every lane of the wavefront holds such a term at the call site (the tracking task: all but the controls' cosh):
// thirty of a lane's thirty-six entries then cost a handful of instructions instead of a call into the switch over the norm types
// (term_of is called where all four lanes of a candidate are: it asks the wavefront)
template <typename R> struct LTerm { R w, p, q; int type, mode; bool four; };
template <typename R> LD LTerm<R> term_of(const LimbModelT<R>& m, const LTask<R>& tk, int t, bool on = true) {  // (on: does this lane use the term)
  LTerm<R> r{tk.weight[t], tk.norm_p[t], tk.norm_q[t], m.term_norm[t], 2, false};
  r.four = r.type == 7 && r.q == 4;
  const int kind = r.type == 0 ? 0 : ((r.type == 6 || (r.type == 7 && (r.q == 2 || r.q == 4))) ? 1 : 2);
  const bool all0 = !qw_any(on && kind != 0), all1 = !qw_any(on && kind != 1);
  r.mode = all0 ? 0 : (all1 ? 1 : 2);
  return r;
}
template <typename R, class T>
LRESID_ATTR R residual_cost(const LimbModelT<R>& m_in, const LTask<R>* tk_in, int lane, const LState<R>* S_in, const R* ctrl_in, const R* tctrl_in, const LSense<R>* f_in, T* rs, long long* stamps) {
  const LimbModelT<R>& m = LREBIND_LDS(LimbModelT<R>, m_in);
  const LimbT<R>& L = m.limb[lane];
  struct { long long* stamps; } pa{stamps};
  long long prof_last = 0;
  LPROF(pa, prof_last, -1);
  LTask<R> tk;
  LPRV_LOAD(tk, tk_in);
  LState<R> S;
  LPRV_LOAD(S, S_in);
  LSense<R> f;
  LPRV_LOAD(f, f_in);
  R ctrl[kLD], tctrl[3];
  LPRV_LOADN(ctrl, ctrl_in, kLD); LPRV_LOADN(tctrl, tctrl_in, 3);
  const int nj = m.nv - 6, nu = m.nu, c0 = nj + nu;
  R cost = 0;
  // Entries are evaluated term by term: the residual's layout fixes which term an entry belongs to (joint velocities, controls, the marker
  // average, one position and one velocity term per marker: limb_model.h bakes the marker terms into the sites), so a term's weight and norm
  // parameters are read once for its entries instead of through a chain of look-ups per entry (that chain was 2 k cycles an entry).
  auto entry = [&](const LTerm<R>& t, int idx, R x) {
    R v;
    if (t.mode == 0) v = R(0.5) * x * x;
    else if (t.mode == 1) {
      const R x2 = x * x, p2 = t.p * t.p;
      const R r2 = sqrt(x2 + p2), r4 = sqrt(sqrt(x2 * x2 + p2 * p2));
      v = (t.four ? r4 : r2) - t.p;
    } else v = norm_entry(x, t.type, t.p, t.q);
    cost += t.w * v;
    if (rs) LREC(rs[idx], (T)x);
  };
  const LTerm<R> tv = term_of(m, tk, m.t_qvel), tc = term_of(m, tk, m.t_ctrl);
  LUNROLL for (int j = 0; j < kLD; j++) {
    const LJointT<R>& J = L.jnt[j];
    if (!J.on) continue;
    entry(tv, J.dof - 6, S.lv[j]);
    if (J.act >= 0) entry(tc, nj + J.act, ctrl[j]);
  }
  if (L.owns_trunk_rows) {
    LUNROLL for (int h = 0; h < 3; h++) {
      const LJointT<R>& J = m.tjnt[h];
      if (!J.on) continue;
      entry(tv, J.dof - 6, S.tv[6 + h]);
      if (J.act >= 0) entry(tc, nj + J.act, tctrl[h]);
    }
  }
  LPROF(pa, prof_last, 29);
  // ComputeInterpolationValues (tracking.cc:29-38)
  const int start = tk.ri[0], last = tk.ri[1];
  const R kFps = 30;
  const R index = (S.time - tk.re[0]) * kFps + R(start);
  const R clamped = index < 0 ? R(0) : (index > R(last) ? R(last) : index);
  const int k0 = (int)floor(clamped);
  const int k1 = k0 + 1 < last ? k0 + 1 : last;
  const R w1 = clamped - R(k0), w0 = R(1) - w1;
  R mp[kLS][3], dv[kLS][3], am[3] = {0, 0, 0}, as[3] = {0, 0, 0};
  LUNROLL for (int s = 0; s < kLS; s++) {   // (unrolled: the four sites' twenty-four key loads are in flight together)
    const LSiteT<R>& St = L.site[s];
    const int mc = St.on ? St.mocap : 0;
    const R* key0 = tk.key_mpos + ((size_t)m.nmocap * k0 + mc) * 3;
    const R* key1 = tk.key_mpos + ((size_t)m.nmocap * k1 + mc) * 3;
    LUNROLL for (int k = 0; k < 3; k++) {
      const R a0 = key0[k], a1 = key1[k];
      R v = a0 * w0;
      v += a1 * w1;
      mp[s][k] = v;
      dv[s][k] = (a1 - a0) * kFps - f.svel[s][k];
      am[k] += St.on ? v : R(0); as[k] += St.on ? f.spos[s][k] : R(0);
    }
  }
  LUNROLL for (int k = 0; k < 3; k++) { am[k] = qd_sum(am[k]) * (R(1) / 16); as[k] = qd_sum(as[k]) * (R(1) / 16); }
  LPROF(pa, prof_last, 30);
  const LTerm<R> ta = term_of(m, tk, m.t_avg);
  if (L.owns_trunk_rows) { LUNROLL for (int k = 0; k < 3; k++) entry(ta, c0 + k, am[k] - as[k]); }
  LUNROLL for (int s = 0; s < kLS; s++) {
    const LSiteT<R>& St = L.site[s];
    const LTerm<R> tp = term_of(m, tk, St.on ? St.tpos : 0, St.on), tw = term_of(m, tk, St.on ? St.tvel : 0, St.on);
    if (!St.on) continue;
    LUNROLL for (int k = 0; k < 3; k++) {
      entry(tp, c0 + 3 + 3 * St.marker + k, (mp[s][k] - am[k]) - (f.spos[s][k] - as[k]));
      entry(tw, c0 + 51 + 3 * St.marker + k, dv[s][k]);
    }
  }
  LPROF(pa, prof_last, 31);
  cost = qd_sum(cost);
  if (!(fabs(tk.risk) < R(1.0e-6))) cost = (exp(tk.risk * cost) - R(1)) / tk.risk;
  return cost;
}

// ---------------------------------------------------------------- candidate generation + Trajectory::Rollout
// Philox4x32-10 + Box-Muller exactly as include/mjpcx.h specifies (device_common.h gaussian_pair; oracle/rng.c): in double whatever R is
LD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  LUNROLL for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
LD double u53(uint32_t hi, uint32_t lo) { const uint64_t k = (((uint64_t)hi << 32) | lo) >> 11; return ((double)k + 0.5) * (1.0 / 9007199254740992.0); }
LD void gaussian_pair(uint64_t seed, uint32_t cand, uint32_t pair, uint32_t iter, double* z) {
  uint32_t o[4];
  philox4x32_10(cand, pair, iter, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  const double u1 = u53(o[0], o[1]), u2 = u53(o[2], o[3]);
  const double r = sqrt(-2.0 * log(u1));
  z[0] = r * cos(6.283185307179586476925286766559 * u2); z[1] = r * sin(6.283185307179586476925286766559 * u2);
}
LD double bernoulli_uniform(uint64_t seed, uint32_t cand, uint32_t iter) {
  uint32_t o[4];
  philox4x32_10(cand, 0u, iter, 1u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return u53(o[0], o[1]);
}

// One lane's share of one candidate's rollout. `state0` = qpos[nq] qvel[nv] of the plan (Planner::SetState). Returns the flag bits (0:
// rolled out; otherwise failure[cand] carries kLFallback and the wavefront-per-candidate kernel takes the candidate over).
template <typename R, class CS, class MS, class SH, class KS>
LD int rollout(const LimbModelT<R>& m, const LTask<R>& tk, const R* state0, R time0, const LArgs<R>& a, int cand, int lane, CS& cs, MS& ms, SH& sh, KS& ks) {
  const LimbT<R>& L = m.limb[lane];
  const int nu = m.nu, P = a.P, H = a.H, nq = m.nq;
  const size_t N = (size_t)a.N;
  // ---- the candidate's spline nodes of the lane's actuators (AddNoiseToPolicy, sampling/planner.cc:326-352)
  auto act_of = [&](int e) { return e < kLD ? L.jnt[e < kLD ? e : 0].act : L.tact[e - kLD < 3 ? e - kLD : 0]; };
  auto lo_of = [&](int e) { return e < kLD ? L.jnt[e < kLD ? e : 0].ctrl_lo : m.tjnt[e - kLD < 3 ? e - kLD : 0].ctrl_lo; };
  auto hi_of = [&](int e) { return e < kLD ? L.jnt[e < kLD ? e : 0].ctrl_hi : m.tjnt[e - kLD < 3 ? e - kLD : 0].ctrl_hi; };
  if (a.noise_mode >= 0) {
    const int gi = a.candidate_offset + cand;
    double std = a.std0;
    if (a.noise_mode == 0 && a.std1 > 0) { if (bernoulli_uniform(a.seed, (uint32_t)gi, a.iteration) < 0.2) std = a.std1; }
    const bool noised = gi != a.nominal_candidate;
    for (int p = 0; p < P; p++)
      LUNROLL for (int e = 0; e < kLD + 3; e++) {
        const int u = act_of(e);
        if (u < 0) continue;
        const int j = p * nu + u;
        double v = (double)a.nominal[j];
        if (noised) {
          double z[2];
          gaussian_pair(a.seed, (uint32_t)gi, (uint32_t)(j >> 1), a.iteration, z);
          const double lo = (double)lo_of(e), hi = (double)hi_of(e);
          double sigma;
          if (a.noise_mode == 0) sigma = 0.5 * (hi - lo) * std;
          else {
            const double fl = gi < a.explore_count ? a.std0 : a.std1;
            const double sd = sqrt(a.param_variance[j]);
            sigma = sd > fl ? sd : fl;
          }
          v = v + sigma * ((j & 1) ? z[1] : z[0]);
          v = v < lo ? lo : (v > hi ? hi : v);
        }
        a.nodes[(size_t)j * N + cand] = (R)v;
      }
  }
#define LNODE(p, u) a.nodes[(size_t)((p) * nu + (u)) * N + cand]
  LState<R> S;
  LUNROLL for (int k = 0; k < 7; k++) S.tq[k] = state0[k];
  LUNROLL for (int h = 0; h < 3; h++) { S.tq[7 + h] = m.tjnt[h].on ? state0[m.tjnt[h].qadr] : R(0); S.tv[6 + h] = m.tjnt[h].on ? state0[nq + m.tjnt[h].dof] : R(0); }
  LUNROLL for (int k = 0; k < 6; k++) S.tv[k] = state0[nq + k];
  LUNROLL for (int j = 0; j < kLD; j++) { S.lq[j] = L.jnt[j].on ? state0[L.jnt[j].qadr] : R(0); S.lv[j] = L.jnt[j].on ? state0[nq + L.jnt[j].dof] : R(0); S.wl[j] = 0; }
  LUNROLL for (int k = 0; k < kTD; k++) S.wt[k] = 0;
  S.time = time0;
  const size_t ds = (size_t)(m.nq + m.nv);
  double total = 0;
  R ctrl[kLD], tctrl[3];
  LUNROLL for (int j = 0; j < kLD; j++) ctrl[j] = 0;
  LUNROLL for (int h = 0; h < 3; h++) tctrl[h] = 0;
  int flags = 0, flag_step = 0, iters_total = 0, up = 0;
  long long prof_last = 0;
  LPROF(a, prof_last, -1);
  for (int t = 0; t < H; t++) {
    flag_step = t;
    const bool last = t == H - 1;
    bool bad = false;
    if (!last) {
      // policy: TimeSpline::Sample + Clamp (SamplingPolicy::Action, sampling/policy.cc:52-59)
      const R now = S.time;
      while (up < P && a.node_times[up] <= now) up++;  // (time only moves forward: the search resumes where the last step's ended)
      R mine[kLD + 3];
      {
        // every candidate of a launch runs on the same clock (state0's time + t steps), so the interval is the same in all lanes: its index
        // goes to a scalar register, the node times are read once, and the lane's node values -- up to four per actuator for the cubic --
        // are fetched together before anything waits for them (fetched inside the per-actuator arithmetic they were 40 k cycles a step)
        const int upu = LUNIFORM(up);
        const bool edge = upu == P || upu == 0;
        const int lo = edge ? (upu == 0 ? 0 : P - 1) : upu - 1, hi = edge ? lo : upu;
        const int im = lo > 0 ? lo - 1 : 0, ip = hi + 1 < P ? hi + 1 : P - 1;
        R pm[kLD + 3], p0[kLD + 3], p1[kLD + 3], p2[kLD + 3];
        LUNROLL for (int e = 0; e < kLD + 3; e++) {
          const int ua = act_of(e), uu = ua < 0 ? 0 : ua;
          p0[e] = LNODE(lo, uu); p1[e] = LNODE(hi, uu);
          pm[e] = p0[e]; p2[e] = p1[e];
          if (a.interp >= 2 && !edge) { pm[e] = LNODE(im, uu); p2[e] = LNODE(ip, uu); }
        }
        const R tl = a.node_times[lo], tu = a.node_times[hi], tp = a.node_times[im], tn = a.node_times[ip];
        const R s = edge ? R(0) : (S.time - tl) / (tu - tl), dt_mid = tu - tl;
        const R s2 = s * s, s3 = s * s * s;
        const R c0 = 2 * s3 - 3 * s2 + 1, c1 = (s3 - 2 * s2 + s) * (tu - tl), c2 = -2 * s3 + 3 * s2, c3 = (s3 - s2) * (tu - tl);
        LUNROLL for (int e = 0; e < kLD + 3; e++) {
          const int ua = act_of(e);
          mine[e] = 0;
          if (ua < 0) continue;
          R u;
          if (edge || a.interp == 0) u = p0[e];
          else if (a.interp == 1) u = p0[e] * (1 - s) + p1[e] * s;
          else {
            const R fwd = (p1[e] - p0[e]) / dt_mid;
            R m0, m1;
            if (lo == 0) m0 = fwd;
            else m0 = R(0.5) * (p1[e] - p0[e]) / dt_mid + R(0.5) * (p0[e] - pm[e]) / (tl - tp);
            if (hi == P - 1) m1 = fwd;
            else m1 = R(0.5) * (p2[e] - p1[e]) / (tn - tu) + R(0.5) * (p1[e] - p0[e]) / dt_mid;
            u = c0 * p0[e] + c1 * m0 + c2 * p1[e] + c3 * m1;
          }
          bad |= lbad(u);
          mine[e] = clampr(u, lo_of(e), hi_of(e));
        }
      }
      LUNROLL for (int j = 0; j < kLD; j++) ctrl[j] = mine[j];
      // the trunk's controls reach every lane from the one that evaluated their splines
      LUNROLL for (int h = 0; h < 3; h++) tctrl[h] = qd_sum(L.tact[h] >= 0 ? mine[kLD + h] : R(0));
      LUNROLL for (int k = 0; k < 10; k++) bad |= lbad(S.tq[k]);
      LUNROLL for (int k = 0; k < kTD; k++) bad |= lbad(S.tv[k]);
      LUNROLL for (int j = 0; j < kLD; j++) bad |= lbad(S.lq[j]) || lbad(S.lv[j]);
    }
    if (qd_or(bad ? 1 : 0)) { flags = kFlagBad; break; }
    LPROF(a, prof_last, 0);
    LDyn<R> D;
    LSense<R> f;
    LPOISON(D); LPOISON(f);
    flags = forward_smooth(m, lane, &S, ctrl, tctrl, cs, ms, sh, ks, &D, &f, a.stamps);
    if (flags) break;
    LPROF(a, prof_last, 1);
    // the sensor stage (residual, cost, traces) does not depend on the constraint solve: evaluated and recorded first
    R* rs = a.residual + ((size_t)cand * H + t) * m.nr;
#ifdef LEXP_NO_RESID_STORE
    rs = nullptr;
#endif
    const R cost = residual_cost(m, &tk, lane, &S, ctrl, tctrl, &f, rs, a.stamps);
    {
      R* st = a.states + ((size_t)cand * H + t) * ds;
      R* ac = a.actions + ((size_t)cand * H + t) * nu;
      LUNROLL for (int j = 0; j < kLD; j++) if (L.jnt[j].on) { LREC(st[L.jnt[j].qadr], S.lq[j]); LREC(st[nq + L.jnt[j].dof], S.lv[j]); if (L.jnt[j].act >= 0) LREC(ac[L.jnt[j].act], ctrl[j]); }
      if (L.owns_trunk_rows) {
        LUNROLL for (int k = 0; k < 7; k++) LREC(st[k], S.tq[k]);
        LUNROLL for (int k = 0; k < 6; k++) LREC(st[nq + k], S.tv[k]);
        LUNROLL for (int h = 0; h < 3; h++) if (m.tjnt[h].on) { LREC(st[m.tjnt[h].qadr], S.tq[7 + h]); LREC(st[nq + m.tjnt[h].dof], S.tv[6 + h]); if (m.tjnt[h].act >= 0) LREC(ac[m.tjnt[h].act], tctrl[h]); }
        LREC(a.times[(size_t)cand * H + t], S.time); LREC(a.costs[(size_t)cand * H + t], cost);
      }
      LUNROLL for (int q = 0; q < kMaxTrace; q++)
        if (q < m.ntrace && (m.trace[q].lane == lane || (m.trace[q].lane == 4 && L.owns_trunk_rows)))
          LUNROLL for (int k = 0; k < 3; k++) LREC(a.trace[((size_t)cand * H + t) * 3 * m.ntrace + 3 * q + k], f.trace[q][k]);
    }
    total += (double)cost;
    LPROF(a, prof_last, 2);
    if (last) break;  // (the last step's mj_forward only feeds the sensor stage)
    LNewtonOut<R> io;
    LPOISON(io);
    flags = newton(m, lane, ks, ms, &D, &S, cs, sh, t > 0, &io, a.stamps);
    if (flags) break;
    LPROF(a, prof_last, 3);
    iters_total += io.iters;
    LUNROLL for (int j = 0; j < kLD; j++) bad |= lbad(io.al[j]);
    LUNROLL for (int k = 0; k < kTD; k++) bad |= lbad(io.at[k]);
    if (qd_or(bad ? 1 : 0)) { flags = kFlagBad; break; }
    euler(m, lane, &S, &D, &io, ms);
    LPROF(a, prof_last, 4);
  }
#undef LNODE
  if (lane == 0) {
    a.total_return[cand] = flags ? 1.0e6 : total / (double)(H > 1 ? H : 1);
    a.failure[cand] = flags ? (kLFallback | flags | (flag_step << 8)) : 0;
    if (a.iters) a.iters[cand] = iters_total;
  }
  return flags;
}

} }  // namespace mjpcx::limb
