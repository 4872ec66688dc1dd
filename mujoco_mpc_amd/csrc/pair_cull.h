// pair_cull.h -- host side of the moving-geom pairs (mjpcx_create): the list MuJoCo's body filters leave
// (engine_collision_driver.c: same weld, parent-child, <exclude>, contype / conaffinity; restated in oracle/contact.inc bake_pairs), the
// class of each pair, and a bake-time PROOF that a pair can never touch.
//
// The kernels collide sphere | capsule pairs and (sphere | capsule) x (box | cylinder) pairs (solid_pairs.h). Two solids (box | cylinder
// both) have no narrow phase anywhere; the quad kernel's layout (quad_model.h) has room for leg-leg and leg-trunk pairs of sphere |
// capsule geoms and for a leg's cylinders against another leg's sphere | capsule geoms only. A pair outside what a kernel covers may be
// dropped only if it is proven apart: with every hinge between the two bodies anywhere in its range widened by a pad (soft joint limits
// give: kPairCullPad, or kPairCullPadTight for ONE joint side if the proof needs that), the distance stays above the margin. The proof is
// a branch and bound over the joint box (at most three hinges between the bodies): the distance d at a cell's centre bounds the distance in
// the cell from below by d - sum_j L_j delta_j, L_j a bound on the lever arm of hinge j over the geom it moves (link lengths + the geom's
// bounding radius), delta_j the cell's half width. The distance is exact for thin-solid pairs (solid_pairs.h) and a lower bound for two
// solids (one replaced by a thin geom that contains it).
//   * wave / tree kernels (wave_model.h): every sphere | capsule pair and every thin-solid pair is collided, proven apart or not; two solids
//     are kept in the list and WATCHED, proven apart or not (a thin geom containing one of them within the margin of the other raises
//     warning bit 128 and the rollout fails, as the oracle's does); the ones that cannot be proven apart are REPORTED at create time (the
//     warning / MJPCX_STRICT_PAIRS refusal), and a model in which two of them already touch at qpos0 is refused outright (qpos0 only: keyframes are not examined -- a model whose solids overlap in a keyframe alone fails its rollouts at run time with warning bit 128).
//   * quad kernel (quad_model.h): pairs outside its layout must be proven apart or the model is declined -- two solids with the wide pad
//     only; a candidate one of whose joints leaves the range the proofs cover (range + pad) is handed to the wavefront-per-candidate
//     kernel at that step (kFlagRange).
// The oracle keeps every pair and raises a warning should two solids come within reach: the parity suites check the proofs at run time.
#pragma once
#include <math.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <vector>

#include "../../include/mjpcx.h"
#include "solid_pairs.h"

namespace mjpcx {

constexpr double kPairCullPad = 0.2;       // [rad] how far past its range a (soft) joint limit may be pushed in a proof ...
constexpr double kPairCullPadTight = 0.1;  // ... and on ONE joint side of a proof that does not hold with the wider pad

namespace cull_detail {
inline void q2m(double* m, const double* q0) {
  double q[4], n = sqrt(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
  for (int k = 0; k < 4; k++) q[k] = n > 0 ? q0[k] / n : (k == 0 ? 1.0 : 0.0);
  const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  const double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
inline void mm(double* r, const double* a, const double* b) {
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  for (int k = 0; k < 9; k++) r[k] = t[k];
}
inline void mv(double* r, const double* a, const double* v) {
  const double t0 = a[0] * v[0] + a[1] * v[1] + a[2] * v[2], t1 = a[3] * v[0] + a[4] * v[1] + a[5] * v[2], t2 = a[6] * v[0] + a[7] * v[1] + a[8] * v[2];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
inline double norm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
inline void axis_angle(double* R, const double* ax0, double th) {
  double ax[3], n = norm3(ax0);
  for (int k = 0; k < 3; k++) ax[k] = n > 0 ? ax0[k] / n : 0.0;
  const double c = cos(th), s = sin(th), v = 1 - c;
  R[0] = c + ax[0] * ax[0] * v; R[1] = ax[0] * ax[1] * v - ax[2] * s; R[2] = ax[0] * ax[2] * v + ax[1] * s;
  R[3] = ax[1] * ax[0] * v + ax[2] * s; R[4] = c + ax[1] * ax[1] * v; R[5] = ax[1] * ax[2] * v - ax[0] * s;
  R[6] = ax[2] * ax[0] * v - ax[1] * s; R[7] = ax[2] * ax[1] * v + ax[0] * s; R[8] = c + ax[2] * ax[2] * v;
}
inline bool is_thin(int t) { return t == MJPCX_GEOM_SPHERE || t == MJPCX_GEOM_CAPSULE; }
inline bool is_solid(int t) { return t == MJPCX_GEOM_CYLINDER || t == MJPCX_GEOM_BOX; }
inline double bound_radius(const mjpcx_model* m, int g) {
  const double* s = m->geom_size + 3 * g;
  switch (m->geom_type[g]) {
    case MJPCX_GEOM_SPHERE: return s[0];
    case MJPCX_GEOM_CAPSULE: return s[0] + s[1];
    case MJPCX_GEOM_CYLINDER: return sqrt(s[0] * s[0] + s[1] * s[1]);
    default: return norm3(s);
  }
}

struct Chain {
  std::vector<int> bodies;  // from the child of the common ancestor down to the geom's body
};

struct Problem {
  const mjpcx_model* m;
  int g1, g2;               // g2 is a solid; g1 thin, or a solid taken as the thin geom that contains it
  Chain c[2];
  std::vector<int> jnt;     // the hinges between the two bodies, chain 1's first
  std::vector<double> lo, hi, lever;
  int evals = 0, budget = 0;
  double closest = 1e300, margin = 0;

  // pose of chain s's last body (or of the common ancestor's frame itself for an empty chain) in the common ancestor's frame
  void pose(int s, const double* theta, double* p, double* R) const {
    for (int k = 0; k < 3; k++) p[k] = 0;
    for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    int ji = 0;
    if (s == 1) for (int b : c[0].bodies) ji += m->body_jntnum[b];
    for (int b : c[s].bodies) {
      double v[3], Rb[9];
      mv(v, R, m->body_pos + 3 * b);
      for (int k = 0; k < 3; k++) p[k] += v[k];
      q2m(Rb, m->body_quat + 4 * b);
      mm(R, R, Rb);
      for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++, ji++) {
        double Rj[9], ra[3], d[3];
        axis_angle(Rj, m->jnt_axis + 3 * j, theta[ji]);
        mv(ra, Rj, m->jnt_pos + 3 * j);
        for (int k = 0; k < 3; k++) d[k] = m->jnt_pos[3 * j + k] - ra[k];  // rotation about the anchor: x -> anchor + Rj (x - anchor)
        mv(v, R, d);
        for (int k = 0; k < 3; k++) p[k] += v[k];
        mm(R, R, Rj);
      }
    }
  }
  double distance(const double* theta) {
    evals++;
    double p[2][3], R[2][9], gp[2][3], gR[2][9];
    for (int s = 0; s < 2; s++) {
      pose(s, theta, p[s], R[s]);
      const int g = s == 0 ? g1 : g2;
      double v[3], Rg[9];
      mv(v, R[s], m->geom_pos + 3 * g);
      for (int k = 0; k < 3; k++) gp[s][k] = p[s][k] + v[k];
      q2m(Rg, m->geom_quat + 4 * g);
      mm(gR[s], R[s], Rg);
    }
    const int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    const double* s1 = m->geom_size + 3 * g1;
    double h = t1 == MJPCX_GEOM_CAPSULE ? s1[1] : 0.0, r = s1[0];
    if (is_solid(t1)) solid::solid_as_thin<double>(t1 == MJPCX_GEOM_CYLINDER ? solid::kSolidCylinder : solid::kSolidBox, s1, h, r);
    double rel[3], pl[3], al[3], n[3], cc[3];
    for (int k = 0; k < 3; k++) rel[k] = gp[0][k] - gp[1][k];
    for (int k = 0; k < 3; k++) {
      pl[k] = gR[1][k] * rel[0] + gR[1][3 + k] * rel[1] + gR[1][6 + k] * rel[2];
      al[k] = gR[1][k] * gR[0][2] + gR[1][3 + k] * gR[0][5] + gR[1][6 + k] * gR[0][8];
    }
    const double d = solid::thin_vs_solid<double>(t2 == MJPCX_GEOM_CYLINDER ? solid::kSolidCylinder : solid::kSolidBox, m->geom_size + 3 * g2, pl, al, h, r, n, cc);
    if (d < closest) closest = d;
    return d;
  }
  bool cell(std::vector<double>& c0, std::vector<double>& half) {
    if (evals >= budget) return false;
    const double d = distance(c0.data());
    double slack = 0, worst = -1;
    int wj = 0;
    for (size_t j = 0; j < jnt.size(); j++) { const double s = lever[j] * half[j]; slack += s; if (s > worst) { worst = s; wj = (int)j; } }
    if (d - slack > margin) return true;
    if (slack < 2.5e-4 || d <= margin) return false;  // (resolved to a quarter of a millimetre and still not clear: the pair stays)
    const double keep_c = c0[wj], keep_h = half[wj];
    half[wj] = 0.5 * keep_h;
    bool ok = true;
    for (int side = -1; side <= 1 && ok; side += 2) { c0[wj] = keep_c + side * half[wj]; ok = cell(c0, half); }
    c0[wj] = keep_c; half[wj] = keep_h;
    return ok;
  }
};
}  // namespace cull_detail

// true: geoms g1, g2 (on two moving bodies; g2 a box or a cylinder, g1 any of sphere | capsule | cylinder | box) can be PROVEN never to come
// within `margin` of each other while the hinges between their bodies stay within their ranges widened by `pad`. false: no proof (too many
// joints in between, a joint that is not a hinge, the evaluation budget spent, or the two can in fact touch).
inline bool pair_never_touches(const mjpcx_model* m, int g1, int g2, double margin, double pad, int* evals = nullptr, double* closest = nullptr,
                               int tight_jnt = -1, int tight_side = 0, double tight_pad = 0, std::vector<int>* joints = nullptr) {
  using namespace cull_detail;
  if (evals) *evals = 0;
  if (!is_solid(m->geom_type[g2])) return false;
  Problem P;
  P.m = m; P.g1 = g1; P.g2 = g2; P.margin = margin; P.budget = 400000;
  const int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
  // common ancestor
  std::vector<char> anc(m->nbody, 0);
  for (int b = b1;; b = m->body_parentid[b]) { anc[b] = 1; if (b == 0) break; }
  int lca = b2;
  while (!anc[lca]) lca = m->body_parentid[lca];
  for (int s = 0; s < 2; s++) {
    std::vector<int> up;
    for (int b = s == 0 ? b1 : b2; b != lca; b = m->body_parentid[b]) up.push_back(b);
    P.c[s].bodies.assign(up.rbegin(), up.rend());
  }
  for (int s = 0; s < 2; s++) {
    const int g = s == 0 ? g1 : g2;
    const std::vector<int>& bs = P.c[s].bodies;
    for (size_t i = 0; i < bs.size(); i++) {
      const int b = bs[i];
      for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
        if (m->jnt_type[j] != MJPCX_JNT_HINGE) return false;
        P.jnt.push_back(j);
        const bool lim = m->jnt_limited[j] != 0;
        const double ref = m->qpos0[m->jnt_qposadr[j]];  // (the hinge turns by qpos - qpos0; the range is one of qpos)
        P.lo.push_back(lim ? m->jnt_range[2 * j] - ref - (j == tight_jnt && tight_side == 0 ? tight_pad : pad) : -3.14159265358979323846);
        P.hi.push_back(lim ? m->jnt_range[2 * j + 1] - ref + (j == tight_jnt && tight_side == 1 ? tight_pad : pad) : 3.14159265358979323846);
        double L = norm3(m->jnt_pos + 3 * j);
        for (int jj = j + 1; jj < m->body_jntadr[b] + m->body_jntnum[b]; jj++) L += 2 * norm3(m->jnt_pos + 3 * jj);
        for (size_t k = i + 1; k < bs.size(); k++) {
          L += norm3(m->body_pos + 3 * bs[k]);
          // (a joint further down rotates about its own anchor: the anchor's offset is part of the path)
          for (int jj = m->body_jntadr[bs[k]]; jj < m->body_jntadr[bs[k]] + m->body_jntnum[bs[k]]; jj++) L += 2 * norm3(m->jnt_pos + 3 * jj);
        }
        L += norm3(m->geom_pos + 3 * g) + bound_radius(m, g);
        P.lever.push_back(L);
      }
    }
  }
  if (joints) *joints = P.jnt;
  if (P.jnt.empty() || P.jnt.size() > 3) return false;
  std::vector<double> c0(P.jnt.size()), half(P.jnt.size());
  for (size_t j = 0; j < P.jnt.size(); j++) { c0[j] = 0.5 * (P.lo[j] + P.hi[j]); half[j] = 0.5 * (P.hi[j] - P.lo[j]); }
  const bool ok = P.cell(c0, half);
  if (evals) *evals = P.evals;
  if (closest) *closest = P.closest;
  return ok;
}

// Two solids at the model's reference pose qpos0: the kernels' watch (the thin geom that contains g1 against the solid g2; oracle
// pair_thin_solid with as_thin) evaluated on the host. true: the pair is within its margin there -- every rollout that starts near
// qpos0 would fail with warning bit 128 at its first step, so mjpcx_create refuses such a model with one clear message instead.
inline bool solids_touch_at_qpos0(const mjpcx_model* m, int g1, int g2) {
  using namespace cull_detail;
  if (!is_solid(m->geom_type[g1]) || !is_solid(m->geom_type[g2])) return false;
  double gp[2][3], gR[2][9];
  for (int s = 0; s < 2; s++) {
    const int g = s == 0 ? g1 : g2;
    std::vector<int> chain;
    for (int b = m->geom_bodyid[g]; b > 0; b = m->body_parentid[b]) chain.push_back(b);
    double p[3] = {0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (auto it = chain.rbegin(); it != chain.rend(); ++it) {
      const int b = *it;
      const double* bp = m->body_pos + 3 * b; const double* bq = m->body_quat + 4 * b;
      // (a free joint's qpos0 IS the body's pose; hinges and slides are at zero displacement at qpos0, a ball joint at its reference quaternion)
      if (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] == MJPCX_JNT_FREE) { bp = m->qpos0 + m->jnt_qposadr[m->body_jntadr[b]]; bq = bp + 3; }
      double v[3], Rb[9];
      mv(v, R, bp);
      for (int k = 0; k < 3; k++) p[k] += v[k];
      q2m(Rb, bq);
      mm(R, R, Rb);
    }
    double v[3], Rg[9];
    mv(v, R, m->geom_pos + 3 * g);
    for (int k = 0; k < 3; k++) gp[s][k] = p[k] + v[k];
    q2m(Rg, m->geom_quat + 4 * g);
    mm(gR[s], R, Rg);
  }
  const int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  double h = 0, r = 0;
  solid::solid_as_thin<double>(t1 == MJPCX_GEOM_CYLINDER ? solid::kSolidCylinder : solid::kSolidBox, m->geom_size + 3 * g1, h, r);
  double rel[3], pl[3], al[3], n[3], cc[3];
  for (int k = 0; k < 3; k++) rel[k] = gp[0][k] - gp[1][k];
  for (int k = 0; k < 3; k++) {
    pl[k] = gR[1][k] * rel[0] + gR[1][3 + k] * rel[1] + gR[1][6 + k] * rel[2];
    al[k] = gR[1][k] * gR[0][2] + gR[1][3 + k] * gR[0][5] + gR[1][6 + k] * gR[0][8];
  }
  const double d = solid::thin_vs_solid<double>(t2 == MJPCX_GEOM_CYLINDER ? solid::kSolidCylinder : solid::kSolidBox, m->geom_size + 3 * g2, pl, al, h, r, n, cc);
  const double margin = m->geom_margin[g1] > m->geom_margin[g2] ? m->geom_margin[g1] : m->geom_margin[g2];
  return d < margin;
}

// the moving-geom pairs of a model in MuJoCo's order (lower geom type first, then lower index)
enum { kPairThin = 0,        // sphere | capsule both: the kernels' original narrow phase
       kPairThinSolid = 1,   // (sphere | capsule, box | cylinder): solid_pairs.h
       kPairSolids = 2,      // two solids: no narrow phase
       kPairOther = 3        // a geom type outside sphere | capsule | cylinder | box (MuJoCo's convex collider): no narrow phase
};
struct MovingPair {
  int g1, g2, kind;
  int apart;                  // proven never to touch (kinds with a solid only)
  int tight_jnt, tight_side;  // the joint side (0 lower, 1 upper) whose pad the proof had to take as kPairCullPadTight, or -1
};

// `moving[b]`: body b has a dof at or above it. Pairs with a solid are put to the proof only if `prove` is set.
inline void moving_pairs(const mjpcx_model* m, const std::vector<char>& moving, bool prove, std::vector<MovingPair>& out) {
  using namespace cull_detail;
  out.clear();
  const int ng = m->ngeom, nb = m->nbody;
  // (the proofs of a model cost ~0.1 s for the A1; a process that creates many contexts of one model pays once)
  static std::mutex mu;
  static std::map<uint64_t, std::vector<MovingPair>> cache;
  uint64_t key = 1469598103934665603ull;
  {
    auto mix = [&](const void* ptr, size_t bytes) { const unsigned char* c = (const unsigned char*)ptr; if (!ptr) return; for (size_t i = 0; i < bytes; i++) { key ^= c[i]; key *= 1099511628211ull; } };
    const int hdr[4] = {ng, nb, m->njnt, (prove ? 1 : 0) | (m->nexclude << 1)};
    mix(hdr, sizeof hdr);
    mix(moving.data(), moving.size());
    mix(m->geom_type, 4 * ng); mix(m->geom_bodyid, 4 * ng); mix(m->geom_contype, 4 * ng); mix(m->geom_conaffinity, 4 * ng);
    mix(m->geom_size, 24 * ng); mix(m->geom_pos, 24 * ng); mix(m->geom_quat, 32 * ng); mix(m->geom_margin, 8 * ng);
    mix(m->body_parentid, 4 * nb); mix(m->body_weldid, 4 * nb); mix(m->body_pos, 24 * nb); mix(m->body_quat, 32 * nb);
    mix(m->body_jntnum, 4 * nb); mix(m->body_jntadr, 4 * nb);
    mix(m->qpos0, 8 * (size_t)m->nq); mix(m->jnt_qposadr, 4 * m->njnt);
    mix(m->jnt_type, 4 * m->njnt); mix(m->jnt_limited, 4 * m->njnt); mix(m->jnt_range, 16 * m->njnt); mix(m->jnt_pos, 24 * m->njnt); mix(m->jnt_axis, 24 * m->njnt);
    mix(m->exclude_signature, 4 * (size_t)m->nexclude);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) { out = it->second; return; }
  }
  std::vector<int> weld(nb);  // (a caller that passes no body_weldid: every body is its own weld, as for a model without welds)
  for (int b = 0; b < nb; b++) weld[b] = m->body_weldid ? m->body_weldid[b] : b;
  for (int a = 0; a < ng; a++)
    for (int b = a + 1; b < ng; b++) {
      const int b1 = m->geom_bodyid[a], b2 = m->geom_bodyid[b];
      if (!moving[b1] || !moving[b2]) continue;
      if (!((m->geom_contype[a] & m->geom_conaffinity[b]) || (m->geom_contype[b] & m->geom_conaffinity[a]))) continue;
      const int w1 = weld[b1], w2 = weld[b2];
      if (w1 == w2) continue;
      const int pw1 = weld[m->body_parentid[w1]], pw2 = weld[m->body_parentid[w2]];
      if (w1 != 0 && w2 != 0 && (w1 == pw2 || w2 == pw1)) continue;
      const int sig = ((b1 < b2 ? b1 : b2) << 16) + (b1 < b2 ? b2 : b1);
      bool excluded = false;
      for (int e = 0; e < m->nexclude; e++) excluded |= m->exclude_signature[e] == sig;
      if (excluded) continue;
      const int ta = m->geom_type[a], tb = m->geom_type[b];
      MovingPair p;
      p.g1 = ta > tb ? b : a; p.g2 = ta > tb ? a : b;
      const int t1 = m->geom_type[p.g1], t2 = m->geom_type[p.g2];
      p.apart = 0; p.tight_jnt = -1; p.tight_side = 0;
      if (is_thin(t1) && is_thin(t2)) p.kind = kPairThin;
      else if ((is_thin(t1) || is_solid(t1)) && is_solid(t2)) {
        p.kind = is_thin(t1) ? kPairThinSolid : kPairSolids;
        const double margin = m->geom_margin[p.g1] > m->geom_margin[p.g2] ? m->geom_margin[p.g1] : m->geom_margin[p.g2];
        if (prove) {
          std::vector<int> js;
          int evals = 0;
          double closest = 0;
          if (pair_never_touches(m, p.g1, p.g2, margin, kPairCullPad, &evals, &closest, -1, 0, 0, &js)) p.apart = 1;
          else if (js.size() <= 3)  // (the proof may hold with one joint side held tighter)
            for (size_t k = 0; k < 2 * js.size() && !p.apart; k++)
              if (m->jnt_limited[js[k / 2]] && pair_never_touches(m, p.g1, p.g2, margin, kPairCullPad, nullptr, nullptr, js[k / 2], (int)(k & 1), kPairCullPadTight)) {
                p.apart = 1; p.tight_jnt = js[k / 2]; p.tight_side = (int)(k & 1);
              }
        }
      } else p.kind = kPairOther;
      out.push_back(p);
    }
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() > 64) cache.clear();
  cache[key] = out;
}

}  // namespace mjpcx
