// solid_pairs.h -- narrow phase of a THIN geom (sphere | capsule: a segment of half length h swept by a ball of radius r) against a
// SOLID geom (box | cylinder) between two moving bodies: the pairs MuJoCo collides with mjc_SphereBox / mjc_CapsuleBox /
// mjc_SphereCylinder and, for (capsule, cylinder), its general convex collider (engine_collision_driver.c's table; the call sites are
// mj_step / mj_forward in mjpc/trajectory.cc:158,198). The A1 of the north-star task carries such pairs: the trunk's boxes and
// cylinders and the hips' cylinders against thigh / calf / foot geoms (mjpc/tasks/quadruped/a1.xml.patch).
//
// One algorithm for all four pairs, in the solid's frame: the point c(t) = p + t a, |t| <= h of the thin geom's axis nearest to the
// solid is the minimiser of the convex function D(c(t)) (D: distance of a point to the solid, 0 inside); the contact is that of the
// ball at c(t*) with the solid. t* is the root of g(t) = 1/2 d/dt D^2, which is monotone:
//   box       g is piecewise linear with <= 6 kinks (the faces' planes): bracket narrowed kink by kink, then one interpolation -- exact;
//   cylinder  kinks where the axis crosses the cap planes (<= 2) and the side's surface (<= 2); between kinks g is smooth: bracket
//             narrowed the same way, then a fixed number of safeguarded Newton steps (the rim distance has no closed form).
// A centre inside the solid leaves through the nearest face (as the sphere-box test of the static pairs does); an axis that crosses the
// solid takes the middle of the crossing; an axis parallel to a face (a whole interval of nearest points) takes the interval's middle.
// One contact per pair (MuJoCo's capsule-box test can add a second one for an axis parallel to a face). The same arithmetic is restated in C in oracle/contact.inc (thin_vs_solid); tests/test_solid_pairs.py holds the
// two against each other and against a brute-force distance.
//
// Host and device: a plain template over the scalar type; SP_FN carries the function qualifiers.
#pragma once
#include <math.h>

#ifndef SP_FN
#ifdef __HIPCC__
#define SP_FN __host__ __device__ __forceinline__
#else
#define SP_FN static inline
#endif
#endif

namespace mjpcx { namespace solid {

enum { kSolidBox = 0, kSolidCylinder = 1 };

template <class T> SP_FN T sp_tiny() { return sizeof(T) == 8 ? (T)1e-14 : (T)1e-7; }
template <class T> SP_FN T sp_excess(T x, T s) { return x > s ? x - s : (x < -s ? x + s : (T)0); }

// ---- box (half sizes s)
template <class T> SP_FN T box_g(const T* s, const T* p, const T* a, T t) {
  return sp_excess(p[0] + t * a[0], s[0]) * a[0] + sp_excess(p[1] + t * a[1], s[1]) * a[1] + sp_excess(p[2] + t * a[2], s[2]) * a[2];
}
template <class T> SP_FN T box_segment_param(const T* s, const T* p, const T* a, T h) {
  T lo = -h, hi = h;
  T glo = box_g(s, p, a, lo), ghi = box_g(s, p, a, hi);
  if (glo >= 0) return lo;
  if (ghi <= 0) return hi;
  for (int k = 0; k < 3; k++) {
    if (!(fabs(a[k]) > sp_tiny<T>())) continue;
    const T inv = (T)1 / a[k];
    for (int e = 0; e < 2; e++) {
      const T tb = ((e ? s[k] : -s[k]) - p[k]) * inv;
      if (tb > lo && tb < hi) {
        const T gb = box_g(s, p, a, tb);
        if (gb <= 0) { lo = tb; glo = gb; } else { hi = tb; ghi = gb; }
      }
    }
  }
  return lo - glo * (hi - lo) / (ghi - glo);
}
// the middle of the part of the axis inside the box (called when c(t*) is inside)
template <class T> SP_FN T box_crossing_mid(const T* s, const T* p, const T* a, T h) {
  T tl = -h, th = h;
  for (int k = 0; k < 3; k++) {
    if (!(fabs(a[k]) > sp_tiny<T>())) continue;
    const T inv = (T)1 / a[k];
    const T t1 = (-s[k] - p[k]) * inv, t2 = (s[k] - p[k]) * inv;
    const T a1 = t1 < t2 ? t1 : t2, a2 = t1 < t2 ? t2 : t1;
    tl = a1 > tl ? a1 : tl; th = a2 < th ? a2 : th;
  }
  return th >= tl ? (T)0.5 * (tl + th) : tl;
}

// ---- cylinder (radius R, half length H, axis z)
template <class T> SP_FN T cyl_g(T R, T H, const T* p, const T* a, T t, T& dg) {
  const T x = p[0] + t * a[0], y = p[1] + t * a[1], z = p[2] + t * a[2];
  const T rho2 = x * x + y * y, w = x * a[0] + y * a[1];
  T g = 0; dg = 0;
  if (rho2 > R * R) {
    const T rho = sqrt(rho2), f = (T)1 - R / rho;
    g = f * w; dg = R * w * w / (rho2 * rho) + f * (a[0] * a[0] + a[1] * a[1]);
  }
  const T ez = sp_excess(z, H);
  if (ez != 0) { g += ez * a[2]; dg += a[2] * a[2]; }
  return g;
}
template <class T> SP_FN T cyl_segment_param(T R, T H, const T* p, const T* a, T h) {
  T lo = -h, hi = h, d0;
  T glo = cyl_g(R, H, p, a, lo, d0), ghi = cyl_g(R, H, p, a, hi, d0);
  if (glo >= 0) return lo;
  if (ghi <= 0) return hi;
  T tb[4]; int nb = 0;
  if (fabs(a[2]) > sp_tiny<T>()) { const T inv = (T)1 / a[2]; tb[nb++] = (-H - p[2]) * inv; tb[nb++] = (H - p[2]) * inv; }
  const T A = a[0] * a[0] + a[1] * a[1];
  if (A > sp_tiny<T>()) {
    const T B = p[0] * a[0] + p[1] * a[1], Cc = p[0] * p[0] + p[1] * p[1] - R * R, disc = B * B - A * Cc;
    if (disc > 0) { const T sq = sqrt(disc), inv = (T)1 / A; tb[nb++] = (-B - sq) * inv; tb[nb++] = (-B + sq) * inv; }
  }
  for (int i = 0; i < nb; i++)
    if (tb[i] > lo && tb[i] < hi) {
      const T gb = cyl_g(R, H, p, a, tb[i], d0);
      if (gb <= 0) { lo = tb[i]; glo = gb; } else { hi = tb[i]; ghi = gb; }
    }
  T t = lo - glo * (hi - lo) / (ghi - glo);
  for (int it = 0; it < 6; it++) {
    T dg;
    const T gt = cyl_g(R, H, p, a, t, dg);
    if (gt <= 0) { lo = t; glo = gt; } else { hi = t; ghi = gt; }
    T tn = dg > sp_tiny<T>() ? t - gt / dg : t;
    if (!(tn >= lo && tn <= hi)) tn = lo - glo * (hi - lo) / (ghi - glo);
    t = tn;
  }
  return t;
}
template <class T> SP_FN T cyl_crossing_mid(T R, T H, const T* p, const T* a, T h) {
  T tl = -h, th = h;
  if (fabs(a[2]) > sp_tiny<T>()) {
    const T inv = (T)1 / a[2];
    const T t1 = (-H - p[2]) * inv, t2 = (H - p[2]) * inv;
    const T a1 = t1 < t2 ? t1 : t2, a2 = t1 < t2 ? t2 : t1;
    tl = a1 > tl ? a1 : tl; th = a2 < th ? a2 : th;
  }
  const T A = a[0] * a[0] + a[1] * a[1];
  if (A > sp_tiny<T>()) {
    const T B = p[0] * a[0] + p[1] * a[1], Cc = p[0] * p[0] + p[1] * p[1] - R * R, disc = B * B - A * Cc;
    if (disc > 0) { const T sq = sqrt(disc), inv = (T)1 / A; const T a1 = (-B - sq) * inv, a2 = (-B + sq) * inv; tl = a1 > tl ? a1 : tl; th = a2 < th ? a2 : th; }
  }
  return th >= tl ? (T)0.5 * (tl + th) : tl;
}

// The contact of a thin geom (axis point p, unit axis a, half length h -- 0 for a sphere --, radius r; all in the SOLID's frame) with a
// solid (kind, size: box half sizes | cylinder radius, half length). Returns the distance (negative: penetration); n: unit normal from the
// thin geom to the solid, c: the point of the thin geom's axis the contact belongs to (both in the solid's frame). The contact position
// is c + n (r + dist / 2).
template <class T> SP_FN T thin_vs_solid(int kind, const T* size, const T* p, const T* a, T h, T r, T* n, T* c) {
  T t = 0;
  if (h > 0) {
    // from both ends of the axis: where the minimum is attained on a whole interval (an axis parallel to a face, to the side or to a cap)
    // the two searches stop at its two ends and the contact takes the middle; otherwise they agree
    T te[2];
    for (int e = 0; e < 2; e++) {
      const T sg = e ? (T)-1 : (T)1;
      const T ae[3] = {sg * a[0], sg * a[1], sg * a[2]};
      te[e] = sg * (kind == kSolidBox ? box_segment_param(size, p, ae, h) : cyl_segment_param(size[0], size[1], p, ae, h));
    }
    t = (T)0.5 * (te[0] + te[1]);
  }
  for (int k = 0; k < 3; k++) c[k] = p[k] + t * a[k];
  if (kind == kSolidBox) {
    T e[3] = {sp_excess(c[0], size[0]), sp_excess(c[1], size[1]), sp_excess(c[2], size[2])};
    if (e[0] == 0 && e[1] == 0 && e[2] == 0) {  // the axis point is inside the box: out through the nearest face
      if (h > 0) { t = box_crossing_mid(size, p, a, h); for (int k = 0; k < 3; k++) c[k] = p[k] + t * a[k]; }
      int best = 0; T bd = size[0] - fabs(c[0]);
      for (int k = 1; k < 3; k++) { const T dd = size[k] - fabs(c[k]); if (dd < bd) { bd = dd; best = k; } }
      for (int k = 0; k < 3; k++) n[k] = 0;
      n[best] = c[best] >= 0 ? (T)-1 : (T)1;
      return -bd - r;
    }
    const T D = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), inv = (T)1 / D;
    for (int k = 0; k < 3; k++) n[k] = -e[k] * inv;
    return D - r;
  }
  const T R = size[0], H = size[1];
  T rho = sqrt(c[0] * c[0] + c[1] * c[1]);
  T er = rho > R ? rho - R : (T)0, ez = sp_excess(c[2], H);
  if (er == 0 && ez == 0) {  // inside the cylinder: out through the side or the nearer cap
    if (h > 0) { t = cyl_crossing_mid(R, H, p, a, h); for (int k = 0; k < 3; k++) c[k] = p[k] + t * a[k]; rho = sqrt(c[0] * c[0] + c[1] * c[1]); }
    const T dr = R - rho, dz = H - fabs(c[2]);
    if (dr < dz) {
      if (rho > sp_tiny<T>()) { n[0] = -c[0] / rho; n[1] = -c[1] / rho; } else { n[0] = -1; n[1] = 0; }
      n[2] = 0;
      return -dr - r;
    }
    n[0] = n[1] = 0; n[2] = c[2] >= 0 ? (T)-1 : (T)1;
    return -dz - r;
  }
  const T D = sqrt(er * er + ez * ez), inv = (T)1 / D;
  const T sc = er > 0 ? er / rho : (T)0;
  n[0] = -c[0] * sc * inv; n[1] = -c[1] * sc * inv; n[2] = -ez * inv;
  return D - r;
}

// A lower bound of the distance between two SOLIDS (cylinder | box pairs have no narrow phase here): solid 1 is replaced by a thin geom
// that contains it (cylinder: the capsule of its radius and half length; box: the ball about its centre). Used to PROVE separation (the
// bake-time cull of pair_cull.h; the oracle's run-time check), never to make a contact.
template <class T> SP_FN void solid_as_thin(int kind, const T* size, T& h, T& r) {
  if (kind == kSolidCylinder) { h = size[1]; r = size[0]; }
  else { h = 0; r = sqrt(size[0] * size[0] + size[1] * size[1] + size[2] * size[2]); }
}

}}  // namespace mjpcx::solid
