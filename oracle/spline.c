/* spline.c -- oracle restatement of mjpc::spline::TimeSpline
 * (mjpc/spline/spline.cc). TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Pinned by the reference's own known answers, mjpc/test/spline/spline_test.cc
 * (ported in tests/test_oracle_spline.py).
 *
 * The reference keeps node values in a ring buffer and node times in a deque;
 * only the observable behaviour (time-ordered nodes, Sample, DiscardBefore,
 * ShiftTime, AddNode at either end) is restated, over flat arrays. */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

void ospline_init(OSpline* s, int dim, int interp) {
  memset(s, 0, sizeof *s);
  s->dim = dim;
  s->interp = interp;
}
void ospline_free(OSpline* s) {
  free(s->times);
  free(s->values);
  memset(s, 0, sizeof *s);
}
void ospline_clear(OSpline* s) { s->size = 0; } /* spline.cc:199-204 */
static void reserve(OSpline* s, int n) {
  if (n <= s->cap) return;
  int cap = s->cap ? s->cap : 4;
  while (cap < n) cap *= 2;
  s->times = (double*)realloc(s->times, sizeof(double) * cap);
  s->values = (double*)realloc(s->values, sizeof(double) * cap * (s->dim > 0 ? s->dim : 1));
  s->cap = cap;
}
void ospline_copy(OSpline* dst, const OSpline* src) {
  dst->dim = src->dim;
  dst->interp = src->interp;
  dst->size = 0;
  dst->cap = 0;
  free(dst->times); free(dst->values);
  dst->times = dst->values = NULL;
  reserve(dst, src->size);
  dst->size = src->size;
  memcpy(dst->times, src->times, sizeof(double) * src->size);
  memcpy(dst->values, src->values, sizeof(double) * src->size * src->dim);
}
void ospline_set_interpolation(OSpline* s, int interp) { s->interp = interp; }

/* spline.cc:213-249: nodes may only be appended after the last or before the first */
int ospline_add_node(OSpline* s, double time, const double* values) {
  int dim = s->dim;
  if (s->size && !(time > s->times[s->size - 1] || time < s->times[0])) return -1;
  reserve(s, s->size + 1);
  int idx;
  if (s->size == 0 || time > s->times[s->size - 1]) {
    idx = s->size;
  } else {
    memmove(s->times + 1, s->times, sizeof(double) * s->size);
    memmove(s->values + dim, s->values, sizeof(double) * s->size * dim);
    idx = 0;
  }
  s->times[idx] = time;
  if (values) memcpy(s->values + idx * dim, values, sizeof(double) * dim);
  else memset(s->values + idx * dim, 0, sizeof(double) * dim);
  s->size++;
  return idx;
}

/* std::upper_bound(times, time): first index with times[i] > time */
static int upper_bound(const OSpline* s, double time) {
  int lo = 0, hi = s->size;
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (s->times[mid] > time) hi = mid; else lo = mid + 1;
  }
  return lo;
}

/* spline.cc:269-287 */
static double slope(const OSpline* s, int node, int k) {
  int dim = s->dim;
  const double* v = s->values;
  const double* t = s->times;
  if (node == 0) return (v[dim + k] - v[k]) / (t[1] - t[0]);
  if (node == s->size - 1)
    return (v[node * dim + k] - v[(node - 1) * dim + k]) / (t[node] - t[node - 1]);
  return 0.5 * (v[(node + 1) * dim + k] - v[node * dim + k]) / (t[node + 1] - t[node]) +
         0.5 * (v[node * dim + k] - v[(node - 1) * dim + k]) / (t[node] - t[node - 1]);
}

/* spline.cc:103-156 */
void ospline_sample(const OSpline* s, double time, double* out) {
  int dim = s->dim;
  if (s->size == 0) { memset(out, 0, sizeof(double) * dim); return; }
  int up = upper_bound(s, time);
  if (up == s->size) { memcpy(out, s->values + (up - 1) * dim, sizeof(double) * dim); return; }
  if (up == 0) { memcpy(out, s->values, sizeof(double) * dim); return; }
  int lo = up - 1;
  double tl = s->times[lo], tu = s->times[up];
  double t = (time - tl) / (tu - tl);
  const double* vl = s->values + lo * dim;
  const double* vu = s->values + up * dim;
  switch (s->interp) {
    case MJPCX_SPLINE_ZERO:
      memcpy(out, vl, sizeof(double) * dim);
      return;
    case MJPCX_SPLINE_LINEAR:
      for (int i = 0; i < dim; i++) out[i] = vl[i] * (1 - t) + vu[i] * t;
      return;
    case MJPCX_SPLINE_CUBIC: { /* spline.cc:251-267 */
      double c0 = 2.0 * t * t * t - 3.0 * t * t + 1.0;
      double c1 = (t * t * t - 2.0 * t * t + t) * (tu - tl);
      double c2 = -2.0 * t * t * t + 3 * t * t;
      double c3 = (t * t * t - t * t) * (tu - tl);
      for (int i = 0; i < dim; i++) {
        double m0 = slope(s, lo, i), m1 = slope(s, up, i);
        out[i] = c0 * vl[i] + c1 * m0 + c2 * vu[i] + c3 * m1;
      }
      return;
    }
  }
}

/* spline.cc:164-187 */
int ospline_discard_before(OSpline* s, double time) {
  int last = upper_bound(s, time);
  if (last == 0) return 0;
  int keep = s->interp == MJPCX_SPLINE_CUBIC ? 1 : 0;
  last--;
  while (last != 0 && keep) { last--; keep--; }
  int n = last;
  if (n > 0) {
    memmove(s->times, s->times + n, sizeof(double) * (s->size - n));
    memmove(s->values, s->values + n * s->dim, sizeof(double) * (s->size - n) * s->dim);
    s->size -= n;
  }
  return n;
}

/* spline.cc:189-197 */
void ospline_shift_time(OSpline* s, double start_time) {
  if (!s->size) return;
  double shift = start_time - s->times[0];
  for (int i = 0; i < s->size; i++) s->times[i] += shift;
}
