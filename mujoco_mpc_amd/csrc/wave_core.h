// wave_core.h -- per-precision part of rollout_wave.h (no include guard: included once per working type, see there):
// LDS state of one candidate, wave-level primitives (sync, reductions, broadcasts), small math, register-resident
// Cholesky, impedance and norm helpers.
namespace mjpcx { namespace WAVE_NS {
typedef WaveModelT<wreal> WModel;
typedef WaveTaskT<wreal> WTask;
typedef wreal w_acc4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int w_mfma_row(int lk, int rg) { return sizeof(wreal) == 8 ? lk + 4 * rg : 4 * lk + rg; }


// Single-wavefront workgroups: the ordering point between dependent LDS phases only has to (a) stop the compiler from
// moving LDS accesses across it and (b) wait for this wave's outstanding LDS operations -- no s_barrier, and no wait
// on outstanding global stores (which __syncthreads() would add through its global-memory fence).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only (gfx9 encoding: vmcnt = max, expcnt = max)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#define WSYNC() wave_lds_sync()

struct WaveContact {
  int g1, g2, dim, dim0, efc;
  int nrow;          // rows built: dim (frictionless / elliptic) or up to 2 (dim0 - 1) pyramid edges
  unsigned dofmask;  // dofs with a non-zero Jacobian column (the chains of the two bodies, minus their common part)
  wreal dist, margin, includemargin, mu;
  wreal pos[3], frame[9], friction[5], solref[2], solimp[5];
};
enum { kEfcFriction = 0, kEfcLimit = 1, kEfcNormal = 2, kEfcElliptic = 3, kEfcConeRow = 4, kEfcTendon = 5, kEfcPyramid = 6 };
enum { kZoneTop = 0, kZoneMiddle = 1, kZoneBottom = 2 };

// LDS state of one candidate ("mjData")
struct WaveData {
  wreal *qpos, *qvel, *ctrl;
  wreal *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *site_xpos;
  wreal *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc, *cfrc_sub, *subtree_linvel;
  wreal *M, *L, *H, *Ldinv, *dinv;
  wreal *qfrc_passive, *qfrc_bias, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qacc, *qfrc_constraint;
  wreal *actuator_force, *grad, *search, *Ma, *Ms, *tmpv, *qacc_warm;
  wreal *efc_J, *efc_pos, *efc_margin, *efc_D, *efc_R, *efc_aref, *efc_floss, *efc_force, *jar, *jv;
  int *efc_type, *efc_id, *efc_zone;
  wreal* coneH;  // kWaveMaxCon x 21: lower triangle (j >= k at j (j + 1) / 2 + k) of each cone's symmetric Hessian block
  wreal* foot_xpos;  // geom_xpos of the geoms the residual reads (4 x 3)
  wreal* residual;
  wreal* terms;
  WaveContact* con;
  int* counters;  // [0] ncon [1] nefc [2] warning
  wreal* scal;   // scratch scalars
  wreal* xfrc;   // 6 nbody: force, torque at each body's centre of mass (mjData.xfrc_applied); nullptr outside NoisyRollout
};

// Sum over the 64 lanes, same value returned in every lane. DPP row shifts / row broadcasts (6 steps of two 32-bit DPP
// moves + one add, zero fill at the row boundaries) and one v_readlane of lane 63 -- the ds_bpermute butterfly of
// __shfl_xor costs a dependent LDS-crossbar round trip per step, and the Newton solver reduces ~16 times per iteration.
__device__ __forceinline__ wreal wave_sum(wreal v) {
  v += dpp_move<0x111, 0xf>(v);  // row_shr:1
  v += dpp_move<0x112, 0xf>(v);  // row_shr:2
  v += dpp_move<0x114, 0xf>(v);  // row_shr:4
  v += dpp_move<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of every row of 16 holds the row total
  v += dpp_move<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
  return w_readlane(v, 63);
}
__device__ __forceinline__ void q_mul(wreal* r, const wreal* a, const wreal* b) {
  const wreal w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const wreal x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const wreal y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  const wreal z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
__device__ __forceinline__ void q2mat(wreal* m, const wreal* q) {
  const wreal q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  const wreal q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
__device__ __forceinline__ void q_rot(wreal* r, const wreal* v, const wreal* q) {  // rotate v by q
  wreal m[9];
  q2mat(m, q);
  const wreal x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
               z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void q_norm(wreal* q) {
  const wreal n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < kMinVal) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { const wreal s = WL(1.0) / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
__device__ __forceinline__ void aa2quat(wreal* q, const wreal* axis, wreal angle) {
  wreal s, c;
  w_sincos(WL(0.5) * angle, &s, &c);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
__device__ __forceinline__ void mv3(wreal* r, const wreal* m, const wreal* v) {
  const wreal x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
               z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void cr3(wreal* r, const wreal* a, const wreal* b) {
  const wreal x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
// spatial helpers on plain pointers (same formulas as device_common.h's array versions)
__device__ __forceinline__ void w_inert_com(wreal* res, const wreal* inert, const wreal* mat, const wreal* dif, wreal mass) {
  wreal tmp[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[3 * i + j] = mat[3 * i + j] * inert[j];
  res[0] = tmp[0] * mat[0] + tmp[1] * mat[1] + tmp[2] * mat[2];
  res[1] = tmp[3] * mat[3] + tmp[4] * mat[4] + tmp[5] * mat[5];
  res[2] = tmp[6] * mat[6] + tmp[7] * mat[7] + tmp[8] * mat[8];
  res[3] = tmp[0] * mat[3] + tmp[1] * mat[4] + tmp[2] * mat[5];
  res[4] = tmp[0] * mat[6] + tmp[1] * mat[7] + tmp[2] * mat[8];
  res[5] = tmp[3] * mat[6] + tmp[4] * mat[7] + tmp[5] * mat[8];
  res[0] += mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] += mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] += mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] -= mass * dif[0] * dif[1];
  res[4] -= mass * dif[0] * dif[2];
  res[5] -= mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2];
  res[9] = mass;
}
__device__ __forceinline__ void w_mul_inert(wreal* res, const wreal* i, const wreal* v) {
  res[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  res[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  res[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  res[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  res[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  res[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ void w_cross_motion(wreal* res, const wreal* vel, const wreal* v) {
  res[0] = -vel[2] * v[1] + vel[1] * v[2];
  res[1] = vel[2] * v[0] - vel[0] * v[2];
  res[2] = -vel[1] * v[0] + vel[0] * v[1];
  res[3] = -vel[2] * v[4] + vel[1] * v[5];
  res[4] = vel[2] * v[3] - vel[0] * v[5];
  res[5] = -vel[1] * v[3] + vel[0] * v[4];
  res[3] += -vel[5] * v[1] + vel[4] * v[2];
  res[4] += vel[5] * v[0] - vel[3] * v[2];
  res[5] += -vel[4] * v[0] + vel[3] * v[1];
}
__device__ __forceinline__ void w_cross_force(wreal* res, const wreal* vel, const wreal* f) {
  res[0] = -vel[2] * f[1] + vel[1] * f[2];
  res[1] = vel[2] * f[0] - vel[0] * f[2];
  res[2] = -vel[1] * f[0] + vel[0] * f[1];
  res[3] = -vel[2] * f[4] + vel[1] * f[5];
  res[4] = vel[2] * f[3] - vel[0] * f[5];
  res[5] = -vel[1] * f[3] + vel[0] * f[4];
  res[0] += -vel[5] * f[4] + vel[4] * f[5];
  res[1] += vel[5] * f[3] - vel[3] * f[5];
  res[2] += -vel[4] * f[3] + vel[3] * f[4];
}
__device__ __forceinline__ wreal w_dot6(const wreal* a, const wreal* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

// ------------------------------------------------------------------ dense SPD algebra, lane i owns row i (n <= NMAX <= 32)
// Register-resident: lane i keeps ROW i of the matrix in NMAX registers; values of other rows arrive through
// v_readlane (uniform source lane), so a factorisation or a triangular solve makes no LDS round trips on its
// dependent chain. Pivots are applied as reciprocals (one v_rsq-based 1/sqrt per column instead of a sqrt and a
// division per row), which differs from the oracle's divisions by <= 2 ulp.
__device__ __forceinline__ wreal wbcast(wreal v, int src) { return w_readlane(v, src); }  // src must be wave-uniform
// position of the n-th (0-based) set bit of m (n < popcount(m))
__device__ __forceinline__ int w_nth_bit(unsigned m, int n) {
  int base = 0, c;
  c = __popc(m & 0xFFFFu); if (n >= c) { n -= c; m >>= 16; base += 16; }
  c = __popc(m & 0xFFu);   if (n >= c) { n -= c; m >>= 8; base += 8; }
  c = __popc(m & 0xFu);    if (n >= c) { n -= c; m >>= 4; base += 4; }
  c = __popc(m & 0x3u);    if (n >= c) { n -= c; m >>= 2; base += 2; }
  c = (int)(m & 1u);       if (n >= c) base += 1;
  return base;
}
// in-place: on exit the lower triangle of A (LDS, ld n) holds L with A = L L', dinv[j] = 1 / L[j][j].
// (not inlined: five call sites per step, and the step loop has to stay inside the instruction cache)
template <int NMAX>
__device__ __noinline__ bool wave_chol(wreal* A, wreal* dinv, int n_, int lane) {
  // (arguments of an out-of-line function arrive in VGPRs: make the size scalar again, or every guard below becomes a
  // vector compare + exec-mask branch)
  const int n = __builtin_amdgcn_readfirstlane(n_);
  wreal row[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; k++) row[k] = (lane < n && k <= lane) ? A[lane * n + k] : WL(0.0);
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    if (j < n && ok) {
      const wreal djj = wbcast(row[j], j);
      if (!(djj > kMinVal)) {
        ok = false;
      } else {
        const wreal inv = rsqrt(djj);
        const wreal lij = lane == j ? djj * inv : row[j] * inv;
        row[j] = lij;
        if (lane == j) dinv[j] = inv;
        // no guards in the update: lanes >= n hold zero rows (their broadcast l_kj is 0), lanes < k update an
        // upper-triangle slot nobody reads
#pragma unroll
        for (int k = j + 1; k < NMAX; k++) row[k] -= lij * wbcast(lij, k);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NMAX; k++) if (lane < n && k <= lane) A[lane * n + k] = row[k];
  WSYNC();
  return ok;
}
// x := (L L')^-1 x, x in LDS
template <int NMAX>
__device__ __noinline__ void wave_chol_solve(wreal* x, const wreal* L, const wreal* dinv, int n_, int lane) {
  const int n = __builtin_amdgcn_readfirstlane(n_);
  wreal row[NMAX], col[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; k++) {
    row[k] = (lane < n && k < lane) ? L[lane * n + k] : WL(0.0);
    col[k] = (lane < n && k > lane && k < n) ? L[k * n + lane] : WL(0.0);
  }
  wreal b = lane < n ? x[lane] : WL(0.0);
  const wreal mydinv = lane < n ? dinv[lane] : WL(0.0);
  // unguarded sweeps: for j >= n the broadcast pivot reciprocal is 0, so the step is a no-op
#pragma unroll
  for (int j = 0; j < NMAX; j++) {
    const wreal yj = wbcast(b, j) * wbcast(mydinv, j);
    b = lane == j ? yj : (lane > j ? b - row[j] * yj : b);
  }
#pragma unroll
  for (int j = NMAX - 1; j >= 0; j--) {
    const wreal xj = wbcast(b, j) * wbcast(mydinv, j);
    b = lane == j ? xj : (lane < j ? b - col[j] * xj : b);
  }
  if (lane < n) x[lane] = b;
  WSYNC();
}

// solimp -> impedance at violation `dist` (oracle impedance()). The general exponent needs pow() (~1000 instructions each): that path
// is an out-of-line call taken only for non-default solimp; the default (power 2) and linear cases are inline, so the callers' live
// registers are not saved and restored around a call on every contact.
__device__ __noinline__ wreal w_impedance_pow(wreal x, wreal mid, wreal power) {
  if (x <= mid) return pow(x, power) / pow(mid, power - 1);
  return 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
}
__device__ __forceinline__ wreal w_impedance(const wreal* solimp, wreal dist) {
  wreal dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = fmin(fmax(dmin, kMinImp), kMaxImp);
  dmax = fmin(fmax(dmax, kMinImp), kMaxImp);
  if (power < 1) power = 1;
  mid = fmin(fmax(mid, kMinImp), kMaxImp);
  if (dmin == dmax || width <= kMinVal) return WL(0.5) * (dmin + dmax);
  const wreal x = fabs(dist) / width;
  if (x >= 1) return dmax;
  if (x <= 0) return dmin;
  wreal y;
  if (power == 1) y = x;
  else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);  // MuJoCo's default
  else y = w_impedance_pow(x, mid, power);
  return dmin + y * (dmax - dmin);
}
template <class MODEL>
__device__ __forceinline__ void w_solref_kb(const MODEL& m, const wreal* solref, const wreal* solimp, wreal& k, wreal& b) {
  const wreal dmax = fmin(fmax(solimp[1], kMinImp), kMaxImp);
  if (solref[0] > 0) {
    wreal tc = solref[0];
    if (!(m.disableflags & MJPCX_DSBL_REFSAFE) && tc < 2 * m.timestep) tc = 2 * m.timestep;
    k = WL(1.0) / (dmax * dmax * tc * tc * solref[1] * solref[1]);
    b = WL(2.0) / (dmax * tc);
  } else {
    k = -solref[0] / (dmax * dmax);
    b = -solref[1] / dmax;
  }
}
__device__ __forceinline__ void w_make_frame(wreal* frame) {
  wreal* x = frame; wreal* y = frame + 3; wreal* z = frame + 6;
  if (x[1] < WL(0.5) && x[1] > -WL(0.5)) { y[0] = 0; y[1] = 1; y[2] = 0; }
  else { y[0] = 0; y[1] = 0; y[2] = 1; }
  const wreal dt = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
  for (int k = 0; k < 3; k++) y[k] -= dt * x[k];
  const wreal n = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
  for (int k = 0; k < 3; k++) y[k] /= n;
  cr3(z, x, y);
}
constexpr wreal kMinMu = WL(1e-5);

// mjpc::Norm value (mjpc/norm.cc:50-210) in two stages so that the transcendental part runs one lane per residual ENTRY:
// every norm is g(sum_i f(x_i)); w_norm_elem is f, w_norm_finish is g applied to the (sequentially accumulated) sum.
__device__ __forceinline__ wreal w_norm_elem(wreal x, int type, wreal p, wreal q) {
  switch (type) {
    case -1: return x;
    case 0: case 1: case 2: return x * x;
    case 3: return p * p * (cosh(x / p) - WL(1.0));
    case 5: return pow(fabs(x), p);
    case 6: return sqrt(x * x + p * p) - p;
    case 7: return pow(pow(fabs(x), q) + pow(p, q), 1 / q) - p;
    case 8: return p > 0 ? p * log(1 + exp(x / p)) : (x > 0 ? x : WL(0.0));
    default: return 0;
  }
}
__device__ __forceinline__ wreal w_norm_finish(wreal c, int type, wreal p, wreal q) {
  switch (type) {
    case 0: return c * WL(0.5);
    case 1: return pow(pow(c, q / 2) + pow(p, q), 1 / q) - p;
    case 2: return sqrt(c + p * p) - p;
    default: return c;
  }
}
__device__ __forceinline__ wreal w_norm_value(const wreal* x, int n, int type, wreal p, wreal q) {  // serial form
  wreal c = 0;
  for (int i = 0; i < n; i++) c += w_norm_elem(x[i], type, p, q);
  return w_norm_finish(c, type, p, q);
}

} }  // namespace mjpcx::WAVE_NS
