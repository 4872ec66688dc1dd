// quad_kernel.h -- the gfx950 launch shape of the QUAD kernel family (quad_step.h): FOUR LANES PER CANDIDATE, one per leg of
// the floating-base quadruped, sixteen candidates per wavefront. The north-star workload (Quadruped, 16384 candidates) is
// 1024 wavefronts = one per SIMD of the 256 CUs, each with the whole 512-entry register file of its SIMD; the per-candidate
// state that the wavefront-per-candidate kernels keep in a 17.6 KB LDS arena is the four lanes' registers here, and a
// reduction over the legs is two DPP quad permutes instead of an LDS round trip.
//
//   qd_sum    v + quad_perm[1,0,3,2](v), then + quad_perm[2,3,0,1]: every lane adds the same pairs in the same order, so the four
//             lanes hold bit-identical sums and every branch on one is quad-uniform
//   qd_rot<D> lane (l + D) mod 4 of the quad;  qd_bcast<K> lane K;  qd_or bitwise or
//
// The model (quad_model.h, ~12 KB) is staged into LDS once per workgroup (a workgroup is ONE wavefront: no barrier on the step
// path) and read with lane-dependent (per-leg) addresses; the pair-parameter tables stay in global memory (read when a contact
// is instantiated). A candidate the quad form does not cover is flagged in failure[] (kQFallback) and rolled out again, from
// the start, by rollout_tree_kernel (tree_kernel.h, mode bit 32): results do not depend on which kernel produced them.
#pragma once
#include <hip/hip_runtime.h>

#include "quad_model.h"

namespace mjpcx { namespace quad {
template <int CTRL> __device__ __forceinline__ double qdpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
} }
__device__ __forceinline__ double qd_sum(double v) {
  v += mjpcx::quad::qdpp<0xB1>(v);  // quad_perm:[1,0,3,2]
  v += mjpcx::quad::qdpp<0x4E>(v);  // quad_perm:[2,3,0,1]
  return v;
}
template <int K> __device__ __forceinline__ double qd_bcast(double v) { return mjpcx::quad::qdpp<K * 0x55>(v); }
template <int D> __device__ __forceinline__ double qd_rot(double v) {
  return mjpcx::quad::qdpp<(((0 + D) & 3) | (((1 + D) & 3) << 2) | (((2 + D) & 3) << 4) | (((3 + D) & 3) << 6))>(v);
}
// lane (l xor p) of the quad, p in {0, 1, 2, 3} at run time (quad-uniform): the LDS crossbar (ds_bpermute), no LDS memory
__device__ __forceinline__ double qd_partner(double v, int p) {
  const int src = ((int)(threadIdx.x & 63) ^ p) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// lane (l + d) mod 4 of the quad, d in {1, 2, 3} at run time (quad-uniform): the LDS crossbar again
__device__ __forceinline__ double qd_rotv(double v, int d) {
  const int lane = (int)(threadIdx.x & 63);
  const int src = ((lane & ~3) | ((lane + d) & 3)) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// true in every lane of the wavefront if pred holds in any (the four-lane quads of a wavefront share its instruction stream)
__device__ __forceinline__ bool qw_any(bool pred) { return __ballot(pred) != 0; }
// the largest value of v (0..8) over the ACTIVE lanes of the wavefront, as a scalar (ballots: lanes that left a loop earlier do not take
// part, and a butterfly of shuffles would lose values behind them)
__device__ __forceinline__ int qw_max(int v) {
  int r = 0;
  for (int k = 1; k <= 8; k++) r += __ballot(v >= k) != 0 ? 1 : 0;
  return r;
}
__device__ __forceinline__ int qd_or(int v) {
  v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  return v;
}

// ---- the lane's contact store: the first kQLdsSlots records in LDS ([slot][field][lane]: a wavefront's access to one field is one
// conflict-free ds_read_b64 / ds_write_b64), further ones in a private array (scratch: touched only by a lane with that many contacts -- 40 % of
// the wavefront-steps of the bench's gait). QEXP_OVF_SLAB moves that array to a per-wavefront slab in global memory (QArgs::ovf_slab, the LDS
// store's layout): measured 1.6 ms of 63 SLOWER on the gait (same box, back to back: 64.2 against 62.6 ms -- 64-bit address arithmetic per
// access where scratch has an immediate offset), so the private array stays although it is 2.3 KB of the 5.4 KB private segment.
// ---- the store of M while the solver runs: leg block + coupling per lane ([entry][lane]), the trunk block once per QUAD ([entry][quad]:
// its four lanes write the same value and read it back as an LDS broadcast).
namespace mjpcx { namespace quad {
struct QContact;
#ifndef QEXP_LDS_SLOTS
#define QEXP_LDS_SLOTS 3   // (a fourth slot would need 28.7 KB more of the CU's LDS; 0.8 KB are free)
#endif
constexpr int kQLdsSlots = QEXP_LDS_SLOTS;
typedef __attribute__((address_space(3))) double qlds_f64;  // a typed LDS pointer: ds_read / ds_write instead of FLAT accesses
typedef __attribute__((address_space(5))) double qprv_f64;  // a typed private pointer: scratch_load / scratch_store
#ifndef QEXP_OVF_SLAB
struct LdsStore { qlds_f64* lds; qprv_f64* ovf; };
#define QOVF_STRIDE 1
#else
struct LdsStore { qlds_f64* lds; double* ovf; };
#define QOVF_STRIDE 64
#endif
struct LdsM { qlds_f64* ml; qlds_f64* mt; };
struct QProf { long long* buf; long long last; };
} }
__device__ __forceinline__ void qcs_load(const mjpcx::quad::LdsStore& cs, int slot, mjpcx::quad::QContact& c);
__device__ __forceinline__ void qcs_store(mjpcx::quad::LdsStore& cs, int slot, const mjpcx::quad::QContact& c);
__device__ __forceinline__ void qcs_store_jar(mjpcx::quad::LdsStore& cs, int slot, const mjpcx::quad::QContact& c);
__device__ __forceinline__ double qms_l(const mjpcx::quad::LdsM& m, int i) { return m.ml[i * 64]; }
__device__ __forceinline__ double qms_b(const mjpcx::quad::LdsM& m, int j, int k) { return m.ml[(6 + 6 * j + k) * 64]; }
__device__ __forceinline__ double qms_t(const mjpcx::quad::LdsM& m, int i) { return m.mt[i * 16]; }
__device__ __forceinline__ void qms_set_l(mjpcx::quad::LdsM& m, int i, double v) { m.ml[i * 64] = v; }
__device__ __forceinline__ void qms_set_b(mjpcx::quad::LdsM& m, int j, int k, double v) { m.ml[(6 + 6 * j + k) * 64] = v; }
__device__ __forceinline__ void qms_set_t(mjpcx::quad::LdsM& m, int i, double v) { m.mt[i * 16] = v; }
// phase cycle stamps of wavefront 0 (pf.buf != nullptr only there): s_memtime deltas accumulated per phase
#define QPROF(pf, idx) do { if ((pf).buf) { const long long now_ = __builtin_readcyclecounter(); (pf).buf[idx] += now_ - (pf).last; (pf).last = now_; } } while (0)
#define QPROF_COUNT(pf, idx, n) do { if ((pf).buf) (pf).buf[idx] += (n); } while (0)
// buf[idx + k] += 1 if any lane of the wavefront has v > bound_k (bounds 0 1 2 3 4 6 8 12), buf[idx + 8] += the sum of v over the lanes, buf[idx + 9] += the largest
#define QPROF_WAVE_HIST(pf, idx, v) do { if ((pf).buf) { const int b_[8] = {0, 1, 2, 3, 4, 6, 8, 12}; \
    for (int k_ = 0; k_ < 8; k_++) if (__ballot((v) > b_[k_])) (pf).buf[(idx) + k_] += 1; \
    int s_ = (v), m_ = (v); for (int o_ = 32; o_ > 0; o_ >>= 1) { s_ += __shfl_xor(s_, o_); const int t_ = __shfl_xor(m_, o_); m_ = t_ > m_ ? t_ : m_; } \
    (pf).buf[(idx) + 8] += s_; (pf).buf[(idx) + 9] += m_; } } while (0)

// per-wavefront totals (a.wave_times): [1] += the slowest candidate's Newton iterations, [2] += 1 if the general solver ran, [3] += the
// largest per-lane contact count; [0] (cycles) is written by the kernel at the end
#define QWAVE_TIMES(a, iters, general, ncon) do { if ((a).wave_times) { int mi_ = (iters), mc_ = (ncon); \
    for (int o_ = 32; o_ > 0; o_ >>= 1) { const int t_ = __shfl_xor(mi_, o_); mi_ = t_ > mi_ ? t_ : mi_; const int u_ = __shfl_xor(mc_, o_); mc_ = u_ > mc_ ? u_ : mc_; } \
    long long* w_ = (a).wave_times + 4 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6); \
    if ((threadIdx.x & 63) == 0) { w_[1] += mi_; w_[2] += (general) ? 1 : 0; w_[3] += mc_; } } } while (0)

// per-class cycle totals of the wavefront (a.wave_class, quad_abi.h): written by the wavefront's first lane
#ifdef QEXP_CLASS_GENERAL   // (tuning: the histogram's "ovf" bit counts the steps through the multi-pattern solver instead)
#define QCLASS_BIT2(pmask, ncon) ((((pmask) >> 1) & 1) + (((pmask) >> 2) & 1) + (((pmask) >> 3) & 1) >= 2)
#else
#define QCLASS_BIT2(pmask, ncon) ((ncon) > mjpcx::quad::kQLdsSlots)
#endif
#define QCLASS_NOW(a) ((a).wave_class ? (long long)__builtin_readcyclecounter() : 0ll)
#define QCLASS_ADD(a, base, have_rel, pmask, ncon, t0) do { if ((a).wave_class) { \
    const int c_ = (__ballot((have_rel) != 0) ? 1 : 0) | (__ballot((pmask) != 0) ? 2 : 0) | (__ballot(QCLASS_BIT2(pmask, ncon)) ? 4 : 0) | (__ballot((ncon) > mjpcx::quad::kQLineSlots) ? 8 : 0); \
    if ((threadIdx.x & 63) == 0) { long long* w_ = (a).wave_class + 64 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) + (base) + 2 * c_; \
      w_[0] += 1; w_[1] += (long long)__builtin_readcyclecounter() - (t0); } } } while (0)

// every fixed-trip loop over a small array is unrolled: a loop the compiler keeps rolled indexes its array at run time, and a private array
// indexed at run time lives in scratch
#define QUNROLL _Pragma("unroll")
#define QNOUNROLL _Pragma("nounroll")
#define QNOINLINE __device__ __noinline__
#define QD __device__ __forceinline__
#define QFAST_MATH 1
// what an out-of-line function of the step gets by reference: the model image lives in LDS, the caller's locals in its private segment.
// Saying so turns the FLAT loads of a generic pointer (LDS and vector-memory path, both counters) into ds_read / scratch_load.
#define QREBIND_LDS(T, ref) (*(const T*)(const __attribute__((address_space(3))) T*)(&(ref)))
#define QREBIND_PRIVATE(T, ptr) ((__attribute__((address_space(5))) T*)(ptr))
#ifndef QEXP_NO_NT
// trajectory buffers are written once and read by the host or a later kernel: stream them past the caches the spill traffic lives in
#define QREC(dst, v) __builtin_nontemporal_store((v), &(dst))
#endif
#ifndef QEXP_NO_UNIFORM_TIME
// every lane of a launch is at the same time: the node search of the policy becomes scalar loads and a scalar loop
__device__ __forceinline__ double q_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
#define QUNIFORM_TIME(t) q_uniform(t)
#endif
// x > 0, normal: Newton refinement of the hardware estimates (v_rsq_f64 / v_rcp_f64 are good to about 2^-26)
__device__ __forceinline__ void q_sqrt_rsqrt(double x, double& s, double& r) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = fma(y, fma(-hx * y, y, 0.5), y);   // y (1.5 - 0.5 x y^2)
  y = fma(y, fma(-hx * y, y, 0.5), y);
  double g = x * y;
  g = fma(fma(-g, g, x), 0.5 * y, g);    // one correction of the root itself
  s = g; r = y;
}
__device__ __forceinline__ double q_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}
#include "quad_step.h"
#undef QD

// record layout: n 0-2, off 3-5, D0 6, jar 7-12, (depth, friction set, self-collision fields) 13
__device__ __forceinline__ void qcs_load(const mjpcx::quad::LdsStore& cs, int slot, mjpcx::quad::QContact& c) {
  using namespace mjpcx::quad;
  double v[kQConRec];
  if (slot < kQLdsSlots) { const qlds_f64* p = cs.lds + slot * kQConRec * 64; QUNROLL for (int f = 0; f < kQConRec; f++) v[f] = p[f * 64]; }
  else { const auto* p = cs.ovf + (size_t)(slot - kQLdsSlots) * kQConRec * QOVF_STRIDE; QUNROLL for (int f = 0; f < kQConRec; f++) v[f] = p[f * QOVF_STRIDE]; }
  QUNROLL for (int k = 0; k < 3; k++) { c.n[k] = v[k]; c.off[k] = v[3 + k]; }
  c.D0 = v[6];
  QUNROLL for (int k = 0; k < 6; k++) c.jar[k] = v[7 + k];
  const int meta = (int)v[13];
  c.depth = meta & 3; c.fid = (meta >> 2) & 7; c.rel = (meta >> 5) & 1; c.sgn = (meta & 64) ? 1 : -1; c.pd = (meta >> 7) & 3; c.px = (meta >> 9) & 3; c.self = (meta >> 11) & 1;
}
__device__ __forceinline__ void qcs_store(mjpcx::quad::LdsStore& cs, int slot, const mjpcx::quad::QContact& c) {
  using namespace mjpcx::quad;
  double v[kQConRec];
  QUNROLL for (int k = 0; k < 3; k++) { v[k] = c.n[k]; v[3 + k] = c.off[k]; }
  v[6] = c.D0;
  QUNROLL for (int k = 0; k < 6; k++) v[7 + k] = c.jar[k];
  v[13] = (double)(c.depth | (c.fid << 2) | (c.rel << 5) | (c.sgn > 0 ? 64 : 0) | (c.pd << 7) | (c.px << 9) | (c.self << 11));
  if (slot < kQLdsSlots) { qlds_f64* p = cs.lds + slot * kQConRec * 64; QUNROLL for (int f = 0; f < kQConRec; f++) p[f * 64] = v[f]; }
  else { auto* p = cs.ovf + (size_t)(slot - kQLdsSlots) * kQConRec * QOVF_STRIDE; QUNROLL for (int f = 0; f < kQConRec; f++) p[f * QOVF_STRIDE] = v[f]; }
}
__device__ __forceinline__ void qcs_store_jar(mjpcx::quad::LdsStore& cs, int slot, const mjpcx::quad::QContact& c) {
  using namespace mjpcx::quad;
  if (slot < kQLdsSlots) { qlds_f64* p = cs.lds + slot * kQConRec * 64; QUNROLL for (int k = 0; k < 6; k++) p[(7 + k) * 64] = c.jar[k]; }
  else { auto* p = cs.ovf + (size_t)(slot - kQLdsSlots) * kQConRec * QOVF_STRIDE; QUNROLL for (int k = 0; k < 6; k++) p[(7 + k) * QOVF_STRIDE] = c.jar[k]; }
}

namespace mjpcx { namespace quad {

constexpr size_t kQWaveCon = (size_t)kQLdsSlots * kQConRec * 64, kQWaveMl = 24 * 64, kQWaveMt = 21 * 16;  // doubles per wavefront: contacts, M
constexpr size_t kQWaveLds = (kQWaveCon + kQWaveMl + kQWaveMt) * sizeof(double);

// Workgroup = W wavefronts (W = 4 for the large batches: one per SIMD of a CU, sharing one model image; W = 1 spreads small batches
// over the CUs). stats[0]: candidates handed to the fallback kernel, stats[1 + log2(flag)]: by reason.
template <bool FEEDBACK>
__device__ __forceinline__ void quad_kernel_body(const QuadModel* __restrict__ gm, const QuadTables* __restrict__ tab, const double* __restrict__ blob,
                                                 const QBlob& bo, const QArgs& a, const QFeedback& fb, int* __restrict__ stats) {
  __shared__ QuadModel sm;
  __shared__ QStaticPose sp[kQStatic];
  extern __shared__ __attribute__((aligned(16))) double con_lds[];
  {
    static_assert(sizeof(QuadModel) % 8 == 0, "QuadModel is staged in 8-byte words");
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(gm);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&sm);
    for (unsigned i = threadIdx.x; i < sizeof(QuadModel) / 8; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  if ((int)threadIdx.x < sm.nstatic) static_pose(sm, blob + bo.off_mocap, threadIdx.x, sp[threadIdx.x]);
  __syncthreads();
  // candidates per wavefront: 16 fills the chip at N = 16384; a smaller batch is dealt over more wavefronts (a.cpw of them per wavefront, in
  // its first lanes), so that its lock-step is over fewer candidates and it still uses every SIMD
  const int cpw = a.cpw > 0 ? a.cpw : 16;
  const int quad = (threadIdx.x & 63) >> 2, leg = threadIdx.x & 3;
  const int cand = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * cpw + quad;
  if (quad >= cpw || cand >= a.N) return;  // (whole quads leave together)
  QTask tk;
  tk.mocap = blob + bo.off_mocap; tk.weight = blob + bo.off_weight; tk.norm_p = blob + bo.off_normp; tk.norm_q = blob + bo.off_normq;
  tk.param = blob + bo.off_param; tk.re = blob + bo.off_rreal; tk.ri = reinterpret_cast<const int*>(blob + bo.off_rint); tk.risk = blob[bo.off_risk];
  qlds_f64* wave_lds = (qlds_f64*)con_lds + (threadIdx.x >> 6) * (kQWaveLds / sizeof(double));
#ifndef QEXP_OVF_SLAB
  double ovf_prv[(kQMaxCon - kQLdsSlots) * kQConRec];
  LdsStore cs{wave_lds + (threadIdx.x & 63), (qprv_f64*)ovf_prv};
#else
  double* ovf = a.ovf_slab + (size_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * ((kQMaxCon - kQLdsSlots) * kQConRec * 64) + (threadIdx.x & 63);
  LdsStore cs{wave_lds + (threadIdx.x & 63), ovf};
#endif
  LdsM ms{wave_lds + kQWaveCon + (threadIdx.x & 63), wave_lds + kQWaveCon + kQWaveMl + ((threadIdx.x & 63) >> 2)};
  QProf pf{nullptr, 0};
  if (a.stamps && blockIdx.x == 0 && threadIdx.x < 64) { pf.buf = a.stamps; pf.last = __builtin_readcyclecounter(); }
  const long long wave_t0 = a.wave_times ? __builtin_readcyclecounter() : 0;
  const int flags = rollout<FEEDBACK>(sm, *tab, sp, tk, blob, blob[bo.off_time], a, fb, cand, leg, cs, ms, pf);
  if (a.wave_times && (threadIdx.x & 63) == 0) a.wave_times[4 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = __builtin_readcyclecounter() - wave_t0;
  if (flags && leg == 0 && stats) {
    atomicAdd(stats, 1);
    QUNROLL for (int b = 0; b < 7; b++) if (flags & (1 << b)) atomicAdd(stats + 1 + b, 1);
  }
}
__global__ __launch_bounds__(256) void rollout_quad_kernel(const QuadModel* __restrict__ gm, const QuadTables* __restrict__ tab, const double* __restrict__ blob,
                                                           const QBlob bo, const QArgs a, int* __restrict__ stats) {
  quad_kernel_body<false>(gm, tab, blob, bo, a, QFeedback{}, stats);
}
// The iLQG rollouts (the nominal under iLQGPolicy::Action, the line search's under the index policy) on the same step function: one or ten
// candidates, ONE per wavefront (QArgs::cpw = 1) -- pure per-step latency, and the quad form's step is the shortest of the kernels' (0.145 ms
// against 0.2 of the wavefront-per-candidate form: tools/latency_probe.py). A candidate it does not cover is flagged as in rollout_quad_kernel
// and rolled out by rollout_feedback_tree_kernel.
__global__ __launch_bounds__(256) void rollout_feedback_quad_kernel(const QuadModel* __restrict__ gm, const QuadTables* __restrict__ tab, const double* __restrict__ blob,
                                                                    const QBlob bo, const QArgs a, const QFeedback fb, int* __restrict__ stats) {
  quad_kernel_body<true>(gm, tab, blob, bo, a, fb, stats);
}

} }  // namespace mjpcx::quad
