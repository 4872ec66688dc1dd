// limb_kernel.h -- the gfx950 launch shape of the LIMB kernel family (limb_step.h): FOUR LANES PER CANDIDATE, one per limb of the Humanoid
// (BASELINE configs[3]), up to sixteen candidates per wavefront. What the wavefront-per-candidate kernel keeps in a 12.7 KB LDS arena per
// candidate is the four lanes' registers here (fp32: the whole 512-entry file of a SIMD holds a lane's arrowhead Hessian, its chain and
// the solver's vectors); a reduction over the limbs is two DPP quad permutes.
//
//   qd_sum     v + quad_perm[1,0,3,2](v), then + quad_perm[2,3,0,1]: the four lanes hold bit-identical sums
//   qd_bcasti  lane K's integer;  qd_or  bitwise or over the quad;  qw_any  a ballot over the wavefront
//   ld_sync    a wavefront-level LDS fence (the quad's shared block is written and read by lanes of ONE wavefront: no s_barrier)
//
// LDS per wavefront: the lanes' floor contacts ([slot][field][lane]), M ([entry][lane]; the trunk block once per quad), and one SHARED
// block per candidate (world poses of the moving geoms, the contacts between them). The model image is staged once per workgroup.
#pragma once
#include <hip/hip_runtime.h>

#include "limb_model.h"

namespace mjpcx { namespace limb {
template <int CTRL> __device__ __forceinline__ float ldpp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> __device__ __forceinline__ double ldpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
} }
__device__ __forceinline__ float qd_sum(float v) { v += mjpcx::limb::ldpp<0xB1>(v); v += mjpcx::limb::ldpp<0x4E>(v); return v; }
__device__ __forceinline__ double qd_sum(double v) { v += mjpcx::limb::ldpp<0xB1>(v); v += mjpcx::limb::ldpp<0x4E>(v); return v; }
template <int K> __device__ __forceinline__ int qd_bcasti(int v) { return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xf, 0xf, true); }
__device__ __forceinline__ int qd_or(int v) {
  v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  return v;
}
__device__ __forceinline__ bool qw_any(bool pred) { return __ballot(pred) != 0; }
__device__ __forceinline__ void ld_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

namespace mjpcx { namespace limb {
template <typename R> struct LContact;
template <typename R> struct LCross;
template <typename R> struct LKin;
// LS: lanes of a wavefront that work (4 x candidates per wavefront: 64, or 32 for batches that fill the chip at eight candidates per
// wavefront) = the stride of the per-lane arrays -- a half-filled wavefront takes half the LDS, so four of them fit a CU
// (typed LDS pointers: ds_read / ds_write instead of FLAT accesses -- a third of the solver's memory instructions were FLAT before)
#ifdef LEXP_NO_LDSPTR
template <typename R> using lds_ptr = R*;
#else
template <typename R> using lds_ptr = __attribute__((address_space(3))) R*;
#endif
template <typename R, int LS> struct LdsCS { lds_ptr<R> p; };   // + lane
template <typename R, int LS> struct LdsMS { lds_ptr<R> ml; lds_ptr<R> mt; };  // + lane / + quad
template <typename R> struct LdsSH { lds_ptr<R> p; };   // the candidate's shared block
// the dof axes about the centre of mass (limb_step.h LKin), in LDS: the limb's six per lane ([axis][component][lane]), the trunk's nine once
// per quad. The solver reads them element by element through these proxies (kin.cdof[j][c]): until round 6 they travelled from the
// forward stage to the solver through the private segment -- 90 stores and ~1500 FLAT loads per wavefront-step
template <typename R, int ST> struct LdsVec { lds_ptr<R> p; __device__ __forceinline__ R operator[](int c) const { return p[c * ST]; } };
template <typename R, int ST> struct LdsRows { lds_ptr<R> base; __device__ __forceinline__ LdsVec<R, ST> operator[](int j) const { return LdsVec<R, ST>{base + 6 * j * ST}; } };
template <typename R, int LS> struct LdsKin { LdsRows<R, LS> cdof; LdsRows<R, LS / 4> cdofT; };
constexpr int kShGeom = kNG * 6, kShCross = kMaxX * 12;
constexpr int kShStride = ((kShGeom + kShCross) | 1) + 2;  // (odd: the sixteen candidates' blocks start in different banks)
} }
template <typename R, int LS> __device__ __forceinline__ void lcs_load(const mjpcx::limb::LdsCS<R, LS>& cs, int i, mjpcx::limb::LContact<R>& c);
template <typename R, int LS> __device__ __forceinline__ void lcs_store(mjpcx::limb::LdsCS<R, LS>& cs, int i, const mjpcx::limb::LContact<R>& c);
template <typename R, int LS> __device__ __forceinline__ void lcs_store_jar(mjpcx::limb::LdsCS<R, LS>& cs, int i, const mjpcx::limb::LContact<R>& c);
template <typename R, int LS> __device__ __forceinline__ R lms_l(const mjpcx::limb::LdsMS<R, LS>& m, int i) { return m.ml[i * LS]; }
template <typename R, int LS> __device__ __forceinline__ R lms_b(const mjpcx::limb::LdsMS<R, LS>& m, int j, int k) { return m.ml[(21 + mjpcx::limb::kTD * j + k) * LS]; }
template <typename R, int LS> __device__ __forceinline__ R lms_t(const mjpcx::limb::LdsMS<R, LS>& m, int i) { return m.mt[i * (LS / 4)]; }
template <typename R, int LS> __device__ __forceinline__ void lms_set_l(mjpcx::limb::LdsMS<R, LS>& m, int i, R v) { m.ml[i * LS] = v; }
template <typename R, int LS> __device__ __forceinline__ void lms_set_b(mjpcx::limb::LdsMS<R, LS>& m, int j, int k, R v) { m.ml[(21 + mjpcx::limb::kTD * j + k) * LS] = v; }
template <typename R, int LS> __device__ __forceinline__ void lms_set_t(mjpcx::limb::LdsMS<R, LS>& m, int i, R v) { m.mt[i * (LS / 4)] = v; }
template <typename R, int LS> __device__ __forceinline__ void lkin_store(mjpcx::limb::LdsKin<R, LS>& ks, const mjpcx::limb::LKin<R>& k);
template <typename R> __device__ __forceinline__ void lsh_set_geom(mjpcx::limb::LdsSH<R>& sh, int g, const R* pos, const R* axis) {
#pragma unroll
  for (int k = 0; k < 3; k++) { sh.p[6 * g + k] = pos[k]; sh.p[6 * g + 3 + k] = axis[k]; }
}
template <typename R> __device__ __forceinline__ void lsh_get_geom(const mjpcx::limb::LdsSH<R>& sh, int g, R* pos, R* axis) {
#pragma unroll
  for (int k = 0; k < 3; k++) { pos[k] = sh.p[6 * g + k]; axis[k] = sh.p[6 * g + 3 + k]; }
}
template <typename R> __device__ __forceinline__ R lsh_xget(const mjpcx::limb::LdsSH<R>& sh, int r, int f) { return sh.p[mjpcx::limb::kShGeom + 12 * r + f]; }
template <typename R> __device__ __forceinline__ void lsh_xset(mjpcx::limb::LdsSH<R>& sh, int r, int f, R v) { sh.p[mjpcx::limb::kShGeom + 12 * r + f] = v; }
template <typename R> __device__ __forceinline__ void lsh_set_cross(mjpcx::limb::LdsSH<R>& sh, int r, const mjpcx::limb::LCross<R>& c);
template <typename R> __device__ __forceinline__ void lsh_get_cross(const mjpcx::limb::LdsSH<R>& sh, int r, mjpcx::limb::LCross<R>& c);

#define LUNROLL _Pragma("unroll")
// what an out-of-line function of the step gets by reference / pointer: the model image lives in LDS, the caller's locals in its private segment
#ifndef LEXP_NO_REBIND
#define LREBIND_LDS(T, ref) (*(const T*)(const __attribute__((address_space(3))) T*)(&(ref)))
#endif
#ifndef LEXP_NO_REBIND_PRV
#define LREBIND_PRV(T, ptr) ((T*)(__attribute__((address_space(5))) T*)(ptr))
#endif
#ifndef LEXP_NO_PRV
#define LPRV_LOAD(dst, src) __builtin_memcpy(&(dst), (const __attribute__((address_space(5))) decltype(dst)*)(src), sizeof(dst))
#define LPRV_STORE(dst, src) __builtin_memcpy((__attribute__((address_space(5))) decltype(src)*)(dst), &(src), sizeof(src))
#define LPRV_LOADN(dst, src, n) __builtin_memcpy((dst), (const __attribute__((address_space(5))) decltype((dst)[0] + 0)*)(src), (n) * sizeof((dst)[0]))
#define LPRV_STOREN(dst, src, n) __builtin_memcpy((__attribute__((address_space(5))) decltype((src)[0] + 0)*)(dst), (src), (n) * sizeof((src)[0]))
#endif
#define LD __device__ __forceinline__
#ifdef LEXP_ALL_INLINE
#define LNOINLINE __device__ __forceinline__
#else
#define LNOINLINE __device__ __noinline__
#endif
#define LREC(dst, v) __builtin_nontemporal_store((v), &(dst))
#define LUNIFORM(i) __builtin_amdgcn_readfirstlane(i)
// phase cycle stamps of wavefront 0 (a.stamps != nullptr: MJPCX_LIMB_STAMPS=1): [idx] += cycles since the previous stamp; idx -1 starts the clock
#define LPROF(a, last, idx) do { if ((a).stamps && blockIdx.x == 0 && threadIdx.x < 64) { const long long now_ = __builtin_readcyclecounter(); \
    if ((idx) >= 0 && (threadIdx.x & 63) == 0) (a).stamps[(idx) < 0 ? 0 : (idx)] += now_ - (last); (last) = now_; } } while (0)
#define LPROF_COUNT(a, idx) do { if ((a).stamps && blockIdx.x == 0 && threadIdx.x == 0) (a).stamps[idx] += 1; } while (0)
#include "limb_step.h"
#undef LD

template <typename R, int LS> __device__ __forceinline__ void lcs_load(const mjpcx::limb::LdsCS<R, LS>& cs, int i, mjpcx::limb::LContact<R>& c) {
  using namespace mjpcx::limb;
  const lds_ptr<R> p = cs.p + i * kLConRec * LS;
  LUNROLL for (int k = 0; k < 3; k++) c.off[k] = p[k * LS];
  c.D = p[3 * LS]; c.mu = p[4 * LS];
  LUNROLL for (int k = 0; k < 4; k++) c.jar[k] = p[(5 + k) * LS];
  const int meta = (int)p[9 * LS];
  c.body = meta & 3; c.nrow = meta >> 2;
}
template <typename R, int LS> __device__ __forceinline__ void lcs_store(mjpcx::limb::LdsCS<R, LS>& cs, int i, const mjpcx::limb::LContact<R>& c) {
  using namespace mjpcx::limb;
  const lds_ptr<R> p = cs.p + i * kLConRec * LS;
  LUNROLL for (int k = 0; k < 3; k++) p[k * LS] = c.off[k];
  p[3 * LS] = c.D; p[4 * LS] = c.mu;
  LUNROLL for (int k = 0; k < 4; k++) p[(5 + k) * LS] = c.jar[k];
  p[9 * LS] = (R)(c.body | (c.nrow << 2));
}
template <typename R, int LS> __device__ __forceinline__ void lcs_store_jar(mjpcx::limb::LdsCS<R, LS>& cs, int i, const mjpcx::limb::LContact<R>& c) {
  using namespace mjpcx::limb;
  const lds_ptr<R> p = cs.p + i * kLConRec * LS;
  LUNROLL for (int k = 0; k < 4; k++) p[(5 + k) * LS] = c.jar[k];
}
template <typename R> __device__ __forceinline__ void lsh_set_cross(mjpcx::limb::LdsSH<R>& sh, int r, const mjpcx::limb::LCross<R>& c) {
  using namespace mjpcx::limb;
  const lds_ptr<R> p = sh.p + kShGeom + 12 * r;
  LUNROLL for (int k = 0; k < 6; k++) p[k] = c.et[k];
  p[6] = c.D; p[7] = c.b; p[8] = c.kimpx;
  p[9] = (R)(c.la | (c.sa << 3) | (c.lb << 5) | (c.sb << 8));
}
template <typename R> __device__ __forceinline__ void lsh_get_cross(const mjpcx::limb::LdsSH<R>& sh, int r, mjpcx::limb::LCross<R>& c) {
  using namespace mjpcx::limb;
  const lds_ptr<R> p = sh.p + kShGeom + 12 * r;
  LUNROLL for (int k = 0; k < 6; k++) c.et[k] = p[k];
  c.D = p[6]; c.b = p[7]; c.kimpx = p[8];
  const int meta = (int)p[9];
  c.la = meta & 7; c.sa = (meta >> 3) & 3; c.lb = (meta >> 5) & 7; c.sb = (meta >> 8) & 3;
}

template <typename R, int LS> __device__ __forceinline__ void lkin_store(mjpcx::limb::LdsKin<R, LS>& ks, const mjpcx::limb::LKin<R>& k) {
  using namespace mjpcx::limb;
  LUNROLL for (int j = 0; j < kLD; j++) LUNROLL for (int c = 0; c < 6; c++) ks.cdof.base[(6 * j + c) * LS] = k.cdof[j][c];
  LUNROLL for (int j = 0; j < kTD; j++) LUNROLL for (int c = 0; c < 6; c++) ks.cdofT.base[(6 * j + c) * (LS / 4)] = k.cdofT[j][c];  // (the quad's four lanes write the same values)
}

namespace mjpcx { namespace limb {
// reals of LDS per wavefront of LS working lanes: contacts, M (limb part per lane, trunk block per quad), the candidates' shared blocks
constexpr size_t wave_con(int LS) { return (size_t)kMaxPC * kLConRec * LS; }
constexpr size_t wave_ml(int LS) { return (size_t)(21 + kLD * kTD) * LS; }
constexpr size_t wave_mt(int LS) { return (size_t)45 * (LS / 4); }
constexpr size_t wave_sh(int LS) { return (size_t)kShStride * (LS / 4); }
constexpr size_t wave_kin(int LS) { return (size_t)6 * kLD * LS + (size_t)6 * kTD * (LS / 4); }
constexpr size_t wave_reals(int LS) { return wave_con(LS) + wave_ml(LS) + wave_mt(LS) + wave_sh(LS) + wave_kin(LS); }

// Workgroup = W wavefronts sharing one model image. stats[0]: candidates handed to the fallback kernel, stats[1 + b]: by reason bit b.
template <typename R, int LS>
__global__ __launch_bounds__(256) void rollout_limb_kernel(const LimbModelT<R>* __restrict__ gm, const R* __restrict__ blob, const LBlob bo, const LArgs<R> a,
                                                           const R* __restrict__ key_mpos, int* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  LimbModelT<R>& sm = *reinterpret_cast<LimbModelT<R>*>(lds_raw);
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(gm);
    unsigned* dst = reinterpret_cast<unsigned*>(&sm);
    for (unsigned i = threadIdx.x; i < sizeof(LimbModelT<R>) / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  // the cost terms' weights and norm parameters (per-plan data in the blob): staged behind the model image -- every residual entry reads three of
  // them, and from global memory that was 2 k cycles an entry (97 k of a step's 770 k)
  constexpr size_t image_only = (sizeof(LimbModelT<R>) + 15) & ~(size_t)15;
  R* terms = reinterpret_cast<R*>(lds_raw + image_only);
  for (unsigned i = threadIdx.x; i < 3u * kMaxTerm; i += blockDim.x) {
    const unsigned t = i % kMaxTerm, which = i / kMaxTerm;
    terms[i] = (int)t < sm.nterm ? blob[(which == 0 ? bo.off_weight : (which == 1 ? bo.off_normp : bo.off_normq)) + t] : R(0);
  }
  __syncthreads();
  const int cpw = a.cpw > 0 && a.cpw <= LS / 4 ? a.cpw : LS / 4;
  const int wl = threadIdx.x & 63, quad = wl >> 2, lane = wl & 3;
  const int cand = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * cpw + quad;
  if (quad >= cpw || cand >= a.N) return;  // (whole quads leave together)
  LTask<R> tk;
  tk.mocap = blob + bo.off_mocap; tk.weight = terms; tk.norm_p = terms + kMaxTerm; tk.norm_q = terms + 2 * kMaxTerm;
  tk.re = blob + bo.off_rreal; tk.ri = reinterpret_cast<const int*>(blob + bo.off_rint); tk.risk = blob[bo.off_risk]; tk.key_mpos = key_mpos;
  constexpr size_t image = image_only + ((3 * kMaxTerm * sizeof(R) + 15) & ~(size_t)15);
  const lds_ptr<R> wave_lds = (lds_ptr<R>)reinterpret_cast<R*>(lds_raw + image) + (threadIdx.x >> 6) * wave_reals(LS);
  LdsCS<R, LS> cs{wave_lds + wl};
  LdsMS<R, LS> ms{wave_lds + wave_con(LS) + wl, wave_lds + wave_con(LS) + wave_ml(LS) + quad};
  LdsSH<R> sh{wave_lds + wave_con(LS) + wave_ml(LS) + wave_mt(LS) + (size_t)quad * kShStride};
  const lds_ptr<R> kin_lds = wave_lds + wave_con(LS) + wave_ml(LS) + wave_mt(LS) + wave_sh(LS);
  LdsKin<R, LS> ks{LdsRows<R, LS>{kin_lds + wl}, LdsRows<R, LS / 4>{kin_lds + (size_t)6 * kLD * LS + quad}};
  const int flags = rollout(sm, tk, blob, blob[bo.off_time], a, cand, lane, cs, ms, sh, ks);
  if (flags && lane == 0 && stats) {
    atomicAdd(stats, 1);
    LUNROLL for (int b = 0; b < 6; b++) if (flags & (1 << b)) atomicAdd(stats + 1 + b, 1);
  }
}
} }  // namespace mjpcx::limb
