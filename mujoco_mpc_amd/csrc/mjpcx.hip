// mjpcx.hip -- implementation of the C ABI in include/mjpcx.h for gfx950.
//
// Host side of the drop-in boundary: validates the flat model, selects the rollout kernel
// instantiation whose static topology matches it, owns all device buffers, and exposes
// results in the reference's Trajectory layout. No CPU fallback exists: if no kernel
// matches the model, mjpcx_create fails with MJPCX_EUNSUPPORTED.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mjpcx.h"
#include "rollout_lane.h"
#include "lane_registry.h"
#include "ilqg_dense.h"
#include "rollout_wave.h"
#include "quad_launch.h"
#include "limb_launch.h"
#include "wave32_launch.h"
#include <type_traits>
#include <dlfcn.h>
#include <mutex>
#include <chrono>
#include <memory>
#include <thread>
#include <condition_variable>
#include <rccl/rccl.h>

using namespace mjpcx;

// ===================================================================== kernel registry
namespace {

struct TopoKey {
  int nb, nv, nu, nsite, nmocap;
  uint64_t parent, mocap, jtype, jbody, jlimited, actj, siteb;
  bool operator==(const TopoKey& o) const {
    return nb == o.nb && nv == o.nv && nu == o.nu && nsite == o.nsite && nmocap == o.nmocap && parent == o.parent &&
           mocap == o.mocap && jtype == o.jtype && jbody == o.jbody && jlimited == o.jlimited && actj == o.actj &&
           siteb == o.siteb;
  }
};
struct TaskKey {
  int rid, nr, nterm, ntrace;
  uint64_t termdim, tracesite;
  bool operator==(const TaskKey& o) const {
    return rid == o.rid && nr == o.nr && nterm == o.nterm && ntrace == o.ntrace && termdim == o.termdim &&
           tracesite == o.tracesite;
  }
};
template <class TP> constexpr TopoKey topo_key() {
  return {TP::NB, TP::NV, TP::NU, TP::NSITE, TP::NMOCAP, TP::kParent, TP::kMocap, TP::kJtype,
          TP::kJbody, TP::kJlimited, TP::kActj, TP::kSiteb};
}
template <class TK> constexpr TaskKey task_key() {
  return {TK::RID, TK::NR, TK::NTERM, TK::NTRACE, TK::kTermDim, TK::kTraceSite};
}

template <class TP, class TK, typename T>
hipError_t launch_lane(const LaneModel<T>& m, const LaneTask<T>& tk, const RolloutArgs<T>& a, hipStream_t s) {
  return launch_lane_impl<TP, TK, T, RuntimeModel>(m, tk, a, s);
}

struct KernelEntry {
  const char* name;
  TopoKey topo;
  TaskKey task;
  const LaneModel<double>* static_model;  // non-null: instantiation specialised for exactly these constants
  hipError_t (*launch64)(const LaneModel<double>&, const LaneTask<double>&, const RolloutArgs<double>&, hipStream_t);
  hipError_t (*launch32)(const LaneModel<float>&, const LaneTask<float>&, const RolloutArgs<float>&, hipStream_t);
  hipError_t (*feedback64)(const LaneModel<double>&, const LaneTask<double>&, const RolloutArgs<double>&, const FeedbackArgs<double>&, hipStream_t);
  hipError_t (*feedback32)(const LaneModel<float>&, const LaneTask<float>&, const RolloutArgs<float>&, const FeedbackArgs<float>&, hipStream_t);
  hipError_t (*fd64)(const LaneModel<double>&, const LaneTask<double>&, const FdArgs<double>&, hipStream_t);
  hipError_t (*fd32)(const LaneModel<float>&, const LaneTask<float>&, const FdArgs<float>&, hipStream_t);
};
template <class TP, class TK, typename T>
hipError_t launch_feedback(const LaneModel<T>& m, const LaneTask<T>& tk, const RolloutArgs<T>& a, const FeedbackArgs<T>& fb, hipStream_t s) {
  return launch_feedback_impl<TP, TK, T, RuntimeModel>(m, tk, a, fb, s);
}
template <class TP, class TK, typename T>
hipError_t launch_fd(const LaneModel<T>& m, const LaneTask<T>& tk, const FdArgs<T>& f, hipStream_t s) {
  return launch_fd_impl<TP, TK, T, RuntimeModel>(m, tk, f, s);
}
#define MJPCX_LANE_ENTRY(TP, TK) \
  { "rollout_lane<" #TP "," #TK ">", topo_key<TP>(), task_key<TK>(), nullptr, &launch_lane<TP, TK, double>, &launch_lane<TP, TK, float>, \
    &launch_feedback<TP, TK, double>, &launch_feedback<TP, TK, float>, &launch_fd<TP, TK, double>, &launch_fd<TP, TK, float> }
#define MJPCX_STATIC_ENTRY(TP, TK, GEN, FN) \
  { "rollout_lane<" #TP "," #TK "," #GEN ">", topo_key<TP>(), task_key<TK>(), &kHost##GEN, &FN##_f64, &FN##_f32, \
    &FN##_fb_f64, &FN##_fb_f32, &FN##_fd_f64, &FN##_fd_f32 }
const LaneModel<double> kHostStaticCartpole = make_Cartpole<double>();
const LaneModel<double> kHostStaticParticle = make_Particle<double>();
// specialised entries first: mjpcx_create takes the first entry whose key (and constants) match
const KernelEntry kKernels[] = {
    MJPCX_STATIC_ENTRY(TopoCartpole, TaskCartpole, StaticCartpole, launch_static_cartpole),
    MJPCX_STATIC_ENTRY(TopoParticle, TaskParticle, StaticParticle, launch_static_particle),
    MJPCX_STATIC_ENTRY(TopoParticle, TaskParticleCopy, StaticParticle, launch_static_particle_copy),
    MJPCX_LANE_ENTRY(TopoCartpole, TaskCartpole),
    MJPCX_LANE_ENTRY(TopoParticle, TaskParticle),
    MJPCX_LANE_ENTRY(TopoParticle, TaskParticleCopy),
};

// the wavefront-per-candidate family is selected by model FEATURES (free/ball joints, friction loss, contacts),
// not by topology: one generic kernel that reads the model through WaveModel
const KernelEntry kTreeEntryA1 = {"rollout_tree_kernel<A1> (registered model: hot arrays staged in LDS behind compile-time offsets, persistent wavefronts, "
                                   "Jacobian-free Newton contact solver)", TopoKey{}, TaskKey{}, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
const KernelEntry kTreeEntryHumanoid = {"rollout_tree_kernel<Humanoid> (registered model: hot arrays staged in LDS behind compile-time offsets, persistent wavefronts, "
                                         "Jacobian-free Newton contact solver: pyramidal cones, tendon limits)", TopoKey{}, TaskKey{}, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
const KernelEntry kQuadEntryA1 = {"rollout_quad_kernel (four lanes per candidate, one per leg: 16 candidates per wavefront, arrowhead Newton contact solver; "
                                   "candidates it hands on run rollout_tree_kernel<A1>)", TopoKey{}, TaskKey{}, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
const KernelEntry kLimbEntryHumanoid = {"rollout_limb_kernel (four lanes per candidate, one per limb: up to 16 candidates per wavefront, arrowhead Newton contact solver with "
                                         "Woodbury terms for contacts between moving geoms; candidates it hands on run rollout_tree_kernel<Humanoid>)",
                                         TopoKey{}, TaskKey{}, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
const KernelEntry kWaveEntry = {"rollout_wave_kernel (wavefront per candidate, model in LDS/L1; Newton contact solver)",
                                TopoKey{}, TaskKey{}, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

// ===================================================================== small kernels
// argmin / top-k over total_return: replaces std::partial_sort (sampling/planner.cc:184-188).
// Keys are (return, index) with ties broken by index, so the result is deterministic.
struct RetIdx { double r; int i; };
__device__ __forceinline__ bool less_ri(const RetIdx& a, const RetIdx& b) {
  // NaN returns sort last (the reference's operator< would make the order unspecified)
  const bool an = a.r != a.r, bn = b.r != b.r;
  if (an != bn) return bn;
  if (a.r != b.r && !an) return a.r < b.r;
  return a.i < b.i;
}
__global__ __launch_bounds__(1024) void argmin_kernel(const double* __restrict__ ret, int n, RetIdx* out) {
  __shared__ RetIdx sm[16];
  RetIdx best{INFINITY, 0x7fffffff};
  best.r = ret[0] != ret[0] ? ret[0] : INFINITY;  // keep NaN ordering consistent when everything is NaN
  best.i = 0x7fffffff;
  bool have = false;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    RetIdx c{ret[i], i};
    if (!have || less_ri(c, best)) { best = c; have = true; }
  }
  if (!have) { best.r = NAN; best.i = 0x7fffffff; }
  // wave-level reduction over 64 lanes
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    RetIdx o;
    o.r = __shfl_down(best.r, off, 64);
    o.i = __shfl_down(best.i, off, 64);
    if (less_ri(o, best)) best = o;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sm[wave] = best;
  __syncthreads();
  if (wave == 0) {
    const int nw = blockDim.x >> 6;
    best = lane < nw ? sm[lane] : RetIdx{NAN, 0x7fffffff};
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      RetIdx o;
      o.r = __shfl_down(best.r, off, 64);
      o.i = __shfl_down(best.i, off, 64);
      if (less_ri(o, best)) best = o;
    }
    if (lane == 0) out[0] = best;
  }
}
// The policy update of Predictive Sampling needs, from one rollout batch: the winner's index
// and return, its spline values (candidate_policy[winner], sampling/planner.cc:534-543) and the
// nominal candidate's return (`improvement`, planner.cc:207-208). One launch writes them all
// into a pinned, device-mapped host record, so the host pays exactly one stream sync per plan step.
struct BestRecord { double best_return, ref_return; int best_index, ref_failure; double spline[1]; };
template <typename T>
__global__ __launch_bounds__(1024) void best_kernel(const double* __restrict__ ret, const int* __restrict__ fail,
                                                     const T* __restrict__ nodes, int n, int np, int ref, BestRecord* out) {
  __shared__ RetIdx sm[16];
  __shared__ int winner;
  RetIdx best{NAN, 0x7fffffff};
  bool have = false;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    RetIdx c{ret[i], i};
    if (!have || less_ri(c, best)) { best = c; have = true; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    RetIdx o;
    o.r = __shfl_down(best.r, off, 64);
    o.i = __shfl_down(best.i, off, 64);
    if (less_ri(o, best)) best = o;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sm[wave] = best;
  __syncthreads();
  if (wave == 0) {
    const int nw = blockDim.x >> 6;
    best = lane < nw ? sm[lane] : RetIdx{NAN, 0x7fffffff};
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      RetIdx o;
      o.r = __shfl_down(best.r, off, 64);
      o.i = __shfl_down(best.i, off, 64);
      if (less_ri(o, best)) best = o;
    }
    if (lane == 0) {
      winner = best.i;
      out->best_return = best.r;
      out->best_index = best.i;
      out->ref_return = (ref >= 0 && ref < n) ? ret[ref] : NAN;
      out->ref_failure = (ref >= 0 && ref < n) ? fail[ref] : 0;
    }
  }
  __syncthreads();
  const int w = winner;
  if (w >= 0 && w < n)
    for (int j = threadIdx.x; j < np; j += blockDim.x) out->spline[j] = (double)nodes[(size_t)j * n + w];
}

// elite moments: one workgroup per spline parameter j, fixed-shape tree reduction (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void elite_moments_kernel(const T* __restrict__ nodes, const double* __restrict__ ret,
                                                            const int* __restrict__ cand, int n, int N, int np,
                                                            const double* __restrict__ mean, double* out) {
  __shared__ double sm[256];
  const int j = blockIdx.x;  // j == np: the return sum
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = cand[i];
    if (j < np) {
      const double p = (double)nodes[(size_t)j * N + c];
      if (mean) { const double d = p - mean[j]; acc += d * d; } else acc += p;
    } else {
      acc += ret[c];
    }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[j] = sm[0];
}

// single-workgroup bitonic sort of (return, index) pairs in global memory (n2 = pow2 >= n)
__global__ __launch_bounds__(1024) void sort_kernel(const double* __restrict__ ret, int n, int n2, RetIdx* buf) {
  for (int i = threadIdx.x; i < n2; i += blockDim.x) buf[i] = i < n ? RetIdx{ret[i], i} : RetIdx{NAN, 0x7fffffff};
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          RetIdx a = buf[i], b = buf[l];
          if (less_ri(b, a) == up) { buf[i] = b; buf[l] = a; }
        }
      }
      __threadfence_block();
      __syncthreads();
    }
}

// gather one candidate out of the [t][field][candidate] SoA into a packed row-major staging
// buffer: [states H*DS | actions H*NU | times H | residual H*NR | costs H | trace H*3*NTR]
template <typename T>
__global__ void gather_traj_kernel(const T* states, const T* actions, const T* times, const T* residual,
                                   const T* costs, const T* trace, int N, int H, int ds, int nu, int nr, int ntr3,
                                   int cand, double* out, int candidate_major) {
  const int row = ds + nu + 1 + nr + 1 + ntr3;
  const int total = H * row;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    // i indexes the packed output
    int o = i;
    const T* src; int width;
    if (o < H * ds) { src = states; width = ds; }
    else if ((o -= H * ds) < H * nu) { src = actions; width = nu; }
    else if ((o -= H * nu) < H) { src = times; width = 1; }
    else if ((o -= H) < H * nr) { src = residual; width = nr; }
    else if ((o -= H * nr) < H) { src = costs; width = 1; }
    else { o -= H; src = trace; width = ntr3; }
    // o = t*width + k -> [(t*width+k)*N + cand] (lane-per-candidate kernels: a wavefront stores 64 consecutive candidates),
    // or [(cand*H + t)*width + k] (wavefront-per-candidate kernels: a wavefront stores one candidate's row)
    out[i] = (double)(candidate_major ? src[(size_t)cand * H * width + o] : src[(size_t)o * N + cand]);
  }
}
template <typename T>
__global__ void gather_spline_kernel(const T* nodes, int N, int np, int cand, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np) out[i] = (double)nodes[(size_t)i * N + cand];
}
template <typename T>
__global__ void scatter_nodes_kernel(const double* __restrict__ in, T* nodes, int N, int np) {
  // in: candidate-major [N][np] (the reference's per-candidate TimeSpline values) -> [np][N]
  const size_t total = (size_t)N * np;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t j = i / N, c = i % N;
    nodes[i] = (T)in[c * np + j];
  }
}

}  // namespace

// ===================================================================== context
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct mjpcx_ctx {
  int device = 0, precision = 64;
  const KernelEntry* kernel = nullptr;
  hipStream_t stream = nullptr;
  std::string last_error;
  // model/task dims
  int nq = 0, nv = 0, nu = 0, na = 0, nmocap = 0, nr = 0, nterm = 0, ntrace = 0, nparam = 0;
  std::vector<int> num_norm_parameter, dim_norm_residual;
  std::vector<int> ctrllimited;
  std::vector<double> ctrlrange;
  // host mirrors of the device structs (both precisions kept; only one uploaded)
  LaneModel<double> hm64{}; LaneModel<float> hm32{};
  LaneTask<double> ht64{}; LaneTask<float> ht32{};
  const void* cur_times = nullptr; const void* cur_nominal = nullptr;  // device views of the last plan inputs
  // plan inputs (node times, nominal spline, CE variances) go through a small ring of pinned
  // staging slots: one truly asynchronous H2D copy per rollout, no host sync
  static constexpr int kSlots = 4;
  struct Slot { void* host = nullptr; DevBuf dev; size_t cap = 0; hipEvent_t done = nullptr; bool pending = false; };
  Slot slots[kSlots];
  int next_slot = 0;
  // pinned + device-mapped result record of mjpcx_best
  void* best_host = nullptr; void* best_dev = nullptr; size_t best_cap = 0;
  // rollout buffers
  DevBuf d_nodes, d_in_nodes, d_ilqg, d_ilqg_out, d_wblob;
  DevBuf d_states, d_actions, d_times, d_residual, d_costs, d_trace, d_ret, d_fail, d_sort, d_stage;
  bool traj_candidate_major = false;
  int nsite_model = 0;
  double xfrc_std = 0, xfrc_rate = 1; uint64_t xfrc_seed = 0; int xfrc_offset = 0;  // pending NoisyRollout request (0: plain Rollout)  // layout of the last rollout's Trajectory buffers (true: wavefront-per-candidate kernels)
  int N = 0, H = 0, P = 0;  // shape of the last rollout
  bool have_rollout = false;
  // wavefront-per-candidate family
  bool wave = false;
  WaveHost wh;
  std::vector<unsigned char> blob_scratch;
  std::vector<double> h_stage;
  // tuning aids read from the environment ONCE, in mjpcx_create: MJPCX_STAMPS=<step> (phase cycle stamps of candidate 0)
  int stamp_step = -1;
  bool no_tree = false;  // MJPCX_NO_TREE=1: keep the row-table constraint path (A/B runs)
  bool no_lds_model = false;  // MJPCX_NO_LDS_MODEL=1: generic kernel even for a registered model (A/B runs)
  int max_waves = 8;          // MJPCX_TREE_WAVES=<1..8, fp32: 1..12>: wavefronts per workgroup of the registered-model kernel (fp32 default 12: wave32.hip)
  int tree_mode = 16;         // tree_kernel.h mode bits: 16 dynamic candidate hand-out (default), 8 arena poison, 2 image self-check (MJPCX_TREE_MODE)
  bool no_second_pass = false;  // MJPCX_TREE_ONE_PASS=1: leave list overflows as failures (tuning: counts them)
  // multi-GPU (mjpcx_comm_*): the RCCL communicator of this context's rank and its staging buffers
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  DevBuf d_comm_send, d_comm_recv;
  std::vector<double> h_comm;
  DevBuf d_work;              // its self-check counter
  DevBuf d_ovf;               // its global slabs for cones beyond the LDS list
  bool no_cone_slabs = false; // MJPCX_TREE_NO_SLABS=1: overflow goes to the second pass instead (A/B runs)
  int num_cu = 256;
  // quad kernel (quad_kernel.h): four lanes per candidate; fp64 contexts of a model quad_build accepts
  bool quad_ok = false;       // MJPCX_NO_QUAD=1 keeps the wavefront-per-candidate kernels (A/B runs)
  bool quad_stamps = false;   // MJPCX_QUAD_STAMPS=1: phase cycle stamps of wavefront 0 (tuning aid; synchronises every rollout)
  int quad_min_n = 2048;          // batches below this go to rollout_tree_kernel<A1> (MJPCX_QUAD_MIN_N). 2048 = a rank's share of configs[2]'s 16384 candidates over 8 GPUs: there the
                                  // two kernels are level (gait steps, same box: 39.7 ms here at 4 candidates per wavefront, 37.8 there), and a sharded run then executes the
                                  // SAME kernel as one rank -- equal results bit for bit, not to a tolerance
  int quad_cpw = 0;               // candidates per wavefront of the quad kernel (0: chosen from the batch size; MJPCX_QUAD_CPW)
  int quad_con_cap = 0;           // MJPCX_QUAD_CON_CAP=<n>: hand on candidates with more than n contacts in a lane (tests of the hand-on path)
  bool quad_no_fallback = false;  // MJPCX_QUAD_NO_FALLBACK=1: leave the handed-on candidates flagged (tuning: failure[] then carries reason and step)
  bool no_quad_feedback = false;  // MJPCX_NO_QUAD_FEEDBACK=1: the iLQG rollouts stay on the wavefront-per-candidate kernel (A/B runs, tests)
  bool quad_stats = false;    // MJPCX_QUAD_STATS=1: print how many candidates each rollout handed to the fallback kernel, by reason
  std::string quad_why;       // why quad_build declined (mjpcx_create_error after MJPCX_OK carries it when MJPCX_QUAD_STATS is set)
  // limb kernel (limb_kernel.h): four lanes per candidate, one per limb of the Humanoid class limb_build accepts; both precisions
  bool limb_ok = false;       // MJPCX_NO_LIMB=1 keeps the wavefront-per-candidate kernel (A/B runs)
  int limb_min_n = 2560;      // batches below this go to rollout_tree_kernel<Humanoid> (MJPCX_LIMB_MIN_N). The limb kernel's launch is one wavefront's latency, ~20 ms for
                              // any batch up to 8192; the tree kernel takes 13.2 / 13.9 / 15.5 ms at 512 / 1024 / 2048 candidates and 27.1 at 3072 (limb: 18.9), fp32 on MI355X
  int limb_cpw = 0;           // candidates per wavefront (0: chosen from the batch size; MJPCX_LIMB_CPW)
  bool limb_no_fallback = false;  // MJPCX_LIMB_NO_FALLBACK=1: leave the handed-on candidates flagged (tuning)
  std::string limb_why;       // why limb_build declined
  int limb_ids[32];           // residual_int[2..33] the limb model was built for (tracking sites, mocap bodies)
  DevBuf d_limb;              // the model image in the context's precision
  int quad_ids[7] = {0, 0, 0, 0, 0, 0, 0};  // residual_int[1..7] the quad model was built for (torso, head site, goal mocap, feet)
  DevBuf d_qmodel, d_qtab, d_qstats, d_qstamps, d_qwave, d_qovf, d_qclass;
  void* h_qstats = nullptr;       // pinned copy of d_qstats (whether the hand-on pass has anything to do)
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<hipEvent_t> events_main;  // recorded right after the rollout's FIRST (dominant) kernel: events[i].first .. events_main[i] is that kernel alone
  hipEvent_t cur_main = nullptr;
  size_t events_used = 0;
};

namespace {
int fail(mjpcx_ctx* c, int code, const std::string& msg) {
  if (c) c->last_error = msg;
  return code;
}
#define HIPCHK(c, expr)                                                                      \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess)                                                                   \
      return fail(c, e__ == hipErrorOutOfMemory ? MJPCX_ENOMEM : MJPCX_EDEVICE,              \
                  std::string(#expr) + ": " + hipGetErrorString(e__));                       \
  } while (0)

size_t esize(const mjpcx_ctx* c) { return c->precision == 64 ? 8 : 4; }

template <typename T>
void fill_model(LaneModel<T>& d, const mjpcx_model* m) {
  std::memset(&d, 0, sizeof d);
  d.timestep = (T)m->timestep;
  for (int k = 0; k < 3; k++) d.gravity[k] = (T)m->gravity[k];
  d.solver_tolerance = (T)m->solver_tolerance;
  d.meaninertia = (T)m->meaninertia;
  d.disableflags = m->disableflags;
  d.solver_iterations = m->solver_iterations;
  std::vector<double> sub(m->nbody);
  for (int b = 0; b < m->nbody; b++) sub[b] = m->body_mass[b];
  for (int b = m->nbody - 1; b > 0; b--) sub[m->body_parentid[b]] += sub[b];
  for (int b = 0; b < m->nbody; b++) {
    for (int k = 0; k < 3; k++) {
      d.body_pos[b][k] = (T)m->body_pos[3 * b + k];
      d.body_ipos[b][k] = (T)m->body_ipos[3 * b + k];
      d.body_inertia[b][k] = (T)m->body_inertia[3 * b + k];
    }
    for (int k = 0; k < 4; k++) {
      d.body_quat[b][k] = (T)m->body_quat[4 * b + k];
      d.body_iquat[b][k] = (T)m->body_iquat[4 * b + k];
    }
    d.body_mass[b] = (T)m->body_mass[b];
    d.root_invmass[b] = (T)(sub[b] > kMinVal ? 1.0 / sub[b] : 0.0);
  }
  d.any_damping = 0;
  d.integrator = m->integrator;
  for (int j = 0; j < m->njnt; j++) {
    for (int k = 0; k < 3; k++) { d.jnt_pos[j][k] = (T)m->jnt_pos[3 * j + k]; d.jnt_axis[j][k] = (T)m->jnt_axis[3 * j + k]; }
    d.jnt_stiffness[j] = (T)m->jnt_stiffness[j];
    d.jnt_range[j][0] = (T)m->jnt_range[2 * j]; d.jnt_range[j][1] = (T)m->jnt_range[2 * j + 1];
    d.jnt_margin[j] = (T)m->jnt_margin[j];
    d.jnt_solref[j][0] = (T)m->jnt_solref[2 * j]; d.jnt_solref[j][1] = (T)m->jnt_solref[2 * j + 1];
    for (int k = 0; k < 5; k++) d.jnt_solimp[j][k] = (T)m->jnt_solimp[5 * j + k];
    d.qpos0[j] = (T)m->qpos0[j]; d.qpos_spring[j] = (T)m->qpos_spring[j];
    d.dof_armature[j] = (T)m->dof_armature[j]; d.dof_damping[j] = (T)m->dof_damping[j];
    d.dof_invweight0[j] = (T)m->dof_invweight0[j];
    if (m->dof_damping[j] > 0) d.any_damping = 1;
  }
  for (int s = 0; s < m->nsite; s++)
    for (int k = 0; k < 3; k++) d.site_pos[s][k] = (T)m->site_pos[3 * s + k];
  for (int u = 0; u < m->nu; u++) {
    d.act_gear[u] = (T)m->actuator_gear[u];
    d.act_gain[u] = (T)m->actuator_gainprm[3 * u];
    for (int k = 0; k < 3; k++) d.act_bias[u][k] = (T)m->actuator_biasprm[3 * u + k];
    for (int k = 0; k < 2; k++) {
      d.act_ctrlrange[u][k] = (T)m->actuator_ctrlrange[2 * u + k];
      d.act_forcerange[u][k] = (T)m->actuator_forcerange[2 * u + k];
    }
    d.act_biastype[u] = m->actuator_biastype[u];
    d.act_ctrllimited[u] = m->actuator_ctrllimited[u];
    d.act_forcelimited[u] = m->actuator_forcelimited[u];
  }
}

bool same_model(const LaneModel<double>& a, const LaneModel<double>& b) {
  // both objects are fully zero-initialised before being filled, so a byte comparison of the
  // value representation is exact equality of every constant (the structs have no padding holes
  // between doubles; the two int members before arrays of doubles are compared as stored)
  return std::memcmp(&a, &b, sizeof a) == 0;
}

template <typename TD, typename TS>
void convert_task(LaneTask<TD>& d, const LaneTask<TS>& s) {
  for (int k = 0; k < kLaneMaxTerm; k++) {
    d.norm[k] = s.norm[k]; d.weight[k] = (TD)s.weight[k]; d.norm_p[k] = (TD)s.norm_p[k]; d.norm_q[k] = (TD)s.norm_q[k];
  }
  for (int k = 0; k < kLaneMaxParam; k++) d.parameters[k] = (TD)s.parameters[k];
  d.risk = (TD)s.risk;
  for (int k = 0; k < kLaneMaxDof; k++) { d.qpos[k] = (TD)s.qpos[k]; d.qvel[k] = (TD)s.qvel[k]; }
  d.time = (TD)s.time;
  for (int i = 0; i < kLaneMaxMocap; i++) {
    for (int k = 0; k < 3; k++) d.mocap_pos[i][k] = (TD)s.mocap_pos[i][k];
    for (int k = 0; k < 4; k++) d.mocap_quat[i][k] = (TD)s.mocap_quat[i][k];
  }
}

void set_norm_params(mjpcx_ctx* c, const double* norm_parameter) {
  int shift = 0;
  for (int k = 0; k < c->nterm; k++) {
    const int np = c->num_norm_parameter[k];
    const double p = np > 0 ? norm_parameter[shift] : 0.0, q = np > 1 ? norm_parameter[shift + 1] : 0.0;
    if (c->wave) { c->wh.norm_p[k] = p; c->wh.norm_q[k] = q; }
    else { c->ht64.norm_p[k] = p; c->ht64.norm_q[k] = q; }
    shift += np;
  }
}
}  // namespace

namespace {
int reserve_rollout(mjpcx_ctx* c, int N, int H, int P) {
  const size_t w = esize(c);
  const size_t ds = c->nq + c->nv + c->na;
  HIPCHK(c, c->d_nodes.reserve((size_t)N * P * c->nu * w));
  HIPCHK(c, c->d_states.reserve((size_t)N * H * ds * w));
  HIPCHK(c, c->d_actions.reserve((size_t)N * H * c->nu * w));
  HIPCHK(c, c->d_times.reserve((size_t)N * H * w));
  HIPCHK(c, c->d_residual.reserve((size_t)N * H * c->nr * w));
  HIPCHK(c, c->d_costs.reserve((size_t)N * H * w));
  HIPCHK(c, c->d_trace.reserve((size_t)N * H * 3 * std::max(c->ntrace, 1) * w));
  HIPCHK(c, c->d_ret.reserve((size_t)N * 8));
  HIPCHK(c, c->d_fail.reserve((size_t)N * 4));
  return MJPCX_OK;
}

// Stage [node_times (T) | nominal (T) | variance (f64)] into the next pinned slot and enqueue ONE
// asynchronous H2D copy. The slot is recycled only after the kernel that reads it has finished.
template <typename T>
int stage_plan_inputs(mjpcx_ctx* c, int P, const double* node_times, const double* nominal, const double* variance,
                      const T** d_times, const T** d_nominal, const double** d_variance, mjpcx_ctx::Slot** used,
                      const void** d_blob = nullptr) {
  const int np = P * c->nu;
  const size_t off_nom = ((size_t)P * sizeof(T) + 15) & ~(size_t)15;
  const size_t off_var = (off_nom + (size_t)np * sizeof(T) + 15) & ~(size_t)15;
  const size_t off_blob = (off_var + (size_t)np * 8 + 15) & ~(size_t)15;
  const size_t bytes = off_blob + (c->wave ? (sizeof(T) == 4 ? c->wh.blob_bytes32 : c->wh.blob_bytes) : 0);
  mjpcx_ctx::Slot& s = c->slots[c->next_slot];
  c->next_slot = (c->next_slot + 1) % mjpcx_ctx::kSlots;
  if (s.pending) { HIPCHK(c, hipEventSynchronize(s.done)); s.pending = false; }
  if (!s.done) HIPCHK(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  if (bytes > s.cap) {
    if (s.host) (void)hipHostFree(s.host);
    s.host = nullptr; s.cap = 0;
    HIPCHK(c, hipHostMalloc(&s.host, bytes, hipHostMallocDefault));
    s.cap = bytes;
  }
  HIPCHK(c, s.dev.reserve(bytes));
  char* h = (char*)s.host;
  T* ht = (T*)h;
  for (int p = 0; p < P; p++) ht[p] = (T)node_times[p];
  T* hn = (T*)(h + off_nom);
  if (nominal) for (int j = 0; j < np; j++) hn[j] = (T)nominal[j];
  if (variance) std::memcpy(h + off_var, variance, (size_t)np * 8);
  if (c->wave) {
    if (sizeof(T) == 4) c->wh.fill_blob32(h + off_blob); else c->wh.fill_blob(h + off_blob);
    if (d_blob) *d_blob = (const char*)s.dev.p + off_blob;
  }
  HIPCHK(c, hipMemcpyAsync(s.dev.p, s.host, bytes, hipMemcpyHostToDevice, c->stream));
  *d_times = (const T*)s.dev.p;
  *d_nominal = (const T*)((char*)s.dev.p + off_nom);
  *d_variance = (const double*)((char*)s.dev.p + off_var);
  *used = &s;
  return MJPCX_OK;
}

// tuning aid (MJPCX_STAMPS=<step>): phase cycle stamps of candidate 0 at one step of the wavefront-per-candidate kernel
void print_wave_stamps(mjpcx_ctx* c, long long* stamps, int stamp_step, size_t lds, bool tree = false) {
  long long h[48];
  (void)hipStreamSynchronize(c->stream);
  (void)hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost);
  static const char* nm[] = {"policy", "kinematics", "compos", "crb", "cholM", "collision", "comvel", "make_constraint", "smooth", "solve",
                             "newton", "residual", "cost+record", "euler"};
  static const char* nt[] = {"policy", "kinematics", "compos", "crb", "cholM", "comvel", "smooth", "solve", "collision", "make_constraint",
                             "newton", "residual", "cost+record", "euler"};
  std::fprintf(stderr, "wave kernel phase cycles (step %d, candidate 0; LDS %zu B):", stamp_step, lds);
  for (int k = 0; k < 14; k++) std::fprintf(stderr, " %s %lld", tree ? nt[k] : nm[k], h[k + 1] - h[k]);
  std::fprintf(stderr, " | newton iters %lld: grad %lld hess %lld chol+solve %lld linesearch %lld\n", h[20], h[21] - h[10], h[22] - h[21],
               h[23] - h[22], h[24] - h[23]);
  std::fprintf(stderr, "  newton totals over iterations: grad %lld coneblocks %lld hess %lld chol+solve %lld jv+q %lld linesearch %lld (%lld trials) update+cost %lld\n",
               h[32], h[33], h[34], h[35], h[36], h[37], h[39], h[38]);
  std::fprintf(stderr, "  hessian parts: copy M %lld diagonal rows %lld row facts %lld simple rows (MFMA) %lld cones = hess - these\n", h[40], h[41], h[42], h[43]);
}

// registered model (lds_model.h / tree_kernel.h): [image | blob | W arenas] of LDS per workgroup, W <= 8 wavefronts, persistent.
// Two launches: the batch with the small contact lists, then the candidates that overflowed them with the large lists.
template <class C, typename T, bool BIG>
hipError_t launch_tree_pass(mjpcx_ctx* c, const WaveModelT<T>& wm, const WaveTaskT<T>& wt, const RolloutArgs<T>& a, const void* image, size_t blob_bytes,
                            int N, int P, bool only_flagged = false) {
  const int caps = BIG ? (sizeof(T) == 8 ? w64::kTreeMaxSimpleBig : w32::kTreeMaxSimpleBig) : (sizeof(T) == 8 ? w64::kTreeMaxSimple : w32::kTreeMaxSimple);
  const int capc = BIG ? (sizeof(T) == 8 ? w64::kTreeMaxConeBig : w32::kTreeMaxConeBig) : (sizeof(T) == 8 ? w64::kTreeMaxCone : w32::kTreeMaxCone);
  const size_t arena = sizeof(T) == 8
      ? (8 * w64::wave_lds_elems_tree(wm.nq, wm.nv, wm.nu, wm.nbody, wm.njnt, wm.nsite, wt.nr, wt.nterm, P, a.xfrc_scale > 0, caps, capc) + 15) & ~(size_t)15
      : (4 * w32::wave_lds_elems_tree(wm.nq, wm.nv, wm.nu, wm.nbody, wm.njnt, wm.nsite, wt.nr, wt.nterm, P, a.xfrc_scale > 0, caps, capc) + 15) & ~(size_t)15;
  const size_t fixed = LdsLayout<C, T>::kBytes + blob_bytes;
  if (fixed + arena > 160 * 1024) return hipErrorInvalidValue;
  int W = (int)std::min<size_t>((size_t)c->max_waves, (160 * 1024 - fixed) / arena);
  W = std::max(1, std::min(W, (N + c->num_cu - 1) / c->num_cu));  // small batches: spread over the CUs first
  int grid = std::min(c->num_cu, (N + W - 1) / W);
  if (BIG) grid = std::min(grid, 64);  // the second pass scans the failure flags; a handful of rollouts at most
  const int mode = (BIG ? (c->tree_mode & 8) : c->tree_mode) | (only_flagged && !BIG ? 32 : 0);  // 32: only the candidates the quad kernel handed on
  const size_t lds = fixed + (size_t)W * arena;
  hipError_t e = c->d_work.reserve(16);
  if (e != hipSuccess) return e;
  if (mode & (2 | 16)) if ((e = hipMemsetAsync(c->d_work.p, 0, 16, c->stream)) != hipSuccess) return e;  // self-check count / work counter
  // first pass: one slab per wavefront for the cones beyond the LDS list (wave_tree.h; a few tens of MB, never touched in the common case)
  void* slabs = nullptr;
  if (!BIG && !c->no_cone_slabs) {
    const size_t per_wave = (size_t)(w64::kTreeMaxConeTotal - capc) * w64::kConeRec * sizeof(T);
    if ((e = c->d_ovf.reserve((size_t)grid * W * per_wave)) != hipSuccess) return e;
    slabs = c->d_ovf.p;
  }
  if constexpr (sizeof(T) == 8) {
    auto kern = w64::rollout_tree_kernel<C, BIG>;
    if ((e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * W), lds, c->stream, wm, wt, a, (const unsigned char*)image, (unsigned)blob_bytes, (unsigned)arena,
                       (int*)c->d_work.p, mode, (T*)slabs);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  } else {  // (the fp32 kernels live in wave32.hip)
    if ((e = launch_tree_kernel_f32(std::is_same<C, TreeCfgA1>::value ? 0 : 1, BIG, grid, 64 * W, lds, wm, wt, a, (const unsigned char*)image,
                                    (unsigned)blob_bytes, (unsigned)arena, (int*)c->d_work.p, mode, (float*)slabs, c->stream)) != hipSuccess) return e;
  }
  if (c->stamp_step >= 0) std::fprintf(stderr, "rollout_tree_kernel%s: %d wavefronts per workgroup, grid %d, LDS %zu B (arena %zu B)\n", BIG ? " (second pass)" : "", W, grid, lds, arena);
  if (mode & 2) {  // self-check launch: report the mismatch count
    int h[2] = {0, 0};
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(h, c->d_work.p, 8, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "rollout_tree_kernel self-check: %d mismatching entries between the LDS image and the model (W = %d, grid = %d, lds = %zu B)\n", h[1], W, grid, lds);
  }
  return hipSuccess;
}
template <class C, typename T>
hipError_t launch_tree(mjpcx_ctx* c, const WaveModelT<T>& wm, const WaveTaskT<T>& wt, const RolloutArgs<T>& a, const void* image, size_t blob_bytes,
                       int N, int P, bool only_flagged = false) {
  hipError_t e = launch_tree_pass<C, T, false>(c, wm, wt, a, image, blob_bytes, N, P, only_flagged);
  // (mjpcx_timing_read_main: the first pass is the kernel that rolls the batch out; the pass over the quad kernel's hand-ons does not own the stamp)
  if (e == hipSuccess && !only_flagged && c->timing && c->cur_main) { if ((e = hipEventRecord(c->cur_main, c->stream)) == hipSuccess) c->cur_main = nullptr; }
  if (e != hipSuccess || (c->tree_mode & 2) || c->no_second_pass) return e;
  RolloutArgs<T> a2 = a;
  a2.noise.mode = -1;  // the first pass left every candidate's spline nodes in a.nodes
  WaveTaskT<T> wt2 = wt;
  wt2.stamps = nullptr;
  return launch_tree_pass<C, T, true>(c, wm, wt2, a2, image, blob_bytes, N, P);
}

// quad kernel (quad_kernel.h), then the wavefront-per-candidate kernel for the candidates it handed on
hipError_t launch_quad(mjpcx_ctx* c, const WaveModel& wm, const WaveTask& wt, const RolloutArgs<double>& a, int N, int P) {
  quad::QArgs q{};
  q.N = a.N; q.H = a.H; q.P = a.P; q.interp = a.interp; q.node_times = a.node_times; q.nodes = a.nodes; q.nominal = a.nominal;
  q.noise_mode = a.noise.mode; q.seed = a.noise.seed; q.iteration = a.noise.iteration; q.candidate_offset = a.noise.candidate_offset;
  q.nominal_candidate = a.noise.nominal_candidate; q.explore_count = a.noise.explore_count; q.std0 = a.noise.std0; q.std1 = a.noise.std1;
  q.param_variance = a.noise.param_variance;
  q.states = a.states; q.actions = a.actions; q.times = a.times; q.residual = a.residual; q.costs = a.costs; q.trace = a.trace;
  q.total_return = a.total_return; q.failure = a.failure; q.con_cap = c->quad_con_cap; q.cpw = c->quad_cpw;
  const quad::QBlob bo{wt.off_time, wt.off_mocap, wt.off_weight, wt.off_normp, wt.off_normq, wt.off_param, wt.off_risk, wt.off_rreal, wt.off_rint};
  hipError_t e;
  if (quad::quad_uses_ovf_slab()) {  // (a build with QEXP_OVF_SLAB: contacts beyond a lane's LDS slots in global memory, 150 KB per wavefront)
    if ((e = c->d_qovf.reserve((size_t)quad::quad_waves(N, c->quad_cpw) * quad::quad_ovf_doubles_per_wave() * sizeof(double))) != hipSuccess) return e;
    q.ovf_slab = (double*)c->d_qovf.p;
  }
  if ((e = hipMemsetAsync(c->d_qstats.p, 0, 32, c->stream)) != hipSuccess) return e;  // (how many candidates are handed on, by reason: mjpcx_quad_stats)
  if (c->quad_stamps) {
    if ((e = hipMemsetAsync(c->d_qstamps.p, 0, 512, c->stream)) != hipSuccess) return e;
    q.stamps = (long long*)c->d_qstamps.p;
    const size_t wb = (size_t)N * 32;  // (one wavefront per candidate at most)
    if ((e = c->d_qwave.reserve(wb)) != hipSuccess || (e = hipMemsetAsync(c->d_qwave.p, 0, wb, c->stream)) != hipSuccess) return e;
    q.wave_times = (long long*)c->d_qwave.p;
  }
  static const bool classes = getenv("MJPCX_QUAD_CLASSES") != nullptr;  // (tuning aid: cycles by class of wavefront-step, printed after every launch)
  const int nwave = quad::quad_waves(N, c->quad_cpw);
  if (classes) {
    const size_t wb = (size_t)nwave * 64 * 8;
    if ((e = c->d_qclass.reserve(wb)) != hipSuccess || (e = hipMemsetAsync(c->d_qclass.p, 0, wb, c->stream)) != hipSuccess) return e;
    q.wave_class = (long long*)c->d_qclass.p;
  }
  if ((e = quad::launch_rollout_quad(c->d_qmodel.p, c->d_qtab.p, wt.blob, bo, q, (int*)c->d_qstats.p,
                                     c->stream)) != hipSuccess) return e;
  if (classes) {
    std::vector<long long> w((size_t)nwave * 64), tot(64, 0);
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(w.data(), c->d_qclass.p, w.size() * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < nwave; i++) for (int k = 0; k < 64; k++) tot[k] += w[(size_t)i * 64 + k];
    long long steps = 0, cyc = 0, scyc = 0;
    for (int k = 0; k < 16; k++) { steps += tot[32 + 2 * k]; cyc += tot[32 + 2 * k + 1]; scyc += tot[2 * k + 1]; }
    std::fprintf(stderr, "rollout_quad_kernel wavefront-steps by class (rel = a contact between two moving geoms in some candidate, leg = a leg-leg one, ovf = a lane beyond its "
                         "LDS slots, bey = beyond the line-search slots): %lld steps, %.0f cycles per step, %.0f of them in the solve\n", steps, steps ? (double)cyc / steps : 0.0,
                 steps ? (double)scyc / steps : 0.0);
    for (int k = 0; k < 16; k++)
      if (tot[32 + 2 * k] > 0)
        std::fprintf(stderr, "  %s%s%s%s%s: %5.1f %% of the steps, %5.1f %% of the cycles; per step %.0f cycles, solve %.0f\n", k == 0 ? "plain" : "", (k & 1) ? "rel " : "", (k & 2) ? "leg " : "",
                     (k & 4) ? "ovf " : "", (k & 8) ? "bey " : "", 100.0 * tot[32 + 2 * k] / steps, 100.0 * tot[32 + 2 * k + 1] / cyc, (double)tot[32 + 2 * k + 1] / tot[32 + 2 * k],
                     tot[2 * k] ? (double)tot[2 * k + 1] / tot[2 * k] : 0.0);
  }
  if (c->timing && c->cur_main) { if ((e = hipEventRecord(c->cur_main, c->stream)) != hipSuccess) return e; c->cur_main = nullptr; }
  if (c->quad_stamps) {
    long long h[64];
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(h, c->d_qstamps.p, 512, hipMemcpyDeviceToHost);
    static const char* nm[13] = {"policy", "kin..rne", "collision", "pairs", "factor+rows", "residual+record", "newton", "euler", "n:grad", "n:hessian", "n:factor+solve",
                                 "n:linesearch", "n:update+eval"};
    std::fprintf(stderr, "rollout_quad_kernel cycles of wavefront 0 (H = %d):", a.H);
    for (int k = 0; k < 13; k++) std::fprintf(stderr, " %s %lld", nm[k], h[k]);
    std::fprintf(stderr, " | newton iterations %lld, line-search trials %lld\n", h[16], h[17]);
    std::fprintf(stderr, "  raw counters 40..63:");
    for (int k = 40; k < 64; k++) std::fprintf(stderr, " %lld", h[k]);
    std::fprintf(stderr, "\n");
    std::fprintf(stderr, "  newton: entry %lld, first pass %lld, warm start %lld; line search: entry %lld, coefficients %lld, trials %lld\n", h[13], h[14], h[15], h[40], h[41], h[42]);
    {
      const int W = quad::quad_waves(N, c->quad_cpw);
      std::vector<long long> w((size_t)W * 4);
      (void)hipMemcpy(w.data(), c->d_qwave.p, w.size() * 8, hipMemcpyDeviceToHost);
      std::vector<int> order(W);
      for (int i = 0; i < W; i++) order[i] = i;
      std::sort(order.begin(), order.end(), [&](int x, int y) { return w[4 * x] < w[4 * y]; });
      std::fprintf(stderr, "  wavefronts by cycles (Newton iterations run, steps in the general solver, sum of largest per-lane contact counts):");
      const double qs[7] = {0.0, 0.1, 0.5, 0.9, 0.99, 0.999, 1.0};
      for (double qq : qs) { const int i = order[(size_t)(qq * (W - 1))]; std::fprintf(stderr, " p%g %lld (%lld, %lld, %lld)", 100 * qq, w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]); }
      std::fprintf(stderr, "\n");
    }
    std::fprintf(stderr, "  steps at which some lane of the wavefront holds > 0 1 2 3 4 6 8 12 contacts: %lld %lld %lld %lld %lld %lld %lld %lld; contacts per lane-step %.2f; "
                 "Newton iterations per candidate-step %.2f (the wavefront runs the slowest's: %.2f)\n", h[18], h[19], h[20], h[21], h[22], h[23], h[24], h[25],
                 (double)h[26] / (64.0 * a.H), (double)h[36] / (64.0 * a.H), (double)h[37] / a.H);
  }
  if (c->quad_no_fallback) return hipSuccess;
  // the pass over the candidates the quad kernel handed on is 16384 wavefronts that each find their candidate unflagged (0.24 ms) when
  // nothing was handed on -- the usual case: the count comes back first (32 bytes, one synchronisation the plan step pays anyway a moment later)
  if (!c->quad_stats) {
    if (!c->h_qstats && hipHostMalloc(&c->h_qstats, 32, hipHostMallocDefault) != hipSuccess) c->h_qstats = nullptr;
    if (c->h_qstats) {
      if ((e = hipMemcpyAsync(c->h_qstats, c->d_qstats.p, 32, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
      if (static_cast<const int*>(c->h_qstats)[0] == 0) return hipSuccess;
    }
  }
  RolloutArgs<double> a2 = a;
  a2.noise.mode = -1;  // the quad kernel left every candidate's spline nodes in a.nodes
  e = launch_tree<TreeCfgA1, double>(c, wm, wt, a2, c->wh.dev_image, c->wh.blob_bytes, N, P, /*only_flagged=*/true);
  if (e == hipSuccess && c->quad_stats) {
    int h[8] = {0};
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(h, c->d_qstats.p, 32, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "rollout_quad_kernel: %d of %d candidates handed to rollout_tree_kernel (contact list full %d, moving-geom pair %d, indefinite Hessian %d, "
                 "non-finite %d, both joint limits %d, trunk-leg pair %d)\n", h[0], N, h[1], h[2], h[3], h[4], h[5], h[6]);
  }
  return e;
}

// limb kernel (limb_kernel.h), then the wavefront-per-candidate kernel for the candidates it handed on
template <typename T>
hipError_t launch_limb(mjpcx_ctx* c, const WaveModelT<T>& wm, const WaveTaskT<T>& wt, const RolloutArgs<T>& a, int N, int P) {
  limb::LArgs<T> q{};
  q.N = a.N; q.H = a.H; q.P = a.P; q.interp = a.interp; q.node_times = a.node_times; q.nodes = a.nodes; q.nominal = a.nominal;
  q.noise_mode = a.noise.mode; q.seed = a.noise.seed; q.iteration = a.noise.iteration; q.candidate_offset = a.noise.candidate_offset;
  q.nominal_candidate = a.noise.nominal_candidate; q.explore_count = a.noise.explore_count; q.std0 = a.noise.std0; q.std1 = a.noise.std1;
  q.param_variance = a.noise.param_variance;
  q.states = a.states; q.actions = a.actions; q.times = a.times; q.residual = a.residual; q.costs = a.costs; q.trace = a.trace;
  q.total_return = a.total_return; q.failure = a.failure; q.cpw = c->limb_cpw;
  const limb::LBlob bo{wt.off_time, wt.off_mocap, wt.off_weight, wt.off_normp, wt.off_normq, wt.off_param, wt.off_risk, wt.off_rreal, wt.off_rint};
  hipError_t e;
  if ((e = hipMemsetAsync(c->d_qstats.p, 0, 32, c->stream)) != hipSuccess) return e;  // (how many candidates are handed on, by reason: mjpcx_quad_stats)
  static const bool stamps = getenv("MJPCX_LIMB_STAMPS") != nullptr;  // (tuning aid: phase cycles of wavefront 0, Newton iterations; synchronises every rollout)
  if (stamps) {
    if ((e = c->d_qstamps.reserve(512)) != hipSuccess || (e = hipMemsetAsync(c->d_qstamps.p, 0, 512, c->stream)) != hipSuccess) return e;
    q.stamps = (long long*)c->d_qstamps.p;
    if ((e = c->d_qwave.reserve((size_t)N * 4)) != hipSuccess) return e;
    q.iters = (int*)c->d_qwave.p;
  }
  if ((e = limb::launch_rollout_limb(c->d_limb.p, wt.blob, bo, q, wm.key_mpos, (int*)c->d_qstats.p, c->stream)) != hipSuccess) return e;
  if (stamps) {
    long long h[64];
    std::vector<int> it((size_t)N);
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(h, c->d_qstamps.p, sizeof h, hipMemcpyDeviceToHost);
    (void)hipMemcpy(it.data(), c->d_qwave.p, (size_t)N * 4, hipMemcpyDeviceToHost);
    long long sum = 0; int mx = 0;
    for (int v : it) { sum += v; mx = v > mx ? v : mx; }
    std::fprintf(stderr, "rollout_limb_kernel cycles of wavefront 0 (H = %d): policy %lld forward %lld residual+record %lld newton %lld euler %lld | Newton iterations per candidate-step %.2f (most over a rollout: %d)\n",
                 a.H, h[0], h[1], h[2], h[3], h[4], (double)sum / ((double)N * (a.H > 1 ? a.H - 1 : 1)), mx);
    std::fprintf(stderr, "  newton: setup %lld | per iteration (the wavefront ran %lld): hessian %lld factor+solve %lld woodbury %lld mul+dots %lld linesearch %lld update+eval %lld\n",
                 h[8], h[15], h[9], h[10], h[11], h[12], h[13], h[14]);
    std::fprintf(stderr, "  line-search derivative evaluations of candidate 0: %lld\n", h[16]);
    std::fprintf(stderr, "  forward: kinematics+inertias %lld M %lld velocity+bias %lld factor+solve %lld sites %lld rows %lld floor %lld pairs %lld pair list %lld\n",
                 h[20], h[21], h[22], h[23], h[24], h[25], h[26], h[27], h[28]);
    std::fprintf(stderr, "  residual: joint entries %lld marker averages %lld marker entries %lld\n", h[29], h[30], h[31]);
    if (h[32] | h[33] | h[34] | h[35] | h[36] | h[37] | h[38] | h[39])  // (stamps a tuning build adds: -DLEXP_XSTAMPS)
      std::fprintf(stderr, "  raw 32..39: %lld %lld %lld %lld %lld %lld %lld %lld\n", h[32], h[33], h[34], h[35], h[36], h[37], h[38], h[39]);
  }
  if (c->timing && c->cur_main) { if ((e = hipEventRecord(c->cur_main, c->stream)) == hipSuccess) c->cur_main = nullptr; else return e; }
  if (c->limb_no_fallback) return hipSuccess;
  if (!c->quad_stats) {  // the count of hand-ons comes back first: nothing handed on (the usual case) -> no second launch
    if (!c->h_qstats && hipHostMalloc(&c->h_qstats, 32, hipHostMallocDefault) != hipSuccess) c->h_qstats = nullptr;
    if (c->h_qstats) {
      if ((e = hipMemcpyAsync(c->h_qstats, c->d_qstats.p, 32, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
      if (static_cast<const int*>(c->h_qstats)[0] == 0) return hipSuccess;
    }
  }
  RolloutArgs<T> a2 = a;
  a2.noise.mode = -1;  // the limb kernel left every candidate's spline nodes in a.nodes
  e = launch_tree<TreeCfgHumanoid, T>(c, wm, wt, a2, sizeof(T) == 8 ? c->wh.dev_image : c->wh.dev_image32, sizeof(T) == 8 ? c->wh.blob_bytes : c->wh.blob_bytes32, N, P, /*only_flagged=*/true);
  if (e == hipSuccess && c->quad_stats) {
    int h[8] = {0};
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(h, c->d_qstats.p, 32, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "rollout_limb_kernel: %d of %d candidates handed to rollout_tree_kernel (floor contact list full %d, more moving-geom contacts than slots %d, "
                 "indefinite matrix %d, non-finite %d, both limits %d, trunk on the floor %d)\n", h[0], N, h[1], h[2], h[3], h[4], h[5], h[6]);
  }
  return e;
}

template <typename T>
int do_rollout(mjpcx_ctx* c, int N, int H, int P, int interp, const double* node_times,
               const double* node_values, const double* nominal, const mjpcx_noise_spec* ns) {
  int rc;
  if ((rc = reserve_rollout(c, N, H, P)) != MJPCX_OK) return rc;
  const int np = P * c->nu;
  RolloutArgs<T> a{};
  a.N = N; a.H = H; a.P = P; a.interp = interp;
  a.nodes = (T*)c->d_nodes.p;
  a.noise.mode = -1;
  const double* d_var = nullptr;
  mjpcx_ctx::Slot* slot = nullptr;
  const bool ce = ns && ns->mode == MJPCX_NOISE_CROSS_ENTROPY;
  if (ce && !ns->param_variance) return fail(c, MJPCX_EINVAL, "cross-entropy noise needs param_variance");
  const void* d_blob = nullptr;
  if ((rc = stage_plan_inputs<T>(c, P, node_times, nominal, ce ? ns->param_variance : nullptr, &a.node_times,
                                 &a.nominal, &d_var, &slot, &d_blob)) != MJPCX_OK) return rc;
  if (node_values) {
    // candidate-major host splines -> [node][actuator][candidate] on the device (not the hot path:
    // the planner generates candidates on the device; this entry serves tests and NominalTrajectory)
    HIPCHK(c, c->d_in_nodes.reserve((size_t)N * np * 8));
    HIPCHK(c, hipMemcpyAsync(c->d_in_nodes.p, node_values, (size_t)N * np * 8, hipMemcpyHostToDevice, c->stream));
    const size_t total = (size_t)N * np;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL((scatter_nodes_kernel<T>), dim3(blocks), dim3(256), 0, c->stream,
                       (const double*)c->d_in_nodes.p, (T*)c->d_nodes.p, N, np);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));  // pageable source buffer: make caller reuse safe
  } else {
    a.noise.mode = ns->mode;
    a.noise.seed = ns->seed; a.noise.iteration = ns->iteration;
    a.noise.candidate_offset = ns->candidate_offset; a.noise.nominal_candidate = ns->nominal_candidate;
    a.noise.explore_count = ns->explore_count; a.noise.std0 = ns->std0; a.noise.std1 = ns->std1;
    a.noise.param_variance = ce ? d_var : nullptr;
  }
  a.states = (T*)c->d_states.p; a.actions = (T*)c->d_actions.p; a.times = (T*)c->d_times.p;
  a.residual = (T*)c->d_residual.p; a.costs = (T*)c->d_costs.p; a.trace = (T*)c->d_trace.p;
  a.total_return = (double*)c->d_ret.p; a.failure = (int*)c->d_fail.p;
  a.xfrc_decay = 0; a.xfrc_scale = 0; a.xfrc_seed = 0;
  if (c->xfrc_std > 0) {  // Ornstein-Uhlenbeck in discrete time (trajectory.cc:149-150), at the planning timestep
    a.xfrc_decay = std::exp(-(c->wave ? c->wh.m.timestep : c->hm64.timestep) / c->xfrc_rate);
    a.xfrc_scale = c->xfrc_std * std::sqrt(1 - a.xfrc_decay * a.xfrc_decay);
    a.xfrc_seed = c->xfrc_seed;
    if (a.noise.mode < 0) a.noise.candidate_offset = c->xfrc_offset;
  }

  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->timing) {
    if (c->events_used == c->events.size()) {
      hipEvent_t x, y;
      HIPCHK(c, hipEventCreate(&x));
      HIPCHK(c, hipEventCreate(&y));
      c->events.emplace_back(x, y);
      hipEvent_t z;
      HIPCHK(c, hipEventCreate(&z));
      c->events_main.push_back(z);
    }
    e0 = c->events[c->events_used].first; e1 = c->events[c->events_used].second;
    c->cur_main = c->events_main[c->events_used];
    c->events_used++;
    HIPCHK(c, hipEventRecord(e0, c->stream));
  }
  hipError_t le;
  if constexpr (sizeof(T) == 8) {
    if (c->wave) {
      WaveTask wt = c->wh.t;
      wt.blob = (const double*)d_blob;
      wt.stamps = nullptr;
      if (c->stamp_step >= 0) {
        HIPCHK(c, c->d_stage.reserve(48 * 8));
        HIPCHK(c, hipMemsetAsync(c->d_stage.p, 0, 48 * 8, c->stream));
        wt.stamps = (long long*)c->d_stage.p;
        wt.stamp_step = c->stamp_step;
      }
      const WaveModel& wm = c->wh.m;
      // A rank's share of the batch decides the kernel: the quad kernel packs 16 candidates into a wavefront, which fills the 1024 SIMDs
      // at N = 16384 but leaves most of them idle below a quarter of that, where one wavefront per candidate (a third of the latency per
      // step) is quicker -- measured cross-over on MI355X: N = 4096 (tools/quad_n_sweep.py; MJPCX_QUAD_MIN_N moves it)
      if (c->wh.registered == 0 && c->quad_ok && a.xfrc_scale == 0 && !wt.stamps && N >= c->quad_min_n) {
        le = launch_quad(c, wm, wt, a, N, P);
      } else if (c->wh.registered == 0) {
        if (c->quad_ok) (void)hipMemsetAsync(c->d_qstats.p, 0, 32, c->stream);  // (mjpcx_quad_stats reports the LAST rollout: nothing was handed on in this one)
        le = launch_tree<TreeCfgA1, double>(c, wm, wt, a, c->wh.dev_image, c->wh.blob_bytes, N, P);
        if (wt.stamps && le == hipSuccess) print_wave_stamps(c, wt.stamps, wt.stamp_step, 0, true);
      } else if (c->wh.registered == 1 && c->limb_ok && a.xfrc_scale == 0 && !wt.stamps && N >= c->limb_min_n) {
        le = launch_limb<double>(c, wm, wt, a, N, P);
      } else if (c->wh.registered == 1 && wm.integrator != MJPCX_INT_RK4) {
        if (c->limb_ok) (void)hipMemsetAsync(c->d_qstats.p, 0, 32, c->stream);
        le = launch_tree<TreeCfgHumanoid, double>(c, wm, wt, a, c->wh.dev_image, c->wh.blob_bytes, N, P);
        if (wt.stamps && le == hipSuccess) print_wave_stamps(c, wt.stamps, wt.stamp_step, 0, true);
      } else {
      const bool tree = c->wh.tree_ok && !c->no_tree;
      const bool rk4 = wm.integrator == MJPCX_INT_RK4;  // one (NMAX = 32) instantiation per kernel family carries mj_RungeKutta
      // Jacobian-free path, Euler: two launches -- short contact lists for everybody, the long ones for the candidates that overflowed
      const bool two_pass = tree && !rk4 && !c->no_second_pass;
      auto tree_lds = [&](int caps, int capc) { return (8 * w64::wave_lds_elems_tree(wm.nq, wm.nv, wm.nu, wm.nbody, wm.njnt, wm.nsite, wt.nr, wt.nterm, P, a.xfrc_scale > 0, caps, capc) + 15) & ~(size_t)15; };
      const size_t lds_big = tree ? tree_lds(w64::kTreeMaxSimpleBig, w64::kTreeMaxConeBig)
                                  : (8 * w64::wave_lds_elems(wm.nq, wm.nv, wm.nu, wm.nbody, wm.njnt, wm.nsite, wt.nr, wt.nterm, P, wm.cone, /*nodes_in_lds=*/false, a.xfrc_scale > 0) + 15) & ~(size_t)15;
      const size_t lds = two_pass ? tree_lds(w64::kTreeMaxSimple, w64::kTreeMaxCone) : lds_big;
      if (lds_big > 160 * 1024) return fail(c, MJPCX_EUNSUPPORTED, "model state does not fit the 160 KB LDS of a CU");
      // the register-resident Cholesky is unrolled to NMAX columns: instantiations that fit the registered models exactly (A1:
      // nv = 18, humanoid: 27) skip the padding columns' updates (humanoid: 30 % of the factorisation's instructions)
      auto kern_big = rk4 ? (tree ? w64::rollout_wave_kernel<32, true, true> : w64::rollout_wave_kernel<32, false, true>)
                : tree ? w64::rollout_wave_kernel<32, true>  // (the shipped tree models are registered: one width serves the unregistered ones)
                : wm.nv <= 18 ? w64::rollout_wave_kernel<18> : wm.nv <= 20 ? w64::rollout_wave_kernel<20>
                : wm.nv <= 28 ? w64::rollout_wave_kernel<28> : w64::rollout_wave_kernel<32>;
      auto kern = !two_pass ? kern_big
                : w64::rollout_wave_kernel<32, true, false, true>;
      le = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (le == hipSuccess) {
        hipLaunchKernelGGL(kern, dim3(N), dim3(64), lds, c->stream, wm, wt, a);
        le = hipGetLastError();
      }
      if (two_pass && le == hipSuccess) {
        if (c->timing && c->cur_main) { if ((le = hipEventRecord(c->cur_main, c->stream)) == hipSuccess) c->cur_main = nullptr; }
        RolloutArgs<double> a2 = a;
        a2.noise.mode = -1;  // the first pass left every candidate's spline nodes in a.nodes
        a2.only_overflowed = 1;
        if (le == hipSuccess) le = hipFuncSetAttribute((const void*)kern_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big);
        if (le == hipSuccess) {
          hipLaunchKernelGGL(kern_big, dim3(N), dim3(64), lds_big, c->stream, wm, wt, a2);
          le = hipGetLastError();
        }
      }
      if (wt.stamps && le == hipSuccess) print_wave_stamps(c, wt.stamps, wt.stamp_step, lds, tree);
      }
    } else {
      le = c->kernel->launch64(c->hm64, c->ht64, a, c->stream);
    }
  } else if (c->wave) {
    // the fp32 instantiation of the wavefront-per-candidate kernel (BASELINE configs[3]'s precision)
    WaveTaskT<float> wt = c->wh.t32;
    wt.blob = (const float*)d_blob;
    wt.stamps = nullptr;
    if (c->stamp_step >= 0) {
      HIPCHK(c, c->d_stage.reserve(48 * 8));
      HIPCHK(c, hipMemsetAsync(c->d_stage.p, 0, 48 * 8, c->stream));
      wt.stamps = (long long*)c->d_stage.p;
      wt.stamp_step = c->stamp_step;
    }
    const WaveModelT<float>& wm = c->wh.m32;
    if (c->wh.registered == 0) {
      le = launch_tree<TreeCfgA1, float>(c, wm, wt, a, c->wh.dev_image32, c->wh.blob_bytes32, N, P);
      if (wt.stamps && le == hipSuccess) print_wave_stamps(c, wt.stamps, wt.stamp_step, 0, true);
    } else if (c->wh.registered == 1 && c->limb_ok && a.xfrc_scale == 0 && !wt.stamps && N >= c->limb_min_n) {
      le = launch_limb<float>(c, wm, wt, a, N, P);
    } else if (c->wh.registered == 1 && wm.integrator != MJPCX_INT_RK4) {
      if (c->limb_ok) (void)hipMemsetAsync(c->d_qstats.p, 0, 32, c->stream);
      le = launch_tree<TreeCfgHumanoid, float>(c, wm, wt, a, c->wh.dev_image32, c->wh.blob_bytes32, N, P);
      if (wt.stamps && le == hipSuccess) print_wave_stamps(c, wt.stamps, wt.stamp_step, 0, true);
    } else {
    // (the Jacobian-free constraint path in fp32 for the models it covers that are not RK4 -- the Humanoid of configs[3] --, in two
    // passes as in fp64)
    const bool tree = c->wh.tree_ok && !c->no_tree && wm.integrator != MJPCX_INT_RK4;
    const bool two_pass = tree && !c->no_second_pass;
    auto tree_lds = [&](int caps, int capc) { return (4 * w32::wave_lds_elems_tree(wm.nq, wm.nv, wm.nu, wm.nbody, wm.njnt, wm.nsite, wt.nr, wt.nterm, P, a.xfrc_scale > 0, caps, capc) + 15) & ~(size_t)15; };
    const size_t lds_big = tree ? tree_lds(w32::kTreeMaxSimpleBig, w32::kTreeMaxConeBig)
                                : (4 * w32::wave_lds_elems(wm.nq, wm.nv, wm.nu, wm.nbody, wm.njnt, wm.nsite, wt.nr, wt.nterm, P, wm.cone, /*nodes_in_lds=*/false, a.xfrc_scale > 0) + 15) & ~(size_t)15;
    const size_t lds = two_pass ? tree_lds(w32::kTreeMaxSimple, w32::kTreeMaxCone) : lds_big;
    if (lds_big > 160 * 1024) return fail(c, MJPCX_EUNSUPPORTED, "model state does not fit the 160 KB LDS of a CU");
    const int kern_big = wm.integrator == MJPCX_INT_RK4 ? kW32Rk4 : tree ? kW32Tree
                       : wm.nv <= 18 ? kW32Rows18 : wm.nv <= 20 ? kW32Rows20 : wm.nv <= 28 ? kW32Rows28 : kW32Rows32;
    le = launch_wave_kernel_f32(two_pass ? kW32TreeSmall : kern_big, N, lds, wm, wt, a, c->stream);  // (the fp32 kernels live in wave32.hip)
    if (two_pass && le == hipSuccess) {
      if (c->timing && c->cur_main) { if ((le = hipEventRecord(c->cur_main, c->stream)) == hipSuccess) c->cur_main = nullptr; }
      RolloutArgs<float> a2 = a;
      a2.noise.mode = -1;
      a2.only_overflowed = 1;
      if (le == hipSuccess) le = launch_wave_kernel_f32(kern_big, N, lds_big, wm, wt, a2, c->stream);
    }
    if (wt.stamps && le == hipSuccess) print_wave_stamps(c, wt.stamps, wt.stamp_step, lds, tree);
    }
  } else {
    convert_task(c->ht32, c->ht64);
    le = c->kernel->launch32(c->hm32, c->ht32, a, c->stream);
  }
  if (le != hipSuccess) return fail(c, MJPCX_EDEVICE, std::string("rollout kernel launch: ") + hipGetErrorString(le));
  if (c->timing && c->cur_main) { HIPCHK(c, hipEventRecord(c->cur_main, c->stream)); c->cur_main = nullptr; }  // (paths with one kernel, or whose first pass is not singled out)
  if (c->timing) HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventRecord(slot->done, c->stream));
  slot->pending = true;
  c->N = N; c->H = H; c->P = P;
  c->have_rollout = true;
  c->traj_candidate_major = c->wave;
  return MJPCX_OK;
}

int check_rollout_args(mjpcx_ctx* c, int N, int H, int P, int interp, const double* node_times) {
  if (!c || !node_times) return fail(c, MJPCX_EINVAL, "null argument");
  if (N < 1 || H < 1 || P < 1) return fail(c, MJPCX_EINVAL, "num_candidates, horizon and num_nodes must be >= 1");
  if (interp < 0 || interp > 2) return fail(c, MJPCX_EINVAL, "unknown interpolation");
  for (int p = 1; p < P; p++)
    if (!(node_times[p] > node_times[p - 1])) return fail(c, MJPCX_EINVAL, "node_times must be strictly increasing");
  const size_t shmem = c->wave ? 0 : ((size_t)P * c->nu * 64 + P) * esize(c);
  if (shmem > 64 * 1024) return fail(c, MJPCX_EUNSUPPORTED, "num_nodes * nu too large for the LDS spline stage");
  if (hipSetDevice(c->device) != hipSuccess) return fail(c, MJPCX_EDEVICE, "hipSetDevice failed");
  return MJPCX_OK;
}

}  // namespace

// ===================================================================== C ABI
extern "C" {

const char* mjpcx_error_string(int code) {
  switch (code) {
    case MJPCX_OK: return "ok";
    case MJPCX_EINVAL: return "invalid argument";
    case MJPCX_EUNSUPPORTED: return "model or task not supported by any device kernel";
    case MJPCX_EDEVICE: return "HIP runtime error";
    case MJPCX_ENOMEM: return "out of memory";
    case MJPCX_ESTATE: return "call out of order";
    default: return "unknown error";
  }
}
const char* mjpcx_last_error(const mjpcx_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
const char* mjpcx_kernel_name(const mjpcx_ctx* ctx) { return ctx && ctx->kernel ? ctx->kernel->name : ""; }

static thread_local std::string g_create_error;
// the environment is read when a context is created, never on the launch path
static int env_stamp_step() { const char* e = getenv("MJPCX_STAMPS"); return e ? std::atoi(e) : -1; }
const char* mjpcx_create_error(void) { return g_create_error.c_str(); }

int mjpcx_create(const mjpcx_model* m, const mjpcx_task* t, int device, int precision, mjpcx_ctx** out) {
  g_create_error.clear();
  auto bad = [&](int code, const std::string& msg) { g_create_error = msg; return code; };
  if (!m || !t || !out) return bad(MJPCX_EINVAL, "null argument");
  *out = nullptr;
  if (precision != 64 && precision != 32) return bad(MJPCX_EINVAL, "precision must be 64 or 32");
  // ---- features the device kernels cover today
  if (m->na != 0) return bad(MJPCX_EUNSUPPORTED, "actuator activations (na > 0) unsupported");
  mjpcx_model m_euler;
  if (m->integrator == MJPCX_INT_IMPLICITFAST) {
    // mj_implicit with qDeriv = d qfrc_smooth / d qvel restricted to passive damping + actuator velocity terms: without the latter the
    // matrix is M + h diag(damping), mj_Euler's own (include/mjpcx.h). The context then runs the Euler path on a copy of the header.
    for (int i = 0; i < m->nu; i++)
      if (m->actuator_biasprm && m->actuator_biastype && m->actuator_biastype[i] == MJPCX_BIAS_AFFINE && m->actuator_biasprm[3 * i + 2] != 0)
        return bad(MJPCX_EUNSUPPORTED, "implicitfast with a velocity-dependent actuator (affine bias, biasprm[2] != 0): only models whose velocity-dependent smooth "
                                       "force is joint damping are integrated (as mj_Euler, which is the same update there)");
    // mj_implicit ignores mjDSBL_EULERDAMP, mj_Euler switches to explicit damping under it: the two updates differ then
    if (m->disableflags & MJPCX_DSBL_EULERDAMP)
      return bad(MJPCX_EUNSUPPORTED, "implicitfast with eulerdamp disabled: mj_implicit keeps the damping implicit, the Euler path this context would run does not");
    m_euler = *m;
    m_euler.integrator = MJPCX_INT_EULER;
    m = &m_euler;
  }
  if (m->integrator != MJPCX_INT_EULER && m->integrator != MJPCX_INT_RK4)
    return bad(MJPCX_EUNSUPPORTED, "integrators: Euler, RK4 and (damping-only models) implicitfast; implicit is not implemented");
  // ---- wavefront-per-candidate family: free/ball joints, friction loss, contacts
  bool needs_wave = false;
  for (int j = 0; j < m->njnt; j++) needs_wave |= m->jnt_type[j] == MJPCX_JNT_FREE || m->jnt_type[j] == MJPCX_JNT_BALL;
  for (int i = 0; i < m->nv; i++) needs_wave |= m->dof_frictionloss[i] > 0 && !(m->disableflags & MJPCX_DSBL_FRICTIONLOSS);
  if (!(m->disableflags & MJPCX_DSBL_CONTACT) && m->ngeom > 0) {
    bool st = false, dy = false;
    for (int g = 0; g < m->ngeom; g++) {
      if (!(m->geom_contype[g] || m->geom_conaffinity[g])) continue;
      bool moving = false;
      for (int b = m->geom_bodyid[g]; b > 0; b = m->body_parentid[b]) moving |= m->body_dofnum[b] > 0;
      (moving ? dy : st) = true;
    }
    needs_wave |= st && dy;
  }
  for (int k = 0; k < m->ntendon; k++) needs_wave |= m->tendon_limited[k] != 0;
  if (needs_wave) {

    for (int j = 0; j < m->njnt; j++)
      if (m->jnt_limited[j] && (m->jnt_type[j] == MJPCX_JNT_FREE || m->jnt_type[j] == MJPCX_JNT_BALL))
        return bad(MJPCX_EUNSUPPORTED, "limits on free/ball joints are not implemented");
    if (t->num_trace * 3 > 64 || m->nu > 64 || t->num_term > 64 || t->num_residual > kWaveMaxEfc * m->nv)
      return bad(MJPCX_EUNSUPPORTED, "task exceeds the wave kernel capacity");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bad(MJPCX_EDEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return bad(MJPCX_EINVAL, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) return bad(MJPCX_EDEVICE, "hipSetDevice failed");
    mjpcx_ctx* c = new (std::nothrow) mjpcx_ctx();
    if (!c) return bad(MJPCX_ENOMEM, "host allocation failed");
    c->device = device; c->precision = precision; c->stamp_step = env_stamp_step(); c->no_tree = getenv("MJPCX_NO_TREE") != nullptr; c->no_lds_model = getenv("MJPCX_NO_LDS_MODEL") != nullptr;
    if (precision == 32) c->max_waves = 12;
    if (const char* e = getenv("MJPCX_TREE_WAVES")) c->max_waves = std::max(1, std::min(precision == 32 ? 12 : 8, std::atoi(e)));
    if (const char* e = getenv("MJPCX_TREE_MODE")) c->tree_mode = std::atoi(e);
    c->no_second_pass = getenv("MJPCX_TREE_ONE_PASS") != nullptr;
    c->no_cone_slabs = getenv("MJPCX_TREE_NO_SLABS") != nullptr;
    c->kernel = &kWaveEntry; c->wave = true;
    c->nq = m->nq; c->nv = m->nv; c->nu = m->nu; c->na = m->na; c->nmocap = m->nmocap; c->nsite_model = m->nsite;
    c->nr = t->num_residual; c->nterm = t->num_term; c->ntrace = t->num_trace; c->nparam = t->num_parameter;
    c->num_norm_parameter.assign(t->num_norm_parameter, t->num_norm_parameter + t->num_term);
    c->dim_norm_residual.assign(t->dim_norm_residual, t->dim_norm_residual + t->num_term);
    c->ctrllimited.assign(m->actuator_ctrllimited, m->actuator_ctrllimited + m->nu);
    c->ctrlrange.assign(m->actuator_ctrlrange, m->actuator_ctrlrange + 2 * m->nu);
    const std::string err = c->wh.build(m, t, precision == 32);
    if (!err.empty()) { mjpcx_destroy(c); return bad(MJPCX_EUNSUPPORTED, err); }
    if (c->wh.tree_ok && !c->no_tree && !c->no_lds_model && lds_model_matches<TreeCfgA1>(m, t, c->wh)) {
      // registered model: the LDS image of its hot arrays (lds_model.h), in the context's precision
      std::vector<unsigned char> img = precision == 64 ? lds_model_image<TreeCfgA1, double>(m, t, c->wh) : lds_model_image<TreeCfgA1, float>(m, t, c->wh);
      void** slot = precision == 64 ? &c->wh.dev_image : &c->wh.dev_image32;
      if (hipMalloc(slot, img.size()) != hipSuccess || hipMemcpy(*slot, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) {
        mjpcx_destroy(c);
        return bad(MJPCX_ENOMEM, "upload of the model's LDS image failed");
      }
      c->wh.registered = 0;
      c->kernel = &kTreeEntryA1;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cu = prop.multiProcessorCount;
      c->quad_stats = getenv("MJPCX_QUAD_STATS") != nullptr;
      c->quad_stamps = getenv("MJPCX_QUAD_STAMPS") != nullptr;
      c->quad_no_fallback = getenv("MJPCX_QUAD_NO_FALLBACK") != nullptr;
      c->no_quad_feedback = getenv("MJPCX_NO_QUAD_FEEDBACK") != nullptr;
      if (const char* e = getenv("MJPCX_QUAD_CON_CAP")) c->quad_con_cap = std::atoi(e);
      if (const char* e = getenv("MJPCX_QUAD_MIN_N")) c->quad_min_n = std::atoi(e);
      if (const char* e = getenv("MJPCX_QUAD_CPW")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) c->quad_cpw = v; }
      if (precision == 64 && !getenv("MJPCX_NO_QUAD")) {
        // the quad kernel family (four lanes per candidate): models of the legged class quad_build accepts
        std::vector<unsigned char> hq, ht;
        c->quad_why = quad::build_images(m, t, hq, ht);
        if (c->quad_why.empty()) {
          if (c->d_qmodel.reserve(hq.size()) != hipSuccess || c->d_qtab.reserve(ht.size()) != hipSuccess || c->d_qstats.reserve(32) != hipSuccess ||
              c->d_qstamps.reserve(512) != hipSuccess || hipMemset(c->d_qstats.p, 0, 32) != hipSuccess ||
              hipMemcpy(c->d_qmodel.p, hq.data(), hq.size(), hipMemcpyHostToDevice) != hipSuccess ||
              hipMemcpy(c->d_qtab.p, ht.data(), ht.size(), hipMemcpyHostToDevice) != hipSuccess) {
            mjpcx_destroy(c);
            return bad(MJPCX_ENOMEM, "upload of the quad kernel's model failed");
          }
          c->quad_ok = true;
          c->kernel = &kQuadEntryA1;
          for (int k = 0; k < 7; k++) c->quad_ids[k] = t->residual_int[1 + k];
        }
      }
    }
    else if (c->wh.tree_ok && !c->no_tree && !c->no_lds_model && lds_model_matches<TreeCfgHumanoid>(m, t, c->wh)) {
      std::vector<unsigned char> img = precision == 64 ? lds_model_image<TreeCfgHumanoid, double>(m, t, c->wh) : lds_model_image<TreeCfgHumanoid, float>(m, t, c->wh);
      void** slot = precision == 64 ? &c->wh.dev_image : &c->wh.dev_image32;
      if (hipMalloc(slot, img.size()) != hipSuccess || hipMemcpy(*slot, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) {
        mjpcx_destroy(c);
        return bad(MJPCX_ENOMEM, "upload of the model's LDS image failed");
      }
      c->wh.registered = 1;
      c->kernel = &kTreeEntryHumanoid;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cu = prop.multiProcessorCount;
      c->quad_stats = getenv("MJPCX_QUAD_STATS") != nullptr;
      c->limb_no_fallback = getenv("MJPCX_LIMB_NO_FALLBACK") != nullptr;
      if (const char* e = getenv("MJPCX_LIMB_MIN_N")) c->limb_min_n = std::atoi(e);
      if (const char* e = getenv("MJPCX_LIMB_CPW")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) c->limb_cpw = v; }
      if (!getenv("MJPCX_NO_LIMB")) {   // (both precisions: the fp64 instantiation spills more but is still ahead of the tree kernel, 78 against 95 ms on configs[3]'s share)
        // the limb kernel family (four lanes per candidate, one per limb): models of the class limb_build accepts
        std::vector<unsigned char> h32, h64;
        c->limb_why = limb::build_images(m, t, h32, h64);
        if (c->limb_why.empty()) {
          const std::vector<unsigned char>& img = precision == 64 ? h64 : h32;
          if (c->d_limb.reserve(img.size()) != hipSuccess || c->d_qstats.reserve(32) != hipSuccess || hipMemset(c->d_qstats.p, 0, 32) != hipSuccess ||
              hipMemcpy(c->d_limb.p, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) {
            mjpcx_destroy(c);
            return bad(MJPCX_ENOMEM, "upload of the limb kernel's model failed");
          }
          c->limb_ok = true;
          c->kernel = &kLimbEntryHumanoid;
          for (int k = 0; k < 32; k++) c->limb_ids[k] = t->residual_int[2 + k];
        }
      }
    }
    set_norm_params(c, t->norm_parameter);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { mjpcx_destroy(c); return bad(MJPCX_EDEVICE, "hipStreamCreate failed"); }
    g_create_error = c->wh.warning;  // (empty, or what this context does NOT model of the caller's mjModel)
    if (!c->wh.warning.empty() && getenv("MJPCX_STRICT_PAIRS")) {  // strict callers (tests, parity runs) refuse a model whose physics is not fully reproduced
      const std::string why = c->wh.warning;
      mjpcx_destroy(c);
      return bad(MJPCX_EUNSUPPORTED, why.c_str());
    }
    // a model the Jacobian-free path could serve, but with no registered configuration (dimensions in tree_registry.h): it runs -- on the
    // generic kernel, at a fraction of the registered kernel's rate. Say so instead of leaving the caller to find out from the clock.
    if (c->wh.tree_ok && c->wh.registered < 0 && !c->no_tree && !c->no_lds_model)
    {
      char dims[320];
      std::snprintf(dims, sizeof dims, " (a configuration for it would read NQ = %d, NV = %d, NU = %d, NB = %d, NJ = %d, NS = %d, NG = %d, NKEY = %d, NMOCAP = %d, "
                    "NSG = %d, NDG = %d, NRAY = %d, NR = %d, NTERM = %d, NTRACE = %d, NT = %d)", m->nq, m->nv, m->nu, c->wh.m.nbody, m->njnt, c->wh.m.nsite, m->ngeom,
                    m->nkey, m->nmocap, (int)c->wh.h_static_geom.size(), (int)c->wh.h_dynamic_geom.size(), (int)c->wh.h_ray_geom.size(), t->num_residual,
                    t->num_term, t->num_trace, m->ntendon);
      g_create_error += std::string(g_create_error.empty() ? "" : "; ") + "note: no registered kernel configuration for this model's dimensions (tree_registry.h): "
                        "served by the generic wavefront-per-candidate kernel" + dims;
    }
    *out = c;
    return MJPCX_OK;
  }
  if (m->nq != m->nv || m->njnt != m->nv) return bad(MJPCX_EUNSUPPORTED, "only slide/hinge joints are implemented (nq == nv == njnt)");
  if (m->ntendon > 0) return bad(MJPCX_EUNSUPPORTED, "tendons are only implemented in the wavefront-per-candidate kernel (limited fixed tendons)");
  if (m->nbody > kLaneMaxBody || m->nv > kLaneMaxDof || m->nu > kLaneMaxAct || m->nsite > kLaneMaxSite ||
      m->nmocap > kLaneMaxMocap || t->num_term > kLaneMaxTerm || t->num_parameter > kLaneMaxParam)
    return bad(MJPCX_EUNSUPPORTED, "model exceeds the small-model kernel capacity");
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_type[j] != MJPCX_JNT_SLIDE && m->jnt_type[j] != MJPCX_JNT_HINGE)
      return bad(MJPCX_EUNSUPPORTED, "only slide/hinge joints are implemented");
    if (m->jnt_dofadr[j] != j || m->jnt_qposadr[j] != j) return bad(MJPCX_EINVAL, "joint/dof addressing is not 1:1");
    if (m->dof_frictionloss[j] > 0) return bad(MJPCX_EUNSUPPORTED, "joint frictionloss unsupported");
    if (m->jnt_limited[j] && !(m->jnt_range[2 * j + 1] - m->jnt_range[2 * j] > 2 * m->jnt_margin[j]))
      return bad(MJPCX_EUNSUPPORTED, "joint range narrower than twice its margin");
  }
  if (!(m->disableflags & MJPCX_DSBL_CONTACT)) return bad(MJPCX_EUNSUPPORTED, "contacts are not implemented: the model must disable them");
  for (int u = 0; u < m->nu; u++)
    if (m->actuator_trnid[u] < 0 || m->actuator_trnid[u] >= m->njnt) return bad(MJPCX_EINVAL, "actuator transmission out of range");

  // ---- static-topology key of the runtime model
  TopoKey tk{m->nbody, m->nv, m->nu, m->nsite, m->nmocap, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < m->nbody; b++) {
    tk.parent |= (uint64_t)(m->body_parentid[b] & 15) << (4 * b);
    tk.mocap |= (uint64_t)((m->body_mocapid[b] < 0 ? 15 : m->body_mocapid[b]) & 15) << (4 * b);
  }
  for (int j = 0; j < m->njnt; j++) {
    tk.jtype |= (uint64_t)(m->jnt_type[j] & 3) << (2 * j);
    tk.jbody |= (uint64_t)(m->jnt_bodyid[j] & 15) << (4 * j);
    tk.jlimited |= (uint64_t)(m->jnt_limited[j] ? 1 : 0) << j;
  }
  for (int u = 0; u < m->nu; u++) tk.actj |= (uint64_t)(m->actuator_trnid[u] & 15) << (4 * u);
  for (int s = 0; s < m->nsite; s++) tk.siteb |= (uint64_t)(m->site_bodyid[s] & 15) << (4 * s);
  TaskKey kk{t->residual_id, t->num_residual, t->num_term, t->num_trace, 0, 0};
  for (int k = 0; k < t->num_term; k++) kk.termdim |= (uint64_t)(t->dim_norm_residual[k] & 15) << (4 * k);
  for (int k = 0; k < t->num_trace; k++) kk.tracesite |= (uint64_t)(t->trace_site[k] & 15) << (4 * k);
  const KernelEntry* entry = nullptr;
  LaneModel<double> probe;
  fill_model(probe, m);
  for (const KernelEntry& e : kKernels)
    if (e.topo == tk && e.task == kk && (!e.static_model || (!getenv("MJPCX_NO_STATIC") && same_model(*e.static_model, probe)))) {
      entry = &e;
      break;
    }
  if (!entry) {
    char buf[512];
    std::snprintf(buf, sizeof buf,
                  "no rollout kernel is instantiated for this model/task topology: Topo<%d,%d,%d,%d,%d,0x%llx,0x%llx,0x%llx,"
                  "0x%llx,0x%llx,0x%llx,0x%llx> TaskTopo<%d,%d,%d,0x%llx,%d,0x%llx> (add it to kKernels in csrc/mjpcx.hip)",
                  tk.nb, tk.nv, tk.nu, tk.nsite, tk.nmocap, (unsigned long long)tk.parent, (unsigned long long)tk.mocap,
                  (unsigned long long)tk.jtype, (unsigned long long)tk.jbody, (unsigned long long)tk.jlimited,
                  (unsigned long long)tk.actj, (unsigned long long)tk.siteb, kk.rid, kk.nr, kk.nterm,
                  (unsigned long long)kk.termdim, kk.ntrace, (unsigned long long)kk.tracesite);
    return bad(MJPCX_EUNSUPPORTED, buf);
  }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bad(MJPCX_EDEVICE, "no HIP device available");
  if (device < 0 || device >= ndev) return bad(MJPCX_EINVAL, "device index out of range");
  if (hipSetDevice(device) != hipSuccess) return bad(MJPCX_EDEVICE, "hipSetDevice failed");

  mjpcx_ctx* c = new (std::nothrow) mjpcx_ctx();
  if (!c) return bad(MJPCX_ENOMEM, "host allocation failed");
  c->device = device; c->precision = precision; c->stamp_step = env_stamp_step(); c->kernel = entry;
  c->nq = m->nq; c->nv = m->nv; c->nu = m->nu; c->na = m->na; c->nmocap = m->nmocap;
  c->nr = t->num_residual; c->nterm = t->num_term; c->ntrace = t->num_trace; c->nparam = t->num_parameter;
  c->num_norm_parameter.assign(t->num_norm_parameter, t->num_norm_parameter + t->num_term);
  c->dim_norm_residual.assign(t->dim_norm_residual, t->dim_norm_residual + t->num_term);
  c->ctrllimited.assign(m->actuator_ctrllimited, m->actuator_ctrllimited + m->nu);
  c->ctrlrange.assign(m->actuator_ctrlrange, m->actuator_ctrlrange + 2 * m->nu);
  fill_model(c->hm64, m);
  fill_model(c->hm32, m);
  std::memset(&c->ht64, 0, sizeof c->ht64);
  for (int k = 0; k < t->num_term; k++) { c->ht64.norm[k] = t->norm[k]; c->ht64.weight[k] = t->weight[k]; }
  set_norm_params(c, t->norm_parameter);
  for (int k = 0; k < t->num_parameter; k++) c->ht64.parameters[k] = t->parameters[k];
  c->ht64.risk = t->risk;
  for (int j = 0; j < m->nq; j++) c->ht64.qpos[j] = m->qpos0[j];
  for (int b = 0; b < m->nbody; b++)
    if (m->body_mocapid[b] >= 0) {
      for (int k = 0; k < 3; k++) c->ht64.mocap_pos[m->body_mocapid[b]][k] = m->body_pos[3 * b + k];
      for (int k = 0; k < 4; k++) c->ht64.mocap_quat[m->body_mocapid[b]][k] = m->body_quat[4 * b + k];
    }
  auto cleanup = [&](int code, const std::string& msg) { mjpcx_destroy(c); return bad(code, msg); };
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return cleanup(MJPCX_EDEVICE, "hipStreamCreate failed");
  *out = c;
  return MJPCX_OK;
}

void mjpcx_destroy(mjpcx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& ev : c->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (auto& ev : c->events_main) (void)hipEventDestroy(ev);
  for (auto& sl : c->slots) {
    if (sl.host) (void)hipHostFree(sl.host);
    if (sl.done) (void)hipEventDestroy(sl.done);
    sl.dev.release();
  }
  if (c->best_host) (void)hipHostFree(c->best_host);
  if (c->h_qstats) (void)hipHostFree(c->h_qstats);
  (void)mjpcx_comm_destroy(c);
  c->wh.release();
  DevBuf* bufs[] = {&c->d_nodes, &c->d_in_nodes, &c->d_ilqg, &c->d_ilqg_out, &c->d_wblob, &c->d_work, &c->d_ovf, &c->d_qmodel, &c->d_qtab, &c->d_qstats, &c->d_qstamps, &c->d_qwave, &c->d_qovf, &c->d_qclass, &c->d_limb, &c->d_comm_send, &c->d_comm_recv,
                    &c->d_states, &c->d_actions, &c->d_times, &c->d_residual, &c->d_costs, &c->d_trace, &c->d_ret,
                    &c->d_fail, &c->d_sort, &c->d_stage};
  for (DevBuf* b : bufs) b->release();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int mjpcx_set_state(mjpcx_ctx* c, const double* state, double time, const double* mocap, const double* userdata) {
  if (!c || !state) return fail(c, MJPCX_EINVAL, "null argument");
  (void)userdata;  // nuserdata == 0 for every supported model
  if (c->wave) {
    std::memcpy(c->wh.state.data(), state, sizeof(double) * (c->nq + c->nv));
    c->wh.time = time;
    if (mocap) std::memcpy(c->wh.mocap.data(), mocap, sizeof(double) * 7 * c->nmocap);
    return MJPCX_OK;
  }
  for (int j = 0; j < c->nq; j++) c->ht64.qpos[j] = state[j];
  for (int j = 0; j < c->nv; j++) c->ht64.qvel[j] = state[c->nq + j];
  c->ht64.time = time;
  if (mocap)
    for (int i = 0; i < c->nmocap; i++) {  // trajectory.cc:121-124
      for (int k = 0; k < 3; k++) c->ht64.mocap_pos[i][k] = mocap[7 * i + k];
      for (int k = 0; k < 4; k++) c->ht64.mocap_quat[i][k] = mocap[7 * i + 3 + k];
    }
  return MJPCX_OK;
}

int mjpcx_set_task_params(mjpcx_ctx* c, const double* weight, const double* norm_parameter,
                          const double* parameters, double risk) {
  if (!c) return MJPCX_EINVAL;
  if (c->wave) {
    if (weight) c->wh.weight.assign(weight, weight + c->nterm);
    if (norm_parameter) set_norm_params(c, norm_parameter);
    if (parameters) c->wh.parameters.assign(parameters, parameters + c->nparam);
    c->wh.risk = risk;
    return MJPCX_OK;
  }
  if (weight) for (int k = 0; k < c->nterm; k++) c->ht64.weight[k] = weight[k];
  if (norm_parameter) set_norm_params(c, norm_parameter);
  if (parameters) for (int k = 0; k < c->nparam; k++) c->ht64.parameters[k] = parameters[k];
  c->ht64.risk = risk;
  return MJPCX_OK;
}

int mjpcx_set_residual_state(mjpcx_ctx* c, const int32_t* residual_int, const double* residual_real) {
  if (!c) return MJPCX_EINVAL;
  if (!c->wave) return (residual_int || residual_real) ? fail(c, MJPCX_EUNSUPPORTED, "this task's residual has no frozen state") : MJPCX_OK;
  if (residual_int) c->wh.residual_int.assign(residual_int, residual_int + c->wh.t.nri);
  if (residual_int && c->quad_ok)  // the quad model bakes the ids Task::Reset resolves; a caller that changes them gets the generic path
    for (int k = 0; k < 7; k++) if (residual_int[1 + k] != c->quad_ids[k]) { c->quad_ok = false; c->kernel = &kTreeEntryA1; }
  if (residual_int && c->limb_ok)  // likewise the limb model: the tracking sites and mocap bodies (the motion's first / last key may change)
    for (int k = 0; k < 32; k++) if (residual_int[2 + k] != c->limb_ids[k]) { c->limb_ok = false; c->kernel = &kTreeEntryHumanoid; }
  if (residual_real) c->wh.residual_real.assign(residual_real, residual_real + c->wh.t.nrr);
  return MJPCX_OK;
}

int mjpcx_rollout_splines(mjpcx_ctx* c, int N, int H, int P, int interp, const double* node_times,
                          const double* node_values) {
  int rc = check_rollout_args(c, N, H, P, interp, node_times);
  if (rc != MJPCX_OK) return rc;
  if (!node_values) return fail(c, MJPCX_EINVAL, "null node_values");
  return c->precision == 64 ? do_rollout<double>(c, N, H, P, interp, node_times, node_values, nullptr, nullptr)
                            : do_rollout<float>(c, N, H, P, interp, node_times, node_values, nullptr, nullptr);
}

int mjpcx_rollout_splines_noisy(mjpcx_ctx* c, int N, int H, int P, int interp, const double* node_times, const double* node_values,
                                double xfrc_std, double xfrc_rate, uint64_t seed, int candidate_offset) {
  int rc = check_rollout_args(c, N, H, P, interp, node_times);
  if (rc != MJPCX_OK) return rc;
  if (!node_values) return fail(c, MJPCX_EINVAL, "null node_values");
  if (!(xfrc_std >= 0) || !(xfrc_rate > 0)) return fail(c, MJPCX_EINVAL, "xfrc_std must be >= 0 and xfrc_rate > 0");
  c->xfrc_std = xfrc_std; c->xfrc_rate = xfrc_rate; c->xfrc_seed = seed; c->xfrc_offset = candidate_offset;
  rc = c->precision == 64 ? do_rollout<double>(c, N, H, P, interp, node_times, node_values, nullptr, nullptr)
                          : do_rollout<float>(c, N, H, P, interp, node_times, node_values, nullptr, nullptr);
  c->xfrc_std = 0;
  return rc;
}

int mjpcx_rollout_noise(mjpcx_ctx* c, int N, int H, int P, int interp, const double* node_times,
                        const double* nominal, const mjpcx_noise_spec* ns) {
  int rc = check_rollout_args(c, N, H, P, interp, node_times);
  if (rc != MJPCX_OK) return rc;
  if (!nominal || !ns) return fail(c, MJPCX_EINVAL, "null argument");
  if (ns->mode != MJPCX_NOISE_SAMPLING && ns->mode != MJPCX_NOISE_CROSS_ENTROPY) return fail(c, MJPCX_EINVAL, "unknown noise mode");
  return c->precision == 64 ? do_rollout<double>(c, N, H, P, interp, node_times, nullptr, nominal, ns)
                            : do_rollout<float>(c, N, H, P, interp, node_times, nullptr, nominal, ns);
}

int mjpcx_sync(mjpcx_ctx* c) {
  if (!c) return MJPCX_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}

int mjpcx_get_returns(mjpcx_ctx* c, double* total_return, int32_t* failure) {
  if (!c) return MJPCX_EINVAL;
  if (!c->have_rollout) return fail(c, MJPCX_ESTATE, "no rollout has been run");
  HIPCHK(c, hipSetDevice(c->device));
  if (total_return) HIPCHK(c, hipMemcpyAsync(total_return, c->d_ret.p, (size_t)c->N * 8, hipMemcpyDeviceToHost, c->stream));
  if (failure) HIPCHK(c, hipMemcpyAsync(failure, c->d_fail.p, (size_t)c->N * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}

int mjpcx_get_return_at(mjpcx_ctx* c, int cand, double* total_return, int32_t* failure) {
  if (!c) return MJPCX_EINVAL;
  if (!c->have_rollout) return fail(c, MJPCX_ESTATE, "no rollout has been run");
  if (cand < 0 || cand >= c->N) return fail(c, MJPCX_EINVAL, "candidate out of range");
  HIPCHK(c, hipSetDevice(c->device));
  if (total_return) HIPCHK(c, hipMemcpyAsync(total_return, (const double*)c->d_ret.p + cand, 8, hipMemcpyDeviceToHost, c->stream));
  if (failure) HIPCHK(c, hipMemcpyAsync(failure, (const int*)c->d_fail.p + cand, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}

int mjpcx_best(mjpcx_ctx* c, int ref_candidate, int32_t* index, double* best_return, double* ref_return,
               double* spline_values) {
  if (!c || !index) return fail(c, MJPCX_EINVAL, "null argument");
  if (!c->have_rollout) return fail(c, MJPCX_ESTATE, "no rollout has been run");
  HIPCHK(c, hipSetDevice(c->device));
  const int np = c->P * c->nu;
  const size_t bytes = sizeof(BestRecord) + (size_t)np * 8;
  if (bytes > c->best_cap) {
    if (c->best_host) (void)hipHostFree(c->best_host);
    c->best_host = nullptr; c->best_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->best_host, bytes, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer(&c->best_dev, c->best_host, 0));
    c->best_cap = bytes;
  }
  if (c->precision == 64)
    hipLaunchKernelGGL((best_kernel<double>), dim3(1), dim3(1024), 0, c->stream, (const double*)c->d_ret.p,
                       (const int*)c->d_fail.p, (const double*)c->d_nodes.p, c->N, np, ref_candidate, (BestRecord*)c->best_dev);
  else
    hipLaunchKernelGGL((best_kernel<float>), dim3(1), dim3(1024), 0, c->stream, (const double*)c->d_ret.p,
                       (const int*)c->d_fail.p, (const float*)c->d_nodes.p, c->N, np, ref_candidate, (BestRecord*)c->best_dev);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const BestRecord* r = (const BestRecord*)c->best_host;
  *index = r->best_index;
  if (best_return) *best_return = r->best_return;
  if (ref_return) *ref_return = r->ref_return;
  if (spline_values) std::memcpy(spline_values, r->spline, (size_t)np * 8);
  return MJPCX_OK;
}

int mjpcx_topk(mjpcx_ctx* c, int k, int32_t* index, double* total_return) {
  if (!c || !index) return fail(c, MJPCX_EINVAL, "null argument");
  if (!c->have_rollout) return fail(c, MJPCX_ESTATE, "no rollout has been run");
  if (k < 1 || k > c->N) return fail(c, MJPCX_EINVAL, "k out of range");
  HIPCHK(c, hipSetDevice(c->device));
  int n2 = 1;
  while (n2 < c->N) n2 <<= 1;
  HIPCHK(c, c->d_sort.reserve((size_t)n2 * sizeof(RetIdx)));
  if (k == 1) {
    hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(1024), 0, c->stream, (const double*)c->d_ret.p, c->N, (RetIdx*)c->d_sort.p);
  } else {
    hipLaunchKernelGGL(sort_kernel, dim3(1), dim3(1024), 0, c->stream, (const double*)c->d_ret.p, c->N, n2, (RetIdx*)c->d_sort.p);
  }
  HIPCHK(c, hipGetLastError());
  std::vector<RetIdx> h(k);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->d_sort.p, (size_t)k * sizeof(RetIdx), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < k; i++) {
    index[i] = h[i].i;
    if (total_return) total_return[i] = h[i].r;
  }
  return MJPCX_OK;
}

int mjpcx_elite_moments(mjpcx_ctx* c, int n, const int32_t* candidates, const double* mean, double* out,
                        double* sum_return) {
  if (!c || !candidates || !out) return fail(c, MJPCX_EINVAL, "null argument");
  if (!c->have_rollout) return fail(c, MJPCX_ESTATE, "no rollout has been run");
  if (n < 0) return fail(c, MJPCX_EINVAL, "negative count");
  for (int i = 0; i < n; i++)
    if (candidates[i] < 0 || candidates[i] >= c->N) return fail(c, MJPCX_EINVAL, "candidate out of range");
  HIPCHK(c, hipSetDevice(c->device));
  const int np = c->P * c->nu;
  // staging: [cand (n i32) | mean (np f64) | out (np+1 f64)]
  const size_t off_mean = (((size_t)n * 4) + 15) & ~(size_t)15;
  const size_t off_out = off_mean + (size_t)np * 8;
  const size_t bytes = off_out + (size_t)(np + 1) * 8;
  HIPCHK(c, c->d_stage.reserve(bytes));
  char* d = (char*)c->d_stage.p;
  if (n) HIPCHK(c, hipMemcpyAsync(d, candidates, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  if (mean) HIPCHK(c, hipMemcpyAsync(d + off_mean, mean, (size_t)np * 8, hipMemcpyHostToDevice, c->stream));
  const double* dmean = mean ? (const double*)(d + off_mean) : nullptr;
  if (c->precision == 64)
    hipLaunchKernelGGL((elite_moments_kernel<double>), dim3(np + 1), dim3(256), 0, c->stream, (const double*)c->d_nodes.p,
                       (const double*)c->d_ret.p, (const int*)d, n, c->N, np, dmean, (double*)(d + off_out));
  else
    hipLaunchKernelGGL((elite_moments_kernel<float>), dim3(np + 1), dim3(256), 0, c->stream, (const float*)c->d_nodes.p,
                       (const double*)c->d_ret.p, (const int*)d, n, c->N, np, dmean, (double*)(d + off_out));
  HIPCHK(c, hipGetLastError());
  std::vector<double> h(np + 1);
  HIPCHK(c, hipMemcpyAsync(h.data(), d + off_out, (size_t)(np + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(out, h.data(), (size_t)np * 8);
  if (sum_return) *sum_return = h[np];
  return MJPCX_OK;
}

int mjpcx_fetch_trajectory(mjpcx_ctx* c, int cand, mjpcx_traj_view* out) {
  if (!c || !out) return fail(c, MJPCX_EINVAL, "null argument");
  if (!c->have_rollout) return fail(c, MJPCX_ESTATE, "no rollout has been run");
  if (cand < 0 || cand >= c->N) return fail(c, MJPCX_EINVAL, "candidate out of range");
  if (out->horizon < c->H) return fail(c, MJPCX_EINVAL, "trajectory view too short");
  HIPCHK(c, hipSetDevice(c->device));
  const int H = c->H, ds = c->nq + c->nv + c->na, nu = c->nu, nr = c->nr, ntr3 = 3 * c->ntrace;
  const int row = ds + nu + 1 + nr + 1 + ntr3;
  const size_t total = (size_t)H * row;
  HIPCHK(c, c->d_stage.reserve((total + 2) * 8));
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 1024);
  if (c->precision == 64)
    hipLaunchKernelGGL((gather_traj_kernel<double>), dim3(blocks), dim3(256), 0, c->stream, (const double*)c->d_states.p,
                       (const double*)c->d_actions.p, (const double*)c->d_times.p, (const double*)c->d_residual.p,
                       (const double*)c->d_costs.p, (const double*)c->d_trace.p, c->N, H, ds, nu, nr, ntr3, cand,
                       (double*)c->d_stage.p, c->traj_candidate_major ? 1 : 0);
  else
    hipLaunchKernelGGL((gather_traj_kernel<float>), dim3(blocks), dim3(256), 0, c->stream, (const float*)c->d_states.p,
                       (const float*)c->d_actions.p, (const float*)c->d_times.p, (const float*)c->d_residual.p,
                       (const float*)c->d_costs.p, (const float*)c->d_trace.p, c->N, H, ds, nu, nr, ntr3, cand,
                       (double*)c->d_stage.p, c->traj_candidate_major ? 1 : 0);
  HIPCHK(c, hipGetLastError());
  c->h_stage.resize(total);
  double ret = 0; int32_t fl = 0;
  HIPCHK(c, hipMemcpyAsync(c->h_stage.data(), c->d_stage.p, total * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&ret, (const double*)c->d_ret.p + cand, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&fl, (const int*)c->d_fail.p + cand, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const double* s = c->h_stage.data();
  if (out->states) std::memcpy(out->states, s, sizeof(double) * H * ds);
  s += (size_t)H * ds;
  if (out->actions) std::memcpy(out->actions, s, sizeof(double) * H * nu);
  s += (size_t)H * nu;
  if (out->times) std::memcpy(out->times, s, sizeof(double) * H);
  s += H;
  if (out->residual) std::memcpy(out->residual, s, sizeof(double) * H * nr);
  s += (size_t)H * nr;
  if (out->costs) std::memcpy(out->costs, s, sizeof(double) * H);
  s += H;
  if (out->trace && ntr3) std::memcpy(out->trace, s, sizeof(double) * H * ntr3);
  out->horizon = H;
  out->total_return = ret;
  out->failure = fl;
  return MJPCX_OK;
}

int mjpcx_fetch_spline(mjpcx_ctx* c, int cand, double* node_values) {
  if (!c || !node_values) return fail(c, MJPCX_EINVAL, "null argument");
  if (!c->have_rollout) return fail(c, MJPCX_ESTATE, "no rollout has been run");
  if (cand < 0 || cand >= c->N) return fail(c, MJPCX_EINVAL, "candidate out of range");
  HIPCHK(c, hipSetDevice(c->device));
  const int np = c->P * c->nu;
  HIPCHK(c, c->d_stage.reserve((size_t)np * 8));
  if (c->precision == 64)
    hipLaunchKernelGGL((gather_spline_kernel<double>), dim3((np + 63) / 64), dim3(64), 0, c->stream,
                       (const double*)c->d_nodes.p, c->N, np, cand, (double*)c->d_stage.p);
  else
    hipLaunchKernelGGL((gather_spline_kernel<float>), dim3((np + 63) / 64), dim3(64), 0, c->stream,
                       (const float*)c->d_nodes.p, c->N, np, cand, (double*)c->d_stage.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(node_values, c->d_stage.p, (size_t)np * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}

int mjpcx_timing_reset(mjpcx_ctx* c) {
  if (!c) return MJPCX_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->events_used = 0;
  c->timing = true;
  return MJPCX_OK;
}

int mjpcx_timing_read(mjpcx_ctx* c, double* kernel_ms, int64_t* launches) {
  if (!c) return MJPCX_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double total = 0;
  for (size_t i = 0; i < c->events_used; i++) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->events[i].first, c->events[i].second));
    total += ms;
  }
  if (kernel_ms) *kernel_ms = total;
  if (launches) *launches = (int64_t)c->events_used;
  c->timing = false;
  return MJPCX_OK;
}

int mjpcx_timing_read_main(mjpcx_ctx* c, double* main_kernel_ms, int64_t* launches) {
  if (!c) return MJPCX_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double total = 0;
  for (size_t i = 0; i < c->events_used; i++) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->events[i].first, c->events_main[i]));
    total += ms;
  }
  if (main_kernel_ms) *main_kernel_ms = total;
  if (launches) *launches = (int64_t)c->events_used;
  return MJPCX_OK;
}

int mjpcx_quad_stats(mjpcx_ctx* c, int32_t* handed_on) {
  if (!c || !handed_on) return MJPCX_EINVAL;
  for (int k = 0; k < 8; k++) handed_on[k] = 0;
  if (!c->quad_ok && !c->limb_ok) return MJPCX_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(handed_on, c->d_qstats.p, 32, hipMemcpyDeviceToHost));
  return MJPCX_OK;
}

int64_t mjpcx_algorithmic_bytes(const mjpcx_ctx* c, int H, int P) {
  if (!c) return 0;
  const int64_t w = c->precision == 64 ? 8 : 4;
  const int64_t ds = c->nq + c->nv + c->na;
  return w * ((int64_t)H * (ds + c->nu + 1 + c->nr + 3 * c->ntrace + 1) + (int64_t)P * c->nu + P + 2);
}

int mjpcx_device_buffer(mjpcx_ctx* c, int which, void** ptr, size_t* bytes) {
  if (!c || !ptr) return MJPCX_EINVAL;
  if (!c->have_rollout) return fail(c, MJPCX_ESTATE, "no rollout has been run");
  if (which == 0) { *ptr = c->d_ret.p; if (bytes) *bytes = (size_t)c->N * 8; }
  else if (which == 1) { *ptr = c->d_fail.p; if (bytes) *bytes = (size_t)c->N * 4; }
  else return fail(c, MJPCX_EINVAL, "unknown buffer");
  return MJPCX_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ iLQG entry points
namespace {
// uploads a list of host fp64 arrays into one device buffer, converting to T; returns device pointers
template <typename T>
int upload_arrays(mjpcx_ctx* c, DevBuf& buf, const std::vector<std::pair<const double*, size_t>>& arrays, std::vector<T*>* out) {
  size_t total = 0;
  for (auto& a : arrays) total += (a.second + 1) & ~(size_t)1;
  HIPCHK(c, buf.reserve(total * sizeof(T)));
  std::vector<T> host(total);
  size_t off = 0;
  out->clear();
  for (auto& a : arrays) {
    for (size_t i = 0; i < a.second; i++) host[off + i] = (T)a.first[i];
    out->push_back((T*)buf.p + off);
    off += (a.second + 1) & ~(size_t)1;
  }
  HIPCHK(c, hipMemcpyAsync(buf.p, host.data(), total * sizeof(T), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}

template <typename T>
int do_feedback(mjpcx_ctx* c, int N, int H, int mode, int representation, int use_state, int Tn, const double* times,
                const double* states, const double* actions, const double* gains, const double* improvement,
                const double* alpha) {
  int rc;
  if ((rc = reserve_rollout(c, N, H, 1)) != MJPCX_OK) return rc;
  const size_t ds = c->nq + c->nv, ndx = 2 * (size_t)c->nv, nu = c->nu;
  std::vector<T*> d;
  if ((rc = upload_arrays<T>(c, c->d_ilqg, {{times, (size_t)Tn}, {states, Tn * ds}, {actions, Tn * nu},
                                            {gains, Tn * nu * ndx}, {improvement, Tn * nu}, {alpha, (size_t)N}}, &d)) != MJPCX_OK)
    return rc;
  RolloutArgs<T> a{};
  a.N = N; a.H = H; a.P = 0; a.interp = 0;
  a.nodes = (T*)c->d_nodes.p; a.noise.mode = -1;
  a.states = (T*)c->d_states.p; a.actions = (T*)c->d_actions.p; a.times = (T*)c->d_times.p;
  a.residual = (T*)c->d_residual.p; a.costs = (T*)c->d_costs.p; a.trace = (T*)c->d_trace.p;
  a.total_return = (double*)c->d_ret.p; a.failure = (int*)c->d_fail.p;
  FeedbackArgs<T> fb{d[0], d[1], d[2], d[3], d[4], d[5], Tn, mode, representation, use_state};
  hipError_t le;
  if constexpr (sizeof(T) == 8) le = c->kernel->feedback64(c->hm64, c->ht64, a, fb, c->stream);
  else { convert_task(c->ht32, c->ht64); le = c->kernel->feedback32(c->hm32, c->ht32, a, fb, c->stream); }
  if (le != hipSuccess) return fail(c, MJPCX_EDEVICE, std::string("feedback kernel launch: ") + hipGetErrorString(le));
  c->N = N; c->H = H; c->P = 0;
  c->have_rollout = true;
  c->traj_candidate_major = false;
  return MJPCX_OK;
}

template <typename T>
int do_transition_fd(mjpcx_ctx* c, int Tn, const double* times, const double* states, const double* actions, double eps,
                     int centered, double* A, double* B, double* C, double* D) {
  int rc;
  const size_t ds = c->nq + c->nv, ndx = 2 * (size_t)c->nv, nu = c->nu, nr = c->nr;
  const size_t nc = 1 + 2 * (ndx + nu);
  std::vector<T*> d;
  std::vector<double> cr(c->ctrlrange);
  if ((rc = upload_arrays<T>(c, c->d_ilqg, {{times, (size_t)Tn}, {states, Tn * ds}, {actions, Tn * nu}, {cr.data(), 2 * nu}}, &d)) != MJPCX_OK)
    return rc;
  // outputs: next [Tn][nc][ndx] (T), sensor [Tn][nc][nr] (T), ctrllimited (int), then A,B,C,D (f64)
  const size_t off_sensor = (Tn * nc * ndx * sizeof(T) + 15) & ~(size_t)15;
  const size_t off_lim = (off_sensor + Tn * nc * nr * sizeof(T) + 15) & ~(size_t)15;
  const size_t off_A = (off_lim + nu * sizeof(int) + 15) & ~(size_t)15;
  const size_t nA = Tn * ndx * ndx, nB = Tn * ndx * nu, nC = Tn * nr * ndx, nD = Tn * nr * nu;
  HIPCHK(c, c->d_ilqg_out.reserve(off_A + (nA + nB + nC + nD) * 8));
  char* base = (char*)c->d_ilqg_out.p;
  HIPCHK(c, hipMemcpyAsync(base + off_lim, c->ctrllimited.data(), nu * sizeof(int), hipMemcpyHostToDevice, c->stream));
  FdArgs<T> f{d[0], d[1], d[2], Tn, (T)eps, (T*)base, (T*)(base + off_sensor)};
  hipError_t le;
  if constexpr (sizeof(T) == 8) le = c->kernel->fd64(c->hm64, c->ht64, f, c->stream);
  else { convert_task(c->ht32, c->ht64); le = c->kernel->fd32(c->hm32, c->ht32, f, c->stream); }
  if (le != hipSuccess) return fail(c, MJPCX_EDEVICE, std::string("fd kernel launch: ") + hipGetErrorString(le));
  double* dA = (double*)(base + off_A);
  double *dB = dA + nA, *dC = dB + nB, *dD = dC + nC;
  const int total = (int)(Tn * (ndx + nr) * (ndx + nu));
  hipLaunchKernelGGL((fd_assemble_kernel<T>), dim3(std::min((total + 255) / 256, 1024)), dim3(256), 0, c->stream,
                     (const T*)base, (const T*)(base + off_sensor), (const T*)d[2], (const T*)d[3],
                     (const int*)(base + off_lim), Tn, (int)ndx, (int)nu, (int)nr, (T)eps, centered, dA, dB, dC, dD);
  HIPCHK(c, hipGetLastError());
  if (A) HIPCHK(c, hipMemcpyAsync(A, dA, nA * 8, hipMemcpyDeviceToHost, c->stream));
  if (B) HIPCHK(c, hipMemcpyAsync(B, dB, nB * 8, hipMemcpyDeviceToHost, c->stream));
  if (C) HIPCHK(c, hipMemcpyAsync(C, dC, nC * 8, hipMemcpyDeviceToHost, c->stream));
  if (D) HIPCHK(c, hipMemcpyAsync(D, dD, nD * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}
}  // namespace

namespace {
// per-plan task blob of a wave context on the device (not the Predictive-Sampling hot path: pageable copy)
int wave_blob(mjpcx_ctx* c, WaveTask* wt) {
  c->blob_scratch.resize(c->wh.blob_bytes);
  c->wh.fill_blob(c->blob_scratch.data());
  HIPCHK(c, c->d_wblob.reserve(c->wh.blob_bytes));
  HIPCHK(c, hipMemcpyAsync(c->d_wblob.p, c->blob_scratch.data(), c->wh.blob_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *wt = c->wh.t;
  wt->blob = (const double*)c->d_wblob.p;
  wt->stamps = nullptr;
  wt->stamp_step = 0;
  return MJPCX_OK;
}
// tree: the iLQG kernels of a model wave_tree.h covers run its forward pass (P = the scratch elements kept in the node-time slot)
size_t wave_lds_bytes(const mjpcx_ctx* c, int P, bool tree = false) {
  const WaveModel& wm = c->wh.m;
  if (tree)
    return (8 * w64::wave_lds_elems_tree(wm.nq, wm.nv, wm.nu, wm.nbody, wm.njnt, wm.nsite, c->wh.t.nr, c->wh.t.nterm, P, false, w64::kTreeMaxSimpleBig,
                                         w64::kTreeMaxConeBig) + 15) & ~(size_t)15;
  return (8 * w64::wave_lds_elems(wm.nq, wm.nv, wm.nu, wm.nbody, wm.njnt, wm.nsite, c->wh.t.nr, c->wh.t.nterm, P, wm.cone) + 15) & ~(size_t)15;
}

int do_feedback_wave(mjpcx_ctx* c, int N, int H, int mode, int representation, int use_state, int Tn, const double* times,
                     const double* states, const double* actions, const double* gains, const double* improvement, const double* alpha) {
  if (c->precision != 64) return fail(c, MJPCX_EUNSUPPORTED, "the iLQG kernels of the wavefront-per-candidate family are fp64 only");
  int rc;
  if ((rc = reserve_rollout(c, N, H, 1)) != MJPCX_OK) return rc;
  const size_t ds = c->nq + c->nv, ndx = 2 * (size_t)c->nv, nu = c->nu;
  std::vector<double*> d;
  if ((rc = upload_arrays<double>(c, c->d_ilqg, {{times, (size_t)Tn}, {states, Tn * ds}, {actions, Tn * nu}, {gains, Tn * nu * ndx},
                                                 {improvement, Tn * nu}, {alpha, (size_t)N}}, &d)) != MJPCX_OK) return rc;
  WaveTask wt;
  if ((rc = wave_blob(c, &wt)) != MJPCX_OK) return rc;
  RolloutArgs<double> a{};
  a.N = N; a.H = H; a.P = 1; a.interp = 0; a.nodes = (double*)c->d_nodes.p; a.noise.mode = -1;
  a.states = (double*)c->d_states.p; a.actions = (double*)c->d_actions.p; a.times = (double*)c->d_times.p;
  a.residual = (double*)c->d_residual.p; a.costs = (double*)c->d_costs.p; a.trace = (double*)c->d_trace.p;
  a.total_return = (double*)c->d_ret.p; a.failure = (int*)c->d_fail.p;
  w64::FeedbackWaveArgs fb{d[0], d[1], d[2], d[3], d[4], d[5], Tn, mode, representation, use_state, 0};
  const bool tree = c->wh.tree_ok && !c->no_tree;
  // a model of the quad kernel's class: the iLQG rollouts are one or ten candidates of pure per-step latency, and the quad form's step is the
  // shortest (one candidate per wavefront; MJPCX_NO_QUAD_FEEDBACK=1 keeps the wavefront-per-candidate kernel for A/B runs). Candidates it
  // hands on are rolled out by that kernel afterwards, as for the sampling rollouts.
  if (c->quad_ok && !c->no_quad_feedback && c->wh.m.integrator == MJPCX_INT_EULER) {
    quad::QArgs q{};
    q.N = N; q.H = H; q.P = 1; q.noise_mode = -1; q.nodes = (double*)c->d_nodes.p;
    q.states = a.states; q.actions = a.actions; q.times = a.times; q.residual = a.residual; q.costs = a.costs; q.trace = a.trace;
    q.total_return = a.total_return; q.failure = a.failure; q.con_cap = c->quad_con_cap; q.cpw = 1;
    const quad::QBlob bo{wt.off_time, wt.off_mocap, wt.off_weight, wt.off_normp, wt.off_normq, wt.off_param, wt.off_risk, wt.off_rreal, wt.off_rint};
    const quad::QFeedback qf{d[0], d[1], d[2], d[3], d[4], d[5], Tn, mode, representation, use_state};
    HIPCHK(c, hipMemsetAsync(c->d_qstats.p, 0, 32, c->stream));
    HIPCHK(c, quad::launch_feedback_quad(c->d_qmodel.p, c->d_qtab.p, wt.blob, bo, q, qf, (int*)c->d_qstats.p, c->stream));
    if (!c->h_qstats && hipHostMalloc(&c->h_qstats, 32, hipHostMallocDefault) != hipSuccess) c->h_qstats = nullptr;
    bool handed_on = true;
    if (c->h_qstats) {
      HIPCHK(c, hipMemcpyAsync(c->h_qstats, c->d_qstats.p, 32, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      handed_on = static_cast<const int*>(c->h_qstats)[0] != 0;
    }
    if (!handed_on) {
      c->N = N; c->H = H; c->P = 1;
      c->have_rollout = true;
      c->traj_candidate_major = true;
      return MJPCX_OK;
    }
    fb.only_flagged = 1;
  }
  const int Ppolicy = tree ? (int)(ndx + 2 * ds) : (int)((ndx + 2 * ds + nu - 1) / nu + 1);
  const size_t lds = wave_lds_bytes(c, Ppolicy, tree);
  const bool rk4 = c->wh.m.integrator == MJPCX_INT_RK4;
  HIPCHK(c, launch_feedback_wave(c->wh.m, wt, a, fb, N, lds, tree, rk4, c->wh.registered == 0 ? c->wh.dev_image : nullptr, c->wh.blob_bytes, c->stream));
  c->N = N; c->H = H; c->P = 1;
  c->have_rollout = true;
  c->traj_candidate_major = true;
  return MJPCX_OK;
}

int do_transition_fd_wave(mjpcx_ctx* c, int Tn, const double* times, const double* states, const double* actions, double eps,
                          int centered, double* A, double* B, double* C, double* D) {
  if (c->precision != 64) return fail(c, MJPCX_EUNSUPPORTED, "the iLQG kernels of the wavefront-per-candidate family are fp64 only");
  int rc;
  const size_t ds = c->nq + c->nv, ndx = 2 * (size_t)c->nv, nu = c->nu, nr = c->nr;
  const size_t nc = 1 + 2 * (ndx + nu);
  std::vector<double*> d;
  std::vector<double> cr(c->ctrlrange);
  if ((rc = upload_arrays<double>(c, c->d_ilqg, {{times, (size_t)Tn}, {states, Tn * ds}, {actions, Tn * nu}, {cr.data(), 2 * nu}}, &d)) != MJPCX_OK)
    return rc;
  WaveTask wt;
  if ((rc = wave_blob(c, &wt)) != MJPCX_OK) return rc;
  // outputs: raw next [Tn][nc][ds], tangent next [Tn][nc][ndx], sensor [Tn][nc][nr], ctrllimited (int), then A,B,C,D
  const size_t off_tan = (Tn * nc * ds * 8 + 15) & ~(size_t)15;
  const size_t off_sensor = (off_tan + Tn * nc * ndx * 8 + 15) & ~(size_t)15;
  const size_t off_lim = (off_sensor + Tn * nc * nr * 8 + 15) & ~(size_t)15;
  const size_t off_A = (off_lim + nu * sizeof(int) + 15) & ~(size_t)15;
  const size_t nA = Tn * ndx * ndx, nB = Tn * ndx * nu, nC = Tn * nr * ndx, nD = Tn * nr * nu;
  HIPCHK(c, c->d_ilqg_out.reserve(off_A + (nA + nB + nC + nD) * 8));
  char* base = (char*)c->d_ilqg_out.p;
  HIPCHK(c, hipMemcpyAsync(base + off_lim, c->ctrllimited.data(), nu * sizeof(int), hipMemcpyHostToDevice, c->stream));
  w64::FdWaveArgs f{d[0], d[1], d[2], Tn, (int)nc, eps, (double*)base, (double*)(base + off_sensor)};
  const bool tree = c->wh.tree_ok && !c->no_tree;
  const size_t lds = wave_lds_bytes(c, 1, tree);
  const bool rk4 = c->wh.m.integrator == MJPCX_INT_RK4;
  HIPCHK(c, launch_transition_fd_wave(c->wh.m, wt, f, (unsigned)(Tn * nc), lds, tree, rk4, c->stream));
  HIPCHK(c, launch_fd_tangent(c->wh.m, (const double*)base, (double*)(base + off_tan), Tn, (int)nc, c->stream));
  double* dA = (double*)(base + off_A);
  double *dB = dA + nA, *dC = dB + nB, *dD = dC + nC;
  const int total = (int)(Tn * (ndx + nr) * (ndx + nu));
  hipLaunchKernelGGL((fd_assemble_kernel<double>), dim3(std::min((total + 255) / 256, 1024)), dim3(256), 0, c->stream,
                     (const double*)(base + off_tan), (const double*)(base + off_sensor), (const double*)d[2], (const double*)d[3],
                     (const int*)(base + off_lim), Tn, (int)ndx, (int)nu, (int)nr, eps, centered, dA, dB, dC, dD);
  HIPCHK(c, hipGetLastError());
  if (A) HIPCHK(c, hipMemcpyAsync(A, dA, nA * 8, hipMemcpyDeviceToHost, c->stream));
  if (B) HIPCHK(c, hipMemcpyAsync(B, dB, nB * 8, hipMemcpyDeviceToHost, c->stream));
  if (C) HIPCHK(c, hipMemcpyAsync(C, dC, nC * 8, hipMemcpyDeviceToHost, c->stream));
  if (D) HIPCHK(c, hipMemcpyAsync(D, dD, nD * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}
}  // namespace

extern "C" {

int mjpcx_rollout_feedback(mjpcx_ctx* c, int N, int H, int mode, int representation, int use_state, int Tn,
                           const double* times, const double* states, const double* actions, const double* gains,
                           const double* improvement, const double* alpha) {
  if (!c || !times || !states || !actions || !gains || !improvement || !alpha) return fail(c, MJPCX_EINVAL, "null argument");
  if (N < 1 || H < 1 || Tn < 1) return fail(c, MJPCX_EINVAL, "N, H and Tn must be >= 1");
  if (mode == 0 && H > Tn) return fail(c, MJPCX_EINVAL, "index policy needs a nominal trajectory at least as long as the horizon");
  if (mode != 0 && mode != 1) return fail(c, MJPCX_EINVAL, "unknown feedback policy mode");
  if (mode == 1 && (representation < 0 || representation > 2)) return fail(c, MJPCX_EINVAL, "iLQG policy representation must be 0 (zero-order), 1 (linear) or 2 (cubic)");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->wave) return do_feedback_wave(c, N, H, mode, representation, use_state, Tn, times, states, actions, gains, improvement, alpha);
  const size_t shmem = (size_t)Tn * (1 + 2 * c->nv + 2 * c->nu + c->nu * 2 * c->nv) * esize(c);
  if (shmem > 120 * 1024) return fail(c, MJPCX_EUNSUPPORTED, "nominal trajectory too large for the LDS stage");
  return c->precision == 64 ? do_feedback<double>(c, N, H, mode, representation, use_state, Tn, times, states, actions, gains, improvement, alpha)
                            : do_feedback<float>(c, N, H, mode, representation, use_state, Tn, times, states, actions, gains, improvement, alpha);
}

int mjpcx_transition_fd(mjpcx_ctx* c, int Tn, const double* times, const double* states, const double* actions, double eps,
                        int centered, double* A, double* B, double* C, double* D) {
  if (!c || !times || !states || !actions) return fail(c, MJPCX_EINVAL, "null argument");
  if (Tn < 1 || !(eps > 0)) return fail(c, MJPCX_EINVAL, "bad horizon or epsilon");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->wave) return do_transition_fd_wave(c, Tn, times, states, actions, eps, centered, A, B, C, D);
  return c->precision == 64 ? do_transition_fd<double>(c, Tn, times, states, actions, eps, centered, A, B, C, D)
                            : do_transition_fd<float>(c, Tn, times, states, actions, eps, centered, A, B, C, D);
}

int mjpcx_kinematics(mjpcx_ctx* c, double* xpos, double* xquat, double* xmat, double* xipos, double* site_xpos, double* subtree_com,
                     double* subtree_linvel) {
  if (!c) return MJPCX_EINVAL;
  if (!c->wave) return fail(c, MJPCX_EUNSUPPORTED, "mjpcx_kinematics is implemented for the wavefront-per-candidate models only");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  WaveTask wt;
  if ((rc = wave_blob(c, &wt)) != MJPCX_OK) return rc;
  const size_t nb = (size_t)c->wh.m.nbody_model, ns = (size_t)c->nsite_model;
  const size_t total = 3 * nb + 4 * nb + 9 * nb + 3 * nb + 3 * ns + 3 * nb + 3 * nb;
  HIPCHK(c, c->d_ilqg_out.reserve(total * 8));
  HIPCHK(c, hipMemsetAsync(c->d_ilqg_out.p, 0, total * 8, c->stream));
  const size_t lds = wave_lds_bytes(c, 1);
  HIPCHK(c, launch_kinematics_wave(c->wh.m, wt, (double*)c->d_ilqg_out.p, (int)nb, (int)ns, lds, c->stream));
  std::vector<double> h(total);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->d_ilqg_out.p, total * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const double* p = h.data();
  auto take = [&](double* dst, size_t n) { if (dst) std::memcpy(dst, p, n * 8); p += n; };
  take(xpos, 3 * nb); take(xquat, 4 * nb); take(xmat, 9 * nb); take(xipos, 3 * nb); take(site_xpos, 3 * ns); take(subtree_com, 3 * nb);
  take(subtree_linvel, 3 * nb);
  return MJPCX_OK;
}

int mjpcx_cost_derivatives(mjpcx_ctx* c, int T, const double* residual, const double* C, const double* D, double* cx,
                           double* cu, double* cxx, double* cxu, double* cuu) {
  if (!c || !residual || !C || !D || !cx || !cu || !cxx || !cxu || !cuu) return fail(c, MJPCX_EINVAL, "null argument");
  if (T < 1) return fail(c, MJPCX_EINVAL, "T must be >= 1");
  if (c->nterm > 32) return fail(c, MJPCX_EUNSUPPORTED, "more than 32 cost terms");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t ndx = 2 * (size_t)c->nv, nu = c->nu, nr = c->nr;
  CostSpec cs{};
  cs.num_term = c->nterm; cs.num_residual = c->nr; cs.risk = c->wave ? c->wh.risk : c->ht64.risk;
  for (int k = 0; k < c->nterm; k++) {
    if (c->dim_norm_residual[k] > 32) return fail(c, MJPCX_EUNSUPPORTED, "cost term wider than 32 residuals");
    cs.dim[k] = c->dim_norm_residual[k];
    if (c->wave) { cs.norm[k] = c->wh.norm_types[k]; cs.weight[k] = c->wh.weight[k]; cs.p[k] = c->wh.norm_p[k]; cs.q[k] = c->wh.norm_q[k]; }
    else { cs.norm[k] = c->ht64.norm[k]; cs.weight[k] = c->ht64.weight[k]; cs.p[k] = c->ht64.norm_p[k]; cs.q[k] = c->ht64.norm_q[k]; }
  }
  std::vector<double*> d;
  int rc;
  if ((rc = upload_arrays<double>(c, c->d_ilqg, {{residual, T * nr}, {C, T * nr * ndx}, {D, T * nr * nu}}, &d)) != MJPCX_OK) return rc;
  const size_t n_out = T * (ndx + nu + ndx * ndx + ndx * nu + nu * nu);
  HIPCHK(c, c->d_ilqg_out.reserve(n_out * 8));
  double* o = (double*)c->d_ilqg_out.p;
  double *dcx = o, *dcu = dcx + T * ndx, *dcxx = dcu + T * nu, *dcxu = dcxx + T * ndx * ndx, *dcuu = dcxu + T * ndx * nu;
  const size_t shmem = (32 + 32 * 32 + 32 * ndx + 32 * nu) * 8;
  hipLaunchKernelGGL(cost_derivatives_kernel, dim3(T), dim3(64), shmem, c->stream, cs, (const double*)d[0], (const double*)d[1],
                     (const double*)d[2], T, (int)ndx, (int)nu, dcx, dcu, dcxx, dcxu, dcuu);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(cx, dcx, T * ndx * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cu, dcu, T * nu * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cxx, dcxx, T * ndx * ndx * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cxu, dcxu, T * ndx * nu * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cuu, dcuu, T * nu * nu * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}

int mjpcx_backward_pass(mjpcx_ctx* c, int n, int m, int T, double mu, int reg_type, int use_limits, const double* A,
                        const double* B, const double* cx, const double* cu, const double* cxx, const double* cxu,
                        const double* cuu, const double* actions, const double* limits, double* Vx, double* Vxx, double* K,
                        double* du, double* dV, int32_t* status, double* kernel_ms) {
  if (!c || !A || !B || !cx || !cu || !cxx || !cxu || !cuu || !actions || !limits || !Vx || !Vxx || !K || !du || !dV || !status)
    return fail(c, MJPCX_EINVAL, "null argument");
  if (n < 1 || n > 48 || m < 1 || m > 16 || T < 2) return fail(c, MJPCX_EUNSUPPORTED, "backward pass kernel covers 1 <= n <= 48, 1 <= m <= 16, T >= 2");
  if (reg_type < 0 || reg_type > 2) return fail(c, MJPCX_EINVAL, "unknown regularization type");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<double*> d;
  int rc;
  const size_t sn = n, sm = m, sT = T;
  if ((rc = upload_arrays<double>(c, c->d_ilqg, {{A, sT * sn * sn}, {B, sT * sn * sm}, {cx, sT * sn}, {cu, sT * sm}, {cxx, sT * sn * sn},
                                                 {cxu, sT * sn * sm}, {cuu, sT * sm * sm}, {actions, sT * sm}, {limits, 2 * sm}}, &d)) != MJPCX_OK)
    return rc;
  const size_t n_out = sT * (sn + sn * sn + sm * sn + sm) + 2 + 2 + 16;
  HIPCHK(c, c->d_ilqg_out.reserve(n_out * 8));
  double* o = (double*)c->d_ilqg_out.p;
  BackwardArgs a{};
  a.n = n; a.m = m; a.T = T; a.mu = mu; a.reg_type = reg_type; a.use_limits = use_limits;
  a.A = d[0]; a.B = d[1]; a.cx = d[2]; a.cu = d[3]; a.cxx = d[4]; a.cxu = d[5]; a.cuu = d[6]; a.actions = d[7]; a.limits = d[8];
  a.Vx = o; a.Vxx = a.Vx + sT * sn; a.K = a.Vxx + sT * sn * sn; a.du = a.K + sT * sm * sn; a.dV = a.du + sT * sm;
  a.status = (int*)(a.dV + 2);
  a.stamps = c->stamp_step >= 0 ? (long long*)(a.dV + 4) : nullptr;
  const int NP = (n + 15) & ~15;
  const size_t lds = (size_t)(5 * NP * NP + 3 * NP + 6 * NP * 16 + 5 * 256 + 16 * 23 + 16 * 13) * 8;  // (the carve of backward_pass_kernel, with slack)
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  // the m x m factorisation / box-QP of a step is unrolled to 12 or 16 columns at compile time (the A1 has 12 controls)
  if (m <= 12) {
    HIPCHK(c, hipFuncSetAttribute((const void*)backward_pass_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(backward_pass_kernel<12>, dim3(1), dim3(64 * kBackwardWaves), lds, c->stream, a);
  } else {
    HIPCHK(c, hipFuncSetAttribute((const void*)backward_pass_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(backward_pass_kernel<16>, dim3(1), dim3(64 * kBackwardWaves), lds, c->stream, a);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipMemcpyAsync(Vx, a.Vx, sT * sn * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(Vxx, a.Vxx, sT * sn * sn * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(K, a.K, sT * sm * sn * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(du, a.du, sT * sm * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(dV, a.dV, 16, hipMemcpyDeviceToHost, c->stream));
  int st = 0;
  HIPCHK(c, hipMemcpyAsync(&st, a.status, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *status = st;
  if (a.stamps) {
    long long h[9];
    (void)hipMemcpy(h, a.stamps, sizeof h, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "backward_pass phase cycles (second step): stage %lld gemm %lld reg %lld du/qp %lld Kcols %lld update %lld write %lld\n",
                 h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6]);
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (kernel_ms) *kernel_ms = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return MJPCX_OK;
}

}  // extern "C"

// ===================================================================== multi-GPU exchange over RCCL
namespace {
// librccl.so.1 is resolved at run time: the library loads (and every single-GPU entry point works) without it
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};
RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) { const char* why = dlerror(); api.error = std::string("librccl.so.1 not found: ") + (why ? why : ""); return; }  // (dlerror() clears itself: one call)
#define MJPCX_SYM(field, sym)                                                              \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, #sym));                \
    if (!api.field) api.error = std::string("RCCL symbol missing: ") + #sym;
    MJPCX_SYM(GetUniqueId, ncclGetUniqueId) MJPCX_SYM(CommInitRank, ncclCommInitRank) MJPCX_SYM(CommDestroy, ncclCommDestroy)
    MJPCX_SYM(AllGather, ncclAllGather) MJPCX_SYM(AllReduce, ncclAllReduce) MJPCX_SYM(Broadcast, ncclBroadcast)
    MJPCX_SYM(GetErrorString, ncclGetErrorString)
#undef MJPCX_SYM
  });
  return &api;
}
#define NCCLCHK(c, expr)                                                                                   \
  do {                                                                                                     \
    ncclResult_t r__ = (expr);                                                                             \
    if (r__ != ncclSuccess) return fail(c, MJPCX_EDEVICE, std::string(#expr) + ": " + rccl()->GetErrorString(r__)); \
  } while (0)

// all-gather of `n` doubles per rank: host in -> host out [world][n]
int comm_all_gather(mjpcx_ctx* c, const double* in, int n, std::vector<double>* out) {
  RcclApi* R = rccl();
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->d_comm_send.reserve((size_t)n * 8));
  HIPCHK(c, c->d_comm_recv.reserve((size_t)n * 8 * c->comm_world));
  HIPCHK(c, hipMemcpyAsync(c->d_comm_send.p, in, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(c, R->AllGather(c->d_comm_send.p, c->d_comm_recv.p, (size_t)n, ncclFloat64, (ncclComm_t)c->comm, c->stream));
  out->resize((size_t)n * c->comm_world);
  HIPCHK(c, hipMemcpyAsync(out->data(), c->d_comm_recv.p, (size_t)n * 8 * c->comm_world, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}
}  // namespace

extern "C" {

int mjpcx_comm_unique_id(void* id_out) {
  if (!id_out) return MJPCX_EINVAL;
  RcclApi* R = rccl();
  if (!R->lib || !R->error.empty()) { g_create_error = R->error; return MJPCX_EUNSUPPORTED; }
  static_assert(sizeof(ncclUniqueId) == MJPCX_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return MJPCX_EDEVICE; }
  std::memcpy(id_out, &id, sizeof id);
  return MJPCX_OK;
}

int mjpcx_comm_init(mjpcx_ctx* c, const void* unique_id, int rank, int world) {
  if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return fail(c, MJPCX_EINVAL, "bad communicator arguments");
  RcclApi* R = rccl();
  if (!R->lib || !R->error.empty()) return fail(c, MJPCX_EUNSUPPORTED, R->error);
  if (c->comm) return fail(c, MJPCX_ESTATE, "the context already has a communicator");
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  // ncclCommInitRank blocks until every rank of the world has joined: a rank that never arrives (a crashed peer, a mismatched id) would
  // hang the planner for good. The call runs on a helper thread with a deadline (MJPCX_COMM_TIMEOUT_S, default 120 s); past it the
  // context stays without a communicator, the caller gets MJPCX_EDEVICE and can fall back to its own transport (bench.py does).
  // A communicator that arrives AFTER the deadline is destroyed by the helper thread itself (the peers' next collective then fails
  // instead of waiting for a rank that has moved on); callers must agree on the fallback collectively, as bench.py does.
  double deadline = 120.0;
  if (const char* e = getenv("MJPCX_COMM_TIMEOUT_S")) {  // (validated BEFORE the helper thread exists: nothing to abandon on a bad value)
    char* end = nullptr;
    const double v = std::strtod(e, &end);
    if (end == e || !(v > 0)) return fail(c, MJPCX_EINVAL, "MJPCX_COMM_TIMEOUT_S must be a positive number of seconds");
    deadline = v;
  }
  struct InitState { std::mutex m; std::condition_variable cv; bool done = false, abandoned = false; ncclResult_t rc = ncclSuccess; ncclComm_t comm = nullptr; };
  auto st = std::make_shared<InitState>();
  const int device = c->device;
  std::thread([st, R, world, id, rank, device]() {
    (void)hipSetDevice(device);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = R->CommInitRank(&comm, world, id, rank);
    bool late;
    {
      std::lock_guard<std::mutex> lk(st->m);
      st->rc = rc; st->comm = comm; st->done = true;
      late = st->abandoned;
      st->cv.notify_all();
    }
    if (late && rc == ncclSuccess && comm) (void)R->CommDestroy(comm);
  }).detach();
  {
    std::unique_lock<std::mutex> lk(st->m);
    if (!st->cv.wait_for(lk, std::chrono::duration<double>(deadline), [&] { return st->done; })) {
      st->abandoned = true;
      return fail(c, MJPCX_EDEVICE, "ncclCommInitRank did not complete within MJPCX_COMM_TIMEOUT_S: a rank of the world is missing");
    }
  }
  NCCLCHK(c, st->rc);
  ncclComm_t comm = st->comm;
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  return MJPCX_OK;
}

int mjpcx_comm_info(const mjpcx_ctx* c, int* rank, int* world) {
  if (!c) return MJPCX_EINVAL;
  if (rank) *rank = c->comm_rank;
  if (world) *world = c->comm_world;
  return MJPCX_OK;
}

int mjpcx_comm_destroy(mjpcx_ctx* c) {
  if (!c) return MJPCX_EINVAL;
  if (c->comm) {
    (void)hipSetDevice(c->device);
    (void)rccl()->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  c->comm_rank = 0;
  c->comm_world = 1;
  return MJPCX_OK;
}

int mjpcx_exchange_best(mjpcx_ctx* c, int32_t* index, double* best_return, double* nominal_return, double* spline_values, int n) {
  if (!c || !index || !best_return || !nominal_return || (n > 0 && !spline_values) || n < 0) return fail(c, MJPCX_EINVAL, "null argument");
  if (!c->comm) return c->comm_world == 1 ? MJPCX_OK : fail(c, MJPCX_ESTATE, "mjpcx_comm_init has not been called");  // no communicator: one rank
  const double rec[3] = {*best_return, (double)*index, *nominal_return};
  int rc;
  if ((rc = comm_all_gather(c, rec, 3, &c->h_comm)) != MJPCX_OK) return rc;
  int owner = 0;
  auto key = [&](int r) { const double v = c->h_comm[3 * r]; return v != v ? (double)INFINITY : v; };
  for (int r = 1; r < c->comm_world; r++)
    if (key(r) < key(owner) || (key(r) == key(owner) && c->h_comm[3 * r + 1] < c->h_comm[3 * owner + 1])) owner = r;
  *best_return = c->h_comm[3 * owner];
  *index = (int32_t)c->h_comm[3 * owner + 1];
  *nominal_return = c->h_comm[2];  // global candidate 0 lives on rank 0
  if (n > 0) {
    RcclApi* R = rccl();
    HIPCHK(c, c->d_comm_send.reserve((size_t)n * 8));
    if (c->comm_rank == owner) HIPCHK(c, hipMemcpyAsync(c->d_comm_send.p, spline_values, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, R->Broadcast(c->d_comm_send.p, c->d_comm_send.p, (size_t)n, ncclFloat64, owner, (ncclComm_t)c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(spline_values, c->d_comm_send.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return MJPCX_OK;
}

int mjpcx_merge_topk(mjpcx_ctx* c, int k, int64_t* index, double* total_return) {
  if (!c || k < 1 || !index || !total_return) return fail(c, MJPCX_EINVAL, "null argument");
  if (!c->comm) return c->comm_world == 1 ? MJPCX_OK : fail(c, MJPCX_ESTATE, "mjpcx_comm_init has not been called");
  std::vector<double> mine(2 * (size_t)k);
  for (int i = 0; i < k; i++) {
    const bool used = index[i] >= 0;
    mine[2 * i] = used && total_return[i] == total_return[i] ? total_return[i] : (double)INFINITY;  // (NaN ranks last, and keeps the comparator a strict weak order)
    mine[2 * i + 1] = used ? (double)index[i] : 4503599627370496.0;  // 2^52: sorts after every real index
  }
  int rc;
  if ((rc = comm_all_gather(c, mine.data(), 2 * k, &c->h_comm)) != MJPCX_OK) return rc;
  const int total = k * c->comm_world;
  std::vector<int> order(total);
  for (int i = 0; i < total; i++) order[i] = i;
  const std::vector<double>& all = c->h_comm;
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    if (all[2 * a] != all[2 * b]) return all[2 * a] < all[2 * b];
    return all[2 * a + 1] < all[2 * b + 1];
  });
  for (int i = 0; i < k; i++) {
    const int o = order[i];
    const bool used = all[2 * o + 1] < 4503599627370496.0;
    index[i] = used ? (int64_t)all[2 * o + 1] : -1;
    total_return[i] = used ? all[2 * o] : 1.0e300;
  }
  return MJPCX_OK;
}

int mjpcx_elite_allreduce(mjpcx_ctx* c, double* values, int n) {
  if (!c || !values || n < 1) return fail(c, MJPCX_EINVAL, "null argument");
  if (!c->comm) return c->comm_world == 1 ? MJPCX_OK : fail(c, MJPCX_ESTATE, "mjpcx_comm_init has not been called");
  RcclApi* R = rccl();
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->d_comm_send.reserve((size_t)n * 8));
  HIPCHK(c, hipMemcpyAsync(c->d_comm_send.p, values, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(c, R->AllReduce(c->d_comm_send.p, c->d_comm_send.p, (size_t)n, ncclFloat64, ncclSum, (ncclComm_t)c->comm, c->stream));
  HIPCHK(c, hipMemcpyAsync(values, c->d_comm_send.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MJPCX_OK;
}

int mjpcx_comm_barrier(mjpcx_ctx* c) {
  if (!c) return MJPCX_EINVAL;
  double one = 1.0;
  return mjpcx_elite_allreduce(c, &one, 1);
}

}  // extern "C"
