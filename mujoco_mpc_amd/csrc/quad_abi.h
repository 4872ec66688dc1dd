// quad_abi.h -- the part of the quad kernel family that the rest of the library sees (mjpcx.hip, tree_kernel.h): the rollout request, the
// failure[] marker of a candidate handed to the wavefront-per-candidate kernel, and the host entry points of the quad kernel's own
// translation unit (quad_kernel.hip). Kept apart from quad_model.h so that a change of the quad kernel's model layout does not
// re-compile the wavefront-per-candidate kernels.
#pragma once
#include <stdint.h>

namespace mjpcx {
constexpr int kQFallback = 0x40000000;  // failure[] marker of a candidate the quad kernel handed on (cleared by the fallback pass)
namespace quad {
// the rollout request (RolloutArgs<double> of rollout_lane.h, flattened so that the CPU emulator can fill it too)
struct QArgs {
  int N, H, P, interp;
  const double* node_times;  // P
  double* nodes;             // [P][nu][N]
  const double* nominal;     // [P][nu]
  int noise_mode;            // -1: candidates given in `nodes`
  uint64_t seed; uint32_t iteration;
  int candidate_offset, nominal_candidate, explore_count;
  double std0, std1;
  const double* param_variance;
  double *states, *actions, *times, *residual, *costs, *trace, *total_return;  // [candidate][step][field]
  int* failure;
  long long* stamps;  // nullptr, or 64 counters: phase cycles of wavefront 0 (tuning aid, MJPCX_QUAD_STAMPS=1)
  long long* wave_times;  // nullptr, or [wavefront][4] (tuning aid, with stamps): cycles; Newton iterations run (each step the slowest candidate's); steps through the general solver; the largest per-lane contact count, summed over the steps
  long long* wave_class;  // nullptr, or [wavefront][64] (tuning aid, MJPCX_QUAD_CLASSES=1): per class of wavefront-step -- bit 0 some candidate has a contact between
                          // two moving geoms, bit 1 a leg-leg one, bit 2 some lane holds more contacts than LDS slots, bit 3 more than line-search slots --
                          // [2 c] steps, [2 c + 1] cycles in the constraint solve, [32 + 2 c] steps, [32 + 2 c + 1] cycles of the whole step
  int cpw;            // candidates per wavefront (1, 2, 4, 8 or 16: the first 4 * cpw lanes of a wavefront work; 0 = 16). Small batches spread over more wavefronts: the lock-step is over fewer candidates and every SIMD gets one
  double* ovf_slab;   // builds with QEXP_OVF_SLAB only (nullptr otherwise): per wavefront, (kQMaxCon - kQLdsSlots) x kQConRec x 64 doubles ([slot][field][lane], as the
                      // LDS store): a lane's contacts beyond the LDS slots (quad_ovf_doubles_per_wave / quad_waves: quad_launch.h)
  int con_cap;        // a lane that collects more contacts than this hands its candidate on (0: kQMaxCon, the store's capacity; MJPCX_QUAD_CON_CAP lowers it, for tests of the hand-on)
};

// the feedback policy of the iLQG rollouts (FeedbackArgs of ilqg_kernels.h): nullptr members = the spline policy of QArgs
struct QFeedback {
  const double *times, *states, *actions, *gains, *improvement, *alpha;  // [Tn], [Tn][nq + nv], [Tn][nu], [Tn][nu][2 nv], [Tn][nu], [N]
  int Tn, mode, representation, use_state;  // mode 0: index policy (RolloutDiscrete, planner.cc:630-692); 1: iLQGPolicy::Action at the time (policy.cc:82-161)
};

// offsets into the per-plan blob (WaveTaskT: wave_model.h)
struct QBlob { int off_time, off_mocap, off_weight, off_normp, off_normq, off_param, off_risk, off_rreal, off_rint; };
}  // namespace quad
}  // namespace mjpcx
