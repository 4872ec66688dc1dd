// limb_abi.h -- the part of the limb kernel family that the rest of the library sees (mjpcx.hip): the rollout request and the failure[] marker
// of a candidate handed to the wavefront-per-candidate kernel. Kept apart from limb_model.h / limb_step.h so that a change of the step
// function does not re-compile the other kernels' translation units.
#pragma once
#include <stdint.h>

namespace mjpcx { namespace limb {
// the rollout request (RolloutArgs<T> of rollout_lane.h, flattened so that the CPU emulator can fill it too)
template <typename R> struct LArgs {
  int N, H, P, interp;
  const R* node_times;  // P
  R* nodes;             // [P][nu][N]
  const R* nominal;     // [P][nu]
  int noise_mode;       // -1: candidates given in `nodes`
  uint64_t seed; uint32_t iteration;
  int candidate_offset, nominal_candidate, explore_count;
  double std0, std1;
  const double* param_variance;
  R *states, *actions, *times, *residual, *costs, *trace;  // [candidate][step][field]
  double* total_return;
  int* failure;
  int cpw;              // candidates per wavefront (1, 2, 4, 8, 16; 0 = 16)
  long long* stamps;    // nullptr, or 32 counters: phase cycles of wavefront 0 (tuning aid)
  int* iters;           // nullptr, or [N]: Newton iterations summed over the steps (tuning aid)
};
constexpr int kLFallback = 0x40000000;  // failure[] marker of a candidate handed on (= kQFallback of quad_abi.h: tree_kernel.h's mode bit 32 reads it)
// offsets into the per-plan blob (WaveTaskT: wave_model.h), in elements of the working precision
struct LBlob { int off_time, off_mocap, off_weight, off_normp, off_normq, off_param, off_risk, off_rreal, off_rint; };
} }  // namespace mjpcx::limb
