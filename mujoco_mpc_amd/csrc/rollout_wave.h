// rollout_wave.h -- wavefront-per-candidate rollout kernel for models with free joints, friction loss and
// contacts (the Quadruped class). ONE 64-lane wavefront integrates ONE candidate; the candidate's whole mjData
// lives in LDS (~40 KB), every stage of the step is spread over the lanes by its natural index:
//
//   kinematics / comVel / RNE   one lane per body of a tree level (levels are walked in order)
//   subtree sums                one lane per body, membership from a baked bit mask (no serial tree walk)
//   CRB, passive, actuation     one lane per dof / actuator
//   Cholesky / solves (nv<=32)  lane i owns row i, columns walked in order
//   collision                   one lane per moving geom against each static geom; contacts are compacted with
//                               ballot + popcount so that their ORDER is the oracle's (geom-pair major)
//   constraint rows             one lane per row (J row . vector products, impedance, penalty zones, line search)
//   Newton Hessian              the 171 lower-triangle entries dealt over the 64 lanes
//
// The step function is the one stated in oracle/physics.c + oracle/contact.inc (mj_step: checkPos/Vel, mj_forward
// with the task residual at the sensor stage, checkAcc, Euler with implicit joint damping); trajectory
// bookkeeping is Trajectory::Rollout (mjpc/trajectory.cc:100-210) as in rollout_lane.h.
#pragma once
#include "device_common.h"
#include "rollout_lane.h"  // RolloutArgs / NoiseArgs
#include "wave_model.h"
#include "lds_model.h"
#include "tree_registry.h"
#include "quad_abi.h"  // kQFallback
#include "ilqg_kernels.h"  // find_interval, fd_assemble_kernel
#include "wave_ilqg_launch.h"  // FdWaveArgs, FeedbackWaveArgs

// The device code below is instantiated twice, textually: namespace mjpcx::w64 with wreal = double (the parity path, also
// the finite-difference and feedback-rollout kernels of iLQG) and mjpcx::w32 with wreal = float (BASELINE configs[3]'s
// precision: half the LDS per candidate, twice the VALU rate). WL() gives floating literals the working type.
namespace mjpcx {
__device__ __forceinline__ void w_sincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ void w_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
// 32- and 64-bit lanes through v_readlane / DPP moves (zero fill at the row boundaries)
__device__ __forceinline__ double w_readlane(double v, int src) {  // src must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float w_readlane(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
// one 16 x 16 x 4 MFMA step in the working type; accumulator register rg of lane l holds row (l >> 4) + 4 rg (f64) or
// 4 (l >> 4) + rg (f32) of the tile, column l & 15
typedef double w_acc4_f64 __attribute__((ext_vector_type(4)));
typedef float w_acc4_f32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w_acc4_f64 w_mfma_16x16x4(double a, double b, w_acc4_f64 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ w_acc4_f32 w_mfma_16x16x4(float a, float b, w_acc4_f32 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
}  // namespace mjpcx

#define WL(x) ((wreal)(x))

#define WAVE_NS w64
#define wreal double
#define WAVE_KERNEL_ATTR
#define MJPCX_WAVE_ILQG 1
#include "wave_core.h"
#include "wave_forward.h"
#include "wave_tree.h"
#include "wave_residual.h"
#include "wave_kernel.h"
#include "tree_kernel.h"
#ifdef MJPCX_WITH_ILQG_WAVE_KERNELS  // (ilqg_wave.hip: the iLQG kernels are instantiated in that translation unit only)
#include "wave_ilqg.h"
#endif
#undef MJPCX_WAVE_ILQG
#undef WAVE_KERNEL_ATTR
#undef wreal
#undef WAVE_NS

// fp32 generic kernels: the Jacobian-free path's first pass needs 12.7 KB of LDS per Humanoid candidate, i.e. 12 candidates per CU, so
// the kernels are held to 168 registers (3 wavefronts per SIMD; 2 until the row table left in round 3: +5 % on configs[3])
#define WAVE_NS w32
#define wreal float
#define WAVE_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#include "wave_core.h"
#include "wave_forward.h"
#include "wave_tree.h"
#include "wave_residual.h"
#include "wave_kernel.h"
#include "tree_kernel.h"
#undef WAVE_KERNEL_ATTR
#undef wreal
#undef WAVE_NS
#undef WL
