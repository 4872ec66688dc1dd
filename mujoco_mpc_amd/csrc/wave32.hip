// wave32.hip -- the fp32 kernels of the wavefront-per-candidate family (namespace mjpcx::w32 of rollout_wave.h: BASELINE configs[3]'s
// precision) as a translation unit of their own; mjpcx.hip reaches them through wave32_launch.h.
#include <hip/hip_runtime.h>

// twelve wavefronts of 170 registers per workgroup instead of eight of 256: once the build switches of build.py had taken the spills
// away (12.3 -> 1.5 GB per launch), the third wavefront per SIMD pays (Humanoid: 155.7 -> 162.0 k rollouts/s; LDS admits eleven arenas)
#define TREE_KERNEL_THREADS 768
#include "rollout_wave.h"
#include "wave32_launch.h"

namespace mjpcx {

hipError_t launch_wave_kernel_f32(int which, int N, size_t lds, const WaveModelT<float>& m, const WaveTaskT<float>& wt, const RolloutArgs<float>& a,
                                  hipStream_t stream) {
  // the register-resident Cholesky is unrolled to NMAX columns: row-table instantiations at the widths of the shipped models; the
  // Jacobian-free path at one width (the shipped tree models are registered), in its small-list and long-list forms
  auto kern = which == kW32Rk4 ? w32::rollout_wave_kernel<32, false, true>
            : which == kW32Tree ? w32::rollout_wave_kernel<32, true>
            : which == kW32TreeSmall ? w32::rollout_wave_kernel<32, true, false, true>
            : which == kW32Rows18 ? w32::rollout_wave_kernel<18> : which == kW32Rows20 ? w32::rollout_wave_kernel<20>
            : which == kW32Rows28 ? w32::rollout_wave_kernel<28> : w32::rollout_wave_kernel<32>;
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(N), dim3(64), lds, stream, m, wt, a);
  return hipGetLastError();
}

hipError_t launch_tree_kernel_f32(int config, bool big, int grid, int threads, size_t lds, const WaveModelT<float>& m, const WaveTaskT<float>& wt,
                                  const RolloutArgs<float>& a, const unsigned char* image, unsigned blob_bytes, unsigned arena_bytes, int* work,
                                  int mode, float* slabs, hipStream_t stream) {
  auto kern = config == 0 ? (big ? w32::rollout_tree_kernel<TreeCfgA1, true> : w32::rollout_tree_kernel<TreeCfgA1, false>)
                          : (big ? w32::rollout_tree_kernel<TreeCfgHumanoid, true> : w32::rollout_tree_kernel<TreeCfgHumanoid, false>);
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, m, wt, a, image, blob_bytes, arena_bytes, work, mode, slabs);
  return hipGetLastError();
}

}  // namespace mjpcx
