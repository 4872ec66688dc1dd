// wave32_launch.h -- host-side entry of the translation unit that holds the fp32 kernels of the wavefront-per-candidate family
// (wave32.hip: every w32:: kernel is instantiated there and nowhere else), so that they compile next to their fp64 twins of mjpcx.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "rollout_lane.h"  // RolloutArgs
#include "wave_model.h"

namespace mjpcx {
// generic kernels (rollout_wave_kernel<NMAX, TREE, RK4, SMALL>): which instantiation a model takes
enum Wave32Kernel { kW32Rk4 = 0, kW32Tree, kW32TreeSmall, kW32Rows18, kW32Rows20, kW32Rows28, kW32Rows32 };
hipError_t launch_wave_kernel_f32(int which, int N, size_t lds, const WaveModelT<float>& m, const WaveTaskT<float>& wt, const RolloutArgs<float>& a,
                                  hipStream_t stream);
// registered models (rollout_tree_kernel<C, BIG>): config 0 = TreeCfgA1, 1 = TreeCfgHumanoid (tree_registry.h)
hipError_t launch_tree_kernel_f32(int config, bool big, int grid, int threads, size_t lds, const WaveModelT<float>& m, const WaveTaskT<float>& wt,
                                  const RolloutArgs<float>& a, const unsigned char* image, unsigned blob_bytes, unsigned arena_bytes, int* work,
                                  int mode, float* slabs, hipStream_t stream);
}  // namespace mjpcx
