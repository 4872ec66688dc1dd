// limb_launch.h -- host-side entry of the limb kernel's translation unit (limb_kernel.hip), so that mjpcx.hip does not re-compile the
// wavefront-per-candidate kernels when the limb step changes (and vice versa).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mjpcx.h"
#include "limb_abi.h"

namespace mjpcx { namespace limb {
// the kernel's view of a model + task as opaque images in both precisions (LimbModelT<float>, LimbModelT<double> of limb_model.h); returns ""
// or why the model is outside the class the limb kernel covers
std::string build_images(const mjpcx_model* m, const mjpcx_task* t, std::vector<unsigned char>& image32, std::vector<unsigned char>& image64);
// wavefronts of a launch of N candidates (cpw: LArgs::cpw, 0 = chosen from the batch size)
int limb_waves(int N, int cpw);
// stats: nullptr, or 8 ints (zeroed by the caller): [0] candidates handed on, [1 + b] by reason (limb_step.h kFlag* bit b)
hipError_t launch_rollout_limb(const void* image, const float* blob, const LBlob& bo, const LArgs<float>& a, const float* key_mpos, int* stats, hipStream_t stream);
hipError_t launch_rollout_limb(const void* image, const double* blob, const LBlob& bo, const LArgs<double>& a, const double* key_mpos, int* stats, hipStream_t stream);
} }
