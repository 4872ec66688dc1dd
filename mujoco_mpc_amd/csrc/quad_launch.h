// quad_launch.h -- host-side entry of the quad kernel's translation unit (quad_kernel.hip), so that mjpcx.hip does not
// re-compile the wavefront-per-candidate kernels when the quad step changes (and vice versa).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mjpcx.h"
#include "quad_abi.h"

namespace mjpcx { namespace quad {
// stats: nullptr, or 8 ints (zeroed by the caller): [0] candidates handed on, [1..5] by reason (quad_step.h kFlag*)
// the kernel's view of a model + task as two opaque images (QuadModel, QuadTables of quad_model.h); returns "" or why the model is
// outside the class the quad kernel covers
std::string build_images(const mjpcx_model* m, const mjpcx_task* t, std::vector<unsigned char>& model, std::vector<unsigned char>& tables);
// wavefronts of a launch of N candidates (cpw: QArgs::cpw, 0 = chosen from the batch size) and the doubles of QArgs::ovf_slab each one needs
int quad_waves(int N, int cpw);
size_t quad_ovf_doubles_per_wave();
bool quad_uses_ovf_slab();  // (whether this build keeps a lane's contacts beyond the LDS slots in QArgs::ovf_slab: QEXP_OVF_SLAB)
// the iLQG feedback rollouts of a.N candidates (one per wavefront); outputs and hand-on as launch_rollout_quad
hipError_t launch_feedback_quad(const void* model, const void* tables, const double* blob, const QBlob& bo, const QArgs& a, const QFeedback& fb, int* stats,
                                hipStream_t stream);
hipError_t launch_rollout_quad(const void* model, const void* tables, const double* blob, const QBlob& bo, const QArgs& a, int* stats, hipStream_t stream);
} }
