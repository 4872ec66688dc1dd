// quad_launch.h -- host-side entry of the quad kernel's translation unit (quad_kernel.hip), so that mjpcx.hip does not
// re-compile the wavefront-per-candidate kernels when the quad step changes (and vice versa).
#pragma once
#include <hip/hip_runtime.h>

#include "quad_model.h"

namespace mjpcx { namespace quad {
// stats: nullptr, or 8 ints (zeroed by the caller): [0] candidates handed on, [1..5] by reason (quad_step.h kFlag*)
hipError_t launch_rollout_quad(const QuadModel* model, const QuadTables* tables, const double* blob, const QBlob& bo, const QArgs& a, int* stats,
                               hipStream_t stream);
} }
