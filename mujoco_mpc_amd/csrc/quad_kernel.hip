// quad_kernel.hip -- translation unit of the quad kernel family (quad_kernel.h / quad_step.h).
#include "quad_kernel.h"
#include "quad_launch.h"

namespace mjpcx { namespace quad {
hipError_t launch_rollout_quad(const QuadModel* model, const QuadTables* tables, const double* blob, const QBlob& bo, const QArgs& a, int* stats,
                               hipStream_t stream) {
  // four wavefronts (64 candidates) per workgroup once every CU has one; single-wavefront workgroups for smaller batches
  const int W = a.N >= 64 * 128 ? 4 : 1;
  const size_t lds = W * kQWaveLds;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)rollout_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * kQWaveLds));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(rollout_quad_kernel, dim3((a.N + 16 * W - 1) / (16 * W)), dim3(64 * W), lds, stream, model, tables, blob, bo, a, stats);
  return hipGetLastError();
}
} }
