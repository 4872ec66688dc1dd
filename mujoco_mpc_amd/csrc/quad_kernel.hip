// quad_kernel.hip -- translation unit of the quad kernel family (quad_kernel.h / quad_step.h).
#include "quad_kernel.h"
#include "quad_launch.h"

namespace mjpcx { namespace quad {
std::string build_images(const mjpcx_model* m, const mjpcx_task* t, std::vector<unsigned char>& model, std::vector<unsigned char>& tables) {
  model.assign(sizeof(QuadModel), 0);
  tables.assign(sizeof(QuadTables), 0);
  return quad_build(m, t, reinterpret_cast<QuadModel*>(model.data()), reinterpret_cast<QuadTables*>(tables.data()));
}
hipError_t launch_rollout_quad(const void* model_, const void* tables_, const double* blob, const QBlob& bo, const QArgs& a, int* stats, hipStream_t stream) {
  const QuadModel* model = static_cast<const QuadModel*>(model_);
  const QuadTables* tables = static_cast<const QuadTables*>(tables_);
  // four wavefronts (64 candidates) per workgroup once every CU has one; single-wavefront workgroups for smaller batches
  const int W = a.N >= 64 * 128 ? 4 : 1;
  const size_t lds = W * kQWaveLds;
  // (the opt-in to more than 64 KB of dynamic LDS is per device and costs nothing next to a 60 ms launch: set on every launch, like the other launchers)
  hipError_t e = hipFuncSetAttribute((const void*)rollout_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * kQWaveLds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(rollout_quad_kernel, dim3((a.N + 16 * W - 1) / (16 * W)), dim3(64 * W), lds, stream, model, tables, blob, bo, a, stats);
  return hipGetLastError();
}
} }
