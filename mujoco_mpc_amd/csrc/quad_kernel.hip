// quad_kernel.hip -- translation unit of the quad kernel family (quad_kernel.h / quad_step.h).
#include "quad_kernel.h"
#include "quad_launch.h"

namespace mjpcx { namespace quad {
std::string build_images(const mjpcx_model* m, const mjpcx_task* t, std::vector<unsigned char>& model, std::vector<unsigned char>& tables) {
  model.assign(sizeof(QuadModel), 0);
  tables.assign(sizeof(QuadTables), 0);
  return quad_build(m, t, reinterpret_cast<QuadModel*>(model.data()), reinterpret_cast<QuadTables*>(tables.data()));
}
static int pick_cpw(int N, int cpw) {
  // candidates per wavefront: as many as it takes to give every SIMD of the 256 CUs one wavefront, 16 at most and 4 at least (cpw > 0: the
  // caller's choice). Below four the lock-step no longer shrinks but the wavefronts multiply, and co-resident wavefronts cost each other more
  // than that buys (a rank's 2048-candidate share of configs[2], gait steps, same box: 39.7 ms at 4 per wavefront = 512 wavefronts, 58.3 ms at 2 = 1024)
  if (cpw <= 0) { cpw = 16; while (cpw > 4 && (N + cpw - 1) / cpw < 1024) cpw >>= 1; }
  return cpw;
}
int quad_waves(int N, int cpw) { cpw = pick_cpw(N, cpw); return (N + cpw - 1) / cpw; }
#ifdef QEXP_OVF_SLAB
bool quad_uses_ovf_slab() { return true; }
#else
bool quad_uses_ovf_slab() { return false; }
#endif
size_t quad_ovf_doubles_per_wave() { return (size_t)(kQMaxCon - kQLdsSlots) * kQConRec * 64; }
hipError_t launch_feedback_quad(const void* model_, const void* tables_, const double* blob, const QBlob& bo, const QArgs& a, const QFeedback& fb, int* stats,
                                hipStream_t stream) {
  const QuadModel* model = static_cast<const QuadModel*>(model_);
  const QuadTables* tables = static_cast<const QuadTables*>(tables_);
  QArgs q = a;
  q.cpw = 1;  // one candidate per wavefront, one wavefront per workgroup: the rollouts of an iLQG iteration are latency, not throughput
  hipError_t e = hipFuncSetAttribute((const void*)rollout_feedback_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * kQWaveLds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(rollout_feedback_quad_kernel, dim3(a.N), dim3(64), kQWaveLds, stream, model, tables, blob, bo, q, fb, stats);
  return hipGetLastError();
}
hipError_t launch_rollout_quad(const void* model_, const void* tables_, const double* blob, const QBlob& bo, const QArgs& a, int* stats, hipStream_t stream) {
  const QuadModel* model = static_cast<const QuadModel*>(model_);
  const QuadTables* tables = static_cast<const QuadTables*>(tables_);
  // four wavefronts per workgroup (one per SIMD of a CU, sharing one model image) once every CU has one
  QArgs q = a;
  q.cpw = pick_cpw(a.N, a.cpw);
  const int waves = (a.N + q.cpw - 1) / q.cpw;
  // (below 1024 wavefronts W = 4 would leave CUs without a workgroup: 512 waves in 128 workgroups on 256 CUs -- ADVICE r05; one wavefront
  // per workgroup spreads a 2048-candidate share over 512 CUs' worth of slots instead)
  const int W = waves >= 4 * 256 ? 4 : 1;
  const size_t lds = W * kQWaveLds;
  // (the opt-in to more than 64 KB of dynamic LDS is per device and costs nothing next to a 60 ms launch: set on every launch, like the other launchers)
  hipError_t e = hipFuncSetAttribute((const void*)rollout_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * kQWaveLds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(rollout_quad_kernel, dim3((waves + W - 1) / W), dim3(64 * W), lds, stream, model, tables, blob, bo, q, stats);
  return hipGetLastError();
}
} }
