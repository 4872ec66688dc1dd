// ilqg_wave.hip -- the iLQG kernels of the wavefront-per-candidate family (wave_ilqg.h: finite-difference sweep, feedback rollouts,
// the kinematics read-back) as a translation unit of their own; mjpcx.hip reaches them through wave_ilqg_launch.h.
#include <hip/hip_runtime.h>

#include <algorithm>

#define MJPCX_WITH_ILQG_WAVE_KERNELS 1
#include "rollout_wave.h"
#include "wave_ilqg_launch.h"

namespace mjpcx {

hipError_t launch_feedback_wave(const WaveModel& m, const WaveTask& wt, const RolloutArgs<double>& a, const w64::FeedbackWaveArgs& fb, int N,
                                size_t lds, bool tree, bool rk4, const void* image, size_t blob_bytes, hipStream_t stream) {
  hipError_t e;
  const size_t fixed = LdsLayout<TreeCfgA1, double>::kBytes + blob_bytes;
  if (image && tree && !rk4 && fixed + lds <= 160 * 1024) {
    // registered model: image + blob + one arena per workgroup (the launch is a handful of wavefronts: nothing to share an image between)
    auto reg = w64::rollout_feedback_tree_kernel<TreeCfgA1>;
    if ((e = hipFuncSetAttribute((const void*)reg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fixed + lds))) != hipSuccess) return e;
    hipLaunchKernelGGL(reg, dim3(N), dim3(64), fixed + lds, stream, m, wt, a, fb, (const unsigned char*)image, (unsigned)blob_bytes);
    return hipGetLastError();
  }
  // (one NMAX = 32 instantiation per family carries mj_RungeKutta)
  auto kern = rk4 ? (tree ? w64::rollout_feedback_wave_kernel<32, true, true> : w64::rollout_feedback_wave_kernel<32, false, true>)
            : tree ? (m.nv <= 18 ? w64::rollout_feedback_wave_kernel<18, true> : w64::rollout_feedback_wave_kernel<32, true>)
            : m.nv <= 20 ? w64::rollout_feedback_wave_kernel<20> : w64::rollout_feedback_wave_kernel<32>;
  if ((e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(N), dim3(64), lds, stream, m, wt, a, fb);
  return hipGetLastError();
}

hipError_t launch_transition_fd_wave(const WaveModel& m, const WaveTask& wt, const w64::FdWaveArgs& f, unsigned items, size_t lds, bool tree,
                                     bool rk4, hipStream_t stream) {
  auto kern = rk4 ? (tree ? w64::transition_fd_wave_kernel<32, true, true> : w64::transition_fd_wave_kernel<32, false, true>)
            : tree ? (m.nv <= 18 ? w64::transition_fd_wave_kernel<18, true> : w64::transition_fd_wave_kernel<32, true>)
            : m.nv <= 20 ? w64::transition_fd_wave_kernel<20> : w64::transition_fd_wave_kernel<32>;
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(items), dim3(64), lds, stream, m, wt, f);
  return hipGetLastError();
}

hipError_t launch_fd_tangent(const WaveModel& m, const double* next, double* tan, int Tn, int ncol, hipStream_t stream) {
  const size_t items = (size_t)Tn * ncol;
  hipLaunchKernelGGL(w64::fd_tangent_kernel, dim3((unsigned)std::min<size_t>((items + 63) / 64, 1024)), dim3(64), 0, stream, m, next, tan, Tn, ncol);
  return hipGetLastError();
}

hipError_t launch_kinematics_wave(const WaveModel& m, const WaveTask& wt, double* out, int nb_model, int ns_model, size_t lds, hipStream_t stream) {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)w64::kinematics_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  hipLaunchKernelGGL(w64::kinematics_wave_kernel, dim3(1), dim3(64), lds, stream, m, wt, out, nb_model, ns_model);
  return hipGetLastError();
}

}  // namespace mjpcx
