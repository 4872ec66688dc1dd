// wave_ilqg_launch.h -- host-side entry of the translation unit that holds the iLQG kernels of the wavefront-per-candidate family
// (ilqg_wave.hip: wave_ilqg.h instantiated there and nowhere else), so that they compile next to the rollout kernels of mjpcx.hip
// instead of after them -- the two halves take about as long as each other.
#pragma once
#include <hip/hip_runtime.h>

#include "rollout_lane.h"  // RolloutArgs
#include "wave_model.h"

namespace mjpcx {
namespace w64 {
struct FdWaveArgs {
  const double *times, *states, *actions;  // [Tn], [Tn][nq+nv], [Tn][nu]
  int Tn, ncol;                            // ncol = 1 + 2 (2 nv + nu): 0 nominal | +x_j | -x_j | +u_k | -u_k
  double eps;
  double* next;    // [Tn][ncol][nq+nv] raw next states
  double* sensor;  // [Tn][ncol][nr]
};
struct FeedbackWaveArgs {
  const double *times, *states, *actions, *gains, *improvement, *alpha;  // as FeedbackArgs (ilqg_kernels.h)
  int Tn, mode, representation, use_state;
  int only_flagged;  // roll out only the candidates whose failure[] carries kQFallback: the ones rollout_feedback_quad_kernel handed on
};
}  // namespace w64

// kernel selection by the model's family (tree: the Jacobian-free forward pass), integrator and width; image != nullptr: the registered
// A1 (tree_registry.h), whose feedback rollouts run on its LDS image. lds: one arena (wave_lds_bytes of the caller).
hipError_t launch_feedback_wave(const WaveModel& m, const WaveTask& wt, const RolloutArgs<double>& a, const w64::FeedbackWaveArgs& fb, int N,
                                size_t lds, bool tree, bool rk4, const void* image, size_t blob_bytes, hipStream_t stream);
hipError_t launch_transition_fd_wave(const WaveModel& m, const WaveTask& wt, const w64::FdWaveArgs& f, unsigned items, size_t lds, bool tree,
                                     bool rk4, hipStream_t stream);
hipError_t launch_fd_tangent(const WaveModel& m, const double* next, double* tan, int Tn, int ncol, hipStream_t stream);
hipError_t launch_kinematics_wave(const WaveModel& m, const WaveTask& wt, double* out, int nb_model, int ns_model, size_t lds, hipStream_t stream);
}  // namespace mjpcx
