// lane_registry.h -- the registered static topologies of the lane-per-candidate kernel family and
// the launchers of the specialised (compile-time model constants) instantiations, which live in
// their own translation unit (lane_static.hip) built with -fno-signed-zeros -ffinite-math-only so
// that arithmetic on exact-zero model constants folds away.
#pragma once
#include <cstdlib>
#include "../../include/mjpcx.h"
#include "generated/static_models.h"
#include "ilqg_kernels.h"
#include "rollout_lane.h"

namespace mjpcx {

constexpr uint64_t pack4() { return 0; }
template <typename... R> constexpr uint64_t pack4(int a, R... r) { return (uint64_t)(a & 15) | (pack4(r...) << 4); }
constexpr uint64_t pack2() { return 0; }
template <typename... R> constexpr uint64_t pack2(int a, R... r) { return (uint64_t)(a & 3) | (pack2(r...) << 2); }

// Cart-pole (mjpc/tasks/cartpole): world -> cart[slide x, limited] -> pole_1[hinge y]; site tip on pole
using TopoCartpole = Topo</*NB*/3, /*NV*/2, /*NU*/1, /*NSITE*/1, /*NMOCAP*/0,
                          /*parent*/pack4(0, 0, 1), /*mocap*/pack4(15, 15, 15), /*jtype*/pack2(kJntSlide, kJntHinge),
                          /*jbody*/pack4(1, 2), /*jlimited*/0x1, /*actj*/pack4(0), /*siteb*/pack4(2)>;
using TaskCartpole = TaskTopo<MJPCX_RESIDUAL_CARTPOLE, 4, 4, pack4(1, 1, 1, 1), 1, pack4(0)>;
// Particle (mjpc/test/testdata/particle.xml): world -> goal[mocap]; world -> pointmass[slide x, slide y]
using TopoParticle = Topo<3, 2, 2, 1, 1, pack4(0, 0, 0), pack4(15, 0, 15), pack2(kJntSlide, kJntSlide),
                          pack4(2, 2), 0x3, pack4(0, 1), pack4(2)>;
using TaskParticle = TaskTopo<MJPCX_RESIDUAL_PARTICLE, 4, 2, pack4(2, 2), 1, pack4(0)>;
using TaskParticleCopy = TaskTopo<MJPCX_RESIDUAL_PARTICLE_COPY, 4, 2, pack4(2, 2), 1, pack4(0)>;

template <class TP, class TK, typename T, class MC>
hipError_t launch_lane_impl(const LaneModel<T>& m, const LaneTask<T>& tk, const RolloutArgs<T>& a, hipStream_t s) {
  static_assert(sizeof(LaneModel<T>) + sizeof(LaneTask<T>) + sizeof(RolloutArgs<T>) <= 4096, "kernarg segment is 4 KiB");
  const int blocks = (a.N + 63) / 64;
  const size_t shmem = ((size_t)a.P * TP::NU * 64 + a.P) * sizeof(T);
  // One fused launch or three (see rollout_lane_kernel). The split form wins while the time loop is issue-bound on a few
  // wavefronts (N = 4096: 166 vs 280 us); its sensor-stage launch re-reads the recorded states and runs at the HBM roofline
  // (N = 65536: 143 us at 6.1 TB/s), so beyond ~10^5 candidates the fused form, with 30 % less traffic, is the faster one
  // (N = 262144: 0.91 vs 1.38 ms). MJPCX_LANE_FUSED / MJPCX_LANE_SPLIT force one form for A/B measurements.
  static const bool force_fused = std::getenv("MJPCX_LANE_FUSED") != nullptr, force_split = std::getenv("MJPCX_LANE_SPLIT") != nullptr;
  const bool fused = force_fused || (!force_split && a.N > 98304);
  if (fused && !(a.xfrc_scale > 0)) {
    hipLaunchKernelGGL((rollout_lane_kernel<TP, TK, T, MC, false>), dim3(blocks), dim3(64), shmem, s, m, tk, a);
    return hipGetLastError();
  }
  // time loop (dynamics only) -> sensor stage of every (step, candidate) -> ordered returns; see rollout_lane_kernel
  if (a.xfrc_scale > 0) hipLaunchKernelGGL((rollout_lane_kernel<TP, TK, T, MC, true, true>), dim3(blocks), dim3(64), shmem, s, m, tk, a);
  else hipLaunchKernelGGL((rollout_lane_kernel<TP, TK, T, MC, true>), dim3(blocks), dim3(64), shmem, s, m, tk, a);
  const size_t items = (size_t)a.N * a.H;
  hipLaunchKernelGGL((cost_lane_kernel<TP, TK, T, MC>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, m, tk, a);
  hipLaunchKernelGGL((return_lane_kernel<T>), dim3(blocks), dim3(64), 0, s, a);
  return hipGetLastError();
}

template <class TP, class TK, typename T, class MC>
hipError_t launch_feedback_impl(const LaneModel<T>& m, const LaneTask<T>& tk, const RolloutArgs<T>& a,
                                const FeedbackArgs<T>& fb, hipStream_t s) {
  static_assert(sizeof(LaneModel<T>) + sizeof(LaneTask<T>) + sizeof(RolloutArgs<T>) + sizeof(FeedbackArgs<T>) <= 4096, "kernarg");
  const int blocks = (a.N + 63) / 64;
  const size_t shmem = (size_t)fb.Tn * (1 + 2 * TP::NV + 2 * TP::NU + TP::NU * 2 * TP::NV) * sizeof(T);
  hipLaunchKernelGGL((rollout_feedback_kernel<TP, TK, T, MC>), dim3(blocks), dim3(64), shmem, s, m, tk, a, fb);
  return hipGetLastError();
}
template <class TP, class TK, typename T, class MC>
hipError_t launch_fd_impl(const LaneModel<T>& m, const LaneTask<T>& tk, const FdArgs<T>& f, hipStream_t s) {
  const int items = f.Tn * fd_columns<TP>();
  hipLaunchKernelGGL((transition_fd_kernel<TP, TK, T, MC>), dim3((items + 63) / 64), dim3(64), 0, s, m, tk, f);
  return hipGetLastError();
}

#define MJPCX_DECLARE_STATIC(FN)                                                                             \
  hipError_t FN##_f64(const LaneModel<double>&, const LaneTask<double>&, const RolloutArgs<double>&, hipStream_t); \
  hipError_t FN##_f32(const LaneModel<float>&, const LaneTask<float>&, const RolloutArgs<float>&, hipStream_t); \
  hipError_t FN##_fb_f64(const LaneModel<double>&, const LaneTask<double>&, const RolloutArgs<double>&, const FeedbackArgs<double>&, hipStream_t); \
  hipError_t FN##_fb_f32(const LaneModel<float>&, const LaneTask<float>&, const RolloutArgs<float>&, const FeedbackArgs<float>&, hipStream_t); \
  hipError_t FN##_fd_f64(const LaneModel<double>&, const LaneTask<double>&, const FdArgs<double>&, hipStream_t); \
  hipError_t FN##_fd_f32(const LaneModel<float>&, const LaneTask<float>&, const FdArgs<float>&, hipStream_t);
MJPCX_DECLARE_STATIC(launch_static_cartpole)
MJPCX_DECLARE_STATIC(launch_static_particle)
MJPCX_DECLARE_STATIC(launch_static_particle_copy)

}  // namespace mjpcx
