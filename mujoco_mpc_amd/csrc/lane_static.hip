// lane_static.hip -- rollout_lane_kernel instantiations specialised for the registered models'
// numeric constants (generated/static_models.h). Built with -fno-signed-zeros -ffinite-math-only:
// with every model constant an immediate, x*0 -> 0 and x+0 -> x fold, which removes most of the
// generic kinematics / inertia arithmetic for axis-aligned models (no reassociation, no approximate
// functions: surviving operations are evaluated exactly as in the runtime-constant kernel).
// NaN/Inf detection in the kernel is an integer bit test (is_bad), so it is unaffected.
#include "lane_registry.h"

namespace mjpcx {
#define MJPCX_DEFINE_STATIC(FN, TP, TK, GEN)                                                                        \
  hipError_t FN##_f64(const LaneModel<double>& m, const LaneTask<double>& tk, const RolloutArgs<double>& a, hipStream_t s) { \
    return launch_lane_impl<TP, TK, double, StaticModel<GEN>>(m, tk, a, s);                                         \
  }                                                                                                                 \
  hipError_t FN##_f32(const LaneModel<float>& m, const LaneTask<float>& tk, const RolloutArgs<float>& a, hipStream_t s) {    \
    return launch_lane_impl<TP, TK, float, StaticModel<GEN>>(m, tk, a, s);                                          \
  }                                                                                                                 \
  hipError_t FN##_fb_f64(const LaneModel<double>& m, const LaneTask<double>& tk, const RolloutArgs<double>& a,      \
                         const FeedbackArgs<double>& fb, hipStream_t s) {                                           \
    return launch_feedback_impl<TP, TK, double, StaticModel<GEN>>(m, tk, a, fb, s);                                 \
  }                                                                                                                 \
  hipError_t FN##_fb_f32(const LaneModel<float>& m, const LaneTask<float>& tk, const RolloutArgs<float>& a,         \
                         const FeedbackArgs<float>& fb, hipStream_t s) {                                            \
    return launch_feedback_impl<TP, TK, float, StaticModel<GEN>>(m, tk, a, fb, s);                                  \
  }                                                                                                                 \
  hipError_t FN##_fd_f64(const LaneModel<double>& m, const LaneTask<double>& tk, const FdArgs<double>& f, hipStream_t s) { \
    return launch_fd_impl<TP, TK, double, StaticModel<GEN>>(m, tk, f, s);                                           \
  }                                                                                                                 \
  hipError_t FN##_fd_f32(const LaneModel<float>& m, const LaneTask<float>& tk, const FdArgs<float>& f, hipStream_t s) {    \
    return launch_fd_impl<TP, TK, float, StaticModel<GEN>>(m, tk, f, s);                                            \
  }
MJPCX_DEFINE_STATIC(launch_static_cartpole, TopoCartpole, TaskCartpole, StaticCartpole)
MJPCX_DEFINE_STATIC(launch_static_particle, TopoParticle, TaskParticle, StaticParticle)
MJPCX_DEFINE_STATIC(launch_static_particle_copy, TopoParticle, TaskParticleCopy, StaticParticle)
}  // namespace mjpcx
