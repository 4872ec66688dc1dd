// limb_kernel.hip -- the limb kernel family's own translation unit (limb_kernel.h, limb_step.h, limb_model.h): compiled apart from
// mjpcx.hip so that a change of the limb step does not re-compile the other kernels.
#include "limb_launch.h"
#include "limb_kernel.h"

namespace mjpcx { namespace limb {
std::string build_images(const mjpcx_model* m, const mjpcx_task* t, std::vector<unsigned char>& image32, std::vector<unsigned char>& image64) {
  LimbModelD* d = new LimbModelD;
  const std::string why = limb_build(m, t, d);
  if (why.empty()) {
    image32.assign(sizeof(LimbModelT<float>), 0);
    limb_cast(*d, *reinterpret_cast<LimbModelT<float>*>(image32.data()));
    image64.assign(sizeof(LimbModelD), 0);
    memcpy(image64.data(), d, sizeof(LimbModelD));
  }
  delete d;
  return why;
}
static int pick_cpw(int N, int cpw) {
  if (cpw == 1 || cpw == 2 || cpw == 4 || cpw == 8 || cpw == 16) return cpw;
  // one wavefront per SIMD of the 256 CUs first (the lock-step is then over fewer candidates), sixteen per wavefront from 16384 candidates on
  int c = 1;
  while (c < 16 && (N + c - 1) / c > 1024) c *= 2;
  return c;
}
int limb_waves(int N, int cpw) { const int c = pick_cpw(N, cpw); return (N + c - 1) / c; }
template <typename R, int LS>
static hipError_t launch_ls(const void* image, const R* blob, const LBlob& bo, const LArgs<R>& q, const R* key_mpos, int* stats, hipStream_t stream) {
  const int waves = (q.N + q.cpw - 1) / q.cpw;
  constexpr size_t img = ((sizeof(LimbModelT<R>) + 15) & ~(size_t)15) + ((3 * kMaxTerm * sizeof(R) + 15) & ~(size_t)15), per_wave = wave_reals(LS) * sizeof(R);
  int W = (int)((160 * 1024 - img) / per_wave);
  if (W < 1) return hipErrorInvalidValue;
  if (W > 4) W = 4;
  if (W == 3) W = 2;
  while (W > 1 && (waves + W - 1) / W < 256) W >>= 1;  // (every CU gets a workgroup before a workgroup gets a second wavefront)
  const size_t lds = img + (size_t)W * per_wave;
  auto kern = rollout_limb_kernel<R, LS>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((waves + W - 1) / W), dim3(64 * W), lds, stream, static_cast<const LimbModelT<R>*>(image), blob, bo, q, key_mpos, stats);
  return hipGetLastError();
}
template <typename R>
static hipError_t launch(const void* image, const R* blob, const LBlob& bo, const LArgs<R>& a, const R* key_mpos, int* stats, hipStream_t stream) {
  LArgs<R> q = a;
  q.cpw = pick_cpw(a.N, a.cpw);
  // up to eight candidates per wavefront: the half-width layout (half the LDS per wavefront: four wavefronts per CU)
  if (q.cpw <= 4) return launch_ls<R, 16>(image, blob, bo, q, key_mpos, stats, stream);
  if (q.cpw <= 8) return launch_ls<R, 32>(image, blob, bo, q, key_mpos, stats, stream);
  return launch_ls<R, 64>(image, blob, bo, q, key_mpos, stats, stream);
}
hipError_t launch_rollout_limb(const void* image, const float* blob, const LBlob& bo, const LArgs<float>& a, const float* key_mpos, int* stats, hipStream_t stream) {
  return launch<float>(image, blob, bo, a, key_mpos, stats, stream);
}
hipError_t launch_rollout_limb(const void* image, const double* blob, const LBlob& bo, const LArgs<double>& a, const double* key_mpos, int* stats, hipStream_t stream) {
#ifdef LIMB_F32_ONLY   // (tuning builds: the fp32 kernel alone compiles in half the time)
  return hipErrorNotSupported;
#else
  return launch<double>(image, blob, bo, a, key_mpos, stats, stream);
#endif
}
} }
