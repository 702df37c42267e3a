// MFMA weight-gradient, exact-f32 instantiations.
#include "wgrad_impl.h"
namespace segmi {
int wgrad_mfma_f32(const WgradParams& p, int ksize, int stride, int ct, int gx, hipStream_t st) {
  return launch_wgrad_mfma_t<float>(p, ksize, stride, ct, gx, st);
}
}  // namespace segmi
