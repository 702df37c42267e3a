// MFMA weight-gradient, bf16 instantiations (ds_read_b64_tr_b16 + v_mfma_f32_16x16x32_bf16).
#include "wgrad_impl.h"
namespace segmi {
int wgrad_mfma_bf16(const WgradParams& p, int ksize, int stride, int ct, int gx, hipStream_t st) {
  return launch_wgrad_mfma_t<bf16_t>(p, ksize, stride, ct, gx, st);
}
}  // namespace segmi
