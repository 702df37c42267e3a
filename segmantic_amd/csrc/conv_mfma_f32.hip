// MFMA conv / transposed-conv forward, exact-f32 instantiations (v_mfma_f32_16x16x4_f32).
#include "conv_fwd_impl.h"
#include "conv_ring_impl.h"
#include "conv_ks_impl.h"
#include "convt_ps_impl.h"
namespace segmi {
int conv_mfma_f32(const ConvParams& p, int ksize, int stride, hipStream_t st) {
  if (conv_ks_ok(SEGMI_F32, p.Cin, ksize, stride)) return launch_conv_ks_t<float, 16>(p, stride, st);
  return launch_conv_mfma_t<float>(p, ksize, stride, st);
}
int convt_mfma_f32(const ConvTParams& p, hipStream_t st) {
  if (convt_ps_ok(SEGMI_F32, p.Cin, p.Cout, p.Wi)) return launch_convt_ps_t<float>(p, st);
  return launch_convt_mfma_t<float>(p, st);
}
}  // namespace segmi
