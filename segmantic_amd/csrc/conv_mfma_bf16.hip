// MFMA conv / transposed-conv forward, bf16 instantiations (v_mfma_f32_16x16x32_bf16).
#include "conv_fwd_impl.h"
#include "conv_ring2_impl.h"
#include "conv_ring3_impl.h"
#include "conv_ks_impl.h"
#include "convt_ps_impl.h"
#include "conv_bnbwd_impl.h"
namespace segmi {
int conv_mfma_bf16(const ConvParams& p, int ksize, int stride, hipStream_t st) {
  if (conv_ring_zsplit(SEGMI_BF16, p.Cin, ksize, stride, p.N, p.Do, p.Ho, p.Wo) > 0)
    return conv_ring3_ok(p) ? launch_conv_ring3(p, st) : launch_conv_ring2(p, st);
  if (conv_ks_ok(SEGMI_BF16, p.Cin, ksize, stride)) return launch_conv_ks_t<bf16_t, 32>(p, stride, st);
  return launch_conv_mfma_t<bf16_t>(p, ksize, stride, st);
}
int conv_s2_bnbwd_bf16(const ConvBnBwdParams& p, hipStream_t st) {
  switch (p.Cout / 16) {
    case 1: return launch_conv_s2_bnbwd<1>(p, st);
    case 2: return launch_conv_s2_bnbwd<2>(p, st);
    case 4: return launch_conv_s2_bnbwd<4>(p, st);
  }
  SEGMI_UNSUPPORTED("bn_act_bwd_apply_conv: %d output channels", p.Cout);
}
int convt_mfma_bf16(const ConvTParams& p, hipStream_t st) {
  if (convt_ps_ok(SEGMI_BF16, p.Cin, p.Cout, p.Wi)) return launch_convt_ps_t<bf16_t>(p, st);
  return launch_convt_mfma_t<bf16_t>(p, st);
}
}  // namespace segmi
