// MFMA weight-gradient, bf16 instantiations (ds_read_b64_tr_b16 + v_mfma_f32_16x16x32_bf16).
#include "wgrad_impl.h"
#include "wgrad_ws_impl.h"
namespace segmi {

// ct < 0: the wave-specialised kernel was chosen (wgrad_ws_gx); tile shapes as listed there
static int launch_wgrad_ws(const WgradParams& p, int stride, int ct, int gx, hipStream_t st) {
  const bool wide = p.Wy > 8;
  if (stride == 1) {
    if (ct == 11) return wide ? launch_wgrad_ws_cfg<3, 1, 1, 1, 4, 8, 16>(p, gx, st)
                              : launch_wgrad_ws_cfg<3, 1, 1, 1, 4, 8, 8>(p, gx, st);
    return wide ? launch_wgrad_ws_cfg<3, 1, 2, 1, 2, 8, 16>(p, gx, st)
                : launch_wgrad_ws_cfg<3, 1, 2, 1, 4, 8, 8>(p, gx, st);
  }
  if (ct == 11) return wide ? launch_wgrad_ws_cfg<3, 2, 1, 1, 2, 4, 16>(p, gx, st)
                            : launch_wgrad_ws_cfg<3, 2, 1, 1, 2, 8, 8>(p, gx, st);
  return wide ? launch_wgrad_ws_cfg<3, 2, 2, 1, 2, 4, 16>(p, gx, st)
              : launch_wgrad_ws_cfg<3, 2, 2, 1, 2, 8, 8>(p, gx, st);
}

int wgrad_mfma_bf16(const WgradParams& p, int ksize, int stride, int ct, int gx, hipStream_t st) {
  if (ct < 0) return launch_wgrad_ws(p, stride, -ct, gx, st);
  return launch_wgrad_mfma_t<bf16_t>(p, ksize, stride, ct, gx, st);
}
}  // namespace segmi
