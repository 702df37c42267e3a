// conv.hip -- C entry points for Conv3d / ConvTranspose3d forward + the direct (non-MFMA)
// kernels used when a channel count is not a multiple of 16 (first layer Cin=1, K=3 heads,
// tiny test networks).
#include "conv_fwd_impl.h"
#include "conv_ring_impl.h"
#include "conv_ks_impl.h"
#include "convt_ps_impl.h"
#include "conv_bnbwd_impl.h"

namespace segmi {

int conv_s2_bnbwd_bf16(const ConvBnBwdParams& p, hipStream_t st);

int conv_mfma_f32(const ConvParams& p, int ksize, int stride, hipStream_t st);
int conv_mfma_bf16(const ConvParams& p, int ksize, int stride, hipStream_t st);
int convt_mfma_f32(const ConvTParams& p, hipStream_t st);
int convt_mfma_bf16(const ConvTParams& p, hipStream_t st);
int bn_stats_launch(int dtype, const segmi_act* x, float* partials, hipStream_t st, const BiasFin* bias_fin = nullptr);
int bn_stats_rows_for(const segmi_act* x);
int stats_reserve_rows();
// the separate finalisation launch with a segmi_bn_fin's arguments (generic / direct paths)
static inline int bn_finalize_with(const float* partials, int rows, int c, const segmi_bn_fin* f, hipStream_t st) {
  return segmi_bn_finalize(partials, rows, c, f->count, f->gamma, f->beta, f->running_mean, f->running_var,
                           f->momentum, f->eps, f->mean, f->invstd, f->scale, f->shift, st);
}
bool conv_small_ok(int cin, int cout, int ksize);
int conv_small_fwd(int dtype, const segmi_act* in, const segmi_act* out, const float* w,
                   const float* bias, const float* alpha, const segmi_act* res, float* stats,
                   int stride, hipStream_t st, const segmi_act* out2 = nullptr,
                   const float* w2 = nullptr, const float* bias2 = nullptr,
                   const segmi_bn_fin* fin = nullptr, const segmi_windows* win = nullptr);
int conv_small_fwd_rows(const segmi_act* out);

struct DirectParams {
  const void* in;
  void* out;
  const float* w;
  const float* bias;
  const float* alpha;
  const void* res;
  int N, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, ldi, ldo, ldr;
  int ks, stride, kind;
};

// one thread per (voxel, co); co fastest so stores coalesce along NDHWC rows.
// kind 0: w[co][ci][tap]; kind 1 (stride-1 dgrad): w[ci][co][flipped tap]
template <typename T>
__global__ void conv_direct_kernel(DirectParams p) {
  const int64_t total = (int64_t)p.N * p.Do * p.Ho * p.Wo * p.Cout;
  const int pad = (p.ks - 1) / 2, nt = p.ks * p.ks * p.ks;
  const T* in = (const T*)p.in;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int co = e % p.Cout;
    int64_t v = e / p.Cout;
    const int ox = v % p.Wo; v /= p.Wo;
    const int oy = v % p.Ho; v /= p.Ho;
    const int oz = v % p.Do;
    const int n = v / p.Do;
    float acc = 0.f;
    for (int kd = 0; kd < p.ks; ++kd) {
      const int z = oz * p.stride - pad + kd;
      if ((unsigned)z >= (unsigned)p.Di) continue;
      for (int kh = 0; kh < p.ks; ++kh) {
        const int y = oy * p.stride - pad + kh;
        if ((unsigned)y >= (unsigned)p.Hi) continue;
        for (int kw = 0; kw < p.ks; ++kw) {
          const int x = ox * p.stride - pad + kw;
          if ((unsigned)x >= (unsigned)p.Wi) continue;
          const int tap = (kd * p.ks + kh) * p.ks + kw;
          const T* ip = in + ((((int64_t)n * p.Di + z) * p.Hi + y) * p.Wi + x) * p.ldi;
          if (p.kind == 0) {
            const float* wp = p.w + ((int64_t)co * p.Cin) * nt + tap;
            for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(Elem<T>::ld(ip + ci), wp[(int64_t)ci * nt], acc);
          } else {
            const float* wp = p.w + (int64_t)co * nt + (nt - 1 - tap);
            for (int ci = 0; ci < p.Cin; ++ci)
              acc = fmaf(Elem<T>::ld(ip + ci), wp[(int64_t)ci * p.Cout * nt], acc);
          }
        }
      }
    }
    if (p.bias) acc += p.bias[co];
    if (p.alpha) acc = acc > 0.f ? acc : (*p.alpha) * acc;
    const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
    if (p.res) acc += Elem<T>::ld((const T*)p.res + vox * p.ldr + co);
    Elem<T>::st((T*)p.out + vox * p.ldo + co, acc);
  }
}

// transposed conv k3 s2 p1, w[ci][co][27]
template <typename T>
__global__ void convt_direct_kernel(DirectParams p) {
  const int64_t total = (int64_t)p.N * p.Do * p.Ho * p.Wo * p.Cout;
  const T* in = (const T*)p.in;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int co = e % p.Cout;
    int64_t v = e / p.Cout;
    const int ox = v % p.Wo; v /= p.Wo;
    const int oy = v % p.Ho; v /= p.Ho;
    const int oz = v % p.Do;
    const int n = v / p.Do;
    float acc = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
      const int tz = oz + 1 - kd;
      if (tz < 0 || (tz & 1) || (tz >> 1) >= p.Di) continue;
      for (int kh = 0; kh < 3; ++kh) {
        const int ty = oy + 1 - kh;
        if (ty < 0 || (ty & 1) || (ty >> 1) >= p.Hi) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int tx = ox + 1 - kw;
          if (tx < 0 || (tx & 1) || (tx >> 1) >= p.Wi) continue;
          const int tap = (kd * 3 + kh) * 3 + kw;
          const T* ip = in + ((((int64_t)n * p.Di + (tz >> 1)) * p.Hi + (ty >> 1)) * p.Wi + (tx >> 1)) * p.ldi;
          const float* wp = p.w + (int64_t)co * 27 + tap;
          for (int ci = 0; ci < p.Cin; ++ci)
            acc = fmaf(Elem<T>::ld(ip + ci), wp[(int64_t)ci * p.Cout * 27], acc);
        }
      }
    }
    if (p.bias) acc += p.bias[co];
    if (p.alpha) acc = acc > 0.f ? acc : (*p.alpha) * acc;
    const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
    if (p.res) acc += Elem<T>::ld((const T*)p.res + vox * p.ldr + co);
    Elem<T>::st((T*)p.out + vox * p.ldo + co, acc);
  }
}

static inline bool mfma_ok(int cin, int cout) { return cin % 16 == 0 && cout % 16 == 0; }

static inline int out_extent(int in, int ks, int stride) {
  const int pad = (ks - 1) / 2;
  return (in + 2 * pad - ks) / stride + 1;
}

}  // namespace segmi

using namespace segmi;

// the first-layer kernel (conv_small.hip) takes the layer: shared by the launch and the rows query
static inline bool small_fwd_eligible(int dtype, const segmi_act* in, const segmi_act* out, int ksize) {
  const int es = dtype_size(dtype);
  return conv_small_ok(in->c, out->c, ksize) && out->ld % 4 == 0 &&
         ((uintptr_t)out->data % (4 * es)) == 0;
}

extern "C" {

int segmi_conv3d_stats_rows(int dtype, const segmi_act* in, const segmi_act* out, int ksize,
                            int stride) {
  if (!in || !out) return 0;
  if (mfma_ok(in->c, out->c)) {
    if (conv_ring_ok(dtype, in->c, ksize, stride, out))
      return conv_ring_rows(dtype, in->c, out) + stats_reserve_rows();
    if (conv_ks_ok(dtype, in->c, ksize, stride)) return conv_ks_rows(out) + stats_reserve_rows();
    return conv_mfma_rows(out, stride) + stats_reserve_rows();
  }
  if (small_fwd_eligible(dtype, in, out, ksize)) return conv_small_fwd_rows(out) + stats_reserve_rows();
  return bn_stats_rows_for(out) + stats_reserve_rows();
}

int segmi_conv3d_pair_ok(int dtype, const segmi_act* in, const segmi_act* out_a,
                         const segmi_act* out_b) {
  if (!act_ok(in) || !act_ok(out_a) || !act_ok(out_b)) return 0;
  if (dtype != SEGMI_F32 && dtype != SEGMI_BF16) return 0;
  return !mfma_ok(in->c, out_a->c) && small_fwd_eligible(dtype, in, out_a, 3) &&
         small_fwd_eligible(dtype, in, out_b, 3) && out_a->c == out_b->c && out_a->n == out_b->n &&
         out_a->d == out_b->d && out_a->h == out_b->h && out_a->w == out_b->w;
}

int segmi_conv3d_fwd_pair(int dtype, const segmi_act* in, const segmi_act* out_a, const float* w_a,
                          const float* bias_a, const float* prelu_alpha_a, float* stats_partials_a,
                          const segmi_act* out_b, const float* w_b, const float* bias_b, int stride,
                          const segmi_bn_fin* stats_fin_a, const segmi_windows* windows, void* stream) {
  SEGMI_CHECK_ARG(!stats_fin_a || (stats_partials_a && bn_fin_ok(stats_fin_a)),
                  "conv3d_fwd_pair: stats_fin_a needs stats_partials_a and its output pointers");
  if (windows) {
    const int es = dtype_size(dtype);
    bool ok = in && in->c == 1 && in->ld == 1 && windows->count == in->n && windows->count >= 1 && windows->count <= 32 &&
              windows->row_stride >= in->w && windows->row_stride % 4 == 0 && in->w % 4 == 0 &&
              windows->plane_stride >= (int64_t)windows->row_stride * in->h && windows->plane_stride % 4 == 0 &&
              ((uintptr_t)in->data % (4 * es)) == 0;
    for (int i = 0; ok && i < windows->count; ++i) ok = windows->offset[i] >= 0 && windows->offset[i] % 4 == 0;
    SEGMI_CHECK_ARG(ok, "conv3d_fwd_pair: bad window views (single channel, <= 32 windows, 4-element aligned "
                        "offsets and strides)");
  }
  SEGMI_CHECK_ARG(segmi_conv3d_pair_ok(dtype, in, out_a, out_b),
                  "conv3d_fwd_pair: not a small-Cin k3 pair (ask segmi_conv3d_pair_ok first)");
  SEGMI_CHECK_ARG(w_a && w_b && (stride == 1 || stride == 2), "conv3d_fwd_pair: bad arguments");
  SEGMI_CHECK_ARG(in->n == out_a->n && out_a->d == out_extent(in->d, 3, stride) &&
                      out_a->h == out_extent(in->h, 3, stride) && out_a->w == out_extent(in->w, 3, stride),
                  "conv3d_fwd_pair: output extent does not match the input");
  SEGMI_CHECK_ARG(!(stats_partials_a && prelu_alpha_a),
                  "conv3d_fwd_pair: fused statistics are taken before PReLU");
  return conv_small_fwd(dtype, in, out_a, w_a, bias_a, prelu_alpha_a, nullptr, stats_partials_a,
                        stride, (hipStream_t)stream, out_b, w_b, bias_b, stats_fin_a, windows);
}

int segmi_conv3d_split_act_ok(int dtype, const segmi_act* in, const segmi_act* out, int ksize,
                              int stride) {
  if (!act_ok(in) || !act_ok(out) || (dtype != SEGMI_F32 && dtype != SEGMI_BF16)) return 0;
  // the tile kernel must take the layer (it is the one that honours act_tiles): k3 stride 2 MFMA
  return mfma_ok(in->c, out->c) && ksize == 3 && stride == 2 && !conv_ks_ok(dtype, in->c, ksize, stride) &&
         !conv_ring_ok(dtype, in->c, ksize, stride, out);
}

int segmi_conv3d_fwd_split_act(int dtype, const segmi_act* in, const segmi_act* out, const void* packed,
                               const float* bias, const float* prelu_alpha, int act_channels, int ksize,
                               int stride, const float* bias_b, float* stats_partials,
                               const segmi_bn_fin* stats_fin, void* stream) {
  SEGMI_CHECK_ARG(segmi_conv3d_split_act_ok(dtype, in, out, ksize, stride),
                  "conv3d_fwd_split_act: layer not eligible (ask segmi_conv3d_split_act_ok)");
  SEGMI_CHECK_ARG(packed && act_channels > 0 && act_channels % 16 == 0 && act_channels <= out->c,
                  "conv3d_fwd_split_act: act_channels must be a multiple of 16 within the output channels");
  SEGMI_CHECK_ARG(in->n == out->n && out->d == out_extent(in->d, ksize, stride) &&
                      out->h == out_extent(in->h, ksize, stride) && out->w == out_extent(in->w, ksize, stride),
                  "conv3d_fwd_split_act: output extent does not match the input");
  const int es = dtype_size(dtype);
  SEGMI_CHECK_ARG(in->ld % (16 / es) == 0 && out->ld % 4 == 0 && ((uintptr_t)in->data % 16) == 0 &&
                      ((uintptr_t)out->data % (4 * es)) == 0,
                  "conv3d_fwd_split_act: MFMA path needs 16-byte aligned input rows");
  ConvParams p{};
  p.in = in->data; p.out = out->data; p.wfrag = packed; p.bias = bias; p.alpha = prelu_alpha;
  p.N = in->n; p.Di = in->d; p.Hi = in->h; p.Wi = in->w;
  p.Do = out->d; p.Ho = out->h; p.Wo = out->w;
  p.Cin = in->c; p.Cout = out->c; p.ldi = in->ld; p.ldo = out->ld;
  p.nchunks = in->c / pick_ck(dtype, in->c);
  p.ntiles_total = out->c / 16;
  p.act_tiles = act_channels / 16;
  p.bias2 = bias_b;
  if (stats_partials) {
    // training: statistics (and their finalisation) of the first act_channels outputs only
    p.stats = stats_partials;
    p.stats_tiles = act_channels / 16;
    if (stats_fin) { p.fin_on = 1; p.bfin = bn_fin_from(stats_fin, act_channels); }
  } else {
    SEGMI_CHECK_ARG(!stats_fin, "conv3d_fwd_split_act: stats_fin needs stats_partials");
  }
  hipStream_t st = (hipStream_t)stream;
  return dtype == SEGMI_F32 ? conv_mfma_f32(p, ksize, stride, st) : conv_mfma_bf16(p, ksize, stride, st);
}

const char* segmi_conv3d_fwd_kernel_name(int dtype, const segmi_act* in, const segmi_act* out,
                                         int ksize, int stride) {
  static thread_local char buf[96];
  if (!act_ok(in) || !act_ok(out)) return "invalid";
  const char* dt = dtype == SEGMI_BF16 ? "bf16" : "f32";
  if (mfma_ok(in->c, out->c)) {
    const int ck = pick_ck(dtype, in->c);
    if (conv_ring_ok(dtype, in->c, ksize, stride, out)) {
      if (conv_ring3_shape_ok(in->c, out->c, in->data, out->data, in->d, in->h, in->w, in->ld, out->d, out->h, out->w, out->ld,
                              out->ld, out->ld))
        snprintf(buf, sizeof buf, "conv_ring3_kernel<%s, CK=16> (LDS-DMA ring)", dt);
      else
        snprintf(buf, sizeof buf, "conv_ring2_kernel<%s, CK=%d, NT=%d>", dt, ck,
                 ck == 32 && (out->c / 16) % 2 == 0 ? 2 : 1);
    }
    else if (conv_ks_ok(dtype, in->c, ksize, stride))
      snprintf(buf, sizeof buf, "conv_fwd_ks_kernel<%s, k%d s%d>", dt, ksize, stride);
    else
      snprintf(buf, sizeof buf, "conv_fwd_mfma_kernel<%s, CK=%d, k%d s%d>", dt, ck, ksize, stride);
    return buf;
  }
  if (small_fwd_eligible(dtype, in, out, ksize)) snprintf(buf, sizeof buf, "conv_small_fwd_kernel<%s>", dt);
  else snprintf(buf, sizeof buf, "conv_direct_kernel<%s>", dt);
  return buf;
}

int segmi_conv3d_in_affine_ok(int dtype, const segmi_act* in, const segmi_act* out, int ksize,
                              int stride) {
  if (!act_ok(in) || !act_ok(out) || dtype != SEGMI_BF16 || !mfma_ok(in->c, out->c)) return 0;
  // forward: the bf16 z-marching ring; weight gradient: the MFMA kernel (any channel blocking)
  return ksize == 3 && stride == 1 && conv_ring_ok(dtype, in->c, ksize, stride, out) ? 1 : 0;
}

int segmi_conv3d_bn_bwd_sums_ok(int dtype, const segmi_act* in, const segmi_act* out, int ksize,
                                int stride) {
  if (!act_ok(in) || !act_ok(out) || dtype != SEGMI_BF16) return 0;
  // the ring kernels' MODE 4: 16 -> 16 (conv_ring3 / conv_ring2<bf16, 16, 1>) and 32 -> 32 (conv_ring2<bf16, 32, 2>)
  return ((in->c == 16 && out->c == 16) || (in->c == 32 && out->c == 32)) && ksize == 3 && stride == 1 &&
                 conv_ring_ok(dtype, in->c, ksize, stride, out) ? 1 : 0;
}

int segmi_bn_act_bwd_apply_conv_ok(int dtype, const segmi_act* dy, const segmi_act* x, const segmi_act* dx,
                                   const segmi_act* out) {
  if (dtype != SEGMI_BF16 || !act_ok(dy) || !act_ok(x) || !act_ok(dx) || !act_ok(out)) return 0;
  const segmi_act* a3[3] = {dy, x, dx};
  for (const segmi_act* a : a3) {
    if (a->n != dy->n || a->d != dy->d || a->h != dy->h || a->w != dy->w || a->c != 16) return 0;
    if (a->ld % 8 != 0 || ((uintptr_t)a->data % 16) != 0) return 0;
  }
  if (out->n != dy->n || out->d != out_extent(dy->d, 3, 2) || out->h != out_extent(dy->h, 3, 2) ||
      out->w != out_extent(dy->w, 3, 2))
    return 0;
  if (!(out->c == 16 || out->c == 32 || out->c == 64) || out->ld % 4 != 0 || ((uintptr_t)out->data % 8) != 0)
    return 0;
  // 32-bit byte offsets inside one sample of dy / x / dx
  const int64_t vox = (int64_t)dy->d * dy->h * dy->w;
  if (vox * dy->ld * 2 >= (1ll << 31) || vox * x->ld * 2 >= (1ll << 31) || vox * dx->ld * 2 >= (1ll << 31)) return 0;
  // the 2 x 4 x 16 output tile of the wide stride-2 configuration
  return out->w >= 16 ? 1 : 0;
}

int segmi_bn_act_bwd_apply_conv(int dtype, const segmi_act* dy, const segmi_act* x, const segmi_act* dx,
                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const float* prelu_alpha, const float* coef, const segmi_act* out,
                                const void* packed, void* stream) {
  SEGMI_CHECK_ARG(segmi_bn_act_bwd_apply_conv_ok(dtype, dy, x, dx, out) && mean && invstd && coef && packed,
                  "bn_act_bwd_apply_conv: not eligible (ask segmi_bn_act_bwd_apply_conv_ok) or bad arguments");
  SEGMI_CHECK_ARG(dx->data != dy->data && dx->data != x->data,
                  "bn_act_bwd_apply_conv: dx must not alias dy or x (halo voxels are re-read by neighbouring tiles)");
  ConvBnBwdParams p{};
  p.dy = dy->data; p.x = x->data; p.dx = dx->data; p.out = out->data; p.wfrag = packed;
  p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.alpha = prelu_alpha; p.coef = coef;
  p.N = dy->n; p.Di = dy->d; p.Hi = dy->h; p.Wi = dy->w; p.Do = out->d; p.Ho = out->h; p.Wo = out->w;
  p.Cout = out->c; p.ldy = dy->ld; p.ldx = x->ld; p.lddx = dx->ld; p.ldo = out->ld;
  p.ntiles_total = out->c / 16;
  return conv_s2_bnbwd_bf16(p, (hipStream_t)stream);
}

int segmi_conv3d_fwd(int dtype, const segmi_act* in, const segmi_act* out, const void* packed,
                     const float* w_src, int w_kind, const float* bias,
                     const float* prelu_alpha, const segmi_act* residual,
                     float* stats_partials, int ksize, int stride, const segmi_in_affine* in_tf,
                     const segmi_bn_bwd_sums* bn_bwd, const segmi_bn_fin* stats_fin, void* stream) {
  SEGMI_CHECK_ARG(!stats_fin || (stats_partials && bn_fin_ok(stats_fin)),
                  "conv3d: stats_fin needs stats_partials and its output pointers");
  if (bn_bwd && bn_bwd->fin)
    SEGMI_CHECK_ARG(bn_bwd->fin->count > 0 && bn_bwd->fin->coef, "conv3d: bn_bwd->fin needs count and coef");
  if (bn_bwd)
    SEGMI_CHECK_ARG(bn_bwd->x && act_ok(bn_bwd->x) && bn_bwd->mean && bn_bwd->invstd && bn_bwd->partials &&
                        segmi_conv3d_bn_bwd_sums_ok(dtype, in, out, ksize, stride) && !prelu_alpha &&
                        !stats_partials && bn_bwd->x->n == out->n && bn_bwd->x->d == out->d &&
                        bn_bwd->x->h == out->h && bn_bwd->x->w == out->w && bn_bwd->x->c == out->c &&
                        bn_bwd->x->ld % 4 == 0 && ((uintptr_t)bn_bwd->x->data % 8) == 0,
                    "conv3d: this launch cannot take BatchNorm-backward sums (ask segmi_conv3d_bn_bwd_sums_ok)");
  if (in_tf)
    SEGMI_CHECK_ARG(in_tf->scale && in_tf->shift && segmi_conv3d_in_affine_ok(dtype, in, out, ksize, stride),
                    "conv3d: this layer cannot take an input transform (ask segmi_conv3d_in_affine_ok)");
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "conv3d: bad dtype %d", dtype);
  SEGMI_CHECK_ARG(act_ok(in) && act_ok(out), "conv3d: bad activation view");
  SEGMI_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2),
                  "conv3d: unsupported ksize/stride %d/%d", ksize, stride);
  SEGMI_CHECK_ARG(!(ksize == 1 && stride != 1), "conv3d: k1 is stride 1 only");
  SEGMI_CHECK_ARG(in->n == out->n && out->d == out_extent(in->d, ksize, stride) &&
                      out->h == out_extent(in->h, ksize, stride) &&
                      out->w == out_extent(in->w, ksize, stride),
                  "conv3d: output extent [%d,%d,%d,%d] does not match input [%d,%d,%d,%d] k%d s%d",
                  out->n, out->d, out->h, out->w, in->n, in->d, in->h, in->w, ksize, stride);
  if (residual)
    SEGMI_CHECK_ARG(act_ok(residual) && residual->n == out->n && residual->d == out->d &&
                        residual->h == out->h && residual->w == out->w && residual->c == out->c,
                    "conv3d: residual shape mismatch");
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype_size(dtype);
  if (mfma_ok(in->c, out->c)) {
    SEGMI_CHECK_ARG(packed, "conv3d: MFMA path needs a fragment-packed weight (segmi_wpack)");
    SEGMI_CHECK_ARG(in->ld % (16 / es) == 0 && out->ld % 4 == 0 &&
                        ((uintptr_t)in->data % 16) == 0 && ((uintptr_t)out->data % (4 * es)) == 0,
                    "conv3d: MFMA path needs 16-byte aligned input rows");
    if (residual)
      SEGMI_CHECK_ARG(residual->ld % 4 == 0 && ((uintptr_t)residual->data % (4 * es)) == 0,
                      "conv3d: residual rows must be 4-element aligned");
    ConvParams p{};
    p.in = in->data; p.out = out->data; p.wfrag = packed; p.bias = bias; p.alpha = prelu_alpha;
    p.res = residual ? residual->data : nullptr; p.stats = stats_partials;
    p.N = in->n; p.Di = in->d; p.Hi = in->h; p.Wi = in->w;
    p.Do = out->d; p.Ho = out->h; p.Wo = out->w;
    p.Cin = in->c; p.Cout = out->c; p.ldi = in->ld; p.ldo = out->ld;
    p.ldr = residual ? residual->ld : 0;
    p.nchunks = in->c / pick_ck(dtype, in->c);
    p.ntiles_total = out->c / 16;
    if (in_tf) { p.in_scale = in_tf->scale; p.in_shift = in_tf->shift; p.in_alpha = in_tf->prelu_alpha; }
    if (bn_bwd) {
      p.bx = bn_bwd->x->data; p.ldbx = bn_bwd->x->ld; p.bmean = bn_bwd->mean; p.binvstd = bn_bwd->invstd;
      p.bgamma = bn_bwd->gamma; p.bbeta = bn_bwd->beta; p.balpha = bn_bwd->prelu_alpha;
      p.bpart = bn_bwd->partials;
      SEGMI_CHECK_ARG((int64_t)p.Ho * p.Wo * p.ldbx < (1ll << 31), "conv3d: plane too large for 32-bit offsets");
      if (bn_bwd->fin) {
        p.fin_on = 1;
        p.bbfin = BnBwdFin{out->c, bn_bwd->fin->count, bn_bwd->fin->dgamma, bn_bwd->fin->dbeta,
                           bn_bwd->fin->dalpha, bn_bwd->fin->coef};
      }
    }
    if (stats_fin) { p.fin_on = 1; p.bfin = bn_fin_from(stats_fin, out->c); }
    return dtype == SEGMI_F32 ? conv_mfma_f32(p, ksize, stride, st)
                              : conv_mfma_bf16(p, ksize, stride, st);
  }
  SEGMI_CHECK_ARG(w_src, "conv3d: direct path (channels not multiples of 16) needs w_src");
  SEGMI_CHECK_ARG(w_kind == 0 || w_kind == 1, "conv3d: bad w_kind %d", w_kind);
  SEGMI_CHECK_ARG(!(stats_partials && small_fwd_eligible(dtype, in, out, ksize) && w_kind != 0),
                  "conv3d: fused statistics on a small-Cin layer need the forward weight kind");
  if (small_fwd_eligible(dtype, in, out, ksize) && w_kind == 0 &&
      (!residual || (residual->ld % 4 == 0 && ((uintptr_t)residual->data % (4 * es)) == 0))) {
    SEGMI_CHECK_ARG(!stats_partials || (!prelu_alpha && !residual),
                    "conv3d: fused statistics are taken before PReLU/residual");
    return conv_small_fwd(dtype, in, out, w_src, bias, prelu_alpha, residual, stats_partials, stride,
                          st, nullptr, nullptr, nullptr, stats_fin);
  }
  DirectParams p{};
  p.in = in->data; p.out = out->data; p.w = w_src; p.bias = bias; p.alpha = prelu_alpha;
  p.res = residual ? residual->data : nullptr;
  p.N = in->n; p.Di = in->d; p.Hi = in->h; p.Wi = in->w; p.Do = out->d; p.Ho = out->h;
  p.Wo = out->w; p.Cin = in->c; p.Cout = out->c; p.ldi = in->ld; p.ldo = out->ld;
  p.ldr = residual ? residual->ld : 0; p.ks = ksize; p.stride = stride; p.kind = w_kind;
  const int64_t total = act_voxels(out) * out->c;
  int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  if (dtype == SEGMI_F32) hipLaunchKernelGGL(conv_direct_kernel<float>, blocks, 256, 0, st, p);
  else hipLaunchKernelGGL(conv_direct_kernel<bf16_t>, blocks, 256, 0, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_fwd(direct)");
  if (stats_partials) {
    SEGMI_CHECK_ARG(!prelu_alpha && !residual,
                    "conv3d: fused statistics are taken before PReLU/residual");
    const int rc = bn_stats_launch(dtype, out, stats_partials, st);
    if (rc != SEGMI_OK || !stats_fin) return rc;
    return bn_finalize_with(stats_partials, segmi_conv3d_stats_rows(dtype, in, out, ksize, stride), out->c,
                            stats_fin, st);       // generic path: a separate (small) launch
  }
  return SEGMI_OK;
}

int segmi_convT3d_stats_rows(int dtype, const segmi_act* in, const segmi_act* out) {
  if (!in || !out) return 0;
  if (mfma_ok(in->c, out->c)) return convt_mfma_rows(dtype, in, out->c) + stats_reserve_rows();
  return bn_stats_rows_for(out) + stats_reserve_rows();
}

int segmi_convT3d_fwd(int dtype, const segmi_act* in, const segmi_act* out, const void* packed,
                      const float* w_src, const float* bias, const float* prelu_alpha,
                      const segmi_act* residual, float* stats_partials, const segmi_bn_fin* stats_fin,
                      void* stream) {
  SEGMI_CHECK_ARG(!stats_fin || (stats_partials && bn_fin_ok(stats_fin)),
                  "convT3d: stats_fin needs stats_partials and its output pointers");
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "convT3d: bad dtype %d", dtype);
  SEGMI_CHECK_ARG(act_ok(in) && act_ok(out), "convT3d: bad activation view");
  SEGMI_CHECK_ARG(in->n == out->n, "convT3d: batch mismatch");
  const int di[3] = {in->d, in->h, in->w}, dout[3] = {out->d, out->h, out->w};
  for (int k = 0; k < 3; ++k)
    SEGMI_CHECK_ARG(dout[k] == 2 * di[k] || dout[k] == 2 * di[k] - 1,
                    "convT3d: output extent %d must be 2*in or 2*in-1 (in %d)", dout[k], di[k]);
  if (residual)
    SEGMI_CHECK_ARG(act_ok(residual) && residual->n == out->n && residual->d == out->d &&
                        residual->h == out->h && residual->w == out->w && residual->c == out->c,
                    "convT3d: residual shape mismatch");
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype_size(dtype);
  if (mfma_ok(in->c, out->c)) {
    SEGMI_CHECK_ARG(packed, "convT3d: MFMA path needs a kind-2 fragment pack (segmi_wpack)");
    SEGMI_CHECK_ARG(in->ld % (16 / es) == 0 && out->ld % 4 == 0 &&
                        ((uintptr_t)in->data % 16) == 0 && ((uintptr_t)out->data % (4 * es)) == 0,
                    "convT3d: MFMA path needs 16-byte aligned input rows");
    if (residual)
      SEGMI_CHECK_ARG(residual->ld % 4 == 0 && ((uintptr_t)residual->data % (4 * es)) == 0,
                      "convT3d: residual rows must be 4-element aligned");
    ConvTParams p{};
    p.in = in->data; p.out = out->data; p.wfrag = packed; p.bias = bias; p.alpha = prelu_alpha;
    p.res = residual ? residual->data : nullptr; p.stats = stats_partials;
    p.N = in->n; p.Di = in->d; p.Hi = in->h; p.Wi = in->w;
    p.Do = out->d; p.Ho = out->h; p.Wo = out->w;
    p.Cin = in->c; p.Cout = out->c; p.ldi = in->ld; p.ldo = out->ld;
    p.ldr = residual ? residual->ld : 0;
    p.nchunks = in->c / pick_ck(dtype, in->c);
    p.ntiles_total = out->c / 16;
    if (stats_fin) { p.fin_on = 1; p.bfin = bn_fin_from(stats_fin, out->c); }
    return dtype == SEGMI_F32 ? convt_mfma_f32(p, st) : convt_mfma_bf16(p, st);
  }
  SEGMI_CHECK_ARG(w_src, "convT3d: direct path (channels not multiples of 16) needs w_src");
  DirectParams p{};
  p.in = in->data; p.out = out->data; p.w = w_src; p.bias = bias; p.alpha = prelu_alpha;
  p.res = residual ? residual->data : nullptr;
  p.N = in->n; p.Di = in->d; p.Hi = in->h; p.Wi = in->w; p.Do = out->d; p.Ho = out->h;
  p.Wo = out->w; p.Cin = in->c; p.Cout = out->c; p.ldi = in->ld; p.ldo = out->ld;
  p.ldr = residual ? residual->ld : 0; p.ks = 3; p.stride = 2; p.kind = 2;
  const int64_t total = act_voxels(out) * out->c;
  int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  if (dtype == SEGMI_F32) hipLaunchKernelGGL(convt_direct_kernel<float>, blocks, 256, 0, st, p);
  else hipLaunchKernelGGL(convt_direct_kernel<bf16_t>, blocks, 256, 0, st, p);
  SEGMI_LAUNCH_CHECK("convT3d_fwd(direct)");
  if (stats_partials) {
    SEGMI_CHECK_ARG(!prelu_alpha && !residual,
                    "convT3d: fused statistics are taken before PReLU/residual");
    const int rc = bn_stats_launch(dtype, out, stats_partials, st);
    if (rc != SEGMI_OK || !stats_fin) return rc;
    return bn_finalize_with(stats_partials, segmi_convT3d_stats_rows(dtype, in, out), out->c, stats_fin, st);
  }
  return SEGMI_OK;
}

}  // extern "C"
