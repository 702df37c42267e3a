// wgrad.hip -- weight / bias gradients: C entry points, the direct fallback, slab reduction.
#include "wgrad_impl.h"
#include "reduce_fin.h"

#include <atomic>
namespace segmi {
static inline int clamp_cus(int n) {
  n = n / 8 * 8;
  return n < 8 ? 8 : (n > 256 ? 256 : n);
}
// the `cus` argument of a weight-gradient call -> CUs its grid is sized for: a multiple of the 8 XCDs in
// [8, 256]; <= 0 = the whole chip.  No process-global state (round 3's segmi_wgrad_set_cus let two engines /
// threads with different schedules overwrite each other); SEGMI_WGRAD_CUS overrides for A/B runs.
int wgrad_cus(int cus) {
  static const int env = [] {
    const char* e = getenv("SEGMI_WGRAD_CUS");
    return e ? clamp_cus(atoi(e)) : 0;
  }();
  if (env) return env;
  return cus > 0 ? clamp_cus(cus) : 256;
}
}  // namespace segmi
namespace segmi {

int wgrad_mfma_f32(const WgradParams& p, int ksize, int stride, int ct, int gx, hipStream_t st);
int wgrad_mfma_bf16(const WgradParams& p, int ksize, int stride, int ct, int gx, hipStream_t st);
int bn_stats_launch(int dtype, const segmi_act* x, float* partials, hipStream_t st, const BiasFin* bias_fin = nullptr);
int bn_stats_rows_for(const segmi_act* x);
bool conv_small_ok(int cin, int cout, int ksize);
int conv_small_wgrad_slabs(const segmi_act* dy);
int conv_small_wgrad(int dtype, const segmi_act* x, const segmi_act* dy, float* partials,
                     int stride, hipStream_t st);

struct WgDirectParams {
  const void* x;
  const void* dy;
  float* partials;
  int N, Dx, Hx, Wx, Dy, Hy, Wy, Cin, Cout, ldx, ldy, ks, stride;
  int64_t nvox;
  int chunk;
};

// direct fallback (odd channel counts): block b reduces voxel chunks b, b+grid, ...
template <typename T>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WgDirectParams p) {
  const int nt = p.ks * p.ks * p.ks, pad = (p.ks - 1) / 2;
  const int nout = p.Cout * p.Cin * nt;
  const T* x = (const T*)p.x;
  const T* dy = (const T*)p.dy;
  for (int o = threadIdx.x; o < nout; o += 256) {
    const int tap = o % nt;
    const int ci = (o / nt) % p.Cin;
    const int co = o / (nt * p.Cin);
    const int kd = tap / (p.ks * p.ks), kh = (tap / p.ks) % p.ks, kw = tap % p.ks;
    float acc = 0.f;
    for (int64_t c0 = (int64_t)blockIdx.x * p.chunk; c0 < p.nvox; c0 += (int64_t)gridDim.x * p.chunk) {
      const int64_t c1 = c0 + p.chunk < p.nvox ? c0 + p.chunk : p.nvox;
      for (int64_t v = c0; v < c1; ++v) {
        int64_t t = v;
        const int ox = t % p.Wy; t /= p.Wy;
        const int oy = t % p.Hy; t /= p.Hy;
        const int oz = t % p.Dy;
        const int n = t / p.Dy;
        const int z = oz * p.stride - pad + kd, y = oy * p.stride - pad + kh,
                  xx = ox * p.stride - pad + kw;
        if ((unsigned)z >= (unsigned)p.Dx || (unsigned)y >= (unsigned)p.Hx ||
            (unsigned)xx >= (unsigned)p.Wx) continue;
        const float a = Elem<T>::ld(dy + v * p.ldy + co);
        const float b = Elem<T>::ld(x + ((((int64_t)n * p.Dx + z) * p.Hx + y) * p.Wx + xx) * p.ldx + ci);
        acc = fmaf(a, b, acc);
      }
    }
    p.partials[(int64_t)blockIdx.x * nout + o] = acc;
  }
}

// out[gy][e] = sum over the slabs of group gy (blockIdx.y) of partials[b][e]; fixed order, f64
// accumulate.  Many slabs are reduced in two passes (16 groups, then the 16 group sums) so that
// the column sum has enough parallelism instead of one thread walking 1024 slabs.
constexpr int kSlabGroups = 16;
__global__ void slab_reduce_kernel(const float* __restrict__ partials, int nslab, int64_t n,
                                   float* __restrict__ out) {
  const int per = (nslab + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per;
  const int b1 = b0 + per < nslab ? b0 + per : nslab;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
      a0 += (double)partials[(int64_t)b * n + e];
      a1 += (double)partials[(int64_t)(b + 1) * n + e];
      a2 += (double)partials[(int64_t)(b + 2) * n + e];
      a3 += (double)partials[(int64_t)(b + 3) * n + e];
    }
    for (; b < b1; ++b) a0 += (double)partials[(int64_t)b * n + e];
    out[(int64_t)blockIdx.y * n + e] = (float)((a0 + a1) + (a2 + a3));
  }
}

// The two passes of the many-slab case in ONE launch (the weight-gradient stream of a training step is a chain of
// ~20 layers x (weight gradient, reduce, reduce): the second reduce moved 16 rows but cost a dispatch plus the gap in
// front of it, and the main stream waited ~0.2 ms per step for the end of that chain).  Wave g of a 1024-thread
// workgroup is slab group g of the two-pass form, a lane is an element: the group sums meet in LDS as the f32
// values the first pass used to store, and wave 0 adds them in the second pass's order -- the same bits.
__global__ __launch_bounds__(1024) void slab_reduce_fused_kernel(const float* __restrict__ partials, int nslab,
                                                                 int64_t n, float* __restrict__ out) {
  __shared__ float gs[kSlabGroups][64];
  const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per = (nslab + kSlabGroups - 1) / kSlabGroups;
  const int b0 = g * per;
  const int b1 = b0 + per < nslab ? b0 + per : nslab;
  for (int64_t e0 = blockIdx.x * 64ll; e0 < n; e0 += (int64_t)gridDim.x * 64) {
    const int64_t e = e0 + lane;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (e < n) {
      int b = b0;
      for (; b + 3 < b1; b += 4) {
        a0 += (double)partials[(int64_t)b * n + e];
        a1 += (double)partials[(int64_t)(b + 1) * n + e];
        a2 += (double)partials[(int64_t)(b + 2) * n + e];
        a3 += (double)partials[(int64_t)(b + 3) * n + e];
      }
      for (; b < b1; ++b) a0 += (double)partials[(int64_t)b * n + e];
    }
    gs[g][lane] = (float)((a0 + a1) + (a2 + a3));
    __syncthreads();
    if (g == 0 && e < n) {
      double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
      for (int b = 0; b < kSlabGroups; b += 4) {
        c0 += (double)gs[b][lane];
        c1 += (double)gs[b + 1][lane];
        c2 += (double)gs[b + 2][lane];
        c3 += (double)gs[b + 3][lane];
      }
      out[e] = (float)((c0 + c1) + (c2 + c3));
    }
    __syncthreads();
  }
}


static inline bool aligned_rows(const segmi_act* a, int dtype) {
  const int es = dtype_size(dtype);
  return a->ld % (16 / es) == 0 && ((uintptr_t)a->data % 16) == 0;
}
static inline bool wg_mfma_ok(int dtype, const segmi_act* x, const segmi_act* dy, int ksize) {
  return x->c % 16 == 0 && dy->c % 16 == 0 && aligned_rows(x, dtype) && aligned_rows(dy, dtype) &&
         (ksize == 3 || ksize == 1);
}
static inline bool wg_small_ok(int dtype, const segmi_act* x, const segmi_act* dy, int ksize) {
  const int es = dtype_size(dtype);
  (void)es;
  return conv_small_ok(x->c, dy->c, ksize) && aligned_rows(dy, dtype);
}
static inline int wg_direct_blocks(const segmi_act* dy) {
  const int64_t b = cdiv64(act_voxels(dy), 512);
  return (int)(b > 1024 ? 1024 : b);
}
static inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }
// The wave-specialised kernel addresses each operand through ONE buffer descriptor (32-bit offsets, < 4 GiB).  A
// layer that misses it only for that (160^3 x 32 channels x batch 8 inside a 64-channel skip buffer = 4.2 GB) is
// cut along the batch into 2 / 4 / 8 parts that fit: one launch per part, each with its own slabs, one reduce.
static inline segmi_act batch_part(const segmi_act* a, int parts, int i, int dtype) {
  segmi_act s = *a;
  s.n = a->n / parts;
  s.data = (char*)a->data + (int64_t)i * s.n * a->d * a->h * a->w * a->ld * dtype_size(dtype);
  return s;
}
static inline int wg_batch_parts(int dtype, const segmi_act* x, const segmi_act* dy, int ksize, int stride, int cus) {
  if (!wg_mfma_ok(dtype, x, dy, ksize) || wgrad_ws_gx(dtype, x, dy, ksize, stride, cus) > 0) return 1;
  const int64_t lim = 0xfff00000ll;
  if (act_voxels(x) * x->ld * 2 < lim && act_voxels(dy) * dy->ld * 2 < lim) return 1;   // not the size that keeps it out
  for (int parts = 2; parts <= 8 && parts <= x->n; parts *= 2) {
    if (x->n % parts) break;
    const segmi_act xs = batch_part(x, parts, 0, dtype), ys = batch_part(dy, parts, 0, dtype);
    if (wgrad_ws_gx(dtype, &xs, &ys, ksize, stride, cus) > 0) return parts;
  }
  return 1;
}
static inline int wg_slabs(int dtype, const segmi_act* x, const segmi_act* dy, int ksize,
                           int stride, int cus) {
  const int parts = wg_batch_parts(dtype, x, dy, ksize, stride, cus);
  if (parts > 1) {
    const segmi_act xs = batch_part(x, parts, 0, dtype), ys = batch_part(dy, parts, 0, dtype);
    return parts * wgrad_gx(dtype, &xs, &ys, ksize, stride, cus);
  }
  if (wg_mfma_ok(dtype, x, dy, ksize)) return wgrad_gx(dtype, x, dy, ksize, stride, cus);
  if (wg_small_ok(dtype, x, dy, ksize)) return conv_small_wgrad_slabs(dy);
  return wg_direct_blocks(dy);
}

}  // namespace segmi

using namespace segmi;

extern "C" {

int segmi_wgrad_cus(int cus) { return segmi::wgrad_cus(cus); }

int64_t segmi_conv3d_wgrad_workspace(int dtype, const segmi_act* x, const segmi_act* dy,
                                     int ksize, int stride, int cus) {
  if (!x || !dy) return 0;
  const int64_t nout = (int64_t)x->c * dy->c * ksize * ksize * ksize;
  const int slabs = wg_slabs(dtype, x, dy, ksize, stride, cus);
  const int64_t bias = ((int64_t)bn_stats_rows_for(dy) + 2 * kFinScratchRows + 1) * 2 * dy->c * 4;  // + f64 tail
  return align256(slabs * nout * 4) + align256((int64_t)kSlabGroups * nout * 4) + align256(bias);
}

int segmi_bias_grad(int dtype, const segmi_act* dy, float* db, void* workspace, void* stream) {
  SEGMI_CHECK_ARG(act_ok(dy) && db && workspace, "bias_grad: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // the channel sums are folded by the last workgroup of the statistics launch (fin_tail.h): one launch, not two
  const BiasFin fin{dy->c, db};
  return bn_stats_launch(dtype, dy, (float*)workspace, st, &fin);
}

int segmi_conv3d_wgrad(int dtype, const segmi_act* x, const segmi_act* dy, float* dw,
                       float* db, int ksize, int stride, void* workspace,
                       const segmi_in_affine* in_tf, int cus, void* stream) {
  if (in_tf)
    SEGMI_CHECK_ARG(in_tf->scale && in_tf->shift && dtype == SEGMI_BF16 && act_ok(x) && act_ok(dy) &&
                        wg_mfma_ok(dtype, x, dy, ksize),
                    "wgrad: an input transform needs the bf16 MFMA kernel");
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "wgrad: bad dtype");
  SEGMI_CHECK_ARG(act_ok(x) && act_ok(dy) && dw && workspace, "wgrad: bad arguments");
  SEGMI_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "wgrad: k/s");
  SEGMI_CHECK_ARG(!(ksize == 1 && stride != 1), "wgrad: k1 is stride 1 only");
  const int pad = (ksize - 1) / 2;
  SEGMI_CHECK_ARG(x->n == dy->n && dy->d == (x->d + 2 * pad - ksize) / stride + 1 &&
                      dy->h == (x->h + 2 * pad - ksize) / stride + 1 &&
                      dy->w == (x->w + 2 * pad - ksize) / stride + 1,
                  "wgrad: dy extent does not match x for k%d s%d", ksize, stride);
  hipStream_t st = (hipStream_t)stream;
  const int64_t nout = (int64_t)x->c * dy->c * ksize * ksize * ksize;
  float* partials = (float*)workspace;
  const int slabs = wg_slabs(dtype, x, dy, ksize, stride, cus);
  if (wg_mfma_ok(dtype, x, dy, ksize)) {
    const int parts = wg_batch_parts(dtype, x, dy, ksize, stride, cus);
    for (int part = 0; part < parts; ++part) {
      const segmi_act xs = batch_part(x, parts, part, dtype), ys = batch_part(dy, parts, part, dtype);
      const int gx = slabs / parts;
      WgradParams p{};
      p.x = xs.data; p.dy = ys.data; p.partials = partials + (int64_t)part * gx * nout;
      p.N = xs.n; p.Dx = x->d; p.Hx = x->h; p.Wx = x->w; p.Dy = dy->d; p.Hy = dy->h; p.Wy = dy->w;
      p.Cin = x->c; p.Cout = dy->c; p.ldx = x->ld; p.ldy = dy->ld;
      if (in_tf) { p.in_scale = in_tf->scale; p.in_shift = in_tf->shift; p.in_alpha = in_tf->prelu_alpha; }
      const int ct = wgrad_ct_for(dtype, &xs, &ys, ksize, stride, cus);
      const bool ws = wgrad_ws_gx(dtype, &xs, &ys, ksize, stride, cus) > 0;
      const int rc = dtype == SEGMI_F32 ? wgrad_mfma_f32(p, ksize, stride, ct, gx, st)
                                        : wgrad_mfma_bf16(p, ksize, stride, ws ? -ct : ct, gx, st);
      if (rc) return rc;
    }
  } else if (wg_small_ok(dtype, x, dy, ksize)) {
    const int rc = conv_small_wgrad(dtype, x, dy, partials, stride, st);
    if (rc) return rc;
  } else {
    WgDirectParams p{};
    p.x = x->data; p.dy = dy->data; p.partials = partials;
    p.N = x->n; p.Dx = x->d; p.Hx = x->h; p.Wx = x->w; p.Dy = dy->d; p.Hy = dy->h; p.Wy = dy->w;
    p.Cin = x->c; p.Cout = dy->c; p.ldx = x->ld; p.ldy = dy->ld; p.ks = ksize; p.stride = stride;
    p.nvox = act_voxels(dy); p.chunk = 512;
    if (dtype == SEGMI_F32) hipLaunchKernelGGL(wgrad_direct_kernel<float>, slabs, 256, 0, st, p);
    else hipLaunchKernelGGL(wgrad_direct_kernel<bf16_t>, slabs, 256, 0, st, p);
    SEGMI_LAUNCH_CHECK("conv3d_wgrad(direct)");
  }
  const int rb = (int)(cdiv64(nout, 256) > 2048 ? 2048 : cdiv64(nout, 256));
  float* gsum = (float*)((char*)workspace + align256((int64_t)slabs * nout * 4));
  static const bool two_launches = getenv("SEGMI_SLAB_REDUCE2") && atoi(getenv("SEGMI_SLAB_REDUCE2")) == 1;   // A/B
  if (slabs > 2 * kSlabGroups && !two_launches) {
    const int fb = (int)(cdiv64(nout, 64) > 4096 ? 4096 : cdiv64(nout, 64));
    hipLaunchKernelGGL(slab_reduce_fused_kernel, dim3(fb), 1024, 0, st, (const float*)partials, slabs, nout, dw);
  } else if (slabs > 2 * kSlabGroups) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb, kSlabGroups), 256, 0, st, (const float*)partials,
                       slabs, nout, gsum);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb, 1), 256, 0, st, (const float*)gsum, kSlabGroups,
                       nout, dw);
  } else {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb, 1), 256, 0, st, (const float*)partials, slabs,
                       nout, dw);
  }
  SEGMI_LAUNCH_CHECK("conv3d_wgrad(reduce)");
  if (db) {
    float* bws = (float*)((char*)gsum + align256((int64_t)kSlabGroups * nout * 4));
    return segmi_bias_grad(dtype, dy, db, bws, stream);
  }
  return SEGMI_OK;
}

}  // extern "C"
