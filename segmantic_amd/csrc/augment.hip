// augment.hip -- on-device training augmentation (reference src/segmantic/seg/monai_unet.py:178-217):
//   * warp_crop  : RandRotated x3 + RandZoomd + SpatialPadd + RandCropByLabelClassesd + RandFlipd
//                  composed into ONE gather: patch voxel -> (flip, crop origin) -> augmented-space
//                  index -> 3x4 affine -> continuous source index; image trilinear with border
//                  clamping, label nearest; the SpatialPad region (outside the volume) is 0.
//   * patch_minmax / adjust_contrast / histogram_shift / bias_field : RandAdjustContrastd,
//                  RandHistogramShiftd, RandBiasFieldd on f32 patches, in place.
// All HBM-bound elementwise / gather kernels.
#include "common.h"

namespace segmi {

constexpr int kMaxCrops = 16;

struct WarpList {
  int n;
  int b[kMaxCrops], z[kMaxCrops], y[kMaxCrops], x[kMaxCrops];
  unsigned char flip[kMaxCrops];
  float m[12];   // augmented index (x,y,z,1) -> source index (x,y,z), row-major 3x4
};

template <typename TD>
__global__ void warp_crop_kernel(const float* __restrict__ img, const float* __restrict__ lab,
                                 WarpList wl, int D, int H, int W, int C, int ldi,
                                 TD* __restrict__ oimg, float* __restrict__ olab, int rd, int rh,
                                 int rw, int ldo) {
  const int64_t per = (int64_t)rd * rh * rw;
  const int64_t total = per * wl.n;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t t = e;
    const int x = t % rw; t /= rw;
    const int y = t % rh; t /= rh;
    const int z = t % rd;
    const int w = (int)(t / rd);
    const unsigned char f = wl.flip[w];
    const int az = wl.z[w] + ((f & 1) ? rd - 1 - z : z);
    const int ay = wl.y[w] + ((f & 2) ? rh - 1 - y : y);
    const int ax = wl.x[w] + ((f & 4) ? rw - 1 - x : x);
    const bool in = (unsigned)az < (unsigned)D && (unsigned)ay < (unsigned)H && (unsigned)ax < (unsigned)W;
    float lv = 0.f;
    if (!in) {
      for (int c = 0; c < C; ++c) Elem<TD>::st(oimg + e * ldo + c, 0.f);
    } else {
      float cx = wl.m[0] * ax + wl.m[1] * ay + wl.m[2] * az + wl.m[3];
      float cy = wl.m[4] * ax + wl.m[5] * ay + wl.m[6] * az + wl.m[7];
      float cz = wl.m[8] * ax + wl.m[9] * ay + wl.m[10] * az + wl.m[11];
      // padding_mode="border": clamp the sample position into the volume
      cx = fminf(fmaxf(cx, 0.f), (float)(W - 1));
      cy = fminf(fmaxf(cy, 0.f), (float)(H - 1));
      cz = fminf(fmaxf(cz, 0.f), (float)(D - 1));
      const int x0 = (int)cx, y0 = (int)cy, z0 = (int)cz;
      const int x1 = x0 + 1 < W ? x0 + 1 : W - 1, y1 = y0 + 1 < H ? y0 + 1 : H - 1,
                z1 = z0 + 1 < D ? z0 + 1 : D - 1;
      const float fx = cx - x0, fy = cy - y0, fz = cz - z0;
      const int64_t base = (int64_t)wl.b[w] * D;
      const int64_t r00 = ((base + z0) * H + y0) * W, r01 = ((base + z0) * H + y1) * W;
      const int64_t r10 = ((base + z1) * H + y0) * W, r11 = ((base + z1) * H + y1) * W;
      for (int c = 0; c < C; ++c) {
        const float v000 = img[(r00 + x0) * ldi + c], v001 = img[(r00 + x1) * ldi + c];
        const float v010 = img[(r01 + x0) * ldi + c], v011 = img[(r01 + x1) * ldi + c];
        const float v100 = img[(r10 + x0) * ldi + c], v101 = img[(r10 + x1) * ldi + c];
        const float v110 = img[(r11 + x0) * ldi + c], v111 = img[(r11 + x1) * ldi + c];
        const float a0 = v000 + fx * (v001 - v000), a1 = v010 + fx * (v011 - v010);
        const float a2 = v100 + fx * (v101 - v100), a3 = v110 + fx * (v111 - v110);
        const float b0 = a0 + fy * (a1 - a0), b1 = a2 + fy * (a3 - a2);
        Elem<TD>::st(oimg + e * ldo + c, b0 + fz * (b1 - b0));
      }
      if (lab) {
        const int nx = (int)(cx + 0.5f) < W ? (int)(cx + 0.5f) : W - 1;
        const int ny = (int)(cy + 0.5f) < H ? (int)(cy + 0.5f) : H - 1;
        const int nz = (int)(cz + 0.5f) < D ? (int)(cz + 0.5f) : D - 1;
        lv = lab[((base + nz) * H + ny) * W + nx];
      }
    }
    if (olab) olab[e] = lv;
  }
}

// ---- per-patch min / max (fixed-order: per-workgroup partials, then one wave per patch)
__global__ __launch_bounds__(256) void patch_minmax_partial(const float* __restrict__ x, int64_t per,
                                                            int chunks, float* __restrict__ part) {
  __shared__ float smn[256], smx[256];
  const int pidx = blockIdx.y, chunk = blockIdx.x;
  const int64_t lo = per * chunk / chunks, hi = per * (chunk + 1) / chunks;
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = x[(int64_t)pidx * per + i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + o]);
      smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[((int64_t)pidx * chunks + chunk) * 2 + 0] = smn[0];
    part[((int64_t)pidx * chunks + chunk) * 2 + 1] = smx[0];
  }
}
__global__ void patch_minmax_final(const float* __restrict__ part, int chunks, float* __restrict__ mm) {
  const int pidx = blockIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  for (int c = threadIdx.x; c < chunks; c += 64) {
    mn = fminf(mn, part[((int64_t)pidx * chunks + c) * 2 + 0]);
    mx = fmaxf(mx, part[((int64_t)pidx * chunks + c) * 2 + 1]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  if (threadIdx.x == 0) { mm[2 * pidx] = mn; mm[2 * pidx + 1] = mx; }
}

struct IntensityParams {
  int n;                       // patches
  unsigned char on[kMaxCrops]; // apply to patch i
  float gamma[kMaxCrops];      // adjust_contrast
  float ctrl[kMaxCrops][16];   // histogram_shift: floating control points in [0,1], ascending
  int nctrl;
  float coef[kMaxCrops][20];   // bias_field: degree-3 Legendre coefficients
};

// MONAI AdjustContrast: ((x - min) / (max - min + 1e-7)) ** gamma * (max - min) + min
__global__ void adjust_contrast_kernel(float* __restrict__ x, int64_t per, IntensityParams p,
                                       const float* __restrict__ mm) {
  const int pidx = blockIdx.y;
  if (!p.on[pidx]) return;
  const float mn = mm[2 * pidx], rng = mm[2 * pidx + 1] - mn;
  const float g = p.gamma[pidx];
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    float* q = x + (int64_t)pidx * per + i;
    *q = powf((*q - mn) / (rng + 1e-7f), g) * rng + mn;
  }
}

// MONAI RandHistogramShift: np.interp(x, reference points (linspace(min, max, n)), floating points)
__global__ void histogram_shift_kernel(float* __restrict__ x, int64_t per, IntensityParams p,
                                       const float* __restrict__ mm) {
  const int pidx = blockIdx.y;
  if (!p.on[pidx]) return;
  const float mn = mm[2 * pidx], rng = mm[2 * pidx + 1] - mn;
  if (!(rng > 0.f)) return;
  const int n = p.nctrl;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    float* q = x + (int64_t)pidx * per + i;
    float u = (*q - mn) / rng * (float)(n - 1);     // position on the reference grid
    u = fminf(fmaxf(u, 0.f), (float)(n - 1));
    int k = (int)u;
    if (k > n - 2) k = n - 2;
    const float f = u - (float)k;
    const float y = p.ctrl[pidx][k] + f * (p.ctrl[pidx][k + 1] - p.ctrl[pidx][k]);
    *q = y * rng + mn;
  }
}

__device__ __forceinline__ void legendre4(float t, float (&L)[4]) {
  L[0] = 1.f; L[1] = t; L[2] = 0.5f * (3.f * t * t - 1.f); L[3] = 0.5f * (5.f * t * t * t - 3.f * t);
}
// MONAI RandBiasField (degree 3): x *= exp(sum_{i+j+k<=3} c_ijk L_i(z) L_j(y) L_k(x)), coords in [-1,1]
__global__ void bias_field_kernel(float* __restrict__ x, int rd, int rh, int rw, int C,
                                  IntensityParams p) {
  const int pidx = blockIdx.y;
  if (!p.on[pidx]) return;
  const int64_t per = (int64_t)rd * rh * rw * C;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / C;   // NDHWC patch: the field is shared by the channels of a voxel
    const int vx = (int)(v % rw), vy = (int)((v / rw) % rh), vz = (int)(v / ((int64_t)rw * rh));
    float Lz[4], Ly[4], Lx[4];
    legendre4(rd > 1 ? -1.f + 2.f * vz / (rd - 1) : 0.f, Lz);
    legendre4(rh > 1 ? -1.f + 2.f * vy / (rh - 1) : 0.f, Ly);
    legendre4(rw > 1 ? -1.f + 2.f * vx / (rw - 1) : 0.f, Lx);
    float s = 0.f;
    int ci = 0;
    // coefficient order of numpy.polynomial.legendre.leggrid3d over the upper "triangle" that
    // MONAI fills: for i in 0..3, j in 0..3-i, k in 0..3-i-j
    for (int a = 0; a <= 3; ++a)
      for (int b = 0; b <= 3 - a; ++b)
        for (int c = 0; c <= 3 - a - b; ++c) s += p.coef[pidx][ci++] * Lz[a] * Ly[b] * Lx[c];
    x[(int64_t)pidx * per + i] *= expf(s);
  }
}

static inline int grid_1d(int64_t total, int cap) {
  const int64_t b = cdiv64(total, 256);
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace segmi

using namespace segmi;

extern "C" {

int segmi_warp_crop_patches(const segmi_act* image, const float* label, const int32_t* starts_host,
                            const uint8_t* flips_host, int count, const double* index_map_host,
                            int dst_dtype, const segmi_act* out_image, float* out_label,
                            void* stream) {
  SEGMI_CHECK_ARG(act_ok(image) && act_ok(out_image) && starts_host && index_map_host,
                  "warp_crop_patches: bad arguments");
  SEGMI_CHECK_ARG(count > 0 && count <= kMaxCrops && out_image->n >= count && out_image->c == image->c,
                  "warp_crop_patches: 1..%d crops per call", kMaxCrops);
  SEGMI_CHECK_ARG(dst_dtype == SEGMI_F32 || dst_dtype == SEGMI_BF16, "warp_crop_patches: bad dtype");
  WarpList wl{};
  wl.n = count;
  for (int i = 0; i < count; ++i) {
    wl.b[i] = starts_host[4 * i]; wl.z[i] = starts_host[4 * i + 1];
    wl.y[i] = starts_host[4 * i + 2]; wl.x[i] = starts_host[4 * i + 3];
    wl.flip[i] = flips_host ? flips_host[i] : 0;
    SEGMI_CHECK_ARG(wl.b[i] >= 0 && wl.b[i] < image->n, "warp_crop_patches: volume index out of range");
  }
  for (int i = 0; i < 12; ++i) wl.m[i] = (float)index_map_host[i];
  const int64_t total = (int64_t)count * out_image->d * out_image->h * out_image->w;
  const int grid = grid_1d(total, 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dst_dtype == SEGMI_F32)
    hipLaunchKernelGGL(warp_crop_kernel<float>, grid, 256, 0, st, (const float*)image->data, label, wl,
                       image->d, image->h, image->w, image->c, image->ld, (float*)out_image->data,
                       out_label, out_image->d, out_image->h, out_image->w, out_image->ld);
  else
    hipLaunchKernelGGL(warp_crop_kernel<bf16_t>, grid, 256, 0, st, (const float*)image->data, label, wl,
                       image->d, image->h, image->w, image->c, image->ld, (bf16_t*)out_image->data,
                       out_label, out_image->d, out_image->h, out_image->w, out_image->ld);
  SEGMI_LAUNCH_CHECK("warp_crop_patches");
  return SEGMI_OK;
}

int64_t segmi_intensity_workspace(int count) { return (int64_t)count * (256 + 1) * 2 * 4; }

int segmi_intensity_augment(float* patches, int count, int rd, int rh, int rw, int c,
                            const uint8_t* contrast_on_host, const float* gamma_host,
                            const uint8_t* hist_on_host, const float* ctrl_host, int nctrl,
                            const uint8_t* bias_on_host, const float* coef_host, void* workspace,
                            void* stream) {
  SEGMI_CHECK_ARG(patches && workspace && count > 0 && count <= kMaxCrops && rd > 0 && rh > 0 && rw > 0 && c > 0,
                  "intensity_augment: bad arguments (1..%d patches)", kMaxCrops);
  SEGMI_CHECK_ARG(!hist_on_host || (ctrl_host && nctrl >= 2 && nctrl <= 16),
                  "intensity_augment: 2..16 histogram control points");
  SEGMI_CHECK_ARG((!contrast_on_host || gamma_host) && (!bias_on_host || coef_host),
                  "intensity_augment: missing parameter array");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = (int64_t)rd * rh * rw * c;   // elements of one dense NDHWC patch
  const int chunks = 256;
  float* part = (float*)workspace;
  float* mm = part + (int64_t)count * chunks * 2;
  const int gx = grid_1d(per, 1024);
  IntensityParams p{};
  p.n = count;
  p.nctrl = nctrl;
  auto any = [&](const uint8_t* on) {
    if (!on) return false;
    for (int i = 0; i < count; ++i) if (on[i]) return true;
    return false;
  };
  // order of the reference: contrast, histogram shift, bias field (monai_unet.py:206-208); min/max
  // are those of the patch as it enters each transform
  if (any(contrast_on_host)) {
    for (int i = 0; i < count; ++i) { p.on[i] = contrast_on_host[i]; p.gamma[i] = gamma_host[i]; }
    hipLaunchKernelGGL(patch_minmax_partial, dim3(chunks, count), 256, 0, st, patches, per, chunks, part);
    hipLaunchKernelGGL(patch_minmax_final, count, 64, 0, st, part, chunks, mm);
    hipLaunchKernelGGL(adjust_contrast_kernel, dim3(gx, count), 256, 0, st, patches, per, p, mm);
  }
  if (any(hist_on_host)) {
    for (int i = 0; i < count; ++i) {
      p.on[i] = hist_on_host[i];
      for (int k = 0; k < nctrl; ++k) p.ctrl[i][k] = ctrl_host[i * nctrl + k];
    }
    hipLaunchKernelGGL(patch_minmax_partial, dim3(chunks, count), 256, 0, st, patches, per, chunks, part);
    hipLaunchKernelGGL(patch_minmax_final, count, 64, 0, st, part, chunks, mm);
    hipLaunchKernelGGL(histogram_shift_kernel, dim3(gx, count), 256, 0, st, patches, per, p, mm);
  }
  if (any(bias_on_host)) {
    for (int i = 0; i < count; ++i) {
      p.on[i] = bias_on_host[i];
      for (int k = 0; k < 20; ++k) p.coef[i][k] = coef_host[i * 20 + k];
    }
    hipLaunchKernelGGL(bias_field_kernel, dim3(gx, count), 256, 0, st, patches, rd, rh, rw, c, p);
  }
  SEGMI_LAUNCH_CHECK("intensity_augment");
  return SEGMI_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// k-space augmentations (RandGibbsNoised, RandKSpaceSpikeNoised; monai_unet.py:209-210).
// The patches are small (<= 160^3) and the transforms fire with probability 0.2, so the 3-D DFT
// is three passes of a direct O(N^2) per-line transform (any N <= 512, no radix restrictions --
// the reference's default patch is 96^3): 16 lines of one axis staged in LDS per workgroup with
// the N twiddles, every thread accumulating N/16 outputs.  ~6 GFLOP per 128^3 patch and pass set.
//
// The reference applies mask / spike to fftshift(fftn(ifftshift(x))) and undoes the shifts after
// the inverse transform.  A circular shift is a unit-modulus phase ramp in k-space, which commutes
// with the pointwise mask and with "replace one bin's magnitude, keep its phase"; so both
// transforms are evaluated on the plain fftn(x) with the mask / spike index un-shifted:
// shifted index i  <->  plain index (i - N/2) mod N.
namespace segmi {

typedef float2 cplx;

__global__ __launch_bounds__(256) void dft_axis_kernel(cplx* __restrict__ buf, int N, int64_t stride,
                                                       int64_t inner, int64_t nlines, int inverse) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  cplx* tw = reinterpret_cast<cplx*>(dsm);          // [N]
  cplx* ln = tw + N;                                 // [16][N + 1]
  const int tid = threadIdx.x;
  for (int m = tid; m < N; m += 256) {
    float s, c;
    sincospif(2.0f * (float)m / (float)N, &s, &c);
    tw[m] = cplx{c, inverse ? s : -s};
  }
  const int64_t l0 = (int64_t)blockIdx.x * 16;
  const int li = tid % 16, k0 = tid / 16;
  // line l -> (outer, in) with base offset outer * N * stride + in   (in < inner == stride)
  for (int e = tid; e < 16 * N; e += 256) {
    const int l = e % 16, n = e / 16;
    const int64_t line = l0 + l;
    cplx v = cplx{0.f, 0.f};
    if (line < nlines) v = buf[(line / inner) * N * stride + (line % inner) + (int64_t)n * stride];
    ln[l * (N + 1) + n] = v;
  }
  __syncthreads();
  const int64_t line = l0 + li;
  const float scale = inverse ? 1.0f / (float)N : 1.0f;
  for (int k = k0; k < N; k += 16) {
    float ar = 0.f, ai = 0.f;
    int idx = 0;
    for (int n = 0; n < N; ++n) {
      const cplx x = ln[li * (N + 1) + n], w = tw[idx];
      ar += x.x * w.x - x.y * w.y;
      ai += x.x * w.y + x.y * w.x;
      idx += k;
      if (idx >= N) idx -= N;
    }
    if (line < nlines)
      buf[(line / inner) * N * stride + (line % inner) + (int64_t)k * stride] = cplx{ar * scale, ai * scale};
  }
}

struct KspaceParams {   // indexed by SLOT: only the patches a transform fired for are transformed
  int n;
  unsigned char src[kMaxCrops];    // slot -> patch index
  unsigned char gibbs_on[kMaxCrops], spike_on[kMaxCrops];
  float gibbs_r[kMaxCrops];        // mask radius (1 - alpha) * max(shape) * sqrt(2) / 2
  int spike_loc[kMaxCrops][3];     // (z, y, x) in the SHIFTED k-space, as the reference draws it
  float spike_u[kMaxCrops];        // U(0,1): intensity = mean(log|K|) * 2.5 * (0.95 + 0.15 u)
};

__global__ void real_to_cplx_kernel(const float* __restrict__ x, cplx* __restrict__ k, int64_t per,
                                    int C, int ch, KspaceParams p) {
  const int pidx = blockIdx.y;
  if (!(p.gibbs_on[pidx] | p.spike_on[pidx])) return;
  const int64_t src = p.src[pidx];
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256)
    k[(int64_t)pidx * per + i] = cplx{x[(src * per + i) * C + ch], 0.f};
}
__global__ void cplx_to_real_kernel(const cplx* __restrict__ k, float* __restrict__ x, int64_t per,
                                    int C, int ch, KspaceParams p) {
  const int pidx = blockIdx.y;
  if (!(p.gibbs_on[pidx] | p.spike_on[pidx])) return;
  const int64_t src = p.src[pidx];
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256)
    x[(src * per + i) * C + ch] = k[(int64_t)pidx * per + i].x;
}
// GibbsNoise._apply_mask on the un-shifted spectrum
__global__ void gibbs_mask_kernel(cplx* __restrict__ k, int rd, int rh, int rw, KspaceParams p) {
  const int pidx = blockIdx.y;
  if (!p.gibbs_on[pidx]) return;
  const int64_t per = (int64_t)rd * rh * rw;
  const float r = p.gibbs_r[pidx];
  const float cz = (rd - 1) * 0.5f, cy = (rh - 1) * 0.5f, cx = (rw - 1) * 0.5f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % rw), y = (int)((i / rw) % rh), z = (int)(i / ((int64_t)rw * rh));
    const float sz = (float)((z + rd / 2) % rd) - cz, sy = (float)((y + rh / 2) % rh) - cy,
                sx = (float)((x + rw / 2) % rw) - cx;       // position in the shifted spectrum
    if (sqrtf(sz * sz + sy * sy + sx * sx) > r) k[(int64_t)pidx * per + i] = cplx{0.f, 0.f};
  }
}
// mean over the spectrum of log(|K| + 1e-10): per-workgroup partials (f32), summed in order
__global__ __launch_bounds__(256) void logabs_partial_kernel(const cplx* __restrict__ k, int64_t per,
                                                             int chunks, KspaceParams p,
                                                             float* __restrict__ part) {
  __shared__ float sm[256];
  const int pidx = blockIdx.y, chunk = blockIdx.x;
  float s = 0.f;
  if (p.spike_on[pidx]) {
    const int64_t lo = per * chunk / chunks, hi = per * (chunk + 1) / chunks;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      const cplx v = k[(int64_t)pidx * per + i];
      s += logf(sqrtf(v.x * v.x + v.y * v.y) + 1e-10f);
    }
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(int64_t)pidx * chunks + chunk] = sm[0];
}
// KSpaceSpikeNoise._set_spike: |K[loc]| := exp(intensity), phase kept
__global__ void spike_kernel(cplx* __restrict__ k, int rd, int rh, int rw, int chunks,
                             const float* __restrict__ part, KspaceParams p) {
  const int pidx = blockIdx.x;
  if (!p.spike_on[pidx] || threadIdx.x != 0) return;
  const int64_t per = (int64_t)rd * rh * rw;
  double s = 0.0;
  for (int c = 0; c < chunks; ++c) s += (double)part[(int64_t)pidx * chunks + c];
  const float mean = (float)(s / (double)per) * 2.5f;
  const float inten = mean * (0.95f + 0.15f * p.spike_u[pidx]);
  const int z = ((p.spike_loc[pidx][0] - rd / 2) % rd + rd) % rd;
  const int y = ((p.spike_loc[pidx][1] - rh / 2) % rh + rh) % rh;
  const int x = ((p.spike_loc[pidx][2] - rw / 2) % rw + rw) % rw;
  cplx* q = k + (int64_t)pidx * per + ((int64_t)z * rh + y) * rw + x;
  const float mag = sqrtf(q->x * q->x + q->y * q->y);
  const float a = expf(inten);
  // angle(0) = 0 in the reference (torch.angle), i.e. a positive real spike
  *q = mag > 0.f ? cplx{q->x / mag * a, q->y / mag * a} : cplx{a, 0.f};
}

static int dft3d(cplx* buf, int count, int rd, int rh, int rw, int inverse, hipStream_t st) {
  const int64_t per = (int64_t)rd * rh * rw;
  const int dims[3] = {rw, rh, rd};
  const int64_t strides[3] = {1, rw, (int64_t)rw * rh};
  for (int a = 0; a < 3; ++a) {
    const int N = dims[a];
    const int64_t nlines = per / N * count;
    // lines of axis a inside the [count * per] buffer: for the contiguous axis a line is
    // `outer`-indexed (inner = 1); for the others consecutive lines are adjacent in memory
    const size_t lds = (size_t)(N + 16 * (N + 1)) * sizeof(cplx);
    hipLaunchKernelGGL(dft_axis_kernel, (unsigned)cdiv64(nlines, 16), 256, lds, st, buf, N,
                       strides[a], strides[a], nlines, inverse);
  }
  SEGMI_LAUNCH_CHECK("kspace_augment(dft)");
  return SEGMI_OK;
}

}  // namespace segmi

extern "C" {

int64_t segmi_kspace_workspace(int count, int rd, int rh, int rw) {
  return (int64_t)count * rd * rh * rw * 8 + (int64_t)count * 256 * 4 + 256;
}

int segmi_kspace_augment(float* patches, int count, int rd, int rh, int rw, int c,
                         const uint8_t* gibbs_on_host, const float* gibbs_alpha_host,
                         const uint8_t* spike_on_host, const int32_t* spike_loc_host,
                         const float* spike_u_host, void* workspace, void* stream) {
  SEGMI_CHECK_ARG(patches && workspace && count > 0 && count <= kMaxCrops && rd > 0 && rh > 0 &&
                      rw > 0 && c > 0, "kspace_augment: bad arguments (1..%d patches)", kMaxCrops);
  SEGMI_CHECK_ARG((!gibbs_on_host || gibbs_alpha_host) && (!spike_on_host || (spike_loc_host && spike_u_host)),
                  "kspace_augment: missing parameter array");
  if (rd > 512 || rh > 512 || rw > 512) SEGMI_UNSUPPORTED("kspace_augment: patch extents up to 512");
  const int mx = rd > rh ? (rd > rw ? rd : rw) : (rh > rw ? rh : rw);
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = (int64_t)rd * rh * rw;
  cplx* buf = (cplx*)workspace;
  float* part = (float*)((char*)workspace + (int64_t)count * per * 8);
  const int chunks = 256;
  const int gx = grid_1d(per, 1024);
  // the reference order is Gibbs then spike, each with its own forward / inverse transform and a
  // real-part projection in between; channel_wise: every channel is transformed on its own
  for (int pass = 0; pass < 2; ++pass) {
    KspaceParams q{};
    int ns = 0;
    for (int i = 0; i < count; ++i) {
      if (pass == 0 && gibbs_on_host && gibbs_on_host[i]) {
        q.src[ns] = (unsigned char)i; q.gibbs_on[ns] = 1;
        q.gibbs_r[ns] = (1.0f - gibbs_alpha_host[i]) * (float)mx * 1.41421356f / 2.0f;
        ++ns;
      } else if (pass == 1 && spike_on_host && spike_on_host[i]) {
        q.src[ns] = (unsigned char)i; q.spike_on[ns] = 1;
        for (int d = 0; d < 3; ++d) q.spike_loc[ns][d] = spike_loc_host[3 * i + d];
        q.spike_u[ns] = spike_u_host[i];
        ++ns;
      }
    }
    if (ns == 0) continue;
    q.n = ns;
    for (int ch = 0; ch < c; ++ch) {
      hipLaunchKernelGGL(real_to_cplx_kernel, dim3(gx, ns), 256, 0, st, patches, buf, per, c, ch, q);
      int rc = dft3d(buf, ns, rd, rh, rw, 0, st);
      if (rc) return rc;
      if (pass == 0) {
        hipLaunchKernelGGL(gibbs_mask_kernel, dim3(gx, ns), 256, 0, st, buf, rd, rh, rw, q);
      } else {
        hipLaunchKernelGGL(logabs_partial_kernel, dim3(chunks, ns), 256, 0, st, buf, per, chunks, q, part);
        hipLaunchKernelGGL(spike_kernel, ns, 64, 0, st, buf, rd, rh, rw, chunks, part, q);
      }
      rc = dft3d(buf, ns, rd, rh, rw, 1, st);
      if (rc) return rc;
      hipLaunchKernelGGL(cplx_to_real_kernel, dim3(gx, ns), 256, 0, st, buf, patches, per, c, ch, q);
    }
  }
  SEGMI_LAUNCH_CHECK("kspace_augment");
  return SEGMI_OK;
}

}  // extern "C"
