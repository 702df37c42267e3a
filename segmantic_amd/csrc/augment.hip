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
