// image.hip -- ITK-semantics 3-D resample (trilinear / nearest) and channel-wise intensity
// normalisation.  Replaces the SimpleITK CPU path of src/segmantic/image/processing.py:49-120
// and MONAI NormalizeIntensityd (monai_unet.py:164).  HBM-bound; index math in f64 as ITK does.
#include "common.h"

namespace segmi {

struct ResampleParams {
  const void* src;
  void* dst;
  int sx, sy, sz, dx, dy, dz;
  double m[12];  // out index (x,y,z,1) -> continuous in index (x,y,z)
  int interp;
  double defval;
};

template <typename P> struct PixelTraits;
template <> struct PixelTraits<float> { static __device__ float cast(double v) { return (float)v; } };
template <> struct PixelTraits<uint8_t> { static __device__ uint8_t cast(double v) { v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v); return (uint8_t)v; } };
template <> struct PixelTraits<uint16_t> { static __device__ uint16_t cast(double v) { v = v < 0.0 ? 0.0 : (v > 65535.0 ? 65535.0 : v); return (uint16_t)v; } };
template <> struct PixelTraits<int16_t> { static __device__ int16_t cast(double v) { v = v < -32768.0 ? -32768.0 : (v > 32767.0 ? 32767.0 : v); return (int16_t)v; } };
template <> struct PixelTraits<int32_t> { static __device__ int32_t cast(double v) { v = v < -2147483648.0 ? -2147483648.0 : (v > 2147483647.0 ? 2147483647.0 : v); return (int32_t)v; } };

template <typename P>
__global__ void resample_kernel(ResampleParams p) {
  const P* src = (const P*)p.src;
  P* dst = (P*)p.dst;
  const int64_t total = (int64_t)p.dx * p.dy * p.dz;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int ox = e % p.dx;
    const int oy = (e / p.dx) % p.dy;
    const int oz = e / ((int64_t)p.dx * p.dy);
    double cx = p.m[0] * ox + p.m[1] * oy + p.m[2] * oz + p.m[3];
    double cy = p.m[4] * ox + p.m[5] * oy + p.m[6] * oz + p.m[7];
    double cz = p.m[8] * ox + p.m[9] * oy + p.m[10] * oz + p.m[11];
    if (p.interp & 2) {  // border padding (MONAI Spacing / grid_sample "border"): clamp the coordinate
      cx = cx < 0.0 ? 0.0 : (cx > p.sx - 1.0 ? p.sx - 1.0 : cx);
      cy = cy < 0.0 ? 0.0 : (cy > p.sy - 1.0 ? p.sy - 1.0 : cy);
      cz = cz < 0.0 ? 0.0 : (cz > p.sz - 1.0 ? p.sz - 1.0 : cz);
    }
    double val = p.defval;
    const bool inside = cx >= -0.5 && cx < p.sx - 0.5 && cy >= -0.5 && cy < p.sy - 0.5 &&
                        cz >= -0.5 && cz < p.sz - 0.5;
    if (inside) {
      if (p.interp & 1) {
        // ITK rounds half up; bit 4 = round half to even (torch grid_sample "nearest" = nearbyint,
        // what MONAI's Spacing(mode="nearest") ends in)
        int ix, iy, iz;
        if (p.interp & 4) { ix = (int)rint(cx); iy = (int)rint(cy); iz = (int)rint(cz); }
        else { ix = (int)floor(cx + 0.5); iy = (int)floor(cy + 0.5); iz = (int)floor(cz + 0.5); }
        ix = ix < 0 ? 0 : (ix > p.sx - 1 ? p.sx - 1 : ix);
        iy = iy < 0 ? 0 : (iy > p.sy - 1 ? p.sy - 1 : iy);
        iz = iz < 0 ? 0 : (iz > p.sz - 1 ? p.sz - 1 : iz);
        val = (double)src[((int64_t)iz * p.sy + iy) * p.sx + ix];
      } else {
        const double fx0 = floor(cx), fy0 = floor(cy), fz0 = floor(cz);
        int bx = (int)fx0, by = (int)fy0, bz = (int)fz0;
        double fx = cx - fx0, fy = cy - fy0, fz = cz - fz0;
        if (bx < 0) fx = 0.0;
        if (by < 0) fy = 0.0;
        if (bz < 0) fz = 0.0;
        const int x0 = bx < 0 ? 0 : bx, y0 = by < 0 ? 0 : by, z0 = bz < 0 ? 0 : bz;
        const int x1 = bx + 1 > p.sx - 1 ? p.sx - 1 : (bx + 1 < 0 ? 0 : bx + 1);
        const int y1 = by + 1 > p.sy - 1 ? p.sy - 1 : (by + 1 < 0 ? 0 : by + 1);
        const int z1 = bz + 1 > p.sz - 1 ? p.sz - 1 : (bz + 1 < 0 ? 0 : bz + 1);
        // corner order and accumulation as oracle/resample_ref.py (bit d of the corner = dim d)
        val = 0.0;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
          const int xi = (corner & 1) ? x1 : x0, yi = (corner & 2) ? y1 : y0, zi = (corner & 4) ? z1 : z0;
          // explicit round-to-nearest mul/add: no FMA contraction, same bits as the numpy oracle
          double w = (corner & 1) ? fx : 1.0 - fx;
          w = __dmul_rn(w, (corner & 2) ? fy : 1.0 - fy);
          w = __dmul_rn(w, (corner & 4) ? fz : 1.0 - fz);
          val = __dadd_rn(val, __dmul_rn(w, (double)src[((int64_t)zi * p.sy + yi) * p.sx + xi]));
        }
      }
    }
    dst[e] = PixelTraits<P>::cast(val);
  }
}

constexpr int kNormChunk = 1 << 16;

__global__ __launch_bounds__(256) void norm_reduce_kernel(const float* __restrict__ x, int64_t nvox,
                                                          int chunks, double* __restrict__ part) {
  __shared__ double red[2][4];
  const int c = blockIdx.y, chunk = blockIdx.x;
  const float* xc = x + (int64_t)c * nvox;
  const int64_t v0 = (int64_t)chunk * kNormChunk;
  const int64_t v1 = v0 + kNormChunk < nvox ? v0 + kNormChunk : nvox;
  double s = 0.0, q = 0.0;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
    const double a = (double)xc[v];
    s += a; q += a * a;
  }
  s = wave_sum(s); q = wave_sum(q);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((int64_t)c * chunks + chunk) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    part[((int64_t)c * chunks + chunk) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ void norm_finalize_kernel(const double* __restrict__ part, int c, int chunks,
                                     int64_t nvox, float* __restrict__ ms) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < chunks; ++k) { s += part[((int64_t)i * chunks + k) * 2]; q += part[((int64_t)i * chunks + k) * 2 + 1]; }
  const double m = s / (double)nvox;
  double var = q / (double)nvox - m * m;
  if (var < 0.0) var = 0.0;
  float sd = (float)sqrt(var);
  if (sd == 0.f) sd = 1.f;
  ms[2 * i] = (float)m;
  ms[2 * i + 1] = sd;
}

__global__ void norm_apply_kernel(float* __restrict__ x, int64_t nvox, const float* __restrict__ ms) {
  const int c = blockIdx.y;
  const float m = ms[2 * c], sd = ms[2 * c + 1];
  float* xc = x + (int64_t)c * nvox;
  for (int64_t v = blockIdx.x * 256ll + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256)
    xc[v] = (xc[v] - m) / sd;
}

}  // namespace segmi

using namespace segmi;

extern "C" {

int segmi_resample3d(int pixel, const void* src, int sx, int sy, int sz, void* dst, int dx,
                     int dy, int dz, const double* index_map_host, int interp,
                     double default_value, void* stream) {
  SEGMI_CHECK_ARG(src && dst && index_map_host, "resample3d: null pointer");
  SEGMI_CHECK_ARG(sx > 0 && sy > 0 && sz > 0 && dx > 0 && dy > 0 && dz > 0, "resample3d: empty image");
  SEGMI_CHECK_ARG(interp >= 0 && interp <= 7 && (!(interp & 4) || (interp & 1)),
                  "resample3d: interp must be 0 (linear) or 1 (nearest), +2 for border padding, +4 (nearest only) to round half to even");
  ResampleParams p{};
  p.src = src; p.dst = dst; p.sx = sx; p.sy = sy; p.sz = sz; p.dx = dx; p.dy = dy; p.dz = dz;
  for (int i = 0; i < 12; ++i) p.m[i] = index_map_host[i];
  p.interp = interp; p.defval = default_value;
  const int64_t total = (int64_t)dx * dy * dz;
  const int grid = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  hipStream_t st = (hipStream_t)stream;
  switch (pixel) {
    case 0: hipLaunchKernelGGL(resample_kernel<float>, grid, 256, 0, st, p); break;
    case 1: hipLaunchKernelGGL(resample_kernel<uint8_t>, grid, 256, 0, st, p); break;
    case 2: hipLaunchKernelGGL(resample_kernel<int16_t>, grid, 256, 0, st, p); break;
    case 3: hipLaunchKernelGGL(resample_kernel<int32_t>, grid, 256, 0, st, p); break;
    case 4: hipLaunchKernelGGL(resample_kernel<uint16_t>, grid, 256, 0, st, p); break;
    default: SEGMI_CHECK_ARG(false, "resample3d: unknown pixel type %d", pixel);
  }
  SEGMI_LAUNCH_CHECK("resample3d");
  return SEGMI_OK;
}

int64_t segmi_normalize_workspace(int c, int64_t nvox) {
  const int64_t chunks = cdiv64(nvox, kNormChunk);
  return c * chunks * 2 * 8 + (int64_t)c * 2 * 4 + 256;
}

int segmi_normalize_intensity(float* x, int c, int64_t nvox, void* workspace, void* stream) {
  SEGMI_CHECK_ARG(x && workspace && c > 0 && nvox > 0, "normalize_intensity: bad arguments");
  const int chunks = (int)cdiv64(nvox, kNormChunk);
  double* part = (double*)workspace;
  float* ms = (float*)((char*)workspace + ((int64_t)c * chunks * 2 * 8 + 255) / 256 * 256);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(norm_reduce_kernel, dim3(chunks, c), 256, 0, st, (const float*)x, nvox, chunks, part);
  hipLaunchKernelGGL(norm_finalize_kernel, cdiv(c, 64), 64, 0, st, (const double*)part, c, chunks, nvox, ms);
  const int gx = (int)(cdiv64(nvox, 256) > 2048 ? 2048 : cdiv64(nvox, 256));
  hipLaunchKernelGGL(norm_apply_kernel, dim3(gx, c), 256, 0, st, x, nvox, (const float*)ms);
  SEGMI_LAUNCH_CHECK("normalize_intensity");
  return SEGMI_OK;
}

}  // extern "C"
