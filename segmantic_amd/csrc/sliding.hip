// sliding.hip -- sliding-window inference data movement (gather / ordered blend / finalise),
// channel argmax, label overlap counts and the on-device patch cropper.  All HBM-bound.
#include <type_traits>

#include "common.h"

namespace segmi {

constexpr int kMaxWin = 16;

struct WinList {
  int n;
  int z[kMaxWin], y[kMaxWin], x[kMaxWin];
};

template <typename TS, typename TD>
__global__ void sw_gather_kernel(const TS* __restrict__ img, TD* __restrict__ win, WinList wl,
                                 int D, int H, int W, int C, int ldi, int rd, int rh, int rw,
                                 int ldw) {
  const int64_t per = (int64_t)rd * rh * rw * C;
  const int64_t total = per * wl.n;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = e % C;
    int64_t t = e / C;
    const int x = t % rw; t /= rw;
    const int y = t % rh; t /= rh;
    const int z = t % rd;
    const int w = t / rd;
    const int gz = wl.z[w] + z, gy = wl.y[w] + y, gx = wl.x[w] + x;
    float v = 0.f;
    if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
      v = Elem<TS>::ld(img + (((int64_t)gz * H + gy) * W + gx) * ldi + c);
    Elem<TD>::st(win + ((((int64_t)w * rd + z) * rh + y) * rw + x) * ldw + c, v);
  }
}

// single-channel images (the usual case): one thread gathers 4 consecutive x of a window row and
// stores them with one 16-byte (f32) / 8-byte (bf16) write
template <typename TS, typename TD>
__global__ void sw_gather4_kernel(const TS* __restrict__ img, TD* __restrict__ win, WinList wl,
                                  int D, int H, int W, int ldi, int rd, int rh, int rw) {
  const int rw4 = rw / 4;
  const int64_t per = (int64_t)rd * rh * rw4;
  const int64_t total = per * wl.n;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t t = e;
    const int x = (int)(t % rw4) * 4; t /= rw4;
    const int y = (int)(t % rh); t /= rh;
    const int z = (int)(t % rd);
    const int w = (int)(t / rd);
    const int gz = wl.z[w] + z, gy = wl.y[w] + y, gx = wl.x[w] + x;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H) {
      const TS* row = img + (((int64_t)gz * H + gy) * W) * ldi;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((unsigned)(gx + j) < (unsigned)W) v[j] = Elem<TS>::ld(row + (int64_t)(gx + j) * ldi);
    }
    store4<TD>(win + ((((int64_t)w * rd + z) * rh + y) * rw + x), v);
  }
}

// one thread per (voxel of the windows' bounding box, channel); windows applied IN ORDER so the
// f32 accumulation order equals the reference's sequential `out[slice] += w * pred`.
template <typename T>
__global__ void sw_scatter_kernel(const T* __restrict__ pred, WinList wl, const float* __restrict__ imp,
                                  float* __restrict__ acc, float* __restrict__ cnt, int D, int H,
                                  int W, int K, int lda, int rd, int rh, int rw, int ldp, int bz0,
                                  int by0, int bx0, int bd, int bh, int bw) {
  const int64_t total = (int64_t)bd * bh * bw * K;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int k = e % K;
    int64_t t = e / K;
    const int x = bx0 + t % bw; t /= bw;
    const int y = by0 + t % bh; t /= bh;
    const int z = bz0 + (int)t;
    const int64_t vox = ((int64_t)z * H + y) * W + x;
    float a = acc[vox * lda + k];
    float c = (cnt && k == 0) ? cnt[vox] : 0.f;
    bool touched = false;
    for (int w = 0; w < wl.n; ++w) {
      const int lz = z - wl.z[w], ly = y - wl.y[w], lx = x - wl.x[w];
      if ((unsigned)lz < (unsigned)rd && (unsigned)ly < (unsigned)rh && (unsigned)lx < (unsigned)rw) {
        const int64_t lv = ((int64_t)lz * rh + ly) * rw + lx;
        const float wt = imp ? imp[lv] : 1.f;
        const float pv = Elem<T>::ld(pred + (((int64_t)w * rd * rh * rw) + lv) * ldp + k);
        a += wt * pv;
        c += wt;
        touched = true;
      }
    }
    if (touched) {
      acc[vox * lda + k] = a;
      if (cnt && k == 0) cnt[vox] = c;
    }
  }
}

// 4-channel vector variant of sw_scatter_kernel (K % 4 == 0, 16-byte aligned rows): one thread
// per (voxel, 4 channels): f32x4 read-modify-write of the accumulator, 8/16-byte prediction loads
template <typename T>
__global__ void sw_scatter4_kernel(const T* __restrict__ pred, WinList wl, const float* __restrict__ imp,
                                   float* __restrict__ acc, float* __restrict__ cnt, int D, int H,
                                   int W, int K, int lda, int rd, int rh, int rw, int ldp, int bz0,
                                   int by0, int bx0, int bd, int bh, int bw) {
  const int kg = K / 4;
  const int64_t total = (int64_t)bd * bh * bw * kg;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int k = (int)(e % kg) * 4;
    int64_t t = e / kg;
    const int x = bx0 + (int)(t % bw); t /= bw;
    const int y = by0 + (int)(t % bh); t /= bh;
    const int z = bz0 + (int)t;
    const int64_t vox = ((int64_t)z * H + y) * W + x;
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    float c = 0.f;
    bool touched = false;
    for (int w = 0; w < wl.n; ++w) {
      const int lz = z - wl.z[w], ly = y - wl.y[w], lx = x - wl.x[w];
      if ((unsigned)lz < (unsigned)rd && (unsigned)ly < (unsigned)rh && (unsigned)lx < (unsigned)rw) {
        const int64_t lv = ((int64_t)lz * rh + ly) * rw + lx;
        const float wt = imp ? imp[lv] : 1.f;
        const f32x4 pv = load4<T>(pred + (((int64_t)w * rd * rh * rw) + lv) * ldp + k);
        if (!touched) {   // first contributing window: fetch the running sums (same add order
          a = *reinterpret_cast<const f32x4*>(acc + vox * lda + k);   // as the scalar kernel)
          c = (cnt && k == 0) ? cnt[vox] : 0.f;
          touched = true;
        }
        a += wt * pv;
        c += wt;
      }
    }
    if (touched) {
      *reinterpret_cast<f32x4*>(acc + vox * lda + k) = a;
      if (cnt && k == 0) cnt[vox] = c;
    }
  }
}

// argmax with K/4 lanes per voxel (K in {4,8,16,32,64,128,256}): 16-byte loads, lane-group
// butterfly keeping torch.argmax semantics (first maximal index; first NaN wins)
__device__ __forceinline__ bool am_better(float av, int ai, float bv, int bi) {
  // true when (bv, bi) should replace (av, ai)
  const bool an = av != av, bn = bv != bv;
  if (an || bn) return bn && (!an || bi < ai);
  return bv > av || (bv == av && bi < ai);
}
template <typename T, typename L>
__global__ void argmax4_kernel(const T* __restrict__ lg, const float* __restrict__ cnt,
                               float* __restrict__ wb, L* __restrict__ labels, int64_t nvox, int K,
                               int ld) {
  const int tpv = K / 4;                       // lanes per voxel (power of two <= 64)
  const int64_t total = nvox * tpv;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e0 = blockIdx.x * 256ll; e0 < total; e0 += stride) {   // whole wave iterates together
    const int64_t e = e0 + threadIdx.x;
    const bool live = e < total;
    const int64_t v = live ? e / tpv : 0;
    const int k = (int)(e % tpv) * 4;
    f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
      x = load4<T>(lg + v * ld + k);
      if (cnt) {
        const float c = cnt[v];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = x[j] / c;
      }
      if (wb) *reinterpret_cast<f32x4*>(wb + v * ld + k) = x;
    }
    float bv = x[0];
    int bi = k;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (am_better(bv, bi, x[j], k + j)) { bv = x[j]; bi = k + j; }
    for (int o = 1; o < tpv; o <<= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (am_better(bv, bi, ov, oi)) { bv = ov; bi = oi; }
    }
    if (live && k == 0) labels[v] = (L)bi;
  }
}

template <typename T, typename L>
__global__ void argmax_kernel(const T* __restrict__ lg, const float* __restrict__ cnt,
                              float* __restrict__ wb, L* __restrict__ labels, int64_t nvox, int K,
                              int ld) {
  for (int64_t v = blockIdx.x * 256ll + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256) {
    const T* p = lg + v * ld;
    const float c = cnt ? cnt[v] : 1.f;
    float best = 0.f;
    int bi = 0;
    for (int k = 0; k < K; ++k) {
      float x = Elem<T>::ld(p + k);
      if (cnt) x = x / c;
      if (wb) wb[v * ld + k] = x;
      // torch.argmax: first maximal index; a NaN counts as maximal and the first one wins
      if (k == 0) { best = x; }
      else if (best == best && (x != x || x > best)) { best = x; bi = k; }
    }
    labels[v] = (L)bi;
  }
}

// ---------------------------------------------------------------- deferred ordered blend
// Every window prediction of the dense schedule is kept (`cache` slot = window index - lo); one
// pass then sums, per output voxel, the covering windows in ascending window index -- the same
// f32 addition sequence as the reference's sequential `out[slice] += w * pred` -- divides by the
// count and takes the argmax.  Each prediction element is read exactly once and no f32
// accumulator is read-modified-written per window group (5.4x less HBM traffic than the
// streaming blend at overlap 0.5).
constexpr int kMaxStarts = 64;
struct BlendSched {
  int n[3];
  int start[3][kMaxStarts];   // per-dimension window origins, ascending (first dim slowest)
};

// G = channels per lane (16 bytes of T); K / G lanes cooperate on one voxel
template <typename T, typename L, int G>
__global__ __launch_bounds__(256) void sw_blend_kernel(
    const T* __restrict__ cache, BlendSched sc, int lo, int hi, const float* __restrict__ imp,
    float* __restrict__ out, float* __restrict__ cnt_out, L* __restrict__ labels, int D, int H,
    int W, int K, int ldo, int rd, int rh, int rw, int ldp, int normalize) {
  // A workgroup takes 256-lane segments of one output row (z, y): z, y and with them the covering
  // windows of those two dimensions are wave-uniform (scalar loops), only x is per lane, and no
  // lane does a 64-bit division (the flat-index form spent more time dividing than loading).
  const int tpv = K / G;
  const int lanes_row = W * tpv;
  const int segs = (lanes_row + 255) / 256;
  const int nseg = D * H * segs;
  for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
    const int seg = s % segs, row = s / segs;
    const int y = row % H, z = row / H;
    const int e = seg * 256 + (int)threadIdx.x;
    const bool live = e < lanes_row;
    const int x = live ? e / tpv : 0;
    const int k = live ? (e % tpv) * G : 0;
    const int64_t v = (int64_t)row * W + x;
    float a[G];
#pragma unroll
    for (int j = 0; j < G; ++j) a[j] = 0.f;
    float c = 0.f;
    if (live) {
      for (int kz = 0; kz < sc.n[0]; ++kz) {
        const int lz = z - sc.start[0][kz];
        if ((unsigned)lz >= (unsigned)rd) continue;
        for (int ky = 0; ky < sc.n[1]; ++ky) {
          const int ly = y - sc.start[1][ky];
          if ((unsigned)ly >= (unsigned)rh) continue;
          for (int kx = 0; kx < sc.n[2]; ++kx) {
            const int lx = x - sc.start[2][kx];
            if ((unsigned)lx >= (unsigned)rw) continue;
            const int w = (kz * sc.n[1] + ky) * sc.n[2] + kx;
            if (w < lo || w >= hi) continue;
            const int64_t lv = ((int64_t)lz * rh + ly) * rw + lx;
            const float wt = imp ? imp[lv] : 1.f;
            const T* pp = cache + ((int64_t)(w - lo) * rd * rh * rw + lv) * ldp + k;
            if constexpr (G == 8) {
              f32x4 p0, p1;
              const u32x4 o = *reinterpret_cast<const u32x4*>(pp);
              p0[0] = __uint_as_float(o[0] << 16); p0[1] = __uint_as_float(o[0] & 0xffff0000u);
              p0[2] = __uint_as_float(o[1] << 16); p0[3] = __uint_as_float(o[1] & 0xffff0000u);
              p1[0] = __uint_as_float(o[2] << 16); p1[1] = __uint_as_float(o[2] & 0xffff0000u);
              p1[2] = __uint_as_float(o[3] << 16); p1[3] = __uint_as_float(o[3] & 0xffff0000u);
#pragma unroll
              for (int j = 0; j < 4; ++j) { a[j] += wt * p0[j]; a[4 + j] += wt * p1[j]; }
            } else if constexpr (G == 4) {
              const f32x4 p0 = load4<T>(pp);
#pragma unroll
              for (int j = 0; j < 4; ++j) a[j] += wt * p0[j];
            } else {
              a[0] += wt * Elem<T>::ld(pp);
            }
            c += wt;
          }
        }
      }
      if (normalize) {
#pragma unroll
        for (int j = 0; j < G; ++j) a[j] = a[j] / c;
      }
      if (out) {
        if constexpr (G >= 4) {
#pragma unroll
          for (int j = 0; j < G; j += 4)
            *reinterpret_cast<f32x4*>(out + v * ldo + k + j) = f32x4{a[j], a[j + 1], a[j + 2], a[j + 3]};
        } else {
          out[v * ldo + k] = a[0];
        }
      }
      if (cnt_out && k == 0) cnt_out[v] = c;
    }
    if (labels) {
      float bv = a[0];
      int bi = k;
#pragma unroll
      for (int j = 1; j < G; ++j)
        if (am_better(bv, bi, a[j], k + j)) { bv = a[j]; bi = k + j; }
      if (tpv <= 64 && (tpv & (tpv - 1)) == 0) {
        for (int o = 1; o < tpv; o <<= 1) {
          const float ov = __shfl_xor(bv, o);
          const int oi = __shfl_xor(bi, o);
          if (am_better(bv, bi, ov, oi)) { bv = ov; bi = oi; }
        }
        if (live && k == 0) labels[v] = (L)bi;
      }
    }
  }
}

__global__ void label_counts_kernel(const int32_t* __restrict__ pred, const int32_t* __restrict__ truth,
                                    int64_t n, int K, unsigned long long* __restrict__ counts) {
  extern __shared__ unsigned int hist[];  // [K][3]
  for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int a = pred[i], b = truth[i];
    if ((unsigned)a < (unsigned)K) atomicAdd(&hist[a * 3 + 1], 1u);
    if ((unsigned)b < (unsigned)K) atomicAdd(&hist[b * 3 + 2], 1u);
    if (a == b && (unsigned)a < (unsigned)K) atomicAdd(&hist[a * 3 + 0], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * K; i += blockDim.x)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

struct CropList {
  int n;
  int b[kMaxWin], z[kMaxWin], y[kMaxWin], x[kMaxWin];
  unsigned char flip[kMaxWin];
};

template <typename TD>
__global__ void crop_kernel(const float* __restrict__ img, const float* __restrict__ lab, CropList cl,
                            int D, int H, int W, int C, int ldi, TD* __restrict__ oimg,
                            float* __restrict__ olab, int rd, int rh, int rw, int ldo) {
  const int64_t per = (int64_t)rd * rh * rw;
  const int64_t total = per * cl.n;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t t = e;
    const int x = t % rw; t /= rw;
    const int y = t % rh; t /= rh;
    const int z = t % rd;
    const int w = t / rd;
    const unsigned char f = cl.flip[w];
    const int sz = (f & 1) ? rd - 1 - z : z, sy = (f & 2) ? rh - 1 - y : y, sx = (f & 4) ? rw - 1 - x : x;
    const int gz = cl.z[w] + sz, gy = cl.y[w] + sy, gx = cl.x[w] + sx;
    const bool in = (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const int64_t gv = (((int64_t)cl.b[w] * D + gz) * H + gy) * W + gx;
    for (int c = 0; c < C; ++c) Elem<TD>::st(oimg + e * ldo + c, in ? img[gv * ldi + c] : 0.f);
    if (olab) olab[e] = (in && lab) ? lab[gv] : 0.f;
  }
}

static inline int grid_for(int64_t total) {
  const int64_t b = cdiv64(total, 256);
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}


// The same blend for schedules in which at most TWO windows cover a coordinate in every dimension (overlap <= 0.5:
// BASELINE config 3 and the reference's defaults): the <= 8 covering windows of a voxel are located first (z, y:
// wave-uniform; x: per lane), all their loads are issued back to back with a non-temporal hint (each of the 23 GB of
// predictions is read exactly once), then summed in ascending window order -- the same f32 sequence as
// sw_blend_kernel, bit for bit.  sw_blend_kernel finds and loads its windows inside three nested loops with
// per-lane conditions: hipcc keeps ONE 16-byte load per wave in flight, 32 KB per CU, and the pass streamed at
// 2.9 TB/s (round 3: 8 ms of a 43 ms volume).
template <typename T, typename L, int G>
__global__ __launch_bounds__(256) void sw_blend2_kernel(
    const T* __restrict__ cache, BlendSched sc, int lo, int hi, const float* __restrict__ imp,
    float* __restrict__ out, float* __restrict__ cnt_out, L* __restrict__ labels, int D, int H,
    int W, int K, int ldo, int rd, int rh, int rw, int ldp, int normalize) {
  typedef typename std::conditional<G == 8, u32x4, f32x4>::type vec_t;
  const int tpv = K / G;
  const int lanes_row = W * tpv;
  const int segs = (lanes_row + 255) / 256;
  const int nseg = D * H * segs;
  const int64_t wvox = (int64_t)rd * rh * rw;
  for (int s = blockIdx.x; s < nseg; s += gridDim.x) {
    const int seg = s % segs, row = s / segs;
    const int y = row % H, z = row / H;
    const int e = seg * 256 + (int)threadIdx.x;
    const bool live = e < lanes_row;
    const int x = live ? e / tpv : 0;
    const int k = live ? (e % tpv) * G : 0;
    const int64_t v = (int64_t)row * W + x;
    // first covering window and how many cover (1 or 2), per dimension
    int kz0 = 0, ky0 = 0;
    while (kz0 + 1 < sc.n[0] && sc.start[0][kz0] + rd <= z) ++kz0;
    while (ky0 + 1 < sc.n[1] && sc.start[1][ky0] + rh <= y) ++ky0;
    const int nzc = kz0 + 1 < sc.n[0] && sc.start[0][kz0 + 1] <= z ? 2 : 1;
    const int nyc = ky0 + 1 < sc.n[1] && sc.start[1][ky0 + 1] <= y ? 2 : 1;
    int kx0 = 0, sx0 = sc.start[2][0], sx1 = sc.start[2][0];
    bool two_x = false;
    for (int kx = 1; kx < sc.n[2]; ++kx) {
      const int st = sc.start[2][kx];
      const bool past = sx0 + rw <= x;              // the current first window ends before x: move on
      if (past) { kx0 = kx; sx0 = st; }
      else if (!two_x && kx == kx0 + 1 && st <= x) { two_x = true; sx1 = st; }
    }
    const int lz0 = z - sc.start[0][kz0], ly0 = y - sc.start[1][ky0];
    const int lz1 = nzc == 2 ? z - sc.start[0][kz0 + 1] : 0, ly1 = nyc == 2 ? y - sc.start[1][ky0 + 1] : 0;
    vec_t pv[8];
    float wt[8];
    bool ok[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int iz = c >> 2, iy = (c >> 1) & 1, ix = c & 1;
      const int w = ((kz0 + iz) * sc.n[1] + ky0 + iy) * sc.n[2] + kx0 + ix;
      ok[c] = live && iz < nzc && iy < nyc && (ix == 0 || two_x) && w >= lo && w < hi;
      const int lz = iz ? lz1 : lz0, ly = iy ? ly1 : ly0, lx = x - (ix ? sx1 : sx0);
      const int64_t lv = ok[c] ? ((int64_t)lz * rh + ly) * rw + lx : 0;
      const int64_t slot = ok[c] ? (int64_t)(w - lo) : 0;
      pv[c] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(cache + (slot * wvox + lv) * ldp + k));
      wt[c] = imp ? imp[lv] : 1.f;
    }
    float a[G];
#pragma unroll
    for (int j = 0; j < G; ++j) a[j] = 0.f;
    float cacc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (ok[c]) {
        if constexpr (G == 8) {
          const u32x4 o = pv[c];
          a[0] += wt[c] * __uint_as_float(o[0] << 16); a[1] += wt[c] * __uint_as_float(o[0] & 0xffff0000u);
          a[2] += wt[c] * __uint_as_float(o[1] << 16); a[3] += wt[c] * __uint_as_float(o[1] & 0xffff0000u);
          a[4] += wt[c] * __uint_as_float(o[2] << 16); a[5] += wt[c] * __uint_as_float(o[2] & 0xffff0000u);
          a[6] += wt[c] * __uint_as_float(o[3] << 16); a[7] += wt[c] * __uint_as_float(o[3] & 0xffff0000u);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] += wt[c] * pv[c][j];
        }
        cacc += wt[c];
      }
    }
    if (live) {
      if (normalize) {
#pragma unroll
        for (int j = 0; j < G; ++j) a[j] = a[j] / cacc;
      }
      if (out) {
#pragma unroll
        for (int j = 0; j < G; j += 4)
          *reinterpret_cast<f32x4*>(out + v * ldo + k + j) = f32x4{a[j], a[j + 1], a[j + 2], a[j + 3]};
      }
      if (cnt_out && k == 0) cnt_out[v] = cacc;
    }
    if (labels) {
      float bv = a[0];
      int bi = k;
#pragma unroll
      for (int j = 1; j < G; ++j)
        if (am_better(bv, bi, a[j], k + j)) { bv = a[j]; bi = k + j; }
      for (int o = 1; o < tpv; o <<= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (am_better(bv, bi, ov, oi)) { bv = ov; bi = oi; }
      }
      if (live && k == 0) labels[v] = (L)bi;
    }
  }
}

}  // namespace segmi

using namespace segmi;

extern "C" {

int segmi_sw_gather(int dtype_src, const segmi_act* image, int img_index,
                    const int32_t* starts_host, int nwin, int dst_dtype,
                    const segmi_act* windows, void* stream) {
  SEGMI_CHECK_ARG(act_ok(image) && act_ok(windows) && starts_host, "sw_gather: bad arguments");
  SEGMI_CHECK_ARG(nwin > 0 && nwin <= kMaxWin && windows->n >= nwin, "sw_gather: 1..%d windows per call", kMaxWin);
  SEGMI_CHECK_ARG(img_index >= 0 && img_index < image->n && image->c == windows->c, "sw_gather: image index / channels");
  WinList wl{};
  wl.n = nwin;
  for (int i = 0; i < nwin; ++i) { wl.z[i] = starts_host[3 * i]; wl.y[i] = starts_host[3 * i + 1]; wl.x[i] = starts_host[3 * i + 2]; }
  const int es = dtype_size(dtype_src);
  const char* base = (const char*)image->data + (int64_t)img_index * image->d * image->h * image->w * image->ld * es;
  const int64_t total = (int64_t)nwin * windows->d * windows->h * windows->w * windows->c;
  const int grid = grid_for(total);
  hipStream_t st = (hipStream_t)stream;
#define GATHER(TS, TD)                                                                          \
  hipLaunchKernelGGL((sw_gather_kernel<TS, TD>), grid, 256, 0, st, (const TS*)base,             \
                     (TD*)windows->data, wl, image->d, image->h, image->w, image->c, image->ld, \
                     windows->d, windows->h, windows->w, windows->ld)
  const int des = dtype_size(dst_dtype);
  if (image->c == 1 && windows->ld == 1 && windows->w % 4 == 0 && ((uintptr_t)windows->data % (4 * des)) == 0) {
    const int g4 = grid_for(total / 4);
#define GATHER4(TS, TD)                                                                          \
    hipLaunchKernelGGL((sw_gather4_kernel<TS, TD>), g4, 256, 0, st, (const TS*)base,              \
                       (TD*)windows->data, wl, image->d, image->h, image->w, image->ld,          \
                       windows->d, windows->h, windows->w)
    if (dtype_src == SEGMI_F32 && dst_dtype == SEGMI_F32) GATHER4(float, float);
    else if (dtype_src == SEGMI_F32 && dst_dtype == SEGMI_BF16) GATHER4(float, bf16_t);
    else if (dtype_src == SEGMI_BF16 && dst_dtype == SEGMI_BF16) GATHER4(bf16_t, bf16_t);
    else if (dtype_src == SEGMI_BF16 && dst_dtype == SEGMI_F32) GATHER4(bf16_t, float);
    else SEGMI_CHECK_ARG(false, "sw_gather: bad dtypes");
#undef GATHER4
    SEGMI_LAUNCH_CHECK("sw_gather");
    return SEGMI_OK;
  }
  if (dtype_src == SEGMI_F32 && dst_dtype == SEGMI_F32) GATHER(float, float);
  else if (dtype_src == SEGMI_F32 && dst_dtype == SEGMI_BF16) GATHER(float, bf16_t);
  else if (dtype_src == SEGMI_BF16 && dst_dtype == SEGMI_BF16) GATHER(bf16_t, bf16_t);
  else if (dtype_src == SEGMI_BF16 && dst_dtype == SEGMI_F32) GATHER(bf16_t, float);
  else SEGMI_CHECK_ARG(false, "sw_gather: bad dtypes");
#undef GATHER
  SEGMI_LAUNCH_CHECK("sw_gather");
  return SEGMI_OK;
}

int segmi_sw_scatter_add(int dtype, const segmi_act* pred, const int32_t* starts_host,
                         int nwin, const float* importance, const segmi_act* acc, float* cnt,
                         void* stream) {
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "sw_scatter_add: bad dtype");
  SEGMI_CHECK_ARG(act_ok(pred) && act_ok(acc) && starts_host && acc->n == 1, "sw_scatter_add: bad arguments");
  SEGMI_CHECK_ARG(nwin > 0 && nwin <= kMaxWin && pred->n >= nwin && pred->c == acc->c, "sw_scatter_add: windows / channels");
  WinList wl{};
  wl.n = nwin;
  int z0 = 1 << 30, y0 = 1 << 30, x0 = 1 << 30, z1 = -(1 << 30), y1 = -(1 << 30), x1 = -(1 << 30);
  for (int i = 0; i < nwin; ++i) {
    wl.z[i] = starts_host[3 * i]; wl.y[i] = starts_host[3 * i + 1]; wl.x[i] = starts_host[3 * i + 2];
    z0 = wl.z[i] < z0 ? wl.z[i] : z0; y0 = wl.y[i] < y0 ? wl.y[i] : y0; x0 = wl.x[i] < x0 ? wl.x[i] : x0;
    z1 = wl.z[i] + pred->d > z1 ? wl.z[i] + pred->d : z1;
    y1 = wl.y[i] + pred->h > y1 ? wl.y[i] + pred->h : y1;
    x1 = wl.x[i] + pred->w > x1 ? wl.x[i] + pred->w : x1;
  }
  z0 = z0 < 0 ? 0 : z0; y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
  z1 = z1 > acc->d ? acc->d : z1; y1 = y1 > acc->h ? acc->h : y1; x1 = x1 > acc->w ? acc->w : x1;
  if (z1 <= z0 || y1 <= y0 || x1 <= x0) return SEGMI_OK;
  const int bd = z1 - z0, bh = y1 - y0, bw = x1 - x0;
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype_size(dtype);
  if (acc->c % 4 == 0 && acc->ld % 4 == 0 && pred->ld % 4 == 0 && ((uintptr_t)acc->data % 16) == 0 &&
      ((uintptr_t)pred->data % (4 * es)) == 0) {
    const int g4 = grid_for((int64_t)bd * bh * bw * (acc->c / 4));
    if (dtype == SEGMI_F32)
      hipLaunchKernelGGL(sw_scatter4_kernel<float>, g4, 256, 0, st, (const float*)pred->data, wl, importance, (float*)acc->data, cnt, acc->d, acc->h, acc->w, acc->c, acc->ld, pred->d, pred->h, pred->w, pred->ld, z0, y0, x0, bd, bh, bw);
    else
      hipLaunchKernelGGL(sw_scatter4_kernel<bf16_t>, g4, 256, 0, st, (const bf16_t*)pred->data, wl, importance, (float*)acc->data, cnt, acc->d, acc->h, acc->w, acc->c, acc->ld, pred->d, pred->h, pred->w, pred->ld, z0, y0, x0, bd, bh, bw);
    SEGMI_LAUNCH_CHECK("sw_scatter_add");
    return SEGMI_OK;
  }
  const int grid = grid_for((int64_t)bd * bh * bw * acc->c);
  if (dtype == SEGMI_F32)
    hipLaunchKernelGGL(sw_scatter_kernel<float>, grid, 256, 0, st, (const float*)pred->data, wl, importance, (float*)acc->data, cnt, acc->d, acc->h, acc->w, acc->c, acc->ld, pred->d, pred->h, pred->w, pred->ld, z0, y0, x0, bd, bh, bw);
  else
    hipLaunchKernelGGL(sw_scatter_kernel<bf16_t>, grid, 256, 0, st, (const bf16_t*)pred->data, wl, importance, (float*)acc->data, cnt, acc->d, acc->h, acc->w, acc->c, acc->ld, pred->d, pred->h, pred->w, pred->ld, z0, y0, x0, bd, bh, bw);
  SEGMI_LAUNCH_CHECK("sw_scatter_add");
  return SEGMI_OK;
}

static int argmax_launch(int dtype, const segmi_act* lg, const float* cnt, int write_back,
                         void* labels, int label_bytes, hipStream_t st) {
  const int64_t nvox = act_voxels(lg);
  float* wb = write_back ? (float*)lg->data : nullptr;
  const int es = dtype_size(dtype);
  const int tpv = lg->c / 4;
  if (lg->c % 4 == 0 && tpv >= 1 && tpv <= 64 && (tpv & (tpv - 1)) == 0 && lg->ld % 4 == 0 &&
      ((uintptr_t)lg->data % (4 * es)) == 0) {
    const int g4 = grid_for(nvox * tpv);
#define ARGMAX4(T, L)                                                                      \
  hipLaunchKernelGGL((argmax4_kernel<T, L>), g4, 256, 0, st, (const T*)lg->data, cnt, wb, \
                     (L*)labels, nvox, lg->c, lg->ld)
    if (dtype == SEGMI_F32) {
      if (label_bytes == 1) ARGMAX4(float, uint8_t);
      else if (label_bytes == 2) ARGMAX4(float, int16_t);
      else ARGMAX4(float, int32_t);
    } else {
      if (label_bytes == 1) ARGMAX4(bf16_t, uint8_t);
      else if (label_bytes == 2) ARGMAX4(bf16_t, int16_t);
      else ARGMAX4(bf16_t, int32_t);
    }
#undef ARGMAX4
    SEGMI_LAUNCH_CHECK("argmax");
    return SEGMI_OK;
  }
  const int grid = grid_for(nvox);
#define ARGMAX(T, L)                                                                     \
  hipLaunchKernelGGL((argmax_kernel<T, L>), grid, 256, 0, st, (const T*)lg->data, cnt, wb, \
                     (L*)labels, nvox, lg->c, lg->ld)
  if (dtype == SEGMI_F32) {
    if (label_bytes == 1) ARGMAX(float, uint8_t);
    else if (label_bytes == 2) ARGMAX(float, int16_t);
    else ARGMAX(float, int32_t);
  } else {
    if (label_bytes == 1) ARGMAX(bf16_t, uint8_t);
    else if (label_bytes == 2) ARGMAX(bf16_t, int16_t);
    else ARGMAX(bf16_t, int32_t);
  }
#undef ARGMAX
  SEGMI_LAUNCH_CHECK("argmax");
  return SEGMI_OK;
}

int segmi_sw_finalize(const segmi_act* acc, const float* cnt, int write_logits, void* labels,
                      int label_bytes, void* stream) {
  SEGMI_CHECK_ARG(act_ok(acc) && cnt && labels, "sw_finalize: bad arguments");
  SEGMI_CHECK_ARG(label_bytes == 1 || label_bytes == 2 || label_bytes == 4, "sw_finalize: label_bytes");
  SEGMI_CHECK_ARG(label_bytes > 1 || acc->c <= 256, "sw_finalize: uint8 labels hold at most 256 classes");
  return argmax_launch(SEGMI_F32, acc, cnt, write_logits, labels, label_bytes, (hipStream_t)stream);
}

int segmi_argmax(int dtype, const segmi_act* logits, void* labels, int label_bytes, void* stream) {
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "argmax: bad dtype");
  SEGMI_CHECK_ARG(act_ok(logits) && labels, "argmax: bad arguments");
  SEGMI_CHECK_ARG(label_bytes == 1 || label_bytes == 2 || label_bytes == 4, "argmax: label_bytes");
  return argmax_launch(dtype, logits, nullptr, 0, labels, label_bytes, (hipStream_t)stream);
}

int segmi_sw_blend(int dtype, const void* cache, int k, int ldp, const int32_t* starts_z, int nz,
                   const int32_t* starts_y, int ny, const int32_t* starts_x, int nx, int win_lo,
                   int win_hi, int rd, int rh, int rw, const float* importance, int d, int h, int w,
                   float* out_logits, int ldo, float* out_count, void* labels, int label_bytes,
                   int normalize, void* stream) {
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "sw_blend: bad dtype");
  SEGMI_CHECK_ARG(cache && starts_z && starts_y && starts_x && k > 0 && ldp >= k && rd > 0 &&
                      rh > 0 && rw > 0 && d > 0 && h > 0 && w > 0,
                  "sw_blend: bad arguments");
  SEGMI_CHECK_ARG(out_logits || labels, "sw_blend: nothing to write");
  SEGMI_CHECK_ARG(!out_logits || ldo >= k, "sw_blend: bad logits row stride");
  if (nz > kMaxStarts || ny > kMaxStarts || nx > kMaxStarts || nz < 1 || ny < 1 || nx < 1)
    SEGMI_UNSUPPORTED("sw_blend: at most %d window origins per dimension", kMaxStarts);
  SEGMI_CHECK_ARG(0 <= win_lo && win_lo < win_hi && (int64_t)win_hi <= (int64_t)nz * ny * nx,
                  "sw_blend: window range [%d, %d) outside the schedule", win_lo, win_hi);
  SEGMI_CHECK_ARG(!labels || normalize, "sw_blend: labels need the normalised blend");
  SEGMI_CHECK_ARG(!labels || label_bytes == 1 || label_bytes == 2 || label_bytes == 4,
                  "sw_blend: label_bytes");
  SEGMI_CHECK_ARG(!labels || label_bytes > 1 || k <= 256, "sw_blend: uint8 labels hold at most 256 classes");
  BlendSched sc{};
  sc.n[0] = nz; sc.n[1] = ny; sc.n[2] = nx;
  for (int i = 0; i < nz; ++i) sc.start[0][i] = starts_z[i];
  for (int i = 0; i < ny; ++i) sc.start[1][i] = starts_y[i];
  for (int i = 0; i < nx; ++i) sc.start[2][i] = starts_x[i];
  const int es = dtype_size(dtype);
  const int gfull = 16 / es;
  const bool vec = k % gfull == 0 && ldp % gfull == 0 && ((uintptr_t)cache % 16) == 0 &&
                   (!out_logits || (ldo % 4 == 0 && ((uintptr_t)out_logits % 16) == 0)) &&
                   (k / gfull) <= 64 && ((k / gfull) & (k / gfull - 1)) == 0;
  SEGMI_CHECK_ARG(vec || !labels || k <= 64,
                  "sw_blend: the scalar path (K %% %d != 0) labels at most 64 classes", gfull);
  hipStream_t st = (hipStream_t)stream;
  const int64_t row_segs = (int64_t)d * h * (((int64_t)w * (vec ? k / gfull : k) + 255) / 256);
  SEGMI_CHECK_ARG(row_segs < (1ll << 31) && (int64_t)w * k < (1ll << 30), "sw_blend: volume too large");
  const int grid = (int)(row_segs < 16384 ? row_segs : 16384);
  // at most two windows cover any coordinate in every dimension (start[i + 2] >= start[i] + roi): the variant that
  // issues all covering loads up front (SEGMI_SW_BLEND2=0: the generic kernel, for A/B)
  static const bool blend2_on = !(getenv("SEGMI_SW_BLEND2") && atoi(getenv("SEGMI_SW_BLEND2")) == 0);
  bool two = blend2_on;
  {
    const int nn[3] = {nz, ny, nx}, rr[3] = {rd, rh, rw};
    for (int dd = 0; dd < 3 && two; ++dd)
      for (int i = 0; i + 2 < nn[dd]; ++i)
        if (sc.start[dd][i + 2] < sc.start[dd][i] + rr[dd]) { two = false; break; }
    for (int dd = 0; dd < 3 && two; ++dd)          // ascending, gap-free coverage (MONAI's dense schedule)
      for (int i = 0; i + 1 < nn[dd]; ++i)
        if (sc.start[dd][i + 1] <= sc.start[dd][i] || sc.start[dd][i + 1] > sc.start[dd][i] + rr[dd]) { two = false; break; }
  }
#define BLEND(TT, LL, GG)                                                                         \
  do {                                                                                            \
    if (two)                                                                                      \
      hipLaunchKernelGGL((sw_blend2_kernel<TT, LL, GG>), grid, 256, 0, st, (const TT*)cache, sc, win_lo, \
                         win_hi, importance, out_logits, out_count, (LL*)labels, d, h, w, k, ldo, rd, \
                         rh, rw, ldp, normalize);                                                 \
    else                                                                                          \
      hipLaunchKernelGGL((sw_blend_kernel<TT, LL, GG>), grid, 256, 0, st, (const TT*)cache, sc, win_lo, \
                         win_hi, importance, out_logits, out_count, (LL*)labels, d, h, w, k, ldo, rd, \
                         rh, rw, ldp, normalize);                                                 \
  } while (0)
  if (vec) {
    if (dtype == SEGMI_BF16) {
      if (label_bytes == 1) BLEND(bf16_t, uint8_t, 8);
      else if (label_bytes == 2) BLEND(bf16_t, uint16_t, 8);
      else BLEND(bf16_t, int32_t, 8);
    } else {
      if (label_bytes == 1) BLEND(float, uint8_t, 4);
      else if (label_bytes == 2) BLEND(float, uint16_t, 4);
      else BLEND(float, int32_t, 4);
    }
  } else {
    // scalar lanes: one channel each; labels (if wanted) come from a second pass over the logits
    SEGMI_CHECK_ARG(!labels || out_logits, "sw_blend: the scalar path labels from the written logits");
    if (dtype == SEGMI_BF16) hipLaunchKernelGGL((sw_blend_kernel<bf16_t, int32_t, 1>), grid, 256, 0, st,
        (const bf16_t*)cache, sc, win_lo, win_hi, importance, out_logits, out_count, (int32_t*)nullptr,
        d, h, w, k, ldo, rd, rh, rw, ldp, normalize);
    else hipLaunchKernelGGL((sw_blend_kernel<float, int32_t, 1>), grid, 256, 0, st,
        (const float*)cache, sc, win_lo, win_hi, importance, out_logits, out_count, (int32_t*)nullptr,
        d, h, w, k, ldo, rd, rh, rw, ldp, normalize);
    SEGMI_LAUNCH_CHECK("sw_blend");
    if (labels) {
      segmi_act lg{out_logits, 1, d, h, w, k, ldo};
      return argmax_launch(SEGMI_F32, &lg, nullptr, 0, labels, label_bytes, st);
    }
    return SEGMI_OK;
  }
#undef BLEND
  SEGMI_LAUNCH_CHECK("sw_blend");
  return SEGMI_OK;
}

int segmi_label_counts(const int32_t* pred, const int32_t* truth, int64_t n, int k,
                       int64_t* counts, void* stream) {
  SEGMI_CHECK_ARG(pred && truth && counts && n > 0 && k > 0 && k <= 4096, "label_counts: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(counts, 0, (size_t)k * 3 * 8, st) != hipSuccess) {
    set_error("label_counts: memset failed");
    return SEGMI_ELAUNCH;
  }
  const int grid = grid_for(n) > 1024 ? 1024 : grid_for(n);
  hipLaunchKernelGGL(label_counts_kernel, grid, 256, (size_t)k * 3 * 4, st, pred, truth, n, k,
                     (unsigned long long*)counts);
  SEGMI_LAUNCH_CHECK("label_counts");
  return SEGMI_OK;
}

int segmi_crop_patches(const segmi_act* image, const float* label, const int32_t* starts_host,
                       const uint8_t* flips_host, int count, int dst_dtype,
                       const segmi_act* out_image, float* out_label, void* stream) {
  SEGMI_CHECK_ARG(act_ok(image) && act_ok(out_image) && starts_host, "crop_patches: bad arguments");
  SEGMI_CHECK_ARG(count > 0 && count <= kMaxWin && out_image->n >= count && out_image->c == image->c,
                  "crop_patches: 1..%d crops per call", kMaxWin);
  CropList cl{};
  cl.n = count;
  for (int i = 0; i < count; ++i) {
    cl.b[i] = starts_host[4 * i]; cl.z[i] = starts_host[4 * i + 1];
    cl.y[i] = starts_host[4 * i + 2]; cl.x[i] = starts_host[4 * i + 3];
    cl.flip[i] = flips_host ? flips_host[i] : 0;
    SEGMI_CHECK_ARG(cl.b[i] >= 0 && cl.b[i] < image->n, "crop_patches: volume index out of range");
  }
  const int64_t total = (int64_t)count * out_image->d * out_image->h * out_image->w;
  const int grid = grid_for(total);
  hipStream_t st = (hipStream_t)stream;
  if (dst_dtype == SEGMI_F32)
    hipLaunchKernelGGL(crop_kernel<float>, grid, 256, 0, st, (const float*)image->data, label, cl, image->d, image->h, image->w, image->c, image->ld, (float*)out_image->data, out_label, out_image->d, out_image->h, out_image->w, out_image->ld);
  else
    hipLaunchKernelGGL(crop_kernel<bf16_t>, grid, 256, 0, st, (const float*)image->data, label, cl, image->d, image->h, image->w, image->c, image->ld, (bf16_t*)out_image->data, out_label, out_image->d, out_image->h, out_image->w, out_image->ld);
  SEGMI_LAUNCH_CHECK("crop_patches");
  return SEGMI_OK;
}

}  // extern "C"
