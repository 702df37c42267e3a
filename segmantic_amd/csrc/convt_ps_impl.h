// convt_ps_impl.h -- persistent ConvTranspose3d k3 s2 p1 forward for the wide, few-channel layers
// (top of the decoder and the dgrad of the first stride-2 convs), where the op is HBM-bound and
// the tile kernel of convt_fwd_impl.h is latency/VALU-bound (1.8-2.0 TB/s measured).
//
//  * every weight fragment of the layer (<= 54 KB: 16 output channels and one or two channel chunks, or 32 -> 32 in bf16) is staged into LDS once
//    per workgroup and reused over `tiles_per_wg` consecutive input tiles (2 x 2 x 16 voxels);
//  * the next tile's halo (3 x 3 x 17 voxels, all channels) is fetched into registers while the
//    current one is multiplied and stored;
//  * one wave owns one 16-voxel x-row and computes all 8 output-parity classes for it from 8
//    voxel fragments (27 MFMAs per 32/16 channels, perfectly balanced over the 4 waves);
//  * the two x-parity classes of an output row are exchanged between lane rows with
//    v_permlane16_swap so that every lane stores 8 consecutive channels: one fully coalesced
//    32-voxel row segment per store instruction instead of two interleaved half-filled ones;
//  * BatchNorm statistics accumulate in registers over all tiles of the workgroup: one partial
//    row per workgroup.
#pragma once
#include "convt_fwd_impl.h"

namespace segmi {

constexpr int kCtPsMaxTiles = 16;     // tiles per workgroup (upper bound)
constexpr int kCtPsTargetWgs = 1024;  // workgroups wanted before tiles are grouped

static inline int convt_ps_tiles(const int n, const int di, const int hi, const int wi) {
  return n * cdiv(di, 2) * cdiv(hi, 2) * cdiv(wi, 16);
}
static inline int convt_ps_per_wg(int ntiles) {
  int per = ntiles / kCtPsTargetWgs;
  return per < 1 ? 1 : (per > kCtPsMaxTiles ? kCtPsMaxTiles : per);
}
static inline int convt_ps_grid(int ntiles) { return cdiv(ntiles, convt_ps_per_wg(ntiles)); }
// layers the persistent kernel takes (everything else: convt_fwd_impl.h)
static inline bool convt_ps_ok(int dtype, int cin, int cout, int wi) {
  const int ck = dtype == SEGMI_F32 ? 16 : 32;
  if (cin % ck || cout % 16 || wi < 16) return false;
  const int nch = cin / ck, nt = cout / 16;
  return (nt == 1 && nch <= 2) || (nt == 2 && nch == 1 && dtype == SEGMI_BF16);
}

// partial-statistics rows the MFMA transposed-conv path writes (one per workgroup)
static inline int convt_mfma_rows(int dtype, const segmi_act* in, int cout) {
  if (convt_ps_ok(dtype, in->c, cout, in->w))
    return convt_ps_grid(convt_ps_tiles(in->n, in->d, in->h, in->w));
  return convt_tile_rows(dtype, in);
}

constexpr int ctps_class_off(int cls, int nch, int nt) {
  int o = 0;
  for (int p = 0; p < cls; ++p) o += nch * ct_ntaps_c(p) * nt * 1024;
  return o;
}
// index (dd*4 + dh*2 + dw) of the voxel fragment tap t of class cls multiplies
constexpr int ctps_tap_a(int cls, int t) {
  const int rw = cls & 1, rh = (cls >> 1) & 1, rd = (cls >> 2) & 1;
  const int nw = 1 + rw, nh = 1 + rh;
  const int tw = t % nw, th = (t / nw) % nh, td = t / (nw * nh);
  return (rd ? td : 0) * 4 + (rh ? th : 0) * 2 + (rw ? tw : 0);
}

template <typename T> struct Vec8;   // 8 consecutive channels held by one lane
template <> struct Vec8<float> {
  __device__ static void load(const float* p, f32x4& a, f32x4& b) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ static void store(float* p, f32x4 a, f32x4 b) {
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};
template <> struct Vec8<bf16_t> {
  __device__ static void load(const bf16_t* p, f32x4& a, f32x4& b) {
    const u32x4 o = *reinterpret_cast<const u32x4*>(p);
    a[0] = __uint_as_float(o[0] << 16); a[1] = __uint_as_float(o[0] & 0xffff0000u);
    a[2] = __uint_as_float(o[1] << 16); a[3] = __uint_as_float(o[1] & 0xffff0000u);
    b[0] = __uint_as_float(o[2] << 16); b[1] = __uint_as_float(o[2] & 0xffff0000u);
    b[2] = __uint_as_float(o[3] << 16); b[3] = __uint_as_float(o[3] & 0xffff0000u);
  }
  __device__ static void store(bf16_t* p, f32x4 a, f32x4 b) {
    u32x4 o;
    o[0] = pack_bf16x2(a[0], a[1]); o[1] = pack_bf16x2(a[2], a[3]);
    o[2] = pack_bf16x2(b[0], b[1]); o[3] = pack_bf16x2(b[2], b[3]);
    *reinterpret_cast<u32x4*>(p) = o;
  }
};

// lanes 16-31 of `a` <-> lanes 0-15 of `b`, lanes 48-63 of `a` <-> lanes 32-47 of `b`
__device__ __forceinline__ void swap_rows16(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

template <typename T, int NT, int NCH>
struct CtPsGeom {
  static constexpr int ES = (int)sizeof(T);
  static constexpr int CK = ES == 2 ? 32 : 16;
  static constexpr int HD = 3, HH = 3, HW = 17, NHV = HD * HH * HW;
  static constexpr int RAWB = CK * NCH * ES, ROWB = RAWB + 16, CPR = RAWB / 16;
  static constexpr int NCHK = NHV * CPR, NLD = (NCHK + 255) / 256;
  static constexpr int WBYTES = NCH * 27 * NT * 1024;
  static constexpr int IN_BYTES = NHV * ROWB;
  static constexpr int RED_BYTES = 4 * 2 * NT * 16 * 4;
  static constexpr int LDS_BYTES = WBYTES + (IN_BYTES > RED_BYTES ? IN_BYTES : RED_BYTES);
};

template <typename T, int NT, int NCH>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void convt_ps_kernel(ConvTParams p, int ntiles, int per_wg) {
  using G = CtPsGeom<T, NT, NCH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;
  char* il = smem + G::WBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int wz = wave >> 1, wy = wave & 1;   // the wave's x-row inside the 2 x 2 x 16 tile

  // all weight fragments -> LDS (the pack is [class][chunk][tap][ntile][lane][16 B], contiguous)
  for (int i = tid; i < G::WBYTES / 16; i += 256)
    reinterpret_cast<frag_t*>(wl)[i] = reinterpret_cast<const frag_t*>(p.wfrag)[i];

  // staging descriptors (tile independent): global offset relative to the tile's first voxel,
  // LDS offset, halo coordinates packed as hz | hy << 2 | hx << 4
  int s_goff[G::NLD], s_loff[G::NLD], s_h[G::NLD];
#pragma unroll
  for (int k = 0; k < G::NLD; ++k) {
    const int i = tid + 256 * k;
    const int v = i / G::CPR, ch = i % G::CPR;
    const int hx = v % G::HW, hy = (v / G::HW) % G::HH, hz = v / (G::HW * G::HH);
    s_loff[k] = v * G::ROWB + ch * 16;
    s_goff[k] = ((hz * p.Hi + hy) * p.Wi + hx) * p.ldi * G::ES + ch * 16;
    s_h[k] = i < G::NCHK ? (hz | (hy << 2) | (hx << 4)) : -1;
  }
  const int vaddr = ((wz * G::HH + wy) * G::HW + r) * G::ROWB + g * 16;

  f32x4 bias4[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4[j] = *reinterpret_cast<const f32x4*>(p.bias + j * 16 + 4 * g);
  }
  const bool has_alpha = p.alpha != nullptr;
  const float alpha = has_alpha ? *p.alpha : 0.f;
  f32x4 ssum[NT], ssq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    ssum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ssq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int t_begin = blockIdx.x * per_wg;
  const int t_end = t_begin + per_wg < ntiles ? t_begin + per_wg : ntiles;

  // Tile coordinates are carried, not decoded: a workgroup walks consecutive tiles (x fastest), so
  // the next tile is an increment with three possible wraps instead of three runtime divisions
  // (SALU divisions are ~20-instruction sequences and every SALU instruction is an issue turn of
  // the wave: the loop had 452 of them around 27 MFMAs).  (nx*) = tile being fetched, (c*) = tile
  // being computed.
  int nx_x, nx_y, nx_z, nx_n;
  {
    int t = t_begin;
    nx_x = t % p.tx; t /= p.tx;
    nx_y = t % p.ty; t /= p.ty;
    nx_z = t % p.tz;
    nx_n = t / p.tz;
  }
  int c_x = nx_x, c_y = nx_y, c_z = nx_z, c_n = nx_n;
  frag_t pf[G::NLD];
  auto fetch = [&]() {
    const int txi = nx_x, tyi = nx_y, tzi = nx_z, n = nx_n;
    const int iz0 = tzi * 2, iy0 = tyi * 2, ix0 = txi * 16;
    const char* tile_in = (const char*)p.in +
        ((((int64_t)n * p.Di + iz0) * p.Hi + iy0) * p.Wi + ix0) * p.ldi * (int64_t)G::ES;
    const bool interior = iz0 + 2 < p.Di && iy0 + 2 < p.Hi && ix0 + 16 < p.Wi;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
      const int h = s_h[k];
      bool ok = h >= 0;
      if (!interior)
        ok = ok && iz0 + (h & 3) < p.Di && iy0 + ((h >> 2) & 3) < p.Hi && ix0 + (h >> 4) < p.Wi;
      pf[k] = frag_t{0u, 0u, 0u, 0u};
      if (ok) pf[k] = *reinterpret_cast<const frag_t*>(tile_in + s_goff[k]);
    }
    // advance to the tile after this one
    if (++nx_x == p.tx) {
      nx_x = 0;
      if (++nx_y == p.ty) {
        nx_y = 0;
        if (++nx_z == p.tz) { nx_z = 0; ++nx_n; }
      }
    }
  };

  // element strides of one output row / plane (loop invariant)
  const int64_t o_dy = (int64_t)p.Wo * p.ldo, o_dz = (int64_t)p.Ho * o_dy;
  const int64_t r_dy = (int64_t)p.Wo * p.ldr, r_dz = (int64_t)p.Ho * r_dy;
  if (t_begin < t_end) fetch();
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();   // weights staged (first pass) / previous tile's fragments consumed
#pragma unroll
    for (int k = 0; k < G::NLD; ++k)
      if (s_h[k] >= 0) *reinterpret_cast<frag_t*>(il + s_loff[k]) = pf[k];
    __syncthreads();
    const int txi = c_x, tyi = c_y, tzi = c_z, n = c_n;    // the tile now in LDS
    c_x = nx_x; c_y = nx_y; c_z = nx_z; c_n = nx_n;
    if (tile + 1 < t_end) fetch();   // in flight under the MFMAs and the stores below

    f32x4 acc[8][NT];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[c8][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      frag_t a[8];
#pragma unroll
      for (int o = 0; o < 8; ++o)
        a[o] = *reinterpret_cast<const frag_t*>(
            il + vaddr + ((((o >> 2) & 1) * G::HH + ((o >> 1) & 1)) * G::HW + (o & 1)) * G::ROWB +
            c * G::CK * G::ES);
#pragma unroll
      for (int cls = 0; cls < 8; ++cls) {
#pragma unroll
        for (int t = 0; t < ct_ntaps_c(cls); ++t) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const frag_t w = *reinterpret_cast<const frag_t*>(
                wl + ctps_class_off(cls, NCH, NT) + ((c * ct_ntaps_c(cls) + t) * NT + j) * 1024 +
                lane * 16);
            acc[cls][j] = mma16<T>(w, a[ctps_tap_a(cls, t)], acc[cls][j]);
          }
        }
      }
    }

    // ---- epilogue: bias, statistics, PReLU, x-parity exchange, residual, store
    const int iz = tzi * 2 + wz, iy = tyi * 2 + wy, ix = txi * 16 + r;
    const int ox_pre0 = 2 * ix, ox_st = 2 * ix + (g & 1);
    T* outp = (T*)p.out;
    const T* resp = (const T*)p.res;
    // per-lane 32-bit element offsets inside an output row; the row base is wave-uniform
    const int lo_out = ox_st * p.ldo + 8 * (g >> 1);
    const int lo_res = ox_st * p.ldr + 8 * (g >> 1);
    // one 64-bit row index per tile; the other three rows are +1 row / +1 plane
    const int64_t row00 = (((int64_t)n * p.Do + 2 * iz) * p.Ho + 2 * iy) * p.Wo;
    T* const orow00 = outp + row00 * p.ldo;
    const T* const rrow00 = resp ? resp + row00 * p.ldr : nullptr;
#pragma unroll
    for (int dh2 = 0; dh2 < 4; ++dh2) {
      const int rd = dh2 >> 1, rh = dh2 & 1;
      const int oz = 2 * iz + rd, oy = 2 * iy + rh;
      const bool zy_ok = oz < p.Do && oy < p.Ho;   // wave-uniform
      T* orow = orow00 + (rd ? o_dz : 0) + (rh ? o_dy : 0);
      const T* rrow = rrow00 ? rrow00 + (rd ? r_dz : 0) + (rh ? r_dy : 0) : nullptr;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 v0 = acc[rd * 4 + rh * 2 + 0][j] + bias4[j];
        f32x4 v1 = acc[rd * 4 + rh * 2 + 1][j] + bias4[j];
        if (p.stats) {
          const float m0 = (zy_ok && ox_pre0 < p.Wo) ? 1.f : 0.f;
          const float m1 = (zy_ok && ox_pre0 + 1 < p.Wo) ? 1.f : 0.f;
          const f32x4 u0 = v0 * m0, u1 = v1 * m1;
          ssum[j] += u0 + u1;
          ssq[j] += u0 * v0 + u1 * v1;
        }
        if (has_alpha) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = v0[e] > 0.f ? v0[e] : alpha * v0[e];
            v1[e] = v1[e] > 0.f ? v1[e] : alpha * v1[e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = v0[e], b = v1[e];
          swap_rows16(a, b);
          v0[e] = a;
          v1[e] = b;
        }
        // lane (r, g) now holds channels j*16 + 8*(g>>1) .. +7 of output voxel 2*ix + (g&1)
        if (zy_ok && ox_st < p.Wo) {
          if (rrow) {
            f32x4 r0, r1;
            Vec8<T>::load(rrow + lo_res + j * 16, r0, r1);
            v0 += r0;
            v1 += r1;
          }
          Vec8<T>::store(orow + lo_out + j * 16, v0, v1);
        }
      }
    }
  }

  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(il);  // [wave][2][NT*16]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = row16_sum(ssum[j][e]);
        const float b = row16_sum(ssq[j][e]);
        if (r == 0) {
          red[(wave * 2 + 0) * NT * 16 + j * 16 + 4 * g + e] = a;
          red[(wave * 2 + 1) * NT * 16 + j * 16 + 4 * g + e] = b;
        }
      }
    __syncthreads();
    if (tid < 2 * NT * 16) {
      const int which = tid / (NT * 16), ch = tid % (NT * 16);
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sacc += red[(w * 2 + which) * NT * 16 + ch];
      fin_store(&p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + ch], sacc);
    }
    fin_tail_run<BnFin, 256, offsetof(ConvTParams, ft), offsetof(ConvTParams, bfin)>(p.stats, smem);
  }
}

template <typename T, int NT, int NCH>
static int launch_convt_ps_cfg(ConvTParams p, hipStream_t st) {
  using G = CtPsGeom<T, NT, NCH>;
  p.tz = cdiv(p.Di, 2);
  p.ty = cdiv(p.Hi, 2);
  p.tx = cdiv(p.Wi, 16);
  const int64_t nt64 = (int64_t)p.N * p.tz * p.ty * p.tx;
  SEGMI_CHECK_ARG(nt64 < (1ll << 31), "convT3d: too many tiles");
  const int ntiles = (int)nt64, per = convt_ps_per_wg(ntiles);
  // 32-bit staging offsets inside one halo tile
  SEGMI_CHECK_ARG((int64_t)3 * p.Hi * p.Wi * p.ldi * G::ES < (1ll << 31),
                  "convT3d: input plane too large for the persistent kernel");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)convt_ps_kernel<T, NT, NCH>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_set = true;
  }
  p.fin_on = p.fin_on && p.stats;
  const size_t lds = fin_tail_arm(p, dim3((unsigned)cdiv(ntiles, per)), 256, 2 * p.Cout, G::LDS_BYTES);
  SEGMI_CHECK_ARG(lds == (size_t)G::LDS_BYTES, "convT3d: LDS budget");
  hipLaunchKernelGGL((convt_ps_kernel<T, NT, NCH>), cdiv(ntiles, per), 256, G::LDS_BYTES, st, p,
                     ntiles, per);
  SEGMI_LAUNCH_CHECK("convT3d_fwd(persistent)");
  return SEGMI_OK;
}

template <typename T>
static int launch_convt_ps_t(const ConvTParams& p, hipStream_t st) {
  constexpr int CK = sizeof(T) == 2 ? 32 : 16;
  const int nch = p.Cin / CK, nt = p.Cout / 16;
  SEGMI_CHECK_ARG((nt == 1 && (nch == 1 || nch == 2)) || (nt == 2 && nch == 1),
                  "convT3d: persistent kernel misuse");
  if constexpr (sizeof(T) == 2) {
    if (nt == 2) return launch_convt_ps_cfg<T, 2, 1>(p, st);
  }
  if (nch == 1) return launch_convt_ps_cfg<T, 1, 1>(p, st);
  return launch_convt_ps_cfg<T, 1, 2>(p, st);
}

}  // namespace segmi
