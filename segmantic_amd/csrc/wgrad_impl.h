// wgrad_impl.h -- MFMA weight gradient of Conv3d k3 (stride 1 or 2).
//
//   dW[co][ci][tap] = sum_{n,vox} dY[n,vox,co] * X[n, vox*S + tap - 1, ci]
//
// GEMM with the huge voxel axis as K: D[co][ci] += A[co][k=vox] * B[k=vox][ci].  Both operands
// need "8 voxels of one channel" per lane, i.e. the transpose of the NDHWC rows staged in LDS:
//   bf16 -> ds_read_b64_tr_b16 (gfx950 transposing LDS read; 2 reads = one 16x16x32 operand)
//   f32  -> ds_read_b32 gathers (one 16x16x4 operand each)
// The k -> voxel assignment is free as long as A and B agree; we use voxel = 16h + 4g + q
// (h = which tr-read, g = lane>>4, q = row the lane addresses) so that every 32-lane LDS
// group touches 256 contiguous bytes (conflict-free for stride-1 convs).
//
// Work split: workgroup = (voxel-tile range) x (16*CT out-channels) x (16*CT in-channels);
// its 4 waves own taps {w, w+4, ...} (7/7/7/6) so no cross-wave reduction is needed, and keep
// their 7*CT*CT accumulators in registers across all tiles of the range (persistent loop).
// Per-workgroup partial slabs are reduced in fixed order by wgrad_reduce_kernel: bitwise
// reproducible, no atomics.
#pragma once
#include "common.h"

// Time-split diagnostic (build with -DSEGMI_WGRAD_DIAG, run with SEGMI_WGRAD_DBG=bits: 1 = no global
// loads, 4 = no LDS-read / MFMA loop; scripts/wgrad_diag.py).  Measured on the 128^3 x 8, 16 x 16 layer
// (cold caches, us): all 444, no loads 316, no MFMA loop 254, neither 137 -- the three phases of a
// tile barely overlap at 2 workgroups per CU; the MFMA-loop phase sits at the LDS instruction
// rate (2 ds_read_b64_tr_b16 per operand, ~2.4 clk each, 2.3 per MFMA).
#ifdef SEGMI_WGRAD_DIAG
#define WGRAD_DBG(p, bit) (((p).dbg & (bit)) != 0)
#else
#define WGRAD_DBG(p, bit) false
#endif

namespace segmi {

struct WgradParams {
  const void* x;
  const void* dy;
  float* partials;
  int N, Dx, Hx, Wx, Dy, Hy, Wy, Cin, Cout, ldx, ldy;
  int tz, ty, tx;
  int ntiles;
  int zs, zper;   // wave-specialised kernel: z-segments per column, z-tiles per segment
  unsigned x_bytes, y_bytes;   // ... and the byte extents of the two tensors (buffer num_records)
  unsigned long long* stamps;  // diag build: s_memtime stamps of workgroup 0, [iter][12 waves][4]
  int ci_chunks;
  int dbg;
  int zmarch;     // wave-specialised kernel: shared input planes of consecutive z-tiles are copied LDS -> LDS (1 = on)
  // optional input transform of X (segmi_in_affine): the BatchNorm-apply + PReLU that produced the
  // forward input is applied while the X tile is committed to LDS (bf16 only)
  const float* in_scale;
  const float* in_shift;
  const float* in_alpha;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) float lds_f32;

template <typename T, int KS_, int S, int CTO, int CTI, int TD, int TH, int TW>
struct WgradGeom {
  static constexpr int KS = KS_, PAD = (KS_ - 1) / 2, NTAPS = KS_ * KS_ * KS_;
  static constexpr int NTW = (NTAPS + 3) / 4;  // taps per wave
  static constexpr int NV = TD * TH * TW;
  static constexpr int NL = NV / 16;  // 16-voxel lines
  static constexpr int HD = (TD - 1) * S + KS, HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
  static constexpr int YROWB = 16 * CTO * (int)sizeof(T), XROWB = 16 * CTI * (int)sizeof(T);
  static constexpr int YCPR = YROWB / 16, XCPR = XROWB / 16;
  static constexpr int YBYTES = NV * YROWB;
  static constexpr int XROWS = HD * HH * HW;
  static constexpr int LDS_BYTES = YBYTES + XROWS * XROWB;
  static constexpr int LPG = sizeof(T) == 2 ? 2 : 1;  // lines per mma16
  static_assert(NL % LPG == 0, "tile must hold whole k-steps");
  static_assert(TW == 8 || TW == 16, "tile width");
};

// row (in X halo rows) of voxel `v` (0..15) of line `line`, before the tap offset
template <int S, int TH, int TW, int HH, int HW>
__device__ __forceinline__ constexpr int wg_line_row(int line) {
  // a line is 16 consecutive voxel indices: TW=16 -> one x-row, TW=8 -> two x-rows
  const int first = line * 16;
  const int y = (first / TW) % TH, z = first / (TW * TH);
  return (z * S * HH + y * S) * HW;
}

template <typename T, int KS, int S, int CTO, int CTI, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradParams p) {
  using G = WgradGeom<T, KS, S, CTO, CTI, TD, TH, TW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ysm = smem;
  char* xsm = smem + G::YBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i16 = lane & 15;

  const int cochunk = blockIdx.y / p.ci_chunks, cichunk = blockIdx.y % p.ci_chunks;
  const int co0 = cochunk * 16 * CTO, ci0 = cichunk * 16 * CTI;

  f32x4 acc[G::NTW][CTO][CTI];
#pragma unroll
  for (int a = 0; a < G::NTW; ++a)
#pragma unroll
    for (int b = 0; b < CTO; ++b)
#pragma unroll
      for (int c = 0; c < CTI; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // wave-uniform tap offsets (bytes into the X tile)
  int tapoff[G::NTW];
#pragma unroll
  for (int ti = 0; ti < G::NTW; ++ti) {
    int tap = wave + 4 * ti;
    if (tap > G::NTAPS - 1) tap = G::NTAPS - 1;
    const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
    tapoff[ti] = ((kd * G::HH + kh) * G::HW + kw) * G::XROWB;
  }

  // per-lane address parts
  int ylane, xlane;
  if constexpr (sizeof(T) == 2) {
    const int q = i16 >> 2, pp = i16 & 3;
    const int v = 4 * g + q;  // voxel within the line
    ylane = v * G::YROWB + 8 * pp;
    xlane = ((v / TW) * S * G::HW + (v % TW) * S) * G::XROWB + 8 * pp;
  } else {
    ylane = g * G::YROWB + 4 * i16;   // + 4u rows per sub-step
    xlane = g * S * G::XROWB + 4 * i16;
  }

  const char* xb = (const char*)p.x;
  const char* yb = (const char*)p.dy;
  // Register-staged software pipeline: the global loads of tile t+1 are issued before the MFMA
  // phase of tile t and land in LDS after it (global latency hides under compute).
  constexpr int NLY = (G::NV * G::YCPR + 255) / 256, NLX = (G::XROWS * G::XCPR + 255) / 256;
  static_assert(NLX <= 32, "one validity bit per staged X chunk");
  frag_t ry[NLY], rx[NLX];
  // Step-invariant per-lane descriptors: 32-bit byte offset inside a tile and packed tile-local
  // coordinates.  Per tile only a wave-uniform base pointer and uniform limits change, so a load
  // costs ~6 VALU ops of bounds checking instead of a 64-bit voxel-address multiply chain.
  int y_goff[NLY], x_goff[NLX];
  unsigned y_pk[NLY], x_pk[NLX];
#pragma unroll
  for (int k = 0; k < NLY; ++k) {
    const int i = tid + 256 * k;
    const int v = i / G::YCPR, ch = i % G::YCPR;
    const int vz = v / (TW * TH), vy = (v / TW) % TH, vx = v % TW;
    y_goff[k] = ((vz * p.Hy + vy) * p.Wy + vx) * p.ldy * (int)sizeof(T) + ch * 16;
    y_pk[k] = i < G::NV * G::YCPR ? (unsigned)(vz | (vy << 8) | (vx << 16)) : 0xffffffffu;
  }
#pragma unroll
  for (int k = 0; k < NLX; ++k) {
    const int i = tid + 256 * k;
    const int v = i / G::XCPR, ch = i % G::XCPR;
    const int hx = v % G::HW, hy = (v / G::HW) % G::HH, hz = v / (G::HW * G::HH);
    x_goff[k] = ((hz * p.Hx + hy) * p.Wx + hx) * p.ldx * (int)sizeof(T) + ch * 16;
    x_pk[k] = i < G::XROWS * G::XCPR ? (unsigned)(hz | (hy << 8) | (hx << 16)) : 0xffffffffu;
  }
  // input transform: this thread's X chunks all sit at channel offset (tid % XCPR) * 8 of the block
  const bool in_tf = sizeof(T) == 2 && p.in_scale != nullptr;
  const bool in_act = in_tf && p.in_alpha != nullptr;
  float tsc[8], tsh[8];
  float in_alpha = in_act ? *p.in_alpha : 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = ci0 + (tid % G::XCPR) * 8 + e;
    tsc[e] = in_tf ? p.in_scale[ch] : 1.f;
    tsh[e] = in_tf ? p.in_shift[ch] : 0.f;
  }
  unsigned xin = 0u;   // bit k: rx[k] of the pending tile lies inside the volume (zero padding stays 0)
  auto fetch = [&](int tile) {
    int t = tile;
    const int txi = t % p.tx; t /= p.tx;
    const int tyi = t % p.ty; t /= p.ty;
    const int tzi = t % p.tz;
    const int n = t / p.tz;
    const int oz0 = tzi * TD, oy0 = tyi * TH, ox0 = txi * TW;
    const int iz0 = oz0 * S - G::PAD, iy0 = oy0 * S - G::PAD, ix0 = ox0 * S - G::PAD;
    const char* ybase = yb + (((((int64_t)n * p.Dy + oz0) * p.Hy + oy0) * p.Wy + ox0) * p.ldy + co0) *
                                 (int64_t)sizeof(T);
    const char* xbase = xb + (((((int64_t)n * p.Dx + iz0) * p.Hx + iy0) * p.Wx + ix0) * p.ldx + ci0) *
                                 (int64_t)sizeof(T);
    const unsigned lz = p.Dy - oz0, ly = p.Hy - oy0, lx = p.Wy - ox0;
#pragma unroll
    for (int k = 0; k < NLY; ++k) {        // dY tile [NV][16*CTO]
      const unsigned pk = y_pk[k];
      ry[k] = frag_t{0u, 0u, 0u, 0u};
      if ((pk & 255u) < lz && ((pk >> 8) & 255u) < ly && (pk >> 16) < lx && !WGRAD_DBG(p, 1))
        ry[k] = *reinterpret_cast<const frag_t*>(ybase + (unsigned)y_goff[k]);
    }
#pragma unroll
    for (int k = 0; k < NLX; ++k) {        // X halo tile [HD*HH*HW][16*CTI]
      const unsigned pk = x_pk[k];
      rx[k] = frag_t{0u, 0u, 0u, 0u};
      const bool inside = pk != 0xffffffffu && !WGRAD_DBG(p, 1) && (unsigned)((int)(pk & 255u) + iz0) < (unsigned)p.Dx &&
          (unsigned)((int)((pk >> 8) & 255u) + iy0) < (unsigned)p.Hx &&
          (unsigned)((int)(pk >> 16) + ix0) < (unsigned)p.Wx;
      if (inside) rx[k] = *reinterpret_cast<const frag_t*>(xbase + (unsigned)x_goff[k]);
      if (in_tf) xin = inside ? (xin | (1u << k)) : (xin & ~(1u << k));
    }
  };
  // (the X loop exists in three copies behind wave-uniform branches: no transform / affine /
  // affine + PReLU -- no per-element selects on the two runtime flags)
  auto commit_with = [&](auto tf) {
#pragma unroll
    for (int k = 0; k < NLY; ++k) {
      const int i = tid + 256 * k;
      if (i < G::NV * G::YCPR)
        *reinterpret_cast<frag_t*>(ysm + (i / G::YCPR) * G::YROWB + (i % G::YCPR) * 16) = ry[k];
    }
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      const int i = tid + 256 * k;
      if (i < G::XROWS * G::XCPR) {
        frag_t val = rx[k];
        if ((xin >> k) & 1u) val = tf(val);
        *reinterpret_cast<frag_t*>(xsm + (i / G::XCPR) * G::XROWB + (i % G::XCPR) * 16) = val;
      }
    }
  };
  auto commit = [&]() {
    if constexpr (sizeof(T) == 2) {
      if (in_act && in_alpha >= 0.f && in_alpha <= 1.f) {
        commit_with([&](frag_t v) { return bn_prelu01_bf16x8(v, tsc, tsh, in_alpha); });
        return;
      }
      if (in_act) { commit_with([&](frag_t v) { return bn_prelu_bf16x8(v, tsc, tsh, in_alpha, true); }); return; }
      if (in_tf) { commit_with([&](frag_t v) { return bn_prelu_bf16x8(v, tsc, tsh, 0.f, false); }); return; }
    }
    commit_with([](frag_t v) { return v; });
  };
  if ((int)blockIdx.x < p.ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    __syncthreads();   // every wave is done reading the previous tile
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) fetch(tile + gridDim.x);

    if (!WGRAD_DBG(p, 4))
#pragma unroll
    for (int lg = 0; lg < G::NL / G::LPG; ++lg) {
      frag_t af[CTO];
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ct = 0; ct < CTO; ++ct) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4*)(ysm + ylane + (2 * lg) * 16 * G::YROWB + ct * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4*)(ysm + ylane + (2 * lg + 1) * 16 * G::YROWB + ct * 32));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          af[ct] = frag_t{l2[0], l2[1], h2[0], h2[1]};
        }
      } else {
#pragma unroll
        for (int ct = 0; ct < CTO; ++ct) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            af[ct][u] = __float_as_uint(*(lds_f32*)(ysm + ylane + (lg * 16 + 4 * u) * G::YROWB + ct * 64));
        }
      }
#pragma unroll
      for (int ti = 0; ti < G::NTW; ++ti) {
        if (KS == 1 && wave != 0) break;  // k1: a single tap, owned by wave 0
        frag_t bf[CTI];
        if constexpr (sizeof(T) == 2) {
          const int r0 = wg_line_row<S, TH, TW, G::HH, G::HW>(2 * lg) * G::XROWB;
          const int r1 = wg_line_row<S, TH, TW, G::HH, G::HW>(2 * lg + 1) * G::XROWB;
#pragma unroll
          for (int it = 0; it < CTI; ++it) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4*)(xsm + xlane + tapoff[ti] + r0 + it * 32));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4*)(xsm + xlane + tapoff[ti] + r1 + it * 32));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            bf[it] = frag_t{l2[0], l2[1], h2[0], h2[1]};
          }
        } else {
          const int r0 = wg_line_row<S, TH, TW, G::HH, G::HW>(lg) * G::XROWB;
#pragma unroll
          for (int it = 0; it < CTI; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              // voxel 4u+g of the line: TW=16 -> same x-row; TW=8 -> rows u>>1
              const int vo = TW == 16 ? 4 * u * S : ((u >> 1) * S * G::HW + 4 * (u & 1) * S);
              bf[it][u] = __float_as_uint(
                  *(lds_f32*)(xsm + xlane + tapoff[ti] + r0 + vo * G::XROWB + it * 64));
            }
          }
        }
#pragma unroll
        for (int ct = 0; ct < CTO; ++ct)
#pragma unroll
          for (int it = 0; it < CTI; ++it) acc[ti][ct][it] = mma16<T>(af[ct], bf[it], acc[ti][ct][it]);
      }
    }
  }

  // partial slab [block][Cout][Cin][27]
  float* slab = p.partials + (int64_t)blockIdx.x * p.Cout * p.Cin * G::NTAPS;
#pragma unroll
  for (int ti = 0; ti < G::NTW; ++ti) {
    const int tap = wave + 4 * ti;
    if (tap < G::NTAPS) {
#pragma unroll
      for (int ct = 0; ct < CTO; ++ct)
#pragma unroll
        for (int it = 0; it < CTI; ++it)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co = co0 + ct * 16 + 4 * g + e, ci = ci0 + it * 16 + i16;
            slab[((int64_t)co * p.Cin + ci) * G::NTAPS + tap] = acc[ti][ct][it][e];
          }
    }
  }
}

template <typename T, int KS, int S, int CTO, int CTI, int TD, int TH, int TW>
static int launch_wgrad_cfg(WgradParams p, int gx_hint, hipStream_t st) {
  using G = WgradGeom<T, KS, S, CTO, CTI, TD, TH, TW>;
  p.tz = cdiv(p.Dy, TD);
  p.ty = cdiv(p.Hy, TH);
  p.tx = cdiv(p.Wy, TW);
  p.ntiles = p.N * p.tz * p.ty * p.tx;
  // 32-bit byte offsets inside one staged tile
  SEGMI_CHECK_ARG((int64_t)G::HD * p.Hx * p.Wx * p.ldx * (int64_t)sizeof(T) < (1ll << 31) &&
                      (int64_t)TD * p.Hy * p.Wy * p.ldy * (int64_t)sizeof(T) < (1ll << 31),
                  "conv3d_wgrad: plane too large for the MFMA kernel's 32-bit tile offsets");
  p.ci_chunks = p.Cin / (16 * CTI);
  static const int dbg = getenv("SEGMI_WGRAD_DBG") ? atoi(getenv("SEGMI_WGRAD_DBG")) : 0;
  p.dbg = dbg;
  const int co_chunks = p.Cout / (16 * CTO);
  dim3 grid((unsigned)gx_hint, (unsigned)(co_chunks * p.ci_chunks));
  auto kern = wgrad_mfma_kernel<T, KS, S, CTO, CTI, TD, TH, TW>;
  static bool attr_done = false;
  if (!attr_done && G::LDS_BYTES > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, 256, G::LDS_BYTES, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_wgrad(mfma)");
  return SEGMI_OK;
}

// how many voxel-range workgroups (= partial slabs) the MFMA path uses
static inline int wgrad_tiles(const segmi_act* dy, int stride, bool one_block) {
  const bool wide = dy->w > 8;
  int td, th, tw;
  if (stride == 1) { td = wide && !one_block ? 2 : 4; th = 8; tw = wide ? 16 : 8; }
  else { td = 2; th = wide ? 4 : 8; tw = wide ? 16 : 8; }
  return dy->n * cdiv(dy->d, td) * cdiv(dy->h, th) * cdiv(dy->w, tw);
}
// channel blocking of one workgroup, encoded 10*CTO + CTI (16-channel tiles of dY x of X).
// bf16 only: 2x2 when both channel counts allow it; with 16 input channels (the wide layers,
// where X is the big tensor) 2 output tiles share one staging of X (4x1 measured slower: 360
// VGPRs).
static inline int wgrad_ct(int dtype, int cin, int cout) {
  if (dtype != SEGMI_BF16) return 11;
  if (cin % 32 == 0 && cout % 32 == 0) return 22;
  if (cin % 32 == 0) return 11;
  if (cout % 32 == 0) return 21;
  return 11;
}
// ---- wave-specialised kernel (wgrad_ws_impl.h): bf16 k3 layers whose two tile buffers fit one CU
// and that have enough tiles per workgroup for its pipeline to matter.  Encodes the choice made by
// the launcher (wgrad_mfma_bf16) so that the workspace query sizes the same number of slabs.
static inline bool wgrad_ws_cfg_ok(int stride, int ct) {
  // 16-channel X (the wide, memory-bound layers).  2x2 channel tiles: stride 1 measured 2x slower
  // than wgrad_mfma_kernel on 32x32 @ 32^3 (latency-bound, 16 tiles per workgroup); stride 2 does
  // not fit (2 x 103 KB of LDS)
  (void)stride;
  return ct == 11 || ct == 21;
}
// CUs the weight-gradient kernels size their grids for (the `cus` argument of segmi_conv3d_wgrad and of its
// workspace query; SEGMI_WGRAD_CUS overrides).  These
// kernels hold a CU exclusively (one 768-thread or 400-register workgroup per CU), so with one workgroup on
// EVERY CU the dependent chain of the main stream gets no CU until they retire; sized for half the chip the two
// streams really run side by side (5.55 -> 5.28 ms per training step, round 3).
int wgrad_cus(int cus);
static inline int wgrad_ws_gx_ct(int dtype, const segmi_act* x, const segmi_act* dy, int ksize, int stride, int cus, int ct) {
  static const bool enabled = !(getenv("SEGMI_WGRAD_WS") && atoi(getenv("SEGMI_WGRAD_WS")) == 0);
  if (!enabled || dtype != SEGMI_BF16 || ksize != 3) return 0;
  if (!wgrad_ws_cfg_ok(stride, ct)) return 0;
  const int cto = ct / 10, cti = ct % 10;
  const int chunks = (x->c / (16 * cti)) * (dy->c / (16 * cto));
  int gx = wgrad_cus(cus) / chunks / 8 * 8;        // one 768-thread workgroup per CU, a multiple of the 8 XCDs
  if (gx < 8) return 0;
  // the tile shapes of launch_wgrad_ws (must match)
  const bool wide = dy->w > 8;
  int td, th, tw;
  if (stride == 1) { td = ct == 11 ? 4 : (wide ? 2 : 4); th = 8; tw = wide ? 16 : 8; }
  else { td = 2; th = wide ? 4 : 8; tw = wide ? 16 : 8; }
  const int64_t nt = (int64_t)dy->n * cdiv(dy->d, td) * cdiv(dy->h, th) * cdiv(dy->w, tw);
  // raw buffer loads: 32-bit offsets / num_records over each tensor
  if (act_voxels(x) * x->ld * 2 >= 0xfff00000ll || act_voxels(dy) * dy->ld * 2 >= 0xfff00000ll) return 0;
  static const int min_tiles = getenv("SEGMI_WGRAD_WS_MINT") ? atoi(getenv("SEGMI_WGRAD_WS_MINT")) : 4;   // A/B
  return nt >= min_tiles * (int64_t)gx ? gx : 0;
}
// Channel tile of THIS layer.  Layers with >= 32 input and output channels took the 2 x 2 tile of wgrad_mfma_kernel
// (220 VGPRs, one workgroup per CU) until round 4; with the 2 x 1 tile (32 output x 16 input channels per workgroup:
// the wave-specialised kernel where the layer has the tiles for it, wgrad_mfma_kernel's 2 x 1 form elsewhere) the
// training step went 4.98 -> 4.75 ms and the 160^3 / 32-label step at batch 4 11.1 -> 9.9 ms, alternating runs
// (the operand read twice costs less than the 2 x 2 tile's occupancy).  SEGMI_WGRAD_CT22 (A/B): 22 = the 2 x 2 tile
// as before (4.98), 2122 = 2 x 1 only where the wave-specialised kernel then takes the layer (4.79), 11 / 1122
// likewise with 1 x 1 (4.82 / 4.82).
static inline int wgrad_ct_for(int dtype, const segmi_act* x, const segmi_act* dy, int ksize, int stride, int cus) {
  const int ct = wgrad_ct(dtype, x->c, dy->c);
  if (ct != 22) return ct;
  static const int mode = getenv("SEGMI_WGRAD_CT22") ? atoi(getenv("SEGMI_WGRAD_CT22")) : 21;
  if (mode == 21 || mode == 11) return mode;
  if (mode == 2122 || mode == 1122) {
    const int alt = mode / 100;
    return wgrad_ws_gx_ct(dtype, x, dy, ksize, stride, cus, alt) > 0 ? alt : 22;
  }
  return ct;
}
static inline int wgrad_ws_gx(int dtype, const segmi_act* x, const segmi_act* dy, int ksize, int stride, int cus) {
  return wgrad_ws_gx_ct(dtype, x, dy, ksize, stride, cus, wgrad_ct_for(dtype, x, dy, ksize, stride, cus));
}
static inline int wgrad_gx(int dtype, const segmi_act* x, const segmi_act* dy, int ksize, int stride, int cus_arg) {
  const int ws = wgrad_ws_gx(dtype, x, dy, ksize, stride, cus_arg);
  if (ws > 0) return ws;
  const int ct = wgrad_ct_for(dtype, x, dy, ksize, stride, cus_arg);
  const int cto = ct / 10, cti = ct % 10;
  const int chunks = (x->c / (16 * cti)) * (dy->c / (16 * cto));
  // workgroups wanted = a multiple of the 256 CUs; one per CU once the kernel holds > 1 channel
  // tile (200-400 VGPRs, 55-110 KB slabs).  Measured on MI355X: 1x1 blocks
  // 704/508/540/608 us at 256/512/1024/2048 workgroups, 2x2 blocks 125/198/348 us at 256/512/1024.
  static const int mfma_env = getenv("SEGMI_WGRAD_CUS_MFMA") ? atoi(getenv("SEGMI_WGRAD_CUS_MFMA")) / 8 * 8 : 0;   // experiments
  const int cus = mfma_env >= 8 ? mfma_env : wgrad_cus(cus_arg);
  static const int mul21 = getenv("SEGMI_WGRAD_MUL21") ? atoi(getenv("SEGMI_WGRAD_MUL21")) : 1;             // A/B
  const int target = cto * cti == 1 ? 2 * cus : (cto * cti == 2 ? mul21 * cus : cus);
  int gx = target / chunks;
  if (gx < 1) gx = 1;
  const int nt = wgrad_tiles(dy, stride, cto * cti == 1);
  return gx < nt ? gx : nt;
}

#define WG_CFG(KS, S, CTO, CTI)                                                         \
  do {                                                                                   \
    if (S == 1 || KS == 1)                                                               \
      return wide ? launch_wgrad_cfg<T, KS, 1, CTO, CTI, 2, 8, 16>(p, gx, st)            \
                  : launch_wgrad_cfg<T, KS, 1, CTO, CTI, 4, 8, 8>(p, gx, st);            \
    return wide ? launch_wgrad_cfg<T, KS, S, CTO, CTI, 2, 4, 16>(p, gx, st)              \
                : launch_wgrad_cfg<T, KS, S, CTO, CTI, 2, 8, 8>(p, gx, st);              \
  } while (0)

template <typename T>
static int launch_wgrad_mfma_t(const WgradParams& p, int ksize, int stride, int ct, int gx,
                               hipStream_t st) {
  const bool wide = p.Wy > 8;
  if (ksize == 1) {
    if constexpr (sizeof(T) == 2) {
      if (ct == 22) WG_CFG(1, 1, 2, 2);
      if (ct == 21) WG_CFG(1, 1, 2, 1);
    }
    WG_CFG(1, 1, 1, 1);
  }
  if constexpr (sizeof(T) == 2) {
    if (stride == 1) {
      if (ct == 22) WG_CFG(3, 1, 2, 2);
      if (ct == 21) WG_CFG(3, 1, 2, 1);
    } else {
      if (ct == 22) WG_CFG(3, 2, 2, 2);
      if (ct == 21) WG_CFG(3, 2, 2, 1);
    }
  }
  // 16 x 16 channel block (144 VGPRs, 50 KB of LDS): 4 output planes per tile halve the per-tile
  // overhead (barriers, tile decode, LDS commit) and take the X halo from 2.8x to 2.1x
  if (stride == 1)
    return wide ? launch_wgrad_cfg<T, 3, 1, 1, 1, 4, 8, 16>(p, gx, st)
                : launch_wgrad_cfg<T, 3, 1, 1, 1, 4, 8, 8>(p, gx, st);
  WG_CFG(3, 2, 1, 1);
}
#undef WG_CFG

}  // namespace segmi
