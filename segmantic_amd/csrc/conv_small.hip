// conv_small.hip -- k3 convolutions with a tiny input-channel count (the network's first layer:
// Cin = num_channels = 1..4, Cout = 16 or 32).  K = 27*Cin is too short for the MFMA path and
// the layer is HBM-bound (reads 1 voxel, writes 16 channels):
//   forward : MFMA with k = (ci, tap) gathered element-wise from the staged halo tile (K = 27 is
//             ONE 16x16x32 bf16 MFMA per 16 voxels x 16 channels), fused bias / statistics / PReLU
//             / residual epilogue.
//   wgrad   : MFMA with the voxel axis as K and (ci, tap) as N (see conv_small_wgrad_kernel);
//             persistent workgroups, per-workgroup partial slabs (deterministic).
#include "common.h"
#include "fin_tail.h"

namespace segmi {

struct SmallConvParams {
  const void* in;
  void* out;
  const float* w;     // torch layout [Cout][Cin][27]
  const float* bias;
  const float* alpha;  // PReLU slope (nullable)
  const void* res;     // residual view (nullable), added after the activation
  float* stats;        // per-workgroup partial sums [grid.x][2][Cout] (nullable)
  int N, Di, Hi, Wi, Do, Ho, Wo, Cout, ldi, ldo, ldr;
  int tz, ty, tx;
  // PAIR: a second convolution of the same input and geometry (the residual convolution of the
  // unit): its own weights / bias / destination, no activation, no statistics
  void* out2;
  const float* w2;
  const float* bias2;
  int ldo2;
  int vec;   // Cin == 1 rows can be staged with 4-element loads (dense, 4-aligned W and base)
  // window views (segmi_windows): sample n of the input is the (Di, Hi, Wi) block that starts woff[n]
  // elements into a larger single-channel volume with row / plane strides wsy / wsz -- the sliding-window
  // driver's windows, read in place instead of being gathered into a batch first.  nwin == 0: dense batch.
  int nwin;
  int wsy;
  int64_t wsz;
  int64_t woff[32];
  // BatchNorm statistics finalised by the last workgroup of this launch (fin_tail.h)
  int fin_on;
  FinTail ft;
  BnFin bfin;
};

// MFMA forward for tiny Cin:  D[co][vox] += W[co][k] * X[k][vox] with k = ci*27 + tap (the torch
// weight row, zero-padded to a multiple of the MFMA K).  The weight operand is built once per wave
// from the f32 weights (8 consecutive k per lane); the voxel operand is gathered from the staged
// halo tile: 8 (bf16) / 4 (f32) single-element LDS reads per lane and k-step -- with K = 27 the
// whole convolution of 16 voxels x 16 channels is ONE v_mfma_f32_16x16x32_bf16 for Cin = 1.
// Tile = 4 x 8 x 16 output voxels (32 voxel tiles, 8 per wave); epilogue as conv_fwd_impl.h.
// compile-time halo offset of k = ci*27 + tap (0 for the zero padding beyond 27*CIN)
template <int CIN, int HD, int HH, int XW>
__host__ __device__ constexpr int small_koff(int k) {
  if (k >= 27 * CIN) return 0;
  const int ci = k / 27, tap = k % 27;
  return ((ci * HD + tap / 9) * HH + (tap / 3) % 3) * XW + tap % 3;
}

template <typename T, int S, int CIN, bool PAIR>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(SmallConvParams p) {
  constexpr int ES = (int)sizeof(T);
  constexpr int KG = Elem<T>::KG;                      // k-values per lane per MFMA operand
  constexpr int KSTEP = 4 * KG;                        // 32 (bf16) / 16 (f32)
  constexpr int NK = (27 * CIN + KSTEP - 1) / KSTEP;   // k-steps
  constexpr int TD = 4, TH = 8, TW = 16;
  constexpr int HD = (TD - 1) * S + 3, HH = (TH - 1) * S + 3, HW = (TW - 1) * S + 3;
  constexpr int XW = (HW + 3 + 3) / 4 * 4;             // LDS row: 3 lead-in elements + HW, 4-padded
  constexpr int NROWS = CIN * HD * HH;                 // halo rows of HW elements
  constexpr int RPI = 256 / HW < HH ? 256 / HW : HH;   // rows staged per iteration (<= one row wrap)
  constexpr int NIT = (NROWS + RPI - 1) / RPI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xs = reinterpret_cast<T*>(smem);                  // [CIN][HD][HH][XW], halo x = 0 at column 3
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  int t = blockIdx.x;
  const int txi = t % p.tx; t /= p.tx;
  const int tyi = t % p.ty; t /= p.ty;
  const int tzi = t % p.tz;
  const int n = t / p.tz;
  const int oz0 = tzi * TD, oy0 = tyi * TH, ox0 = txi * TW;
  const int iz0 = oz0 * S - 1, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int co0 = blockIdx.y * 16;

  // ---- halo tile -> LDS.  The index arithmetic of this layer used to cost more than everything
  // else (212 quarter-rate integer multiplies per wave): a thread now owns one x position and walks
  // the halo rows with running (ci, hz, hy) counters and a running 32-bit element offset from the
  // wave-uniform tile origin; all loads are issued before the first LDS store.
  const bool win = p.nwin > 0;                         // (windows: single channel, ldi = 1)
  const T* tile = win ? (const T*)p.in + p.woff[n] + (int64_t)iz0 * p.wsz + (int64_t)iy0 * p.wsy + ix0
                      : (const T*)p.in + ((((int64_t)n * p.Di + iz0) * p.Hi + iy0) * p.Wi + ix0) * p.ldi;
  const int hx = tid % HW;
  const bool xlive = tid < RPI * HW && (unsigned)(ix0 + hx) < (unsigned)p.Wi;
  T stg[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) stg[k] = (T)0;
  // single-channel dense rows: the halo row [ix0 - 3, ix0 - 3 + XW) starts on a multiple of 4
  // elements (ix0 = TW*S*tx - 1, TW*S % 4 == 0), so it is XW/4 aligned 4-element loads -- 6 loads per
  // thread for the stride-2 first layer instead of 22 two-byte ones
  typedef typename std::conditional<ES == 2, u32x2, f32x4>::type Vec4;
  constexpr int GPR = XW / 4, NGRP = NROWS * GPR, NITV = (NGRP + 255) / 256;
  Vec4 vstg[CIN == 1 ? NITV : 1];
  unsigned vok = 0u;                                   // bit k: vstg[k] lies inside the volume
  const bool vec = CIN == 1 && p.vec;
  if (vec) {
    const int64_t sy = win ? p.wsy : p.Wi, sz = win ? p.wsz : (int64_t)p.Hi * p.Wi;
#pragma unroll
    for (int k = 0; k < NITV; ++k) {
      const int gi = tid + 256 * k;
      const int row = gi / GPR, gq = gi % GPR;
      const int hy = row % HH, hz = row / HH;
      const int x0 = ix0 - 3 + 4 * gq;
      // No branch around the load: a conditional load compiles to an exec-mask region with its own
      // `s_waitcnt vmcnt(0)`, i.e. the six loads of a thread went out ONE AFTER THE OTHER (six global-memory
      // latencies in front of every tile's 16 MFMAs; round 4).  Lanes outside the volume read the tensor's first
      // element group instead (always there, aligned) and drop it.
      const bool ok = (gi < NGRP) & ((unsigned)(iz0 + hz) < (unsigned)p.Di) & ((unsigned)(iy0 + hy) < (unsigned)p.Hi) &
                      ((unsigned)x0 < (unsigned)p.Wi);
      const T* src = ok ? tile + hz * sz + hy * sy + (4 * gq - 3) : (const T*)p.in;
      vstg[k] = *reinterpret_cast<const Vec4*>(src);      // (zeroed where !ok when it goes to LDS: no use of the value here)
      vok |= ok ? (1u << k) : 0u;
    }
  } else {
    static_assert(RPI <= HH, "at most one row wrap per iteration");
    int row = tid / HW;                                // row = (ci * HD + hz) * HH + hy
    int hy = row % HH, hz = (row / HH) % HD;
    const int sy = p.Wi * p.ldi, sz = p.Hi * sy;        // element strides of one halo row / plane
    int off = hz * sz + hy * sy + hx * p.ldi + row / (HH * HD);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      {
        const bool ok = xlive & (row < NROWS) & ((unsigned)(iz0 + hz) < (unsigned)p.Di) & ((unsigned)(iy0 + hy) < (unsigned)p.Hi);
        const T* src = ok ? tile + off : (const T*)p.in;          // (no branch around the load, as above)
        const T got = *src;
        stg[k] = ok ? got : (T)0;
      }
      row += RPI;
      hy += RPI;
      off += RPI * sy;
      if (hy >= HH) {
        hy -= HH;
        off += sz - HH * sy;
        if (++hz == HD) { hz = 0; off += 1 - HD * sz; }   // next input channel
      }
    }
  }
  // ---- weight operand: lane (co = r, g) holds k = KSTEP*s + KG*g + j; the halo offset of that k
  // is a compile-time constant per lane group, plus the lane's voxel (x = r, plane z = wave)
  frag_t wf[NK], wf2[PAIR ? NK : 1];
  int lk[NK][KG];
  const int lane_vox = (wave * S * HH) * XW + 3 + r * S;
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    float wv[KG];
#pragma unroll
    for (int j = 0; j < KG; ++j) {
      const int k = KSTEP * s + KG * g + j;
      wv[j] = k < 27 * CIN ? p.w[(co0 + r) * 27 * CIN + k] : 0.f;
      const int o0 = small_koff<CIN, HD, HH, XW>(KSTEP * s + KG * 0 + j);
      const int o1 = small_koff<CIN, HD, HH, XW>(KSTEP * s + KG * 1 + j);
      const int o2 = small_koff<CIN, HD, HH, XW>(KSTEP * s + KG * 2 + j);
      const int o3 = small_koff<CIN, HD, HH, XW>(KSTEP * s + KG * 3 + j);
      lk[s][j] = lane_vox + (g == 0 ? o0 : g == 1 ? o1 : g == 2 ? o2 : o3);
    }
    if constexpr (ES == 2) {
      wf[s] = frag_t{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3]), pack_bf16x2(wv[4], wv[5]),
                     pack_bf16x2(wv[6], wv[7])};
    } else {
      wf[s] = frag_t{__float_as_uint(wv[0]), __float_as_uint(wv[1]), __float_as_uint(wv[2]),
                     __float_as_uint(wv[3])};
    }
    if constexpr (PAIR) {
#pragma unroll
      for (int j = 0; j < KG; ++j) {
        const int k = KSTEP * s + KG * g + j;
        wv[j] = k < 27 * CIN ? p.w2[(co0 + r) * 27 * CIN + k] : 0.f;
      }
      if constexpr (ES == 2) {
        wf2[s] = frag_t{pack_bf16x2(wv[0], wv[1]), pack_bf16x2(wv[2], wv[3]), pack_bf16x2(wv[4], wv[5]),
                        pack_bf16x2(wv[6], wv[7])};
      } else {
        wf2[s] = frag_t{__float_as_uint(wv[0]), __float_as_uint(wv[1]), __float_as_uint(wv[2]),
                        __float_as_uint(wv[3])};
      }
    }
  }
  f32x4 bias4b = f32x4{0.f, 0.f, 0.f, 0.f};
  if (PAIR && p.bias2) bias4b = *reinterpret_cast<const f32x4*>(p.bias2 + co0 + 4 * g);
  touch_v(bias4b);
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co0 + 4 * g);
  const bool has_alpha = p.alpha != nullptr;
  float alpha = has_alpha ? *p.alpha : 0.f;
  touch_v(bias4);
  touch_s(alpha);
  // ---- the staged halo -> LDS, only now: the weight / bias loads above are in flight beside the halo loads instead
  // of behind their wait
  if (vec) {
#pragma unroll
    for (int k = 0; k < NITV; ++k) {
      const int gi = tid + 256 * k;
      if (gi < NGRP) *reinterpret_cast<Vec4*>(xs + 4 * gi) = ((vok >> k) & 1u) ? vstg[k] : Vec4{};
    }
  } else if (tid < RPI * HW) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int row = tid / HW + RPI * k;
      if (row < NROWS) xs[row * XW + 3 + hx] = stg[k];
    }
  }
  __syncthreads();

  // ---- wave w computes plane z = w of the tile; voxel tile i = row y = i (x = r).  The 8 rows are done in
  // two halves (MFMAs, then the epilogue of those 4 rows): with all 8 (x 2 convolutions) accumulated before
  // the epilogue the kernel held 204 registers -- two workgroups per CU for a launch that is a chain of
  // load -> barrier -> 16 MFMAs -> store latencies (95 us for the 224 MB of the training step's first layer).
  constexpr int HB = 4;
  // epilogue: bias, BN statistics, PReLU, residual, 8/16-byte NDHWC stores; wave-uniform row pointers + one
  // per-lane 32-bit offset
  const int oz = oz0 + wave, ox = ox0 + r;
  const bool zx_ok = oz < p.Do && ox < p.Wo;
  T* orow = (T*)p.out + ((((int64_t)n * p.Do + oz) * p.Ho + oy0) * p.Wo) * p.ldo;
  const T* rrow = p.res ? (const T*)p.res + ((((int64_t)n * p.Do + oz) * p.Ho + oy0) * p.Wo) * p.ldr
                        : nullptr;
  const unsigned lo_out = (unsigned)(ox * p.ldo + co0 + 4 * g), lo_res = (unsigned)(ox * p.ldr + co0 + 4 * g);
  const int ostep = p.Wo * p.ldo, rstep = p.Wo * p.ldr;
  T* orow2 = nullptr;
  unsigned lo_out2 = 0u;
  int ostep2 = 0;
  if constexpr (PAIR) {
    orow2 = (T*)p.out2 + ((((int64_t)n * p.Do + oz) * p.Ho + oy0) * p.Wo) * p.ldo2;
    lo_out2 = (unsigned)(ox * p.ldo2 + co0 + 4 * g);
    ostep2 = p.Wo * p.ldo2;
  }
#pragma unroll
  for (int k = 0; k < NIT; ++k) touch_v(stg[k]);
  if constexpr (CIN == 1) {
#pragma unroll
    for (int k = 0; k < NITV; ++k) touch_v(vstg[k]);
  }
  f32x4 ssum = f32x4{0.f, 0.f, 0.f, 0.f}, ssq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int hb = 0; hb < 8; hb += HB) {
    f32x4 resv[HB];
    if (rrow) {
#pragma unroll
      for (int i = 0; i < HB; ++i)
        resv[i] = (zx_ok && oy0 + hb + i < p.Ho) ? load4<T>(rrow + (hb + i) * rstep + lo_res)
                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 acc[HB], acc2[PAIR ? HB : 1];
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (PAIR) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NK; ++s) {
        frag_t a;
        if constexpr (ES == 2) {
          unsigned short e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = xs[lk[s][j] + (hb + i) * S * XW];
          a = frag_t{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                     (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] = __float_as_uint(xs[lk[s][j] + (hb + i) * S * XW]);
        }
        acc[i] = mma16<T>(wf[s], a, acc[i]);
        if constexpr (PAIR) acc2[i] = mma16<T>(wf2[s], a, acc2[i]);
      }
    }
    if (rrow) {
#pragma unroll
      for (int i = 0; i < HB; ++i) touch_v(resv[i]);
    }
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      f32x4 v = acc[i] + bias4;
      if (zx_ok && oy0 + hb + i < p.Ho) {
        if (p.stats) { ssum += v; ssq += v * v; }
        if (has_alpha) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
        }
        if (rrow) v += resv[i];
        store4<T>(orow + (hb + i) * ostep + lo_out, v);
      }
    }
    if constexpr (PAIR) {
#pragma unroll
      for (int i = 0; i < HB; ++i)
        if (zx_ok && oy0 + hb + i < p.Ho) store4<T>(orow2 + (hb + i) * ostep2 + lo_out2, acc2[i] + bias4b);
    }
  }
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][2][16]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a0 = row16_sum(ssum[e]);
      const float b0 = row16_sum(ssq[e]);
      if (r == 0) {
        red[(wave * 2 + 0) * 16 + 4 * g + e] = a0;
        red[(wave * 2 + 1) * 16 + 4 * g + e] = b0;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid / 16, ch = tid % 16;
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sacc += red[(w * 2 + which) * 16 + ch];
      fin_store(&p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + co0 + ch], sacc);
    }
    fin_tail_run<BnFin, 256, offsetof(SmallConvParams, ft), offsetof(SmallConvParams, bfin)>(p.stats, smem);
  }
}

struct SmallWgradParams {
  const void* x;
  const void* dy;
  float* partials;   // [grid.x][Cout][Cin][27]
  int N, Dx, Hx, Wx, Dy, Hy, Wy, Cout, ldx, ldy;
  int tz, ty, tx, ntiles;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// MFMA weight gradient for tiny Cin:  dW[co][ci][tap] = sum_v dY[v][co] * X[v*S + tap - 1][ci]
// as D[co][combo] += A[co][k = voxel] * B[k = voxel][combo = ci*27 + tap], 16 output channels per
// blockIdx.y, ceil(27*Cin/16) combo tiles.  The voxel axis is K:
//   A: transposing LDS reads of the staged dY tile (as in wgrad_impl.h; bf16 element j of a lane
//      is voxel 4g + (j&3) of line 2s + (j>>2), f32 element u is voxel 4u + g);
//   B: the X halo is staged three times, once per kw, de-strided: P[kw][row][vx] =
//      halo[row][S*vx + kw], so the 4 voxels a lane needs are 4 consecutive elements.
// Tile = 2 x 8 x 16 dY voxels = 16 lines of 16; the 4 waves split the lines, keep their partial
// D in registers over all tiles of the (persistent) workgroup and are summed in fixed order.
template <typename T, int S, int CIN>
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(SmallWgradParams p) {
  constexpr int ES = (int)sizeof(T);
  constexpr int TD = 2, TH = 8, TW = 16, NV = TD * TH * TW, NL = NV / 16;
  constexpr int HD = (TD - 1) * S + 3, HH = (TH - 1) * S + 3, HWX = (TW - 1) * S + 3;
  constexpr int ROWS = HD * HH;
  constexpr int YROWB = 16 * ES;
  constexpr int COMBOS = 27 * CIN, NTL = (COMBOS + 15) / 16;
  constexpr int NLY = NV * YROWB / 16 / 256;             // 16-byte dY loads per thread
  constexpr int NXE = CIN * ROWS * HWX, NLX = (NXE + 255) / 256;
  static_assert(NV * YROWB / 16 % 256 == 0, "dY tile staging");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ysm = smem;                                        // [NV][16] T
  T* xsm = reinterpret_cast<T*>(smem + NV * YROWB);         // [CIN][3][ROWS][16] T
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i16 = lane & 15;
  const int co0 = blockIdx.y * 16;

  // per-lane B addressing: combo tile nt -> (ci, kd, kh, kw) -> element offset of P[kw] row 0
  int boff[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    int cb = nt * 16 + i16;
    if (cb > COMBOS - 1) cb = COMBOS - 1;   // padding lanes read something valid; never stored
    const int ci = cb / 27, tap = cb % 27;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    boff[nt] = ((ci * 3 + kw) * ROWS + kd * HH + kh) * 16;
  }
  int ylane;
  if constexpr (ES == 2) ylane = (4 * g + (i16 >> 2)) * YROWB + 8 * (i16 & 3);
  else ylane = g * YROWB + 4 * i16;

  f32x4 acc[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* x = (const T*)p.x;
  const char* yb = (const char*)p.dy;
  frag_t ry[NLY];
  T rx[NLX];
  auto fetch = [&](int tile) {
    int t = tile;
    const int txi = t % p.tx; t /= p.tx;
    const int tyi = t % p.ty; t /= p.ty;
    const int tzi = t % p.tz;
    const int n = t / p.tz;
    const int oz0 = tzi * TD, oy0 = tyi * TH, ox0 = txi * TW;
#pragma unroll
    for (int k = 0; k < NLY; ++k) {
      const int i = tid + 256 * k;
      const int v = i / (YROWB / 16), ch = i % (YROWB / 16);
      const int z = oz0 + v / (TW * TH), y = oy0 + (v / TW) % TH, xx = ox0 + v % TW;
      ry[k] = frag_t{0u, 0u, 0u, 0u};
      if (z < p.Dy && y < p.Hy && xx < p.Wy) {
        const int64_t e = ((((int64_t)n * p.Dy + z) * p.Hy + y) * p.Wy + xx) * p.ldy + co0;
        ry[k] = *reinterpret_cast<const frag_t*>(yb + e * ES + ch * 16);
      }
    }
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      const int i = tid + 256 * k;
      const int ci = i % CIN, r = i / CIN;
      const int hx = r % HWX, hy = (r / HWX) % HH, hz = r / (HWX * HH);
      const int z = oz0 * S - 1 + hz, y = oy0 * S - 1 + hy, xx = ox0 * S - 1 + hx;
      rx[k] = (T)0;
      if (i < NXE && (unsigned)z < (unsigned)p.Dx && (unsigned)y < (unsigned)p.Hx &&
          (unsigned)xx < (unsigned)p.Wx)
        rx[k] = x[((((int64_t)n * p.Dx + z) * p.Hx + y) * p.Wx + xx) * p.ldx + ci];
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NLY; ++k) *reinterpret_cast<frag_t*>(ysm + (tid + 256 * k) * 16) = ry[k];
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      const int i = tid + 256 * k;
      if (i >= NXE) continue;
      const int ci = i % CIN, r = i / CIN;
      const int hx = r % HWX, row = r / HWX;
      T* pr = xsm + (ci * 3 * ROWS + row) * 16;
      // P[kw][vx] = halo[S*vx + kw]  <=>  hx = S*vx + kw
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int d = hx - kw;
        if (d >= 0 && d % S == 0 && d / S < 16) pr[kw * ROWS * 16 + d / S] = rx[k];
      }
    }
  };

  if ((int)blockIdx.x < p.ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) fetch(tile + gridDim.x);
    if constexpr (ES == 2) {
#pragma unroll
      for (int si = 0; si < NL / 2 / 4; ++si) {
        const int s = wave + 4 * si;                     // k-step = lines 2s, 2s+1
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(ysm + ylane + (2 * s) * 16 * YROWB));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(ysm + ylane + (2 * s + 1) * 16 * YROWB));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        const frag_t af = frag_t{l2[0], l2[1], h2[0], h2[1]};
        const int l0 = 2 * s, l1 = 2 * s + 1;
        const int r0 = ((l0 / TH) * S * HH + (l0 % TH) * S) * 16 + 4 * g;
        const int r1 = ((l1 / TH) * S * HH + (l1 % TH) * S) * 16 + 4 * g;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
          const u32x2 b0 = *reinterpret_cast<const u32x2*>(xsm + boff[nt] + r0);
          const u32x2 b1 = *reinterpret_cast<const u32x2*>(xsm + boff[nt] + r1);
          acc[nt] = mma16<T>(af, frag_t{b0[0], b0[1], b1[0], b1[1]}, acc[nt]);
        }
      }
    } else {
#pragma unroll
      for (int si = 0; si < NL / 4; ++si) {
        const int l = wave + 4 * si;                     // k-step = line l
        frag_t af;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          af[u] = __float_as_uint(*reinterpret_cast<const float*>(ysm + ylane + (l * 16 + 4 * u) * YROWB));
        const int r0 = ((l / TH) * S * HH + (l % TH) * S) * 16 + g;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
          frag_t bf;
#pragma unroll
          for (int u = 0; u < 4; ++u)
            bf[u] = __float_as_uint(*reinterpret_cast<const float*>(xsm + boff[nt] + r0 + 4 * u));
          acc[nt] = mma16<T>(af, bf, acc[nt]);
        }
      }
    }
  }

  // fixed-order sum over the 4 waves -> slab [16 of Cout][CIN][27] at channel offset co0
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);   // [4][NTL][64][4]
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt)
    *reinterpret_cast<f32x4*>(red + ((wave * NTL + nt) * 64 + lane) * 4) = acc[nt];
  __syncthreads();
  float* slab = p.partials + (int64_t)blockIdx.x * p.Cout * COMBOS;
  for (int o = tid; o < 16 * COMBOS; o += 256) {
    const int co = o / COMBOS, cb = o % COMBOS;
    const int nt = cb / 16, n16 = cb % 16;
    // D[m = co][n = combo]: lane (n16, g = co/4), element co%4
    const int src = (nt * 64 + (co >> 2) * 16 + n16) * 4 + (co & 3);
    float sacc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) sacc += red[w * NTL * 256 + src];
    slab[(int64_t)(co0 + co) * COMBOS + cb] = sacc;
  }
}

bool conv_small_ok(int cin, int cout, int ksize) {
  return ksize == 3 && cin >= 1 && cin <= 4 && (cout == 16 || cout == 32);
}

// partial-statistics rows of the small-Cin forward (= spatial workgroups)
int conv_small_fwd_rows(const segmi_act* out) {
  return out->n * cdiv(out->d, 4) * cdiv(out->h, 8) * cdiv(out->w, 16);
}

template <typename T, int S, int CIN, bool PAIR>
static int launch_small_fwd_k(SmallConvParams p, hipStream_t st) {
  constexpr int HD = 3 * S + 3, HH = 7 * S + 3, HW = 15 * S + 3, XW = (HW + 6) / 4 * 4;
  constexpr int stage = CIN * HD * HH * XW * (int)sizeof(T);
  constexpr int lds0 = stage > 4 * 2 * 16 * 4 ? stage : 4 * 2 * 16 * 4;
  dim3 grid((unsigned)(p.N * p.tz * p.ty * p.tx), (unsigned)(p.Cout / 16));
  p.fin_on = p.fin_on && p.stats;
  const int lds = (int)fin_tail_arm(p, grid, 256, 2 * p.Cout, lds0);
  static int attr_set = 0;
  if (lds > 64 * 1024 && attr_set < lds) {
    (void)hipFuncSetAttribute((const void*)conv_small_fwd_kernel<T, S, CIN, PAIR>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = lds;
  }
  hipLaunchKernelGGL((conv_small_fwd_kernel<T, S, CIN, PAIR>), grid, 256, lds, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_fwd(small-cin)");
  return SEGMI_OK;
}
template <typename T, int S, int CIN>
static int launch_small_fwd(const SmallConvParams& p, hipStream_t st) {
  return p.out2 ? launch_small_fwd_k<T, S, CIN, true>(p, st) : launch_small_fwd_k<T, S, CIN, false>(p, st);
}
template <typename T, int S>
static int launch_small_fwd_cin(const SmallConvParams& p, int cin, hipStream_t st) {
  switch (cin) {
    case 1: return launch_small_fwd<T, S, 1>(p, st);
    case 2: return launch_small_fwd<T, S, 2>(p, st);
    case 3: return launch_small_fwd<T, S, 3>(p, st);
    default: return launch_small_fwd<T, S, 4>(p, st);
  }
}

int conv_small_fwd(int dtype, const segmi_act* in, const segmi_act* out, const float* w,
                   const float* bias, const float* alpha, const segmi_act* res, float* stats,
                   int stride, hipStream_t st, const segmi_act* out2, const float* w2,
                   const float* bias2, const segmi_bn_fin* fin, const segmi_windows* win) {
  SmallConvParams p{};
  if (fin && stats) { p.fin_on = 1; p.bfin = bn_fin_from(fin, out->c); }
  if (win) {
    p.nwin = win->count; p.wsy = win->row_stride; p.wsz = win->plane_stride;
    for (int i = 0; i < win->count; ++i) p.woff[i] = win->offset[i];
  }
  if (out2) { p.out2 = out2->data; p.w2 = w2; p.bias2 = bias2; p.ldo2 = out2->ld; }
  p.vec = in->c == 1 && in->ld == 1 && in->w % 4 == 0 &&
          ((uintptr_t)in->data % (4 * (dtype == SEGMI_F32 ? 4 : 2))) == 0;
  if (win) SEGMI_CHECK_ARG(p.vec, "conv3d: window views need the 4-element staging path");
  p.in = in->data; p.out = out->data; p.w = w; p.bias = bias; p.alpha = alpha;
  p.res = res ? res->data : nullptr; p.ldr = res ? res->ld : 0; p.stats = stats;
  p.N = in->n; p.Di = in->d; p.Hi = in->h; p.Wi = in->w; p.Do = out->d; p.Ho = out->h; p.Wo = out->w;
  p.Cout = out->c; p.ldi = in->ld; p.ldo = out->ld;
  p.tz = cdiv(out->d, 4); p.ty = cdiv(out->h, 8); p.tx = cdiv(out->w, 16);
  SEGMI_CHECK_ARG((int64_t)p.N * p.tz * p.ty * p.tx < (1ll << 31), "conv3d: too many tiles");
  if (dtype == SEGMI_F32)
    return stride == 2 ? launch_small_fwd_cin<float, 2>(p, in->c, st)
                       : launch_small_fwd_cin<float, 1>(p, in->c, st);
  return stride == 2 ? launch_small_fwd_cin<bf16_t, 2>(p, in->c, st)
                     : launch_small_fwd_cin<bf16_t, 1>(p, in->c, st);
}

int conv_small_wgrad_slabs(const segmi_act* dy) {
  const int nt = dy->n * cdiv(dy->d, 2) * cdiv(dy->h, 8) * cdiv(dy->w, 16);
  return nt < 1024 ? nt : 1024;
}

template <typename T, int S, int CIN>
static int launch_small_wgrad(const SmallWgradParams& p, int grid, hipStream_t st) {
  constexpr int ES = (int)sizeof(T);
  constexpr int ROWS = ((2 - 1) * S + 3) * ((8 - 1) * S + 3);
  constexpr int NTL = (27 * CIN + 15) / 16;
  constexpr int stage = 256 * 16 * ES + CIN * 3 * ROWS * 16 * ES;
  constexpr int red = 4 * NTL * 64 * 4 * 4;
  constexpr int lds = stage > red ? stage : red;
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_small_wgrad_kernel<T, S, CIN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_small_wgrad_kernel<T, S, CIN>), dim3(grid, p.Cout / 16), 256, lds, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_wgrad(small-cin)");
  return SEGMI_OK;
}

template <typename T, int S>
static int launch_small_wgrad_cin(const SmallWgradParams& p, int cin, int grid, hipStream_t st) {
  switch (cin) {
    case 1: return launch_small_wgrad<T, S, 1>(p, grid, st);
    case 2: return launch_small_wgrad<T, S, 2>(p, grid, st);
    case 3: return launch_small_wgrad<T, S, 3>(p, grid, st);
    default: return launch_small_wgrad<T, S, 4>(p, grid, st);
  }
}

int conv_small_wgrad(int dtype, const segmi_act* x, const segmi_act* dy, float* partials,
                     int stride, hipStream_t st) {
  SmallWgradParams p{};
  p.x = x->data; p.dy = dy->data; p.partials = partials;
  p.N = x->n; p.Dx = x->d; p.Hx = x->h; p.Wx = x->w; p.Dy = dy->d; p.Hy = dy->h; p.Wy = dy->w;
  p.Cout = dy->c; p.ldx = x->ld; p.ldy = dy->ld;
  p.tz = cdiv(dy->d, 2); p.ty = cdiv(dy->h, 8); p.tx = cdiv(dy->w, 16);
  p.ntiles = dy->n * p.tz * p.ty * p.tx;
  const int grid = conv_small_wgrad_slabs(dy);
  if (dtype == SEGMI_F32)
    return stride == 2 ? launch_small_wgrad_cin<float, 2>(p, x->c, grid, st)
                       : launch_small_wgrad_cin<float, 1>(p, x->c, grid, st);
  return stride == 2 ? launch_small_wgrad_cin<bf16_t, 2>(p, x->c, grid, st)
                     : launch_small_wgrad_cin<bf16_t, 1>(p, x->c, grid, st);
}

}  // namespace segmi
