// conv_small.hip -- k3 convolutions with a tiny input-channel count (the network's first layer:
// Cin = num_channels = 1..4, Cout = 16 or 32).  K = 27*Cin is too short for the MFMA path and
// the layer is HBM-bound anyway (reads 1 voxel, writes 16 channels), so these are VALU kernels:
//   forward : one thread per output voxel, all COUT channels in registers, weights through the
//             scalar cache, 16-byte coalesced NDHWC stores.
//   wgrad   : MFMA with the voxel axis as K and (ci, tap) as N (see conv_small_wgrad_kernel);
//             persistent workgroups, per-workgroup partial slabs (deterministic).
#include "common.h"

namespace segmi {

struct SmallConvParams {
  const void* in;
  void* out;
  const float* w;     // torch layout [COUT][Cin][27]
  const float* bias;
  const float* alpha;  // PReLU slope (nullable)
  const void* res;     // residual view (nullable), added after the activation
  int N, Di, Hi, Wi, Do, Ho, Wo, Cin, ldi, ldo, ldr, stride;
};

template <typename T, int COUT>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(SmallConvParams p) {
  extern __shared__ float wsm[];  // [27*Cin][COUT] (tap-major so a tap's COUT weights are contiguous)
  const int nk = 27 * p.Cin;
  for (int i = threadIdx.x; i < nk * COUT; i += 256) {
    const int co = i % COUT, k = i / COUT;         // k = ci*27 + tap
    wsm[i] = p.w[(int64_t)co * nk + k];
  }
  __syncthreads();
  const int64_t total = (int64_t)p.N * p.Do * p.Ho * p.Wo;
  const T* in = (const T*)p.in;
  T* out = (T*)p.out;
  for (int64_t v = blockIdx.x * 256ll + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    int64_t t = v;
    const int ox = t % p.Wo; t /= p.Wo;
    const int oy = t % p.Ho; t /= p.Ho;
    const int oz = t % p.Do;
    const int n = t / p.Do;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = p.bias ? p.bias[c] : 0.f;
    for (int kd = 0; kd < 3; ++kd) {
      const int z = oz * p.stride - 1 + kd;
      if ((unsigned)z >= (unsigned)p.Di) continue;
      for (int kh = 0; kh < 3; ++kh) {
        const int y = oy * p.stride - 1 + kh;
        if ((unsigned)y >= (unsigned)p.Hi) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int x = ox * p.stride - 1 + kw;
          if ((unsigned)x >= (unsigned)p.Wi) continue;
          const T* ip = in + ((((int64_t)n * p.Di + z) * p.Hi + y) * p.Wi + x) * p.ldi;
          const int tap = (kd * 3 + kh) * 3 + kw;
          for (int ci = 0; ci < p.Cin; ++ci) {
            const float a = Elem<T>::ld(ip + ci);
            const float* wr = wsm + (ci * 27 + tap) * COUT;
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = fmaf(a, wr[c], acc[c]);
          }
        }
      }
    }
    if (p.alpha) {
      const float al = *p.alpha;
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[c] = acc[c] > 0.f ? acc[c] : al * acc[c];
    }
    T* op = out + v * p.ldo;
    const T* rp = p.res ? (const T*)p.res + v * p.ldr : nullptr;
#pragma unroll
    for (int c = 0; c < COUT; c += 4) {
      f32x4 o = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
      if (rp) o += load4<T>(rp + c);
      store4<T>(op + c, o);
    }
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

int conv_small_fwd(int dtype, const segmi_act* in, const segmi_act* out, const float* w,
                   const float* bias, const float* alpha, const segmi_act* res, int stride,
                   hipStream_t st) {
  SmallConvParams p{};
  p.in = in->data; p.out = out->data; p.w = w; p.bias = bias; p.alpha = alpha;
  p.res = res ? res->data : nullptr; p.ldr = res ? res->ld : 0;
  p.N = in->n; p.Di = in->d; p.Hi = in->h; p.Wi = in->w; p.Do = out->d; p.Ho = out->h; p.Wo = out->w;
  p.Cin = in->c; p.ldi = in->ld; p.ldo = out->ld; p.stride = stride;
  const int64_t total = act_voxels(out);
  const int grid = (int)(cdiv64(total, 256) > 16384 ? 16384 : cdiv64(total, 256));
  const size_t lds = (size_t)27 * in->c * out->c * sizeof(float);
#define L(TT, CO) hipLaunchKernelGGL((conv_small_fwd_kernel<TT, CO>), grid, 256, lds, st, p)
  if (dtype == SEGMI_F32) { if (out->c == 16) L(float, 16); else L(float, 32); }
  else { if (out->c == 16) L(bf16_t, 16); else L(bf16_t, 32); }
#undef L
  SEGMI_LAUNCH_CHECK("conv3d_fwd(small-cin)");
  return SEGMI_OK;
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
