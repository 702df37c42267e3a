// conv_small.hip -- k3 convolutions with a tiny input-channel count (the network's first layer:
// Cin = num_channels = 1..4, Cout = 16 or 32).  K = 27*Cin is too short for the MFMA path and
// the layer is HBM-bound anyway (reads 1 voxel, writes 16 channels), so these are VALU kernels:
//   forward : one thread per output voxel, all COUT channels in registers, weights through the
//             scalar cache, 16-byte coalesced NDHWC stores.
//   wgrad   : thread = (tap, ci) x voxel-lane; per voxel one LDS read of X and COUT FMAs against
//             the staged dY row; persistent workgroups, per-workgroup partial slabs (deterministic).
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
  float* partials;   // [grid][COUT][Cin][27]
  int N, Dx, Hx, Wx, Dy, Hy, Wy, Cin, ldx, ldy, stride;
  int tz, ty, tx, ntiles;
};

// tile of dY voxels: 2 x 4 x 16 (stride 2 -> X halo 5 x 9 x 33)
template <typename T, int COUT, int S>
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(SmallWgradParams p) {
  constexpr int TD = 2, TH = 4, TW = 16, NV = TD * TH * TW;
  constexpr int HD = (TD - 1) * S + 3, HH = (TH - 1) * S + 3, HW = (TW - 1) * S + 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ysm = reinterpret_cast<float*>(smem);                 // [NV][COUT] f32
  float* xsm = ysm + NV * COUT;                                // [Cin][HD*HH*HW] f32
  float* red = xsm + p.Cin * HD * HH * HW;                     // [256][...] reuse below
  const int tid = threadIdx.x;
  const int combos = 27 * p.Cin;                               // (ci, tap)
  const int lanes = 256 / combos;                              // voxel lanes
  const int combo = tid % combos, vl = tid / combos;
  const bool active = vl < lanes;
  const int ci = combo / 27, tap = combo % 27;
  const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
  const T* x = (const T*)p.x;
  const T* dy = (const T*)p.dy;
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    int t = tile;
    const int txi = t % p.tx; t /= p.tx;
    const int tyi = t % p.ty; t /= p.ty;
    const int tzi = t % p.tz;
    const int n = t / p.tz;
    const int oz0 = tzi * TD, oy0 = tyi * TH, ox0 = txi * TW;
    __syncthreads();
    for (int i = tid; i < NV * COUT / 4; i += 256) {
      const int v = i / (COUT / 4), c4 = (i % (COUT / 4)) * 4;
      const int z = oz0 + v / (TW * TH), y = oy0 + (v / TW) % TH, xx = ox0 + v % TW;
      f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
      if (z < p.Dy && y < p.Hy && xx < p.Wy)
        val = load4<T>(dy + ((((int64_t)n * p.Dy + z) * p.Hy + y) * p.Wy + xx) * p.ldy + c4);
      *reinterpret_cast<f32x4*>(ysm + v * COUT + c4) = val;
    }
    for (int i = tid; i < p.Cin * HD * HH * HW; i += 256) {
      const int c = i / (HD * HH * HW), r = i % (HD * HH * HW);
      const int hx = r % HW, hy = (r / HW) % HH, hz = r / (HW * HH);
      const int z = oz0 * S - 1 + hz, y = oy0 * S - 1 + hy, xx = ox0 * S - 1 + hx;
      float val = 0.f;
      if ((unsigned)z < (unsigned)p.Dx && (unsigned)y < (unsigned)p.Hx && (unsigned)xx < (unsigned)p.Wx)
        val = Elem<T>::ld(x + ((((int64_t)n * p.Dx + z) * p.Hx + y) * p.Wx + xx) * p.ldx + c);
      xsm[i] = val;
    }
    __syncthreads();
    if (active) {
      const float* xs = xsm + ci * HD * HH * HW;
      for (int v = vl; v < NV; v += lanes) {
        const int vz = v / (TW * TH), vy = (v / TW) % TH, vx = v % TW;
        const float a = xs[((vz * S + kd) * HH + vy * S + kh) * HW + vx * S + kw];
        const f32x4* yr = reinterpret_cast<const f32x4*>(ysm + v * COUT);
#pragma unroll
        for (int c4 = 0; c4 < COUT / 4; ++c4) {
          const f32x4 d = yr[c4];
          acc[4 * c4 + 0] = fmaf(a, d[0], acc[4 * c4 + 0]);
          acc[4 * c4 + 1] = fmaf(a, d[1], acc[4 * c4 + 1]);
          acc[4 * c4 + 2] = fmaf(a, d[2], acc[4 * c4 + 2]);
          acc[4 * c4 + 3] = fmaf(a, d[3], acc[4 * c4 + 3]);
        }
      }
    }
  }
  // reduce the voxel lanes (fixed order) -> slab [COUT][Cin][27]
  __syncthreads();
  float* r2 = reinterpret_cast<float*>(smem);   // [256][COUT]  (fits: checked on the host)
#pragma unroll
  for (int c = 0; c < COUT; ++c) r2[tid * COUT + c] = active ? acc[c] : 0.f;
  __syncthreads();
  float* slab = p.partials + (int64_t)blockIdx.x * COUT * combos;
  for (int o = tid; o < COUT * combos; o += 256) {
    const int co = o / combos, cb = o % combos;   // cb = ci*27 + tap
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += r2[(l * combos + cb) * COUT + co];
    slab[o] = s;
  }
  (void)red;
}

template <int S> static constexpr int small_halo() { return ((2 - 1) * S + 3) * ((4 - 1) * S + 3) * ((16 - 1) * S + 3); }

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
  const int nt = dy->n * cdiv(dy->d, 2) * cdiv(dy->h, 4) * cdiv(dy->w, 16);
  return nt < 512 ? nt : 512;
}

int conv_small_wgrad(int dtype, const segmi_act* x, const segmi_act* dy, float* partials,
                     int stride, hipStream_t st) {
  SmallWgradParams p{};
  p.x = x->data; p.dy = dy->data; p.partials = partials;
  p.N = x->n; p.Dx = x->d; p.Hx = x->h; p.Wx = x->w; p.Dy = dy->d; p.Hy = dy->h; p.Wy = dy->w;
  p.Cin = x->c; p.ldx = x->ld; p.ldy = dy->ld; p.stride = stride;
  p.tz = cdiv(dy->d, 2); p.ty = cdiv(dy->h, 4); p.tx = cdiv(dy->w, 16);
  p.ntiles = dy->n * p.tz * p.ty * p.tx;
  const int grid = conv_small_wgrad_slabs(dy);
  const int cout = dy->c;
  const int halo = stride == 2 ? small_halo<2>() : small_halo<1>();
  size_t lds = (size_t)(128 * cout + x->c * halo) * sizeof(float);
  const size_t lds_red = (size_t)256 * cout * sizeof(float);
  if (lds < lds_red) lds = lds_red;
#define L(TT, CO, SS) hipLaunchKernelGGL((conv_small_wgrad_kernel<TT, CO, SS>), grid, 256, lds, st, p)
  if (dtype == SEGMI_F32) {
    if (cout == 16) { if (stride == 2) L(float, 16, 2); else L(float, 16, 1); }
    else { if (stride == 2) L(float, 32, 2); else L(float, 32, 1); }
  } else {
    if (cout == 16) { if (stride == 2) L(bf16_t, 16, 2); else L(bf16_t, 16, 1); }
    else { if (stride == 2) L(bf16_t, 32, 2); else L(bf16_t, 32, 1); }
  }
#undef L
  SEGMI_LAUNCH_CHECK("conv3d_wgrad(small-cin)");
  return SEGMI_OK;
}

}  // namespace segmi
