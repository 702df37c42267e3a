// conv_ks_impl.h -- K-split variant of the MFMA Conv3d forward for the deep layers (many input
// channels, few voxels: 8^3 .. 32^3 per sample).
//
// conv_fwd_impl.h gives every wave 2 voxel tiles and ALL k-steps: each of the 4 waves streams
// the same weight fragments from L2/L1 (4x redundant, 64 B/clk L1) and gets only 2 MFMAs per
// fragment to hide the next fragment's latency -- with one workgroup per CU (16^3 x 128, 8^3 x 256)
// that latency chain is the run time (45-180 TFLOP/s measured).  Here every wave owns ALL 8 voxel
// tiles of the workgroup and a quarter of the k-steps (taps w, w+4, ...): a weight fragment is
// fetched once per workgroup and feeds 8*NT MFMAs, so a one-step-ahead prefetch covers the L2
// latency.  The 4 partial accumulators are summed through LDS in fixed wave order (deterministic),
// then wave w runs the usual epilogue on voxel tiles 2w, 2w+1.
#pragma once
#include "conv_fwd_impl.h"

namespace segmi {

template <typename T, int CK, int KS, int S, int NT, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void conv_fwd_ks_kernel(ConvParams p) {
  using G = ConvGeom<T, CK, KS, S, TD, TH, TW>;
  static_assert(G::SPT == 4, "k-split kernel: one k-step per tap");
  static_assert(G::NVT == 8, "k-split kernel: 8 voxel tiles per workgroup");
  constexpr int NSW = (G::NSTEP + 3) / 4;   // k-steps per wave per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;

  int t = blockIdx.x;
  const int txi = t % p.tx; t /= p.tx;
  const int tyi = t % p.ty; t /= p.ty;
  const int tzi = t % p.tz;
  const int n = t / p.tz;
  const int oz0 = tzi * TD, oy0 = tyi * TH, ox0 = txi * TW;
  const int iz0 = oz0 * S - G::PAD, iy0 = oy0 * S - G::PAD, ix0 = ox0 * S - G::PAD;
  const int nt0 = blockIdx.y * NT;

  f32x4 acc[8][NT];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int vaddr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 16 + r;
    const int x = idx % TW, y = (idx / TW) % TH, z = idx / (TW * TH);
    vaddr[i] = ((z * S * G::HH + y * S) * G::HW + x * S) * G::ROWB + g * 16;
  }
  // LDS offsets of this wave's taps
  int tapoff[NSW];
#pragma unroll
  for (int si = 0; si < NSW; ++si) {
    int tap = wave + 4 * si;
    if (tap > G::NTAPS - 1) tap = G::NTAPS - 1;
    tapoff[si] = (((tap / (KS * KS)) * G::HH + (tap / KS) % KS) * G::HW + tap % KS) * G::ROWB;
  }

  constexpr int NCH = G::HD * G::HH * G::HW * G::CPR;
  constexpr int NLD = (NCH + 255) / 256;
  const char* inb = (const char*)p.in;
  frag_t stg[NLD];
  auto fetch = [&](int c) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 256 * k;
      const int v = i / G::CPR, ch = i % G::CPR;
      const int hx = v % G::HW, hy = (v / G::HW) % G::HH, hz = v / (G::HW * G::HH);
      const int z = iz0 + hz, y = iy0 + hy, x = ix0 + hx;
      stg[k] = frag_t{0u, 0u, 0u, 0u};
      if (i < NCH && (unsigned)z < (unsigned)p.Di && (unsigned)y < (unsigned)p.Hi &&
          (unsigned)x < (unsigned)p.Wi) {
        const int64_t e = ((((int64_t)n * p.Di + z) * p.Hi + y) * p.Wi + x) * p.ldi + c * CK;
        stg[k] = *reinterpret_cast<const frag_t*>(inb + e * (int64_t)sizeof(T) + ch * 16);
      }
    }
  };
  // this wave's weight fragments of a chunk (k-steps wave, wave+4, ...): all fetched at once, one
  // chunk ahead, so a cold L2 costs one exposed latency per kernel instead of one per k-step
  const char* wb0 = (const char*)p.wfrag + (int64_t)nt0 * 1024 + lane * 16;
  frag_t wcur[NSW][NT], wnxt[NSW][NT];
  auto fetch_w = [&](int c, frag_t (&dst)[NSW][NT]) {
#pragma unroll
    for (int si = 0; si < NSW; ++si) {
      int s = wave + 4 * si;
      if (s > G::NSTEP - 1) s = G::NSTEP - 1;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        dst[si][j] = *reinterpret_cast<const frag_t*>(
            wb0 + (((int64_t)c * G::NSTEP + s) * p.ntiles_total + j) * 1024);
    }
  };
  fetch(0);
  fetch_w(0, wcur);
  for (int c = 0; c < p.nchunks; ++c) {
    if (c > 0) __syncthreads();
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 256 * k;
      if (i < NCH) *reinterpret_cast<frag_t*>(smem + (i / G::CPR) * G::ROWB + (i % G::CPR) * 16) = stg[k];
    }
    __syncthreads();
    if (c + 1 < p.nchunks) {
      fetch(c + 1);
      fetch_w(c + 1, wnxt);
    }
#pragma unroll
    for (int si = 0; si < NSW; ++si) {
      const int s = wave + 4 * si;
      if (s < G::NSTEP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const frag_t a = *reinterpret_cast<const frag_t*>(smem + vaddr[i] + tapoff[si]);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mma16<T>(wcur[si][j], a, acc[i][j]);
        }
      }
    }
#pragma unroll
    for (int si = 0; si < NSW; ++si)
#pragma unroll
      for (int j = 0; j < NT; ++j) wcur[si][j] = wnxt[si][j];
  }

  // ---- fixed-order sum of the 4 waves' partial accumulators: red[wave][tile][NT][lane]
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) red[((wave * 8 + i) * NT + j) * 64 + lane] = acc[i][j];
  __syncthreads();
  f32x4 fin[2][NT];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int i = wave * 2 + ii;
      f32x4 v = red[((0 * 8 + i) * NT + j) * 64 + lane];
      v += red[((1 * 8 + i) * NT + j) * 64 + lane];
      v += red[((2 * 8 + i) * NT + j) * 64 + lane];
      v += red[((3 * 8 + i) * NT + j) * 64 + lane];
      fin[ii][j] = v;
    }

  // ---- epilogue (as conv_fwd_impl.h): wave w owns voxel tiles 2w, 2w+1
  f32x4 bias4[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4[j] = *reinterpret_cast<const f32x4*>(p.bias + (nt0 + j) * 16 + 4 * g);
  }
  const bool has_alpha = p.alpha != nullptr;
  const float alpha = has_alpha ? *p.alpha : 0.f;
  f32x4 ssum[NT], ssq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    ssum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ssq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  T* outp = (T*)p.out;
  const T* resp = (const T*)p.res;
  f32x4 resv[2][NT];
  if (resp) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int idx = (wave * 2 + ii) * 16 + r;
      const int oz = oz0 + idx / (TW * TH), oy = oy0 + (idx / TW) % TH, ox = ox0 + idx % TW;
      const bool valid = oz < p.Do && oy < p.Ho && ox < p.Wo;
      const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        resv[ii][j] = valid ? load4<T>(resp + vox * p.ldr + (nt0 + j) * 16 + 4 * g)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < NT; ++j) touch_v(resv[ii][j]);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) touch_v(bias4[j]);
#pragma unroll
  for (int k = 0; k < NLD; ++k) touch_v(stg[k]);
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) {
    const int idx = (wave * 2 + ii) * 16 + r;
    const int oz = oz0 + idx / (TW * TH), oy = oy0 + (idx / TW) % TH, ox = ox0 + idx % TW;
    const bool valid = oz < p.Do && oy < p.Ho && ox < p.Wo;
    const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x4 v = fin[ii][j] + bias4[j];
      if (valid) {
        if (p.stats) {
          ssum[j] += v;
          ssq[j] += v * v;
        }
        if (has_alpha) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
        }
        const int co = (nt0 + j) * 16 + 4 * g;
        if (resp) v += resv[ii][j];
        store4<T>(outp + vox * p.ldo + co, v);
      }
    }
  }
  if (p.stats) {
    __syncthreads();  // everyone is done with the partial sums
    float* rs = reinterpret_cast<float*>(smem);  // [wave][2][NT*16]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = row16_sum(ssum[j][e]);
        const float b = row16_sum(ssq[j][e]);
        if (r == 0) {
          rs[(wave * 2 + 0) * NT * 16 + j * 16 + 4 * g + e] = a;
          rs[(wave * 2 + 1) * NT * 16 + j * 16 + 4 * g + e] = b;
        }
      }
    __syncthreads();
    if (tid < 2 * NT * 16) {
      const int which = tid / (NT * 16), ch = tid % (NT * 16);
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sacc += rs[(w * 2 + which) * NT * 16 + ch];
      fin_store(&p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + nt0 * 16 + ch], sacc);
    }
    fin_tail_run<BnFin, 256, offsetof(ConvParams, ft), offsetof(ConvParams, bfin)>(p.stats, smem);
  }
}

template <typename T, int CK, int KS, int S, int NT, int TD, int TH, int TW>
static int launch_conv_ks_cfg(ConvParams p, hipStream_t st) {
  using G = ConvGeom<T, CK, KS, S, TD, TH, TW>;
  p.tz = cdiv(p.Do, TD);
  p.ty = cdiv(p.Ho, TH);
  p.tx = cdiv(p.Wo, TW);
  const int64_t nb = (int64_t)p.N * p.tz * p.ty * p.tx;
  SEGMI_CHECK_ARG(nb < (1ll << 31), "conv3d: too many tiles");
  dim3 grid((unsigned)nb, (unsigned)(p.Cout / (16 * NT)));
  constexpr int red = 4 * 8 * NT * 64 * 16;
  constexpr int lds0 = G::LDS_BYTES > red ? G::LDS_BYTES : red;
  p.fin_on = p.fin_on && p.stats;
  const int lds = (int)fin_tail_arm(p, grid, 256, 2 * p.Cout, lds0);
  auto kern = conv_fwd_ks_kernel<T, CK, KS, S, NT, TD, TH, TW>;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, 256, lds, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_fwd(mfma, k-split)");
  return SEGMI_OK;
}

// k-split tiles: 128 output voxels (8 voxel tiles) per workgroup, 16 output channels
static inline int conv_ks_rows(const segmi_act* out) {
  const bool wide = out->w > 8;
  const int td = wide ? 2 : 4, th = 4, tw = wide ? 16 : 8;
  return out->n * cdiv(out->d, td) * cdiv(out->h, th) * cdiv(out->w, tw);
}
// Layers the k-split kernel takes: k3 s1 with at least two channel chunks of one-k-step-per-tap
// width.  Measured with cold caches (scripts/conv_microbench.py, batch 8): 128->128 @16^3
// 99 -> 56 us, 256->256 @8^3 60 -> 36 us, 256->128 @16^3 181 -> 98 us, 64->64 @32^3 119 -> 114 us;
// stride 2 does not gain (86 -> 97 us) and stays on conv_fwd_impl.h.
static inline bool conv_ks_ok(int dtype, int cin, int ksize, int stride) {
  static const bool off = getenv("SEGMI_CONV_KS") && atoi(getenv("SEGMI_CONV_KS")) == 0;
  static const bool s2 = getenv("SEGMI_CONV_KS_S2") && atoi(getenv("SEGMI_CONV_KS_S2")) == 1;
  static const bool one = !(getenv("SEGMI_CONV_KS_1CH") && atoi(getenv("SEGMI_CONV_KS_1CH")) == 0);
  if (off || ksize != 3 || (stride != 1 && !s2)) return false;
  const int ck = pick_ck(dtype, cin);
  const int spt = ck / (dtype == SEGMI_F32 ? 4 : 8);
  return spt == 4 && cin / ck >= (one ? 1 : 2);
}

template <typename T, int CK, int S>
static int launch_conv_ks_s(const ConvParams& p, hipStream_t st) {
  const segmi_act o{nullptr, p.N, p.Do, p.Ho, p.Wo, p.Cout, p.Cout};
  const int nt = p.Cout / 16;
  // two output tiles per workgroup (input fragment reuse) unless that leaves < 512 workgroups
  const bool two = nt % 2 == 0 && (int64_t)conv_ks_rows(&o) * (nt / 2) >= 512;
  const bool wide = p.Wo > 8;
  if (two) return wide ? launch_conv_ks_cfg<T, CK, 3, S, 2, 2, 4, 16>(p, st)
                       : launch_conv_ks_cfg<T, CK, 3, S, 2, 4, 4, 8>(p, st);
  return wide ? launch_conv_ks_cfg<T, CK, 3, S, 1, 2, 4, 16>(p, st)
              : launch_conv_ks_cfg<T, CK, 3, S, 1, 4, 4, 8>(p, st);
}
template <typename T, int CK>
static int launch_conv_ks_t(const ConvParams& p, int stride, hipStream_t st) {
  if (stride == 2) return launch_conv_ks_s<T, CK, 2>(p, st);
  return launch_conv_ks_s<T, CK, 1>(p, st);
}

}  // namespace segmi
