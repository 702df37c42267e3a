// convt_fwd_impl.h -- MFMA ConvTranspose3d k3 s2 p1 forward (also the dgrad of a stride-2 conv).
//
// out[o] = sum_{i,k : o = 2i - 1 + k} in[i] * W[k].  Per dim an even output (r=0) takes the
// single tap k=1 of input i=o/2; an odd output (r=1) takes k=2 of i=(o-1)/2 and k=0 of
// i=(o+1)/2.  So the 8 output-parity classes are 8 small GEMMs with 1,2,2,4,2,4,4,8 taps (27
// in total) over the SAME input tile -- no zero insertion, no wasted MACs.
//
// One workgroup stages an input tile of 64 voxels (+1 halo on the high side) per channel
// chunk; the 4 waves split the 8 classes {7} {3,5} {6,1,0} {2,4} (8/8/7/4 taps), each wave
// reusing its weight fragments across the 4 voxel tiles.  D[co][vox] orientation and epilogue
// as in conv_fwd_impl.h; voxel (z,y,x) of class (rd,rh,rw) lands at (2z+rd, 2y+rh, 2x+rw).
#pragma once
#include "common.h"
#include "fin_tail.h"

namespace segmi {

struct ConvTParams {
  const void* in;
  void* out;
  const void* wfrag;
  const float* bias;
  const float* alpha;
  const void* res;
  float* stats;
  int N, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, ldi, ldo, ldr;
  int tz, ty, tx;
  int nchunks, ntiles_total;
  int64_t class_off[8];  // byte offsets of each class in the pack
  // BatchNorm statistics finalised by the last workgroup of this launch (fin_tail.h)
  int fin_on;
  FinTail ft;
  BnFin bfin;
};

constexpr int kCtCls[4][3] = {{7, 0, 0}, {3, 5, 0}, {6, 1, 0}, {2, 4, 0}};
constexpr int kCtNCls[4] = {1, 2, 3, 2};

constexpr int ct_ntaps_c(int p) {
  return (1 + ((p >> 2) & 1)) * (1 + ((p >> 1) & 1)) * (1 + (p & 1));
}
// row offset (in halo rows) of tap t of class p
constexpr int ct_tap_rowoff(int p, int t, int HH, int HW) {
  const int rw = p & 1, rh = (p >> 1) & 1, rd = (p >> 2) & 1;
  const int nw = 1 + rw, nh = 1 + rh;
  const int tw = t % nw, th = (t / nw) % nh, td = t / (nw * nh);
  const int dw = rw ? tw : 0, dh = rh ? th : 0, dd = rd ? td : 0;
  return (dd * HH + dh) * HW + dw;
}

template <typename T, int CK, int TD, int TH, int TW>
struct ConvTGeom {
  static constexpr int KG = Elem<T>::KG;
  static constexpr int SPT = CK / KG;
  static constexpr int HD = TD + 1, HH = TH + 1, HW = TW + 1;
  static constexpr int RAWB = CK * (int)sizeof(T);
  static constexpr int ROWB = RAWB == 32 ? 32 : RAWB + 16;
  static constexpr int CPR = RAWB / 16;
  static constexpr int NVT = TD * TH * TW / 16;
  static constexpr int LDS_BYTES = HD * HH * HW * ROWB;
  static_assert(NVT == 4, "transposed-conv tile is 64 input voxels");
  static_assert(SPT == 2 || SPT == 4, "unsupported chunk width");
};

// all weight fragments of wave W's classes for chunk c (<= 8 k-steps): fetched before the halo
// staging so the L2 latency hides under it
template <typename T, int CK, int NT, int TD, int TH, int TW, int W>
__device__ __forceinline__ void convt_wave_loadw(frag_t (&wall)[8][NT], const char* wfrag,
                                                 const ConvTParams& p, int c, int nt0, int lane) {
  using G = ConvTGeom<T, CK, TD, TH, TW>;
  int idx = 0;
#pragma unroll
  for (int ci = 0; ci < kCtNCls[W]; ++ci) {
    const int cls = kCtCls[W][ci];
    const int nsteps = (ct_ntaps_c(cls) * G::SPT + 3) / 4;
    const char* wb = wfrag + p.class_off[cls] +
                     (((int64_t)c * nsteps) * p.ntiles_total + nt0) * 1024 + lane * 16;
#pragma unroll
    for (int s = 0; s < nsteps; ++s) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wall[idx][j] = *reinterpret_cast<const frag_t*>(wb + ((int64_t)s * p.ntiles_total + j) * 1024);
      ++idx;
    }
  }
}

template <typename T, int CK, int NT, int TD, int TH, int TW, int W>
__device__ __forceinline__ void convt_wave_compute(f32x4 (&acc)[3][4][NT], const char* smem,
                                                   const int (&vaddr)[4],
                                                   const frag_t (&wall)[8][NT], int g) {
  using G = ConvTGeom<T, CK, TD, TH, TW>;
  int idx = 0;
#pragma unroll
  for (int ci = 0; ci < kCtNCls[W]; ++ci) {
    const int cls = kCtCls[W][ci];
    const int ntp = ct_ntaps_c(cls);
    const int nsteps = (ntp * G::SPT + 3) / 4;
#pragma unroll
    for (int s = 0; s < nsteps; ++s) {
      int loff;
      if constexpr (G::SPT == 2) {
        const int t0 = 2 * s, t1 = 2 * s + 1;
        const int o0 = ct_tap_rowoff(cls, t0, G::HH, G::HW);
        const int o1 = t1 < ntp ? ct_tap_rowoff(cls, t1, G::HH, G::HW) : 0;
        loff = ((g >> 1) ? o1 : o0) * G::ROWB + (g & 1) * 16;
      } else {
        const int tap = (4 * s) / G::SPT;
        const int sub0 = (4 * s) % G::SPT;
        loff = ct_tap_rowoff(cls, tap, G::HH, G::HW) * G::ROWB + (sub0 + g) * 16;
      }
#pragma unroll
      for (int vt = 0; vt < 4; ++vt) {
        const frag_t a = *reinterpret_cast<const frag_t*>(smem + vaddr[vt] + loff);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[ci][vt][j] = mma16<T>(wall[idx][j], a, acc[ci][vt][j]);
      }
      ++idx;
    }
  }
}

// Epilogue of one wave: bias, BN statistics, PReLU, residual, NDHWC store.
// FAST (tile completely inside the output): no per-lane bounds checks, 32-bit per-lane offsets
// relative to a wave-uniform tile base -- the slow form spends ~45 VALU ops per 16x16 tile on
// 64-bit index arithmetic, more than the MFMA work of this kernel.
template <typename T, int NT, int TD, int TH, int TW, int W, bool FAST>
__device__ __forceinline__ void convt_wave_store(const f32x4 (&acc)[3][4][NT],
                                                 const ConvTParams& p, int n, int iz0, int iy0,
                                                 int ix0, int nt0, int g, int r,
                                                 f32x4 (&ssum)[NT], f32x4 (&ssq)[NT]) {
  f32x4 bias4[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4[j] = *reinterpret_cast<const f32x4*>(p.bias + (nt0 + j) * 16 + 4 * g);
  }
  const bool has_alpha = p.alpha != nullptr;
  const float alpha = has_alpha ? *p.alpha : 0.f;
  T* outp = (T*)p.out;
  const T* resp = (const T*)p.res;
  // wave-uniform base of the tile's first output voxel, per-lane 32-bit offsets for the 4 tiles
  const int64_t tile_vox = (((int64_t)n * p.Do + 2 * iz0) * p.Ho + 2 * iy0) * p.Wo + 2 * ix0;
  int lvox[4];
#pragma unroll
  for (int vt = 0; vt < 4; ++vt) {
    const int idx = vt * 16 + r;
    lvox[vt] = (2 * (idx / (TW * TH)) * p.Ho + 2 * ((idx / TW) % TH)) * p.Wo + 2 * (idx % TW);
  }
  // residual rows first, all at once (a load -> add -> store chain per tile would serialise the
  // stores on s_waitcnt vmcnt(0); see touch_v in common.h)
  f32x4 resv[3][4][NT];
  if (resp) {
#pragma unroll
    for (int ci = 0; ci < kCtNCls[W]; ++ci) {
      const int cls = kCtCls[W][ci];
      const int rw = cls & 1, rh = (cls >> 1) & 1, rd = (cls >> 2) & 1;
      const int cls_vox = (rd * p.Ho + rh) * p.Wo + rw;
      const T* rb = resp + (tile_vox + cls_vox) * p.ldr + nt0 * 16 + 4 * g;
#pragma unroll
      for (int vt = 0; vt < 4; ++vt) {
        bool valid = true;
        if constexpr (!FAST) {
          const int idx = vt * 16 + r;
          const int oz = 2 * (iz0 + idx / (TW * TH)) + rd;
          const int oy = 2 * (iy0 + (idx / TW) % TH) + rh;
          const int ox = 2 * (ix0 + idx % TW) + rw;
          valid = oz < p.Do && oy < p.Ho && ox < p.Wo;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
          resv[ci][vt][j] = valid ? load4<T>(rb + (int64_t)lvox[vt] * p.ldr + j * 16)
                                  : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int ci = 0; ci < kCtNCls[W]; ++ci)
#pragma unroll
      for (int vt = 0; vt < 4; ++vt)
#pragma unroll
        for (int j = 0; j < NT; ++j) touch_v(resv[ci][vt][j]);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) touch_v(bias4[j]);
#pragma unroll
  for (int ci = 0; ci < kCtNCls[W]; ++ci) {
    const int cls = kCtCls[W][ci];
    const int rw = cls & 1, rh = (cls >> 1) & 1, rd = (cls >> 2) & 1;
    const int cls_vox = (rd * p.Ho + rh) * p.Wo + rw;
    T* ob = outp + (tile_vox + cls_vox) * p.ldo + nt0 * 16 + 4 * g;
#pragma unroll
    for (int vt = 0; vt < 4; ++vt) {
      bool valid = true;
      if constexpr (!FAST) {
        const int idx = vt * 16 + r;
        const int oz = 2 * (iz0 + idx / (TW * TH)) + rd;
        const int oy = 2 * (iy0 + (idx / TW) % TH) + rh;
        const int ox = 2 * (ix0 + idx % TW) + rw;
        valid = oz < p.Do && oy < p.Ho && ox < p.Wo;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 v = acc[ci][vt][j] + bias4[j];
        if (valid) {
          if (p.stats) {
            ssum[j] += v;
            ssq[j] += v * v;
          }
          if (has_alpha) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
          }
          if (resp) v += resv[ci][vt][j];
          store4<T>(ob + (int64_t)lvox[vt] * p.ldo + j * 16, v);
        }
      }
    }
  }
}

template <typename T, int CK, int NT, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void convt_fwd_mfma_kernel(ConvTParams p) {
  using G = ConvTGeom<T, CK, TD, TH, TW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;

  int t = blockIdx.x;
  const int txi = t % p.tx; t /= p.tx;
  const int tyi = t % p.ty; t /= p.ty;
  const int tzi = t % p.tz;
  const int n = t / p.tz;
  const int iz0 = tzi * TD, iy0 = tyi * TH, ix0 = txi * TW;
  const int nt0 = blockIdx.y * NT;

  f32x4 acc[3][4][NT];
#pragma unroll
  for (int a = 0; a < kCtNCls[0] + 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[a][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int vaddr[4];
#pragma unroll
  for (int vt = 0; vt < 4; ++vt) {
    const int idx = vt * 16 + r;
    const int x = idx % TW, y = (idx / TW) % TH, z = idx / (TW * TH);
    vaddr[vt] = ((z * G::HH + y) * G::HW + x) * G::ROWB;
  }

  // staging descriptors: 32-bit offsets relative to the wave-uniform tile base; interior tiles
  // (halo inside the image) load without per-lane bounds checks
  constexpr int NCH = G::HD * G::HH * G::HW * G::CPR;
  constexpr int NLD = (NCH + 255) / 256;
  const char* tile_in = (const char*)p.in +
      ((((int64_t)n * p.Di + iz0) * p.Hi + iy0) * p.Wi + ix0) * p.ldi * (int64_t)sizeof(T);
  const bool interior = iz0 + TD < p.Di && iy0 + TH < p.Hi && ix0 + TW < p.Wi;
  int s_goff[NLD], s_loff[NLD];
  bool s_ok[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = tid + 256 * k;
    const int v = i / G::CPR, ch = i % G::CPR;
    const int hx = v % G::HW, hy = (v / G::HW) % G::HH, hz = v / (G::HW * G::HH);
    s_loff[k] = v * G::ROWB + ch * 16;
    s_goff[k] = ((hz * p.Hi + hy) * p.Wi + hx) * p.ldi * (int)sizeof(T) + ch * 16;
    s_ok[k] = i < NCH && (interior || (iz0 + hz < p.Di && iy0 + hy < p.Hi && ix0 + hx < p.Wi));
  }

  const char* wfrag = (const char*)p.wfrag;
  for (int c = 0; c < p.nchunks; ++c) {
    if (c > 0) __syncthreads();
    frag_t wall[8][NT];
    if (wave == 0) convt_wave_loadw<T, CK, NT, TD, TH, TW, 0>(wall, wfrag, p, c, nt0, lane);
    else if (wave == 1) convt_wave_loadw<T, CK, NT, TD, TH, TW, 1>(wall, wfrag, p, c, nt0, lane);
    else if (wave == 2) convt_wave_loadw<T, CK, NT, TD, TH, TW, 2>(wall, wfrag, p, c, nt0, lane);
    else convt_wave_loadw<T, CK, NT, TD, TH, TW, 3>(wall, wfrag, p, c, nt0, lane);
    const char* cin = tile_in + c * CK * (int)sizeof(T);
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      frag_t val = frag_t{0u, 0u, 0u, 0u};
      if (s_ok[k]) val = *reinterpret_cast<const frag_t*>(cin + s_goff[k]);
      if (tid + 256 * k < NCH) *reinterpret_cast<frag_t*>(smem + s_loff[k]) = val;
    }
    __syncthreads();
    if (wave == 0) convt_wave_compute<T, CK, NT, TD, TH, TW, 0>(acc, smem, vaddr, wall, g);
    else if (wave == 1) convt_wave_compute<T, CK, NT, TD, TH, TW, 1>(acc, smem, vaddr, wall, g);
    else if (wave == 2) convt_wave_compute<T, CK, NT, TD, TH, TW, 2>(acc, smem, vaddr, wall, g);
    else convt_wave_compute<T, CK, NT, TD, TH, TW, 3>(acc, smem, vaddr, wall, g);
  }

  f32x4 ssum[NT], ssq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    ssum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ssq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool full = 2 * (iz0 + TD) <= p.Do && 2 * (iy0 + TH) <= p.Ho && 2 * (ix0 + TW) <= p.Wo;
#define CT_STORE(WV, F) convt_wave_store<T, NT, TD, TH, TW, WV, F>(acc, p, n, iz0, iy0, ix0, nt0, g, r, ssum, ssq)
  if (full) {
    if (wave == 0) CT_STORE(0, true); else if (wave == 1) CT_STORE(1, true);
    else if (wave == 2) CT_STORE(2, true); else CT_STORE(3, true);
  } else {
    if (wave == 0) CT_STORE(0, false); else if (wave == 1) CT_STORE(1, false);
    else if (wave == 2) CT_STORE(2, false); else CT_STORE(3, false);
  }
#undef CT_STORE

  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][2][NT*16]
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
      fin_store(&p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + nt0 * 16 + ch], sacc);
    }
    fin_tail_run<BnFin, 256, offsetof(ConvTParams, ft), offsetof(ConvTParams, bfin)>(p.stats, smem);
  }
}

template <typename T, int CK, int NT, int TD, int TH, int TW>
static int launch_convt_cfg(ConvTParams p, hipStream_t st) {
  using G = ConvTGeom<T, CK, TD, TH, TW>;
  p.tz = cdiv(p.Di, TD);
  p.ty = cdiv(p.Hi, TH);
  p.tx = cdiv(p.Wi, TW);
  const int64_t nb = (int64_t)p.N * p.tz * p.ty * p.tx;
  SEGMI_CHECK_ARG(nb < (1ll << 31), "convT3d: too many tiles");
  dim3 grid((unsigned)nb, (unsigned)(p.Cout / (16 * NT)));
  constexpr int lds0 = G::LDS_BYTES > 4 * 2 * NT * 16 * 4 ? G::LDS_BYTES : 4 * 2 * NT * 16 * 4;
  p.fin_on = p.fin_on && p.stats;
  const int lds = (int)fin_tail_arm(p, grid, 256, 2 * p.Cout, lds0);
  SEGMI_CHECK_ARG(lds <= 64 * 1024, "convT3d: LDS budget");
  hipLaunchKernelGGL((convt_fwd_mfma_kernel<T, CK, NT, TD, TH, TW>), grid, 256, lds, st, p);
  SEGMI_LAUNCH_CHECK("convT3d_fwd(mfma)");
  return SEGMI_OK;
}

template <typename T, int CK>
static int launch_convt_nt(const ConvTParams& p, hipStream_t st) {
  const bool wide = p.Wi > 8;
  const int nt = p.Cout / 16;
  if (nt % 2 == 0) {
    if (wide) return launch_convt_cfg<T, CK, 2, 2, 2, 16>(p, st);
    return launch_convt_cfg<T, CK, 2, 2, 4, 8>(p, st);
  }
  if (wide) return launch_convt_cfg<T, CK, 1, 2, 2, 16>(p, st);
  return launch_convt_cfg<T, CK, 1, 2, 4, 8>(p, st);
}

template <typename T>
static int launch_convt_mfma_t(ConvTParams p, hipStream_t st) {
  constexpr int dt = sizeof(T) == 4 ? SEGMI_F32 : SEGMI_BF16;
  const int ck = pick_ck(dt, p.Cin);
  const PackGeom g = pack_geom(dt, p.Cin, p.Cout, 1);
  int64_t off = 0;
  for (int c = 0; c < 8; ++c) {
    p.class_off[c] = off;
    off += (int64_t)g.nchunks * ((ct_ntaps(c) * g.SPT + 3) / 4) * g.ntiles * 1024;
  }
  if constexpr (sizeof(T) == 2) {
    if (ck == 32) return launch_convt_nt<T, 32>(p, st);
  }
  return launch_convt_nt<T, 16>(p, st);
}

static inline int convt_tile_rows(int dtype, const segmi_act* in) {
  const bool wide = in->w > 8;
  const int td = 2, th = wide ? 2 : 4, tw = wide ? 16 : 8;
  (void)dtype;
  return in->n * cdiv(in->d, td) * cdiv(in->h, th) * cdiv(in->w, tw);
}

}  // namespace segmi
