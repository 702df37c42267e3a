// conv_fwd_impl.h -- MFMA implicit-GEMM Conv3d forward (k in {1,3}, stride in {1,2}).
//
// One 256-thread workgroup (4 waves) computes an output tile of TD x TH x TW voxels for
// NT*16 output channels.  Per input-channel chunk (CK channels) the input halo tile is staged
// NDHWC -> LDS as [voxel][CK] rows; the GEMM is  D[co][vox] += W[co][k] * X[vox][k]  with the
// fragment-packed weights as MFMA A operand (streamed from L1/L2, 1 KiB coalesced per
// fragment) and the LDS rows as B operand (one ds_read_b128 per lane per k-step).  With the
// channel index on the accumulator rows each lane ends up with 4 consecutive channels of one
// voxel: the epilogue (bias, BN statistics, PReLU, residual) stores 8 B (bf16) / 16 B (f32)
// per lane, a fully coalesced NDHWC row per 4 lanes.
#pragma once
#include "common.h"
#include "fin_tail.h"

namespace segmi {

struct ConvParams {
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
  // optional input transform (segmi_in_affine): the producer's BatchNorm-apply + PReLU is applied
  // while the input is staged; kernels that do not implement it must not be dispatched with it
  const float* in_scale;
  const float* in_shift;
  const float* in_alpha;
  // PReLU applies to output-channel tiles < act_tiles only (0 = all): segmi_conv3d_fwd_split_act runs
  // two convolutions of one input as ONE launch with 2c outputs of which the first c are activated.
  // Honoured by the tile kernel (conv_fwd_mfma_kernel) only.  In TRAINING the same pairing carries the
  // BatchNorm statistics of the first convolution only: stats_tiles > 0 = statistics rows of 16 * stats_tiles
  // channels, written (and finalised) by the workgroups of those tiles; bias2 = the second convolution's bias
  // (tiles >= act_tiles read bias2[(tile - act_tiles) * 16 ...]: the two biases are separate arena slots).
  int act_tiles;
  int stats_tiles;
  const float* bias2;
  // optional BatchNorm-backward sums in the epilogue (segmi_bn_bwd_sums; ring kernel, MODE 4): the
  // launch is an input-gradient convolution whose OUTPUT is the gradient g flowing into a training-
  // mode BatchNorm + PReLU; with that layer's forward input bx the epilogue accumulates the three
  // per-channel sums of bn_act_bwd_reduce (sum dz, sum dz*xhat, sum g*z[z<=0]) into bpart[rows][3][c]
  const void* bx;
  int ldbx;
  const float* bmean; const float* binvstd; const float* bgamma; const float* bbeta; const float* balpha;
  float* bpart;
  // finalisation inside this launch (fin_tail.h): the workgroup that finishes last folds `stats`
  // (BnFin: segmi_bn_fin) or `bpart` (BnBwdFin: segmi_bn_bwd_sums.fin); fin_on is set by the C entry
  // point, ft by the launcher (which knows the grid)
  int fin_on;
  FinTail ft;
  BnFin bfin;
  BnBwdFin bbfin;
  int xcd;   // ring2: 1 = XCD-aware blockIdx -> column map (grid.x % 8 == 0)
  int dbg;   // diagnostics only (SEGMI_RING2_DBG): 1 = no staging loads, 2 = no stores, 4 = no MFMA loop
};

template <typename T, int CK, int KS, int S, int TD, int TH, int TW>
struct ConvGeom {
  static constexpr int KG = Elem<T>::KG;
  static constexpr int SPT = CK / KG;
  static constexpr int NTAPS = KS * KS * KS;
  static constexpr int NSLOT = NTAPS * SPT;
  static constexpr int NSTEP = (NSLOT + 3) / 4;
  static constexpr int PAD = (KS - 1) / 2;
  static constexpr int HD = (TD - 1) * S + KS;
  static constexpr int HH = (TH - 1) * S + KS;
  static constexpr int HW = (TW - 1) * S + KS;
  static constexpr int RAWB = CK * (int)sizeof(T);
  static constexpr int ROWB = RAWB == 32 ? 32 : RAWB + 16;
  static constexpr int CPR = RAWB / 16;  // 16-byte chunks per row
  static constexpr int NVT = TD * TH * TW / 16;
  static constexpr int VTW = NVT / 4;
  static constexpr int LDS_BYTES = HD * HH * HW * ROWB;
  static_assert(NVT % 4 == 0, "tile must give each wave whole voxel tiles");
  static_assert(SPT == 2 || SPT == 4 || SPT == 8, "unsupported chunk width");
};

template <typename T, int CK, int KS, int S, int NT, int TD, int TH, int TW>
__global__ __launch_bounds__(256) void conv_fwd_mfma_kernel(ConvParams p) {
  using G = ConvGeom<T, CK, KS, S, TD, TH, TW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;

  int t = blockIdx.x;
  const int txi = t % p.tx; t /= p.tx;
  const int tyi = t % p.ty; t /= p.ty;
  const int tzi = t % p.tz;
  const int n = t / p.tz;
  const int oz0 = tzi * TD, oy0 = tyi * TH, ox0 = txi * TW;
  const int iz0 = oz0 * S - G::PAD, iy0 = oy0 * S - G::PAD, ix0 = ox0 * S - G::PAD;
  const int nt0 = blockIdx.y * NT;

  f32x4 acc[G::VTW][NT];
#pragma unroll
  for (int i = 0; i < G::VTW; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int vaddr[G::VTW];
#pragma unroll
  for (int i = 0; i < G::VTW; ++i) {
    const int idx = (wave * G::VTW + i) * 16 + r;
    const int x = idx % TW, y = (idx / TW) % TH, z = idx / (TW * TH);
    vaddr[i] = ((z * S * G::HH + y * S) * G::HW + x * S) * G::ROWB;
  }

  // Weight fragments come from L2 (1 KiB coalesced per fragment).  When the whole chunk's set
  // fits in <= 64 VGPRs it is fetched BEFORE the halo staging so its latency hides under the
  // staging loads; otherwise the next k-step's fragments are prefetched one step ahead.
  constexpr bool PRE = G::NSTEP * NT <= 16;
  // otherwise a queue of WD k-steps of fragments runs ahead of the MFMAs: between two uses of a
  // layer its weights leave the L2s (GBs of activations stream through), so every fetch pays the
  // HBM latency and a one-step-ahead prefetch serialises NSTEP of them (27 x ~1.5 us measured)
  constexpr int WD = NT <= 2 ? (G::NSTEP < 4 ? G::NSTEP : 4) : 2;
  // Multi-chunk layers (Cin > CK): the halo tile of chunk c+1 is fetched into registers before
  // the MFMA phase of chunk c (register-staged pipeline) when the register budget allows.
  constexpr int NCH = G::HD * G::HH * G::HW * G::CPR;
  constexpr int NLD = (NCH + 255) / 256;
  constexpr bool PF = (G::VTW * NT * 4 + NLD * 4) <= 160;
  const char* inb = (const char*)p.in;
  frag_t stg[PF ? NLD : 1];
  auto fetch = [&](int c) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 256 * k;
      const int v = i / G::CPR, ch = i % G::CPR;
      const int hx = v % G::HW, hy = (v / G::HW) % G::HH, hz = v / (G::HW * G::HH);
      const int z = iz0 + hz, y = iy0 + hy, x = ix0 + hx;
      frag_t val = frag_t{0u, 0u, 0u, 0u};
      if (i < NCH && (unsigned)z < (unsigned)p.Di && (unsigned)y < (unsigned)p.Hi &&
          (unsigned)x < (unsigned)p.Wi) {
        const int64_t e = ((((int64_t)n * p.Di + z) * p.Hi + y) * p.Wi + x) * p.ldi + c * CK;
        val = *reinterpret_cast<const frag_t*>(inb + e * (int64_t)sizeof(T) + ch * 16);
      }
      if constexpr (PF) stg[k] = val;
      else if (i < NCH) *reinterpret_cast<frag_t*>(smem + v * G::ROWB + ch * 16) = val;
    }
  };
  if constexpr (PF) fetch(0);
  for (int c = 0; c < p.nchunks; ++c) {
    if (c > 0) __syncthreads();
    const char* wb = (const char*)p.wfrag +
                     (((int64_t)c * G::NSTEP) * p.ntiles_total + nt0) * 1024 + lane * 16;
    frag_t wall[PRE ? G::NSTEP : 1][NT];
    frag_t wq[PRE ? 1 : WD][NT];
    if constexpr (PRE) {
#pragma unroll
      for (int s = 0; s < G::NSTEP; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          wall[s][j] = *reinterpret_cast<const frag_t*>(wb + ((int64_t)s * p.ntiles_total + j) * 1024);
    } else {   // first WD k-steps: in flight under the halo commit and the barrier
#pragma unroll
      for (int s = 0; s < WD; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          wq[s][j] = *reinterpret_cast<const frag_t*>(wb + ((int64_t)s * p.ntiles_total + j) * 1024);
    }
    // ---- halo tile of chunk c -> LDS
    if constexpr (PF) {
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int i = tid + 256 * k;
        if (i < NCH) *reinterpret_cast<frag_t*>(smem + (i / G::CPR) * G::ROWB + (i % G::CPR) * 16) = stg[k];
      }
    } else {
      fetch(c);
    }
    __syncthreads();
    if constexpr (PF) {
      if (c + 1 < p.nchunks) fetch(c + 1);
    }
#pragma unroll
    for (int s = 0; s < G::NSTEP; ++s) {
      int loff;
      if constexpr (G::SPT == 2) {
        const int t0 = 2 * s, t1 = 2 * s + 1;
        const int o0 = ((t0 / (KS * KS)) * G::HH + (t0 / KS) % KS) * G::HW + t0 % KS;
        const int o1 = t1 < G::NTAPS
                           ? ((t1 / (KS * KS)) * G::HH + (t1 / KS) % KS) * G::HW + t1 % KS
                           : 0;
        loff = ((g >> 1) ? o1 : o0) * G::ROWB + (g & 1) * 16;
      } else {
        const int tap = (4 * s) / G::SPT;
        const int sub0 = (4 * s) % G::SPT;
        const int o0 = ((tap / (KS * KS)) * G::HH + (tap / KS) % KS) * G::HW + tap % KS;
        loff = o0 * G::ROWB + (sub0 + g) * 16;
      }
      frag_t wf[NT];
      if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = wall[s][j];
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = wq[s % WD][j];
        if (s + WD < G::NSTEP) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
            wq[s % WD][j] = *reinterpret_cast<const frag_t*>(
                wb + ((int64_t)(s + WD) * p.ntiles_total + j) * 1024);
        }
      }
#pragma unroll
      for (int i = 0; i < G::VTW; ++i) {
        const frag_t a = *reinterpret_cast<const frag_t*>(smem + vaddr[i] + loff);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma16<T>(wf[j], a, acc[i][j]);
      }
    }
  }

  // ---- epilogue: lane holds channels co = (nt0+j)*16 + 4g + {0..3} of voxel r of each tile
  f32x4 bias4[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias2 && nt0 + j >= p.act_tiles)
      bias4[j] = *reinterpret_cast<const f32x4*>(p.bias2 + (nt0 + j - p.act_tiles) * 16 + 4 * g);
    else if (p.bias) bias4[j] = *reinterpret_cast<const f32x4*>(p.bias + (nt0 + j) * 16 + 4 * g);
  }
  const bool has_alpha = p.alpha != nullptr;
  const float alpha = has_alpha ? *p.alpha : 0.f;
  // (workgroup-uniform) statistics rows: all channel tiles, or the first stats_tiles of a training pair
  const bool do_stats = p.stats != nullptr && (p.stats_tiles == 0 || nt0 < p.stats_tiles);
  const int stats_c = p.stats_tiles > 0 ? 16 * p.stats_tiles : p.Cout;
  f32x4 ssum[NT], ssq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    ssum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ssq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  T* outp = (T*)p.out;
  const T* resp = (const T*)p.res;
  // All loads the epilogue needs are issued and completed up front (see touch_v in common.h):
  // a load -> add -> store chain per tile would put s_waitcnt vmcnt(0) after every store.
  f32x4 resv[G::VTW][NT];
  if (resp) {
#pragma unroll
    for (int i = 0; i < G::VTW; ++i) {
      const int idx = (wave * G::VTW + i) * 16 + r;
      const int oz = oz0 + idx / (TW * TH), oy = oy0 + (idx / TW) % TH, ox = ox0 + idx % TW;
      const bool valid = oz < p.Do && oy < p.Ho && ox < p.Wo;
      const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        resv[i][j] = valid ? load4<T>(resp + vox * p.ldr + (nt0 + j) * 16 + 4 * g)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < G::VTW; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) touch_v(resv[i][j]);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) touch_v(bias4[j]);
  if constexpr (PF) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) touch_v(stg[k]);
  }
#pragma unroll
  for (int i = 0; i < G::VTW; ++i) {
    const int idx = (wave * G::VTW + i) * 16 + r;
    const int oz = oz0 + idx / (TW * TH), oy = oy0 + (idx / TW) % TH, ox = ox0 + idx % TW;
    const bool valid = oz < p.Do && oy < p.Ho && ox < p.Wo;
    const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f32x4 v = acc[i][j] + bias4[j];
      if (valid) {
        if (do_stats && (p.stats_tiles == 0 || nt0 + j < p.stats_tiles)) {
          ssum[j] += v;
          ssq[j] += v * v;
        }
        if (has_alpha && (p.act_tiles == 0 || nt0 + j < p.act_tiles)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
        }
        const int co = (nt0 + j) * 16 + 4 * g;
        if (resp) v += resv[i][j];
        store4<T>(outp + vox * p.ldo + co, v);
      }
    }
  }
  if (do_stats) {
    __syncthreads();  // everyone is done with the staged tile
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
      if (nt0 * 16 + ch < stats_c)
        fin_store(&p.stats[((int64_t)blockIdx.x * 2 + which) * stats_c + nt0 * 16 + ch], sacc);
    }
    fin_tail_run<BnFin, 256, offsetof(ConvParams, ft), offsetof(ConvParams, bfin)>(p.stats, smem);
  }
}

template <typename T, int CK, int KS, int S, int NT, int TD, int TH, int TW>
static int launch_conv_cfg(ConvParams p, hipStream_t st) {
  using G = ConvGeom<T, CK, KS, S, TD, TH, TW>;
  p.tz = cdiv(p.Do, TD);
  p.ty = cdiv(p.Ho, TH);
  p.tx = cdiv(p.Wo, TW);
  const int64_t nb = (int64_t)p.N * p.tz * p.ty * p.tx;
  SEGMI_CHECK_ARG(nb < (1ll << 31), "conv3d: too many tiles");
  dim3 grid((unsigned)nb, (unsigned)(p.Cout / (16 * NT)));
  constexpr int lds0 = G::LDS_BYTES > 4 * 2 * NT * 16 * 4 ? G::LDS_BYTES : 4 * 2 * NT * 16 * 4;
  p.fin_on = p.fin_on && p.stats;
  // statistics of the first stats_tiles channel tiles only (training pair): those workgroups write the rows and
  // take the tickets of the finalisation
  const int stats_c = p.stats_tiles > 0 ? 16 * p.stats_tiles : p.Cout;
  const dim3 sgrid(grid.x, (unsigned)cdiv(stats_c / 16, NT));
  const int lds = (int)fin_tail_arm(p, sgrid, 256, 2 * stats_c, lds0);
  auto kern = conv_fwd_mfma_kernel<T, CK, KS, S, NT, TD, TH, TW>;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, 256, lds, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_fwd(mfma)");
  return SEGMI_OK;
}

// tile selection per (KS, S): big tile when the output row is wide, small otherwise
template <typename T, int CK, int KS, int S, int NT>
static int launch_conv_tiles(const ConvParams& p, hipStream_t st) {
  const bool wide = p.Wo > 8;
  if constexpr (S == 1) {
    if (wide) return launch_conv_cfg<T, CK, KS, S, NT, 4, 8, 16>(p, st);
    return launch_conv_cfg<T, CK, KS, S, NT, 4, 4, 8>(p, st);
  } else {
    if (wide) return launch_conv_cfg<T, CK, KS, S, NT, 2, 4, 16>(p, st);
    return launch_conv_cfg<T, CK, KS, S, NT, 2, 4, 8>(p, st);
  }
}

template <typename T, int CK, int KS, int S>
static int launch_conv_nt(const ConvParams& p, hipStream_t st) {
  const int nt = p.Cout / 16;
  // output-channel tiles per workgroup: as many as divide Cout (input tile reuse), but fewer
  // when the launch would not fill the chip (deep levels: 8^3 .. 16^3 voxels)
  const bool wide = p.Wo > 8;
  const int td = S == 1 ? 4 : 2, th = S == 1 ? (wide ? 8 : 4) : 4, tw = wide ? 16 : 8;
  const int64_t tiles = (int64_t)p.N * cdiv(p.Do, td) * cdiv(p.Ho, th) * cdiv(p.Wo, tw);
  int sel = nt % 4 == 0 ? 4 : (nt % 2 == 0 ? 2 : 1);
  while (sel > 1 && tiles * (nt / sel) < 512) sel /= 2;
  if (sel == 4) return launch_conv_tiles<T, CK, KS, S, 4>(p, st);
  if (sel == 2) return launch_conv_tiles<T, CK, KS, S, 2>(p, st);
  return launch_conv_tiles<T, CK, KS, S, 1>(p, st);
}

template <typename T>
static int launch_conv_mfma_t(const ConvParams& p, int ksize, int stride, hipStream_t st) {
  constexpr int dt = sizeof(T) == 4 ? SEGMI_F32 : SEGMI_BF16;
  const int ck = pick_ck(dt, p.Cin);
  if constexpr (sizeof(T) == 2) {
    if (ck == 32) {
      if (ksize == 3 && stride == 1) return launch_conv_nt<T, 32, 3, 1>(p, st);
      if (ksize == 3 && stride == 2) return launch_conv_nt<T, 32, 3, 2>(p, st);
      if (ksize == 1 && stride == 1) return launch_conv_nt<T, 32, 1, 1>(p, st);
    }
  }
  if (ksize == 3 && stride == 1) return launch_conv_nt<T, 16, 3, 1>(p, st);
  if (ksize == 3 && stride == 2) return launch_conv_nt<T, 16, 3, 2>(p, st);
  if (ksize == 1 && stride == 1) return launch_conv_nt<T, 16, 1, 1>(p, st);
  SEGMI_UNSUPPORTED("conv3d: unsupported ksize/stride %d/%d", ksize, stride);
}

// stats rows of the MFMA path (= spatial workgroups)
static inline int conv_mfma_rows(const segmi_act* out, int stride) {
  const bool wide = out->w > 8;
  int td, th, tw;
  if (stride == 1) { td = 4; th = wide ? 8 : 4; tw = wide ? 16 : 8; }
  else { td = 2; th = 4; tw = wide ? 16 : 8; }
  return out->n * cdiv(out->d, td) * cdiv(out->h, th) * cdiv(out->w, tw);
}

}  // namespace segmi
