// conv_bnbwd_impl.h -- BatchNorm(+PReLU) backward APPLY fused into the stride-2 convolution that consumes it.
//
// The input gradient of a decoder level's transposed convolution is a k3 s2 convolution of du, and du is
// the BatchNorm-backward apply of (dau, u):  du = gamma*invstd*(dz - c0 - xhat*c1)  (bn_act_bwd_apply_kernel).
// As two launches the 16-channel full-resolution tensor du is written (537 MB for 8 x 128^3), read by the
// convolution and read by the weight gradient.  Here the convolution computes du itself while it stages its
// halo tile (two 16-byte loads per chunk instead of one), feeds the bf16-rounded values to its MFMAs -- the
// very bits the separate pass would have stored -- and the workgroup that OWNS a voxel (every input voxel
// sits in exactly one tile's non-halo part: local index >= 1 on each axis) writes it to du once for the
// weight gradient: one read of du less (-0.44 GB per step at the top level) and one launch less on the
// dependent chain.  Tile geometry and MFMA loop are those of conv_fwd_mfma_kernel<bf16, 16, 3, 2, NT, 2, 4, 16>.
#pragma once
#include "conv_fwd_impl.h"

#include <type_traits>
#ifndef SEGMI_BNBWD_KB
#define SEGMI_BNBWD_KB 6
#endif

namespace segmi {

// LDS of the launch: the halo tile rounded up to whole staging passes (the last pass writes past the tile)
constexpr int kBnBwdLds = 12 * 128 * 32;

struct ConvBnBwdParams {
  const void* dy;      // gradient of the activation output, [N, Di, Hi, Wi, 16]
  const void* x;       // forward input of the BatchNorm, same shape
  void* dx;            // apply(dy, x): written once, by the owner of each voxel
  void* out;           // conv_k3s2(dx): [N, Do, Ho, Wo, Cout]
  const void* wfrag;
  const float* mean; const float* invstd; const float* gamma; const float* beta; const float* alpha;
  const float* coef;   // [2][16]: c0, c1 (bn_act_bwd_finalize)
  int N, Di, Hi, Wi, Do, Ho, Wo, Cout, ldy, ldx, lddx, ldo;
  int tz, ty, tx, ntiles_total;
};

template <int NT>
__global__ __launch_bounds__(256, 3) void conv_s2_bnbwd_kernel(ConvBnBwdParams p) {
  using T = bf16_t;
  constexpr int S = 2, KS = 3, TD = 2, TH = 4, TW = 16;
  using G = ConvGeom<T, 16, KS, S, TD, TH, TW>;
  static_assert(G::SPT == 2 && G::ROWB == 32 && G::CPR == 2, "16 bf16 channels per voxel row");
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
  const int iz0 = oz0 * S - 1, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

  // A WAVE stages one channel half (8 channels = one 16-byte chunk per voxel) of the voxels it visits, so
  // the per-channel constants are wave-uniform (SGPRs); as per-lane values they cost the staging 48 VGPRs,
  // i.e. the third workgroup per CU or half of the loads in flight.  Constants are paired for the packed
  // f32 operations; the two that would be a second scalar operand of one instruction (beta in
  // fma(xh, gamma, beta), gamma * invstd) are held in VGPR pairs.
  const int half = wave & 1;
  const int c8 = half * 8;
  f32x2 m2[4], i2[4], g2[4], b2[4], c02[4], c12[4], gi2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ch = c8 + 2 * q + h;
      m2[q][h] = p.mean[ch];
      i2[q][h] = p.invstd[ch];
      g2[q][h] = p.gamma ? p.gamma[ch] : 1.f;
      b2[q][h] = p.beta ? p.beta[ch] : 0.f;
      c02[q][h] = p.coef[ch];
      c12[q][h] = p.coef[16 + ch];
    }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    gi2[q] = g2[q] * i2[q];
    asm volatile("" : "+v"(gi2[q]), "+v"(b2[q]));
  }
  const bool has_alpha = p.alpha != nullptr;
  const float alpha = has_alpha ? *p.alpha : 1.f;

  // ---- halo tile: du = apply(dy, x) -> LDS (+ du to memory for the voxels this tile owns)
  constexpr int NVX = G::HD * G::HH * G::HW;            // voxels of the halo tile
  constexpr int NLD = (NVX + 127) / 128;                // 128 lanes (two waves) per channel half
  constexpr int KB = SEGMI_BNBWD_KB;                    // voxels per batch
  constexpr int NB = NLD / KB;
  static_assert(NLD % KB == 0 && NLD * 128 * G::ROWB <= kBnBwdLds, "batches cover the tile, LDS covers the batches");
  const int vbase = (wave >> 1) * 64 + lane;
  // One buffer resource per operand and sample (wave-uniform base, 32-bit per-lane byte offsets; the entry
  // point checks that a sample stays below 2^31 bytes).  Voxels outside the tensor get an out-of-range
  // offset: the load returns zeros, the store is dropped -- no exec-mask branches in the staging.
  const int64_t svox = (int64_t)p.Di * p.Hi * p.Wi;
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.dy + (n * svox * p.ldy + c8) * 2), 0, (int)(svox * p.ldy * 2 - c8 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)p.x + (n * svox * p.ldx + c8) * 2), 0, (int)(svox * p.ldx * 2 - c8 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((char*)p.dx + (n * svox * p.lddx + c8) * 2), 0, (int)(svox * p.lddx * 2 - c8 * 2), 0x00020000);
  constexpr int kOob = (int)0x80000000;
  // batches are double-buffered: the loads of batch b + 1 are issued before batch b is transformed
  frag_t dzr[2][KB], xr[2][KB];
  int code[2][KB];          // per staged voxel: < 0 = outside the tensor, else 2 * (voxel index in the sample) + owned
  auto issue = [&](int b, frag_t (&dz)[KB], frag_t (&xx)[KB], int (&cd)[KB]) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int v = vbase + 128 * (b * KB + k);
      const int hx = v % G::HW, hy = (v / G::HW) % G::HH, hz = v / (G::HW * G::HH);
      const int z = iz0 + hz, y = iy0 + hy, x = ix0 + hx;
      // (bitwise, not short-circuit: "&&" chains compile to exec-mask branches)
      const bool inside = (v < NVX) & ((unsigned)z < (unsigned)p.Di) & ((unsigned)y < (unsigned)p.Hi) &
                          ((unsigned)x < (unsigned)p.Wi);
      const int vox = (z * p.Hi + y) * p.Wi + x;
      cd[k] = inside ? 2 * vox + (int)((hz >= 1) & (hy >= 1) & (hx >= 1)) : -1;
      dz[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, inside ? vox * p.ldy * 2 : kOob, 0, 0);
      xx[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, inside ? vox * p.ldx * 2 : kOob, 0, 0);
    }
  };
  auto commit = [&](int b, frag_t (&dz)[KB], frag_t (&xx)[KB], int (&cd)[KB], auto ha) {
    constexpr bool HA = decltype(ha)::value;
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int v = vbase + 128 * (b * KB + k);
      frag_t val;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 a{__uint_as_float(xx[k][q] << 16), __uint_as_float(xx[k][q] & 0xffff0000u)};
        const f32x2 d{__uint_as_float(dz[k][q] << 16), __uint_as_float(dz[k][q] & 0xffff0000u)};
        const f32x2 o2 = bn_bwd_apply_elem2g<HA>(a, d, m2[q], i2[q], g2[q], b2[q], c02[q], c12[q], gi2[q], alpha);
        val[q] = cd[k] >= 0 ? pack_bf16x2(o2[0], o2[1]) : 0u;      // zero padding stays zero
      }
      __builtin_amdgcn_raw_buffer_store_b128(val, rs_dx, ((cd[k] >= 0) & ((cd[k] & 1) != 0)) ? (cd[k] >> 1) * p.lddx * 2 : kOob, 0, 0);
      *reinterpret_cast<frag_t*>(smem + v * G::ROWB + half * 16) = val;
    }
  };
  auto stage = [&](auto ha) {
    issue(0, dzr[0], xr[0], code[0]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) issue(b + 1, dzr[(b + 1) & 1], xr[(b + 1) & 1], code[(b + 1) & 1]);
      commit(b, dzr[b & 1], xr[b & 1], code[b & 1], ha);
    }
  };
  if (has_alpha) stage(std::true_type{});
  else stage(std::false_type{});
  // first weight k-steps (L2-resident, shared by every workgroup): issued only now -- held across the
  // staging they would cost it 32 registers, and the staging wants the occupancy (3 workgroups per CU)
  constexpr int WD = NT <= 2 ? 4 : 2;
  const char* wb = (const char*)p.wfrag + lane * 16;
  frag_t wq[WD][NT];
#pragma unroll
  for (int s = 0; s < WD; ++s)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      wq[s][j] = *reinterpret_cast<const frag_t*>(wb + ((int64_t)s * p.ntiles_total + j) * 1024);
  __syncthreads();

  // ---- MFMA loop (as conv_fwd_mfma_kernel, SPT == 2: two taps per k-step)
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
#pragma unroll
  for (int s = 0; s < G::NSTEP; ++s) {
    const int t0 = 2 * s, t1 = 2 * s + 1;
    const int o0 = ((t0 / (KS * KS)) * G::HH + (t0 / KS) % KS) * G::HW + t0 % KS;
    const int o1 = t1 < G::NTAPS ? ((t1 / (KS * KS)) * G::HH + (t1 / KS) % KS) * G::HW + t1 % KS : 0;
    const int loff = ((g >> 1) ? o1 : o0) * G::ROWB + (g & 1) * 16;
    frag_t wf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = wq[s % WD][j];
    if (s + WD < G::NSTEP) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wq[s % WD][j] = *reinterpret_cast<const frag_t*>(wb + ((int64_t)(s + WD) * p.ntiles_total + j) * 1024);
    }
#pragma unroll
    for (int i = 0; i < G::VTW; ++i) {
      const frag_t a = *reinterpret_cast<const frag_t*>(smem + vaddr[i] + loff);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = mma16<T>(wf[j], a, acc[i][j]);
    }
  }

  // ---- epilogue: lane holds channels j*16 + 4g + {0..3} of voxel r of each tile
  T* outp = (T*)p.out;
#pragma unroll
  for (int i = 0; i < G::VTW; ++i) {
    const int idx = (wave * G::VTW + i) * 16 + r;
    const int oz = oz0 + idx / (TW * TH), oy = oy0 + (idx / TW) % TH, ox = ox0 + idx % TW;
    if (oz < p.Do && oy < p.Ho && ox < p.Wo) {
      const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
      for (int j = 0; j < NT; ++j) store4<T>(outp + vox * p.ldo + j * 16 + 4 * g, acc[i][j]);
    }
  }
}

template <int NT>
static int launch_conv_s2_bnbwd(ConvBnBwdParams p, hipStream_t st) {
  using G = ConvGeom<bf16_t, 16, 3, 2, 2, 4, 16>;
  p.tz = cdiv(p.Do, 2);
  p.ty = cdiv(p.Ho, 4);
  p.tx = cdiv(p.Wo, 16);
  const int64_t nb = (int64_t)p.N * p.tz * p.ty * p.tx;
  SEGMI_CHECK_ARG(nb < (1ll << 31), "bn_act_bwd_apply_conv: too many tiles");
  static_assert(G::LDS_BYTES <= kBnBwdLds, "halo tile fits");
  hipLaunchKernelGGL(conv_s2_bnbwd_kernel<NT>, dim3((unsigned)nb), 256, kBnBwdLds, st, p);
  SEGMI_LAUNCH_CHECK("bn_act_bwd_apply_conv");
  return SEGMI_OK;
}

}  // namespace segmi
