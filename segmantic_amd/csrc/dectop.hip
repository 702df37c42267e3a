// dectop.hip -- inference: the full-resolution decoder of the UNet as ONE persistent launch.
//
//   h   = PReLU( ConvTranspose3d(k3, s2, p1, op1; 32 -> 16)(x) * bn_scale + bn_shift )     (BatchNorm folded)
//   out = Conv3d(k3, s1; 16 -> 16)(h) + bias + h                                             (conv-only ResidualUnit)
//
// replaces torch.nn.ConvTranspose3d + ADN + the last ResidualUnit of monai UNet (reference
// src/segmantic/seg/monai_unet.py:114-124, run per window at :354-356 / :665).  As two launches
// (convt_ps_kernel, conv_ring2_kernel) the 16-channel full-resolution tensor h makes a round trip
// through HBM: per 128^3 patch the pair moves 17 + 67 | 67 + 67 = 218 MB; fused it moves 17 + 67.
//
// Structure = the z-marching ring of conv_ring2_impl.h, with the ring planes PRODUCED instead of
// loaded, and the two jobs on different waves (as in wgrad_ws_impl.h): of the 8 waves of a 512-thread
// workgroup (one per CU: 148 KB of LDS) waves 0-3 run the convolution of step s -- 4 output rows x 4
// planes each -- while waves 4-7 produce the ring planes of step s + 1, so that the producers'
// LDS traffic, VALU epilogues and barriers hide under the convolution's MFMAs (with all 8 waves doing
// both jobs in lockstep phases the launch was no faster than the two it replaces: 64 vs 60 us per
// 128^3 patch, 48 % of it outside any MFMA or memory phase).  The workgroup owns a
// 16 x 16 column of output voxels and marches along z, 4 planes per step:
//   * ring of 10 planes of h over the (16+2) x (16+2) footprint of the column (bf16 rows of 32 B);
//   * a ring of 3 coarse input planes (10 x 10 voxels x 32 channels): every step needs the coarse
//     planes {(a-1)/2, (a+1)/2, (a+3)/2} of its first new fine plane a, two of them new;
//   * the 27 transposed-conv weight fragments (A operands, 27 KB) in LDS, the 15 conv weight
//     fragments in registers;
//   * per step, consumers: conv MFMA loop on ring planes zb-1 .. zb+4 -> bias + residual -> stores;
//     producers: coarse planes (loaded one step ahead) to LDS | barrier | transposed-conv MFMAs for
//     fine planes zb+5 .. zb+8 over the footprint, bias + PReLU + zero padding, 8-byte rows into the
//     free ring slots | barrier.
// Transposed-conv tiles are 16 coarse x of one (fine plane, fine row, x-parity): 9 of the 16 lanes
// carry footprint voxels (the simple addressing costs 486 instead of ~324 MFMAs per step; the
// launch stays memory-bound).  Tap order and k-slot layout are those of convt_ps_kernel, the conv
// loop is conv_ring2_kernel's: the result is bit-identical to the two-launch path.
#include <type_traits>
#include "conv_ring_impl.h"

namespace segmi {

struct DecTopParams {
  const void* in;       // [N, Dc, Hc, Wc, 32] bf16
  void* out;            // [N, 2Dc, 2Hc, 2Wc, 16] bf16
  const void* up_frag;  // [27][64][16 B]: tap (kd*3+kh)*3+kw, lane (g, co): W_T[ci = 8g .. 8g+7][co][tap] * bn_scale[co]
  const float* up_bias; // folded: b * bn_scale + bn_shift
  const float* up_alpha;
  const void* cv_frag;  // kind-0 fragment pack of the 16 -> 16 conv
  const float* cv_bias;
  int N, Dc, Hc, Wc, Do, Ho, Wo, ldi, ldo;
  int ty, tx, tz;
  int dbg;   // diag build only (SEGMI_DECTOP_DBG): 1 = no producer MFMAs, 2 = no conv loop, 4 = no output stores, 8 = no coarse loads
};

#ifdef SEGMI_RING2_DIAG
#define DT_DBG(p, bit) (((p).dbg & (bit)) != 0)
#else
#define DT_DBG(p, bit) false
#endif

namespace dectop {
constexpr int TD = 4, TH = 16, TW = 16, HH = TH + 2, HW = TW + 2, R = 10;
constexpr int ROWB = 32, PLANE_B = HH * HW * ROWB, RING_B = R * PLANE_B;           // 103,680
constexpr int CR = 3, CH = TH / 2 + 2, CW = TW / 2 + 2, CROWB = 64;
constexpr int CPLANE_B = CH * CW * CROWB, CAT_B = CR * CPLANE_B + 640;              // + overrun of lanes 10..15
constexpr int WUP_B = 27 * 1024;
constexpr int OFF_CAT = RING_B, OFF_WUP = OFF_CAT + CAT_B;
constexpr int LDS_BYTES = OFF_WUP + WUP_B;                                          // 151,168
constexpr int J = 5, NIT = 6 * J, PD = 2;
constexpr int NRO = 4;                      // output rows per consumer wave
constexpr int NLP = 4;                      // chunks per producer thread of a 2-plane coarse stage (800 / 256)
constexpr int NLC = 3;                      // 16-byte chunks per thread of a 3-plane coarse stage (1200 / 512)
constexpr int PLC = CH * CW * 4;            // chunks per coarse plane
}  // namespace dectop

__global__ __launch_bounds__(512, 1) void dectop_kernel(DecTopParams p) {
  using namespace dectop;
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const catl = smem + OFF_CAT;
  char* const wupl = smem + OFF_WUP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;

  int t = blockIdx.x;
  const int seg = t % p.tz; t /= p.tz;
  const int txi = t % p.tx; t /= p.tx;
  const int tyi = t % p.ty;
  const int n = t / p.ty;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const int Y0 = oy0 >> 1, X0 = ox0 >> 1;
  const int total_steps = p.Do / TD;
  const int seg_steps = (total_steps + p.tz - 1) / p.tz;
  const int z0 = seg * seg_steps * TD;
  const int nsteps_z = total_steps - seg * seg_steps < seg_steps ? total_steps - seg * seg_steps : seg_steps;
  if (nsteps_z <= 0) return;                 // workgroup-uniform

  // ---- transposed-conv weights -> LDS
  for (int i = tid; i < WUP_B / 16; i += 512)
    reinterpret_cast<frag_t*>(wupl)[i] = reinterpret_cast<const frag_t*>(p.up_frag)[i];
  // ---- conv weights -> registers (two taps per k-step, gathered from the standard pack; ring2's layout)
  frag_t wreg[3][J];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      wreg[kd][j] = frag_t{0u, 0u, 0u, 0u};
      const int t9 = 2 * j + (g >> 1);
      if (t9 <= 8) {
        const int tap = kd * 9 + t9;
        const int sp = tap >> 1, gp = (tap & 1) * 2 + (g & 1);
        wreg[kd][j] = *reinterpret_cast<const frag_t*>((const char*)p.cv_frag + (((int64_t)sp) * 64 + gp * 16 + r) * 16);
      }
    }
  int laneoff[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    int t9 = 2 * j + (g >> 1);
    if (t9 > 8) t9 = 8;
    laneoff[j] = ((t9 / 3) * HW + t9 % 3 + r) * ROWB + (g & 1) * 16;
  }
  const int wrow = (wave & 3) * NRO * HW * ROWB;   // consumer wave w: output rows 4w .. 4w+3 (footprint row 4w + kh)

  // ---- coarse staging descriptors (slot k: chunk tid + 512 k of a <= 3-plane stage)
  const char* inb = (const char*)p.in;
  const int64_t cplane_stride = (int64_t)p.Hc * p.Wc * p.ldi * 2;
  const char* img = inb + (int64_t)n * p.Dc * cplane_stride;
  int c_pl[NLC], c_goff[NLC], c_loff[NLC];
#pragma unroll
  for (int k = 0; k < NLC; ++k) {
    const int i = tid + 512 * k;
    const int pl = i / PLC, idx = i % PLC;
    const int vox = idx >> 2, ch = idx & 3;
    const int lr = vox / CW, lc = vox % CW;
    const int cy = Y0 - 1 + lr, cx = X0 - 1 + lc;
    const bool ok = (unsigned)cy < (unsigned)p.Hc && (unsigned)cx < (unsigned)p.Wc;
    c_pl[k] = pl < CR ? pl : -1;
    c_goff[k] = ok ? (cy * p.Wc + cx) * p.ldi * 2 + ch * 16 : -1;
    c_loff[k] = (lr * CW + lc) * CROWB + ch * 16;
  }
  frag_t creg[NLC];
  auto cat_fetch = [&](int c_first, int nplanes) {      // coarse planes c_first .. c_first + nplanes - 1 -> registers
#pragma unroll
    for (int k = 0; k < NLC; ++k) {
      creg[k] = frag_t{0u, 0u, 0u, 0u};
      const int cz = c_first + c_pl[k];
      if (c_pl[k] >= 0 && c_pl[k] < nplanes && (unsigned)cz < (unsigned)p.Dc && c_goff[k] >= 0 && !DT_DBG(p, 8))
        creg[k] = *reinterpret_cast<const frag_t*>(img + (int64_t)cz * cplane_stride + (unsigned)c_goff[k]);
    }
  };
  auto cat_commit = [&](int c_first, int nplanes) {     // ... -> their ring slots (zeros outside the volume)
#pragma unroll
    for (int k = 0; k < NLC; ++k) {
      if (c_pl[k] >= 0 && c_pl[k] < nplanes) {
        const int cz = c_first + c_pl[k];
        const int slot = ((cz % CR) + CR) % CR;
        *reinterpret_cast<frag_t*>(catl + slot * CPLANE_B + c_loff[k]) = creg[k];
      }
    }
  };

  // the same for the producers alone in the steady state: 2 new planes, 256 threads
  const int ptid = tid & 255;
  int d_pl[NLP], d_goff[NLP], d_loff[NLP];
#pragma unroll
  for (int k = 0; k < NLP; ++k) {
    const int i = ptid + 256 * k;
    const int pl = i / PLC, idx = i % PLC;
    const int vox = idx >> 2, ch = idx & 3;
    const int lr = vox / CW, lc = vox % CW;
    const int cy = Y0 - 1 + lr, cx = X0 - 1 + lc;
    const bool ok = (unsigned)cy < (unsigned)p.Hc && (unsigned)cx < (unsigned)p.Wc;
    d_pl[k] = pl < 2 ? pl : -1;
    d_goff[k] = ok ? (cy * p.Wc + cx) * p.ldi * 2 + ch * 16 : -1;
    d_loff[k] = (lr * CW + lc) * CROWB + ch * 16;
  }
  frag_t dreg[NLP];
  auto cat_fetch2 = [&](int c_first) {
#pragma unroll
    for (int k = 0; k < NLP; ++k) {
      dreg[k] = frag_t{0u, 0u, 0u, 0u};
      const int cz = c_first + d_pl[k];
      if (d_pl[k] >= 0 && (unsigned)cz < (unsigned)p.Dc && d_goff[k] >= 0 && !DT_DBG(p, 8))
        dreg[k] = *reinterpret_cast<const frag_t*>(img + (int64_t)cz * cplane_stride + (unsigned)d_goff[k]);
    }
  };
  auto cat_commit2 = [&](int c_first) {
#pragma unroll
    for (int k = 0; k < NLP; ++k) {
      if (d_pl[k] >= 0) {
        const int cz = c_first + d_pl[k];
        const int slot = ((cz % CR) + CR) % CR;
        *reinterpret_cast<frag_t*>(catl + slot * CPLANE_B + d_loff[k]) = dreg[k];
      }
    }
  };

  // ---- producer of the ring planes: h = PReLU(convT(x) + bias) over the footprint, zero outside the volume
  f32x4 ubias = *reinterpret_cast<const f32x4*>(p.up_bias + 4 * g);
  float ualpha = *p.up_alpha;
  touch_v(ubias);
  touch_s(ualpha);
  const int lane_b = r * CROWB + g * 16;              // coarse voxel r of a row, channel chunk g
  const int lane_a = lane * 16;                       // A fragment of a tap
  const int lane_w = (2 * r) * ROWB + 8 * g;          // ring row: footprint column 2r (+1), channels 4g ..
  // fine plane fz, footprint row j: taps of one parity combination (NZ z-taps x NY y-taps), both x parities
  // (an emit is the producers' unit of overhead: 144 per step and workgroup, so it is branch-free: PReLU as
  // fma(alpha, min(v, 0), max(v, 0)) -- the same bits as the select form for any slope -- and the x-range
  // test as a per-lane AND mask computed once)
  const unsigned xmask0 = (unsigned)(ox0 + 2 * r) < (unsigned)p.Wo ? 0xffffffffu : 0u;       // px = 0: fine x = ox0 + 2r
  const unsigned xmask1 = (unsigned)(ox0 + 2 * r - 1) < (unsigned)p.Wo ? 0xffffffffu : 0u;   // px = 1: fine x = ox0 - 1 + 2r
  auto emit = [&](char* dst, f32x4 acc, int px, bool row_in) {
    // px = 0: footprint column 2r + 1; px = 1: column 2r
    u32x2 o = u32x2{0u, 0u};
    if (row_in) {
      f32x4 v = acc + ubias;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(ualpha, fminf(v[e], 0.f), fmaxf(v[e], 0.f));
      const unsigned m = px ? xmask1 : xmask0;
      o[0] = pack_bf16x2(v[0], v[1]) & m;
      o[1] = pack_bf16x2(v[2], v[3]) & m;
    }
    if (r <= TW / 2) *reinterpret_cast<u32x2*>(dst + lane_w + (1 - px) * ROWB) = o;
  };
  // One fine plane, the rows j0, j0 + 8, j0 + 16 of this wave (all of one y-parity: NY y-taps), NZ
  // z-taps: the NZ * NY * 3 weight fragments are read once, the coarse-voxel fragments of the next row
  // are in flight while the current row is multiplied (left as dependent read -> MFMA chains in runtime
  // loops the phase ran at the LDS latency: 70 us per 128^3 patch for the whole launch instead of 25).
  auto produce_plane = [&](auto nz_c, auto ny_c, char* plane, int slotA, int slotB, int j0, bool z_in) {
    constexpr int NZ = decltype(nz_c)::value, NY = decltype(ny_c)::value;
    frag_t Af[NZ][NY][3];
#pragma unroll
    for (int zt = 0; zt < NZ; ++zt)
#pragma unroll
      for (int yt = 0; yt < NY; ++yt) {
        // convt_ps tap order: odd parity (k = 2 on the nearer coarse voxel) then (k = 0); even: k = 1
        const int kd = NZ == 2 ? (zt == 0 ? 2 : 0) : 1, kh = NY == 2 ? (yt == 0 ? 2 : 0) : 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          Af[zt][yt][kw] = *reinterpret_cast<const frag_t*>(wupl + ((kd * 3 + kh) * 3 + kw) * 1024 + lane_a);
      }
    frag_t Bf[2][NZ][NY][2];
    auto load_b = [&](int j, frag_t (&bq)[NZ][NY][2]) {
      // odd fine row (j even): coarse rows j/2 (k = 2), j/2 + 1 (k = 0); even fine row: (j + 1)/2 (k = 1)
      const int lr0 = NY == 2 ? j >> 1 : (j + 1) >> 1;
#pragma unroll
      for (int zt = 0; zt < NZ; ++zt)
#pragma unroll
        for (int yt = 0; yt < NY; ++yt) {
          const char* brow = catl + (zt == 0 ? slotA : slotB) + (lr0 + yt) * CW * CROWB + lane_b;
          bq[zt][yt][0] = *reinterpret_cast<const frag_t*>(brow);            // coarse column r     (local)
          bq[zt][yt][1] = *reinterpret_cast<const frag_t*>(brow + CROWB);    // coarse column r + 1
        }
    };
    load_b(j0, Bf[0]);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int j = j0 + 8 * it;
      if (j >= HH) break;                                   // wave-uniform
      if (j + 8 < HH) load_b(j + 8, Bf[(it + 1) & 1]);
      const int fy = oy0 - 1 + j;
      const bool row_in = z_in && (unsigned)fy < (unsigned)p.Ho;
      char* const dst = plane + j * HW * ROWB;
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      if (row_in && !DT_DBG(p, 1)) {
#pragma unroll
        for (int zt = 0; zt < NZ; ++zt)
#pragma unroll
          for (int yt = 0; yt < NY; ++yt) {
            const frag_t b0 = Bf[it & 1][zt][yt][0], b1 = Bf[it & 1][zt][yt][1];
            // even fine x = 2 (X0 + r): kw = 1 on coarse X0 + r = local column r + 1
            acc0 = mma16<T>(Af[zt][yt][1], b1, acc0);
            // odd fine x = 2 (X0 - 1 + r) + 1: (kw = 2, local column r) then (kw = 0, local column r + 1)
            acc1 = mma16<T>(Af[zt][yt][2], b0, acc1);
            acc1 = mma16<T>(Af[zt][yt][0], b1, acc1);
          }
      }
      emit(dst, acc0, 0, row_in);
      emit(dst, acc1, 1, row_in);
    }
  };
  // `vw` = 0 .. 7: which eighth of the (plane, row) units (prologue: the wave itself; steady state: a
  // producer wave takes two)
  auto produce = [&](int fz_first, int count, int vw) {
    using One = std::integral_constant<int, 1>;
    using Two = std::integral_constant<int, 2>;
    for (int pl = 0; pl < count; ++pl) {
      const int fz = fz_first + pl;
      char* const plane = smem + ((fz - z0 + 1) % R) * PLANE_B;
      const bool z_in = (unsigned)fz < (unsigned)p.Do;
      const bool zodd = (fz & 1) != 0;
      // odd plane: (kd = 2, coarse (fz-1)/2) then (kd = 0, coarse (fz+1)/2); even plane: kd = 1, coarse fz/2
      const int czA = zodd ? (fz - 1) >> 1 : fz >> 1, czB = (fz + 1) >> 1;
      const int slotA = (((czA % CR) + CR) % CR) * CPLANE_B, slotB = (((czB % CR) + CR) % CR) * CPLANE_B;
      const int j0 = (((vw - 3 * pl) % 8) + 8) % 8;       // rows j0, j0 + 8, (j0 + 16): balanced over the eighths
      const bool yodd = (j0 & 1) == 0;                     // footprint row even <=> fine y odd
      if (zodd) {
        if (yodd) produce_plane(Two{}, Two{}, plane, slotA, slotB, j0, z_in);
        else produce_plane(Two{}, One{}, plane, slotA, slotB, j0, z_in);
      } else {
        if (yodd) produce_plane(One{}, Two{}, plane, slotA, slotB, j0, z_in);
        else produce_plane(One{}, One{}, plane, slotA, slotB, j0, z_in);
      }
    }
  };

  // ---- prologue (all 8 waves): ring planes z0-1 .. z0+4 (coarse planes z0/2-1 .. z0/2+2, three at a time)
  const int cz0 = z0 >> 1;
  cat_fetch(cz0 - 1, 3);
  cat_commit(cz0 - 1, 3);
  __syncthreads();
  produce(z0 - 1, 3, wave);
  __syncthreads();
  cat_fetch(cz0 + 2, 1);
  cat_commit(cz0 + 2, 1);
  __syncthreads();
  produce(z0 + 2, 3, wave);
  __syncthreads();

  auto ws_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  if (wave >= 4) {
    // ======================================================= producers: ring planes of step s + 1
    const int pw = wave - 4;
    if (nsteps_z > 1) cat_fetch2((z0 + 5 + 1) >> 1);
    for (int step = 0; step < nsteps_z; ++step) {
      const bool more = step + 1 < nsteps_z;
      const int a = z0 + step * TD + 5;                 // first new fine plane (odd)
      if (more) cat_commit2((a + 1) >> 1);
      ws_barrier();                                    // X: the coarse planes are in LDS
      if (more) {
        if (step + 2 < nsteps_z) cat_fetch2((a + 4 + 1) >> 1);
        for (int vw = 2 * pw; vw < 2 * pw + 2; ++vw) produce(a, 4, vw);
      }
      ws_barrier();                                    // Y: ring planes of step s + 1 complete, step s consumed
    }
    return;
  }

  // ========================================================= consumers: convolution of step s
  f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.cv_bias + 4 * g);
  touch_v(bias4);
  T* outp = (T*)p.out;
  unsigned o_off[NRO];
#pragma unroll
  for (int ro = 0; ro < NRO; ++ro)
    o_off[ro] = (unsigned)(((oy0 + NRO * wave + ro) * p.Wo + ox0 + r) * p.ldo + 4 * g);
  const int64_t oplane = (int64_t)p.Ho * p.Wo * p.ldo;

  for (int step = 0; step < nsteps_z; ++step) {
    const int zb = step * TD;
    ws_barrier();                                      // X (the producers' hand-off among themselves)
    // ---- conv: input plane c (z = z0 + zb - 1 + c) lives in ring slot (zb + c) % R
    f32x4 acc[4][NRO];
    int pofs[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) pofs[c] = ((zb + c) % R) * PLANE_B + wrow;
    frag_t av[PD + 1][NRO];
    auto issue = [&](int it, frag_t (&dst)[NRO]) {
      const int c = it / J, j = it % J;
#pragma unroll
      for (int ro = 0; ro < NRO; ++ro)
        dst[ro] = *reinterpret_cast<const frag_t*>(smem + pofs[c] + laneoff[j] + ro * HW * ROWB);
    };
#pragma unroll
    for (int q = 0; q < PD; ++q) issue(q, av[q]);
    if (!DT_DBG(p, 2))
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it + PD < NIT) issue(it + PD, av[(it + PD) % (PD + 1)]);
      __builtin_amdgcn_sched_barrier(0);
      const int c = it / J, j = it % J;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int zi = c - kd;
        if (zi >= 0 && zi < 4) {
          const bool first = kd == 0 && j == 0;
          const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ro = 0; ro < NRO; ++ro)
            acc[zi][ro] = mma16<T>(wreg[kd][j], av[it % (PD + 1)][ro], first ? zero : acc[zi][ro]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: out = conv + bias + h (identity residual: the centre plane of the ring)
#pragma unroll
    for (int zi = 0; zi < 4; ++zi) {
      const int oz = z0 + zb + zi;
      T* op = outp + ((int64_t)n * p.Do + oz) * oplane;
#pragma unroll
      for (int ro = 0; ro < NRO; ++ro) {
        const u32x2 hres = *reinterpret_cast<const u32x2*>(smem + pofs[zi + 1] + ((ro + 1) * HW + r + 1) * ROWB + 8 * g);
        f32x4 v = acc[zi][ro] + bias4;
        v += Raw4<T>::cvt(hres);
        if (!DT_DBG(p, 4)) store4<T>(op + o_off[ro], v);
      }
    }
    ws_barrier();                                      // Y
  }
}

}  // namespace segmi

using namespace segmi;

static inline int dectop_tz(int n, int Do, int Ho, int Wo) {
  const int columns = n * (Ho / dectop::TH) * (Wo / dectop::TW);
  const int steps = Do / dectop::TD;
  int zs = 256 / columns;
  if (zs < 1) zs = 1;
  if (zs > steps / 2) zs = steps / 2;
  return zs < 1 ? 1 : zs;
}

extern "C" {

int segmi_dectop_ok(int dtype, const segmi_act* in, const segmi_act* out) {
  if (!act_ok(in) || !act_ok(out) || dtype != SEGMI_BF16) return 0;
  if (in->c != 32 || out->c != 16 || in->n != out->n) return 0;
  if (out->d != 2 * in->d || out->h != 2 * in->h || out->w != 2 * in->w) return 0;
  if (out->d % dectop::TD || out->h % dectop::TH || out->w % dectop::TW) return 0;
  if (in->ld % 8 || out->ld % 4 || ((uintptr_t)in->data % 16) || ((uintptr_t)out->data % 8)) return 0;
  if ((int64_t)in->h * in->w * in->ld * 2 >= (1ll << 31) || (int64_t)out->h * out->w * out->ld >= (1ll << 31)) return 0;
  return 1;
}

int segmi_dectop_fwd(int dtype, const segmi_act* in, const segmi_act* out, const void* up_frag,
                     const float* up_bias, const float* up_alpha, const void* conv_packed,
                     const float* conv_bias, void* stream) {
  SEGMI_CHECK_ARG(segmi_dectop_ok(dtype, in, out), "dectop: layer not eligible (ask segmi_dectop_ok)");
  SEGMI_CHECK_ARG(up_frag && up_bias && up_alpha && conv_packed && conv_bias, "dectop: null operand");
  DecTopParams p{};
  p.in = in->data; p.out = out->data; p.up_frag = up_frag; p.up_bias = up_bias; p.up_alpha = up_alpha;
  p.cv_frag = conv_packed; p.cv_bias = conv_bias;
  p.N = in->n; p.Dc = in->d; p.Hc = in->h; p.Wc = in->w; p.Do = out->d; p.Ho = out->h; p.Wo = out->w;
  p.ldi = in->ld; p.ldo = out->ld;
  p.ty = out->h / dectop::TH; p.tx = out->w / dectop::TW;
  p.tz = dectop_tz(in->n, out->d, out->h, out->w);
  static const int dbg = getenv("SEGMI_DECTOP_DBG") ? atoi(getenv("SEGMI_DECTOP_DBG")) : 0;
  p.dbg = dbg;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dectop_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, dectop::LDS_BYTES);
    attr_done = true;
  }
  const unsigned grid = (unsigned)(p.N * p.ty * p.tx * p.tz);
  hipLaunchKernelGGL(dectop_kernel, grid, 512, dectop::LDS_BYTES, (hipStream_t)stream, p);
  SEGMI_LAUNCH_CHECK("dectop_fwd");
  return SEGMI_OK;
}

}  // extern "C"
