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
#include "convt_ps_impl.h"

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
  int dbg;   // SEGMI_DECTOP_DBG, timing probes: 1 = producers idle, 2 = consumers idle (wrong output)
  int alpha01;   // the caller asserts 0 <= *up_alpha <= 1: PReLU as max(v, slope v)
};

namespace dectop {
constexpr int TD = 4, TH = 16, TW = 16, HH = TH + 2, HW = TW + 2;
constexpr int R = 11;                       // fine ring: 6 planes read + 1 held + 4 written per step
constexpr int ROWB = 32, PLANE_B = HH * HW * ROWB, RING_B = R * PLANE_B;           // 114,048
constexpr int CR = 3, CH = TH / 2 + 2, CW = TW / 2 + 2, CROWB = 64;                 // coarse ring: 3 planes of 10 x 10 voxels
constexpr int CPLANE_B = CH * CW * CROWB;
constexpr int CAT_B = CR * CPLANE_B + 1536;    // + overrun: idle lanes (base index up to 111) and the (+1, +1) neighbour reads
constexpr int OFF_CAT = RING_B, OFF_DUMP = OFF_CAT + CAT_B;
constexpr int LDS_BYTES = OFF_DUMP + 16;                                            // 133,968
constexpr int J = 5, NIT = 6 * J, PD = 4;
constexpr int NRO = 4;                      // output rows per consumer wave
constexpr int NLP = 4;                      // 16-byte chunks per producer thread of a 2-plane coarse stage (800 / 256)
constexpr int PLC = CH * CW * 4;            // chunks per coarse plane
constexpr int NBASE = CH * CW;              // base voxels per coarse plane
constexpr int NTILE = (NBASE + 15) / 16;    // 16-voxel tiles per coarse plane (7; 12 idle lanes in the last)
}  // namespace dectop

// Round 3: the producers are re-tiled the way convt_ps_kernel works.  A producer tile is 16 BASE voxels of
// one coarse plane (the 10 x 10 coarse footprint of the column enumerated linearly: 7 tiles per plane, 100 of
// 112 lanes busy instead of 9 of 16), and for a tile the wave computes ALL 8 output-parity classes from 8
// neighbour fragments: 27 MFMAs per 8 emits, the 27 weight fragments of the transposed convolution in 108
// VGPRs for the whole launch (they were re-read from LDS per plane: 380 KB of LDS reads per step on top
// of the convolution's 480 KB).  A step produces the 4 fine planes of 2 coarse base planes, so the new
// planes are 2c .. 2c + 3: the fine ring holds one plane more (11) and runs one plane ahead of round 2's.
// Per step and producer wave: ~95 MFMAs, 28 neighbour reads, 28 emits of ~20 instructions (was: 121 MFMAs,
// ~100 reads, 36 emits inside ~2000 instructions).
template <bool A01>
__global__ __launch_bounds__(512, 1) void dectop_kernel(DecTopParams p) {
  using namespace dectop;
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const catl = smem + OFF_CAT;
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
  const int cz0 = z0 >> 1;
  // fine plane f lives in ring slot (f - z0 + 2) % R, coarse plane c in slot c mod CR
  auto ws_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  if (wave >= 4) {
    // =================================================================== producers
    const int pw = wave - 4, ptid = tid & 255;
    // ---- transposed-conv weights -> registers: A fragment of tap (kd, kh, kw)
    frag_t Af[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) Af[k] = *reinterpret_cast<const frag_t*>((const char*)p.up_frag + k * 1024 + lane * 16);
    f32x4 ubias = *reinterpret_cast<const f32x4*>(p.up_bias + 4 * g);
    float ualpha = *p.up_alpha;
    touch_v(ubias);
    touch_s(ualpha);

    // ---- coarse staging (256 threads, <= 2 planes per stage): slot k = chunk ptid + 256 k
    const char* inb = (const char*)p.in;
    const int64_t cplane_stride = (int64_t)p.Hc * p.Wc * p.ldi * 2;
    const char* img = inb + (int64_t)n * p.Dc * cplane_stride;
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
    auto cat_fetch = [&](int c_first, int nplanes) {          // coarse planes c_first .. -> registers
#pragma unroll
      for (int k = 0; k < NLP; ++k) {
        dreg[k] = frag_t{0u, 0u, 0u, 0u};
        const int cz = c_first + d_pl[k];
        if (d_pl[k] >= 0 && d_pl[k] < nplanes && (unsigned)cz < (unsigned)p.Dc && d_goff[k] >= 0)
          dreg[k] = *reinterpret_cast<const frag_t*>(img + (int64_t)cz * cplane_stride + (unsigned)d_goff[k]);
      }
    };
    auto cat_commit = [&](int c_first, int nplanes) {         // ... -> their ring slots (zeros outside the volume)
#pragma unroll
      for (int k = 0; k < NLP; ++k) {
        if (d_pl[k] >= 0 && d_pl[k] < nplanes) {
          const int cz = c_first + d_pl[k];
          const int slot = ((cz % CR) + CR) % CR;
          *reinterpret_cast<frag_t*>(catl + slot * CPLANE_B + d_loff[k]) = dreg[k];
        }
      }
    };

    // ---- per-lane geometry of the wave's tiles, computed once: a production round is 2 planes x 7 tiles
    // dealt to the 4 producer waves (wave pw: i = pw, pw + 4, pw + 8, pw + 12 < 14; plane i / 7, tile i % 7),
    // so slot q of the wave always holds the same 16 base voxels of the coarse footprint.
    constexpr int NSL = 4;
    int t_baddr[NSL], t_waddr[NSL];
    unsigned t_mask[NSL];           // bits 0-3: write enable of (py, px) = inside the footprint; bits 4-7: inside the volume
#pragma unroll
    for (int q = 0; q < NSL; ++q) {
      const int i = pw + 4 * q, k = i % NTILE;
      const int v = 16 * k + r;
      const bool lane_ok = v < NBASE && i < 2 * NTILE;
      const int cyl = v / CW, cxl = v - cyl * CW;
      const int jy0 = 2 * cyl - 1, jx0 = 2 * cxl - 1;          // footprint row / column of the even-parity output
      t_baddr[q] = v * CROWB + g * 16;                          // the base voxel's channel chunk g in a coarse plane
      t_waddr[q] = (jy0 * HW + jx0) * ROWB + 8 * g;             // its even-parity output voxel in a ring plane
      unsigned m = 0;
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int jy = jy0 + py, jx = jx0 + px;
          if (lane_ok && (unsigned)jy < (unsigned)HH && (unsigned)jx < (unsigned)HW) m |= 1u << (py * 2 + px);
          if ((unsigned)(oy0 - 1 + jy) < (unsigned)p.Ho && (unsigned)(ox0 - 1 + jx) < (unsigned)p.Wo) m |= 16u << (py * 2 + px);
        }
      t_mask[q] = m;
    }
    // ---- one tile: its 16 base voxels of a coarse plane -> the 8 fine voxels of each.  slotA / slotB: LDS
    // offsets of that coarse plane and the next; pl0 / pl1: ring offsets of the fine planes 2 cb, 2 cb + 1;
    // zin0 / zin1: those planes lie inside the volume (else zeros) -- all wave-uniform
    auto produce_tile = [&](int q, int slotA, int slotB, int pl0, int pl1, bool zin0, bool zin1) {
      const int baddr = t_baddr[q];
      frag_t Bf[8];                                             // neighbour (dz, dy, dx) = bits of the index
#pragma unroll
      for (int o = 0; o < 8; ++o)
        Bf[o] = *reinterpret_cast<const frag_t*>(catl + ((o & 4) ? slotB : slotA) + baddr +
                                                 (((o >> 1) & 1) * CW + (o & 1)) * CROWB);
      // 27 MFMAs into 8 accumulators, round-robin over the classes (tap qq of every class that has one):
      // within a class the taps keep convt_ps_kernel's order (x fastest; odd parity = (k = 2, this voxel)
      // then (k = 0, next voxel)), so the sums are its bits
      f32x4 acc[8];
#pragma unroll
      for (int qq = 0; qq < 8; ++qq)
#pragma unroll
        for (int cls = 0; cls < 8; ++cls) {
          if (qq < ct_ntaps_c(cls)) {
            const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
            const int nw = 1 + px, nh = 1 + py;
            const int tw = qq % nw, th = (qq / nw) % nh, td = qq / (nw * nh);
            const int kw = px ? (tw ? 0 : 2) : 1, kh = py ? (th ? 0 : 2) : 1, kd = pz ? (td ? 0 : 2) : 1;
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[cls] = mma16<T>(Af[(kd * 3 + kh) * 3 + kw], Bf[ctps_tap_a(cls, qq)], qq == 0 ? zero : acc[cls]);
          }
        }
      const unsigned m = t_mask[q];
      const int waddr = t_waddr[q];
#pragma unroll
      for (int cls = 0; cls < 8; ++cls) {
        const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
        f32x4 vv = acc[cls] + ubias;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // A01 (0 <= slope <= 1, every PReLU in practice): max(v, slope v) == the select form bit for bit
          if constexpr (A01) vv[e] = fmaxf(vv[e], ualpha * vv[e]);
          else vv[e] = vv[e] > 0.f ? vv[e] : ualpha * vv[e];
        }
        const bool keep = (pz ? zin1 : zin0) && (m & (16u << (py * 2 + px))) != 0;
        u32x2 o;
        o[0] = keep ? pack_bf16x2(vv[0], vv[1]) : 0u;
        o[1] = keep ? pack_bf16x2(vv[2], vv[3]) : 0u;
        const bool wr = (m & (1u << (py * 2 + px))) != 0;
        const int dst = wr ? (pz ? pl1 : pl0) + waddr + (py * HW + px) * ROWB : OFF_DUMP;
        *reinterpret_cast<u32x2*>(smem + dst) = o;
      }
    };

    // ---- rounds.  Round rr produces the base planes cb = cz0 - 1 + 2 rr, cb + 1 (fine planes 2 cb .. 2 cb + 3)
    // from the coarse planes cb .. cb + 2, of which cb + 1, cb + 2 are new.  Rounds 0, 1 are the prologue
    // (fine planes z0-2 .. z0+5), round s + 2 runs beside the convolution of step s; every round is
    //   commit the new coarse planes | barrier X | fetch the next round's | produce | barrier Y
    // and the last step has no round beside it (barriers only).
    cat_fetch(cz0 - 1, 1);
    cat_commit(cz0 - 1, 1);
    cat_fetch(cz0, 2);
    for (int rr = 0; rr <= nsteps_z + 1; ++rr) {
      const bool work = rr <= nsteps_z;
      const int cb0 = cz0 - 1 + 2 * rr;
      if (work) cat_commit(cb0 + 1, 2);
      ws_barrier();                                  // X: the coarse planes are in LDS
      if (work && !(p.dbg & 1)) {
        if (rr + 1 <= nsteps_z) cat_fetch(cb0 + 3, 2);
        int cslot[3], pl[4];
        bool zin[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) cslot[c] = ((((cb0 + c) % CR) + CR) % CR) * CPLANE_B;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int fz = 2 * cb0 + f;                  // >= z0 - 2 always
          pl[f] = ((fz - z0 + 2) % R) * PLANE_B;
          zin[f] = (unsigned)fz < (unsigned)p.Do;
        }
#pragma unroll
        for (int q = 0; q < NSL; ++q) {
          const int i = pw + 4 * q;                    // wave-uniform: plane i / 7 of the round, tile i % 7
          if (i < 2 * NTILE) {
            const bool hi = i >= NTILE;
            produce_tile(q, hi ? cslot[1] : cslot[0], hi ? cslot[2] : cslot[1], hi ? pl[2] : pl[0],
                         hi ? pl[3] : pl[1], hi ? zin[2] : zin[0], hi ? zin[3] : zin[1]);
          }
        }
      }
      ws_barrier();                                  // Y: the round's ring planes are complete, the step beside it consumed
    }
    return;
  }

  // ===================================================================== consumers: convolution of step s
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
  const int wrow = wave * NRO * HW * ROWB;             // consumer wave w: output rows 4w .. 4w+3 (footprint row 4w + kh)
  f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.cv_bias + 4 * g);
  touch_v(bias4);
  T* outp = (T*)p.out;
  unsigned o_off[NRO];
#pragma unroll
  for (int ro = 0; ro < NRO; ++ro)
    o_off[ro] = (unsigned)(((oy0 + NRO * wave + ro) * p.Wo + ox0 + r) * p.ldo + 4 * g);
  const int64_t oplane = (int64_t)p.Ho * p.Wo * p.ldo;
  ws_barrier(); ws_barrier(); ws_barrier(); ws_barrier();   // B1 .. B4 of the producers' prologue

  for (int step = 0; step < nsteps_z; ++step) {
    const int zb = step * TD;
    ws_barrier();                                      // X (the producers' hand-off among themselves)
    if (p.dbg & 2) { ws_barrier(); continue; }         // timing probes only
    // ---- conv: input plane c (z = z0 + zb - 1 + c) lives in ring slot (zb + c + 1) % R
    f32x4 acc[4][NRO];
    int pofs[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) pofs[c] = ((zb + c + 1) % R) * PLANE_B + wrow;
    frag_t av[PD + 1][NRO];
    auto issue = [&](int it, frag_t (&dst)[NRO]) {
      const int c = it / J, j = it % J;
#pragma unroll
      for (int ro = 0; ro < NRO; ++ro)
        dst[ro] = *reinterpret_cast<const frag_t*>(smem + pofs[c] + laneoff[j] + ro * HW * ROWB);
    };
#pragma unroll
    for (int q = 0; q < PD; ++q) issue(q, av[q]);
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
        store4<T>(op + o_off[ro], v);
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
                     const float* up_bias, const float* up_alpha, int up_alpha_in_unit_range,
                     const void* conv_packed, const float* conv_bias, void* stream) {
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
  p.alpha01 = up_alpha_in_unit_range ? 1 : 0;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dectop_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, dectop::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dectop_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, dectop::LDS_BYTES);
    attr_done = true;
  }
  const unsigned grid = (unsigned)(p.N * p.ty * p.tx * p.tz);
  if (p.alpha01) hipLaunchKernelGGL(dectop_kernel<true>, grid, 512, dectop::LDS_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(dectop_kernel<false>, grid, 512, dectop::LDS_BYTES, (hipStream_t)stream, p);
  SEGMI_LAUNCH_CHECK("dectop_fwd");
  return SEGMI_OK;
}

}  // extern "C"
