// wgrad_ws_impl.h -- wave-specialised, double-buffered form of the MFMA weight gradient
// (wgrad_impl.h) for bf16 k3 layers whose X operand has 16 channels per workgroup (the wide,
// memory-bound layers).
//
// wgrad_mfma_kernel runs its three per-tile phases -- global loads in flight, LDS commit, MFMA loop
// -- one after the other in every wave; two co-resident workgroups start together and stay in
// lockstep, so the phases of the chip line up instead of overlapping (444 us on the 16x16 @ 128^3
// layer = 137 stage + 179 MFMA loop + 117 load wait, against 215 us of HBM time).  Here one
// 768-thread workgroup per CU splits the roles:
//   waves 0-3  (one per SIMD)  consumers: the transposing LDS reads + MFMAs of wgrad_impl.h on tile i
//                              (454 instructions per tile: 256 ds_read_b64_tr_b16, 112 MFMA)
//   waves 4-11 (two per SIMD) producers: each takes an eighth of every tile's 16-byte chunks; in
//              iteration i it transforms (segmi_in_affine) and writes tile i + 1 to the free LDS buffer
//              and issues the global loads of tile i + 3 into the registers that tile leaves
// with ONE s_barrier per tile: two tiles per CU are always in flight, and the producers' VALU work
// issues beside the consumers' MFMAs.  (A first version with ONE producer wave per
// SIMD doing a tile every iteration was bound by that wave's instruction issue -- a wave issues
// about one instruction per 4-5 clk: 650 instructions per tile -- not by memory.)
// The barrier is an explicit s_waitcnt lgkmcnt(0) + s_barrier, not __syncthreads(): barriers do
// not drain VMEM.
//
// Zero padding comes from the buffer-load range check: a chunk outside the volume gets an offset
// past the buffer's num_records and reads 0 -- no select at commit.  Only the fused input
// transform needs to know which chunks are padding (it must leave them 0): a per-lane mask, built
// only for tiles that touch the volume border; interior tiles (62 % at 128^3) take a path without
// any per-chunk bounds arithmetic.
//
// Tile order ("z-marching through L2"): the work is cut into units = (image, z-segment, y-tile,
// x-tile) columns of tiles.  The grid is cut into 8 groups by blockIdx % 8 (workgroups b and b + 8
// share an XCD under the round-robin placement -- a speed assumption only); group k owns the k-th
// eighth of the unit list, its workgroups take neighbouring units and each walks its unit along z.
// The halo planes a tile shares with its z-predecessor were fetched one iteration earlier by the
// same CU, and the y / x halos belong to units that neighbouring workgroups of the same XCD walk at
// the same time: both come from that XCD's L2 instead of HBM (measured with FETCH_SIZE: 1.10 GB
// per launch for 1.07 GB of tensors; the 4x8x16 tile's halo is 2.1x of X).
#pragma once
#include <type_traits>
#include "wgrad_impl.h"

#ifdef SEGMI_WGRAD_DIAG
#define WS_STAMP(it, slot)                                                                      \
  do {                                                                                          \
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (it) < 128) \
      p.stamps[(((it) * 12) + (threadIdx.x >> 6)) * 4 + (slot)] = __builtin_amdgcn_s_memtime();    \
  } while (0)
#else
#define WS_STAMP(it, slot) do {} while (0)
#endif

namespace segmi {

__device__ __forceinline__ void ws_barrier() {
  // LDS writes / reads of this wave are complete, then rendezvous; global loads stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int KS, int S, int CTO, int CTI, int TD, int TH, int TW>
__global__ __launch_bounds__(768) void wgrad_ws_kernel(WgradParams p) {
  using T = bf16_t;
  using G = WgradGeom<T, KS, S, CTO, CTI, TD, TH, TW>;
  constexpr int BUF = (G::LDS_BYTES + 255) / 256 * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int cochunk = blockIdx.y / p.ci_chunks, cichunk = blockIdx.y % p.ci_chunks;
  const int co0 = cochunk * 16 * CTO, ci0 = cichunk * 16 * CTI;

  // ---- unit schedule (workgroup-uniform): unit u = ((n * zs + seg) * ty + y) * tx + x covers the
  // z-tiles [seg * zper, min((seg + 1) * zper, tz)) of one (n, y, x) column
  const int nx = gridDim.x, bid = blockIdx.x;
  const int cols = p.ty * p.tx;
  const int nunits = p.N * p.zs * cols;
  int u_begin, u_end, u_stride;
  if ((nx & 7) == 0 && !(p.dbg & 64)) {        // SEGMI_WGRAD_DBG bit 64: the plain unit order (A/B of the XCD grouping)
    const int xcd = bid & 7, slot = bid >> 3;
    u_begin = (int)((int64_t)nunits * xcd / 8) + slot;
    u_end = (int)((int64_t)nunits * (xcd + 1) / 8);
    u_stride = nx >> 3;
  } else {
    u_begin = bid; u_end = nunits; u_stride = nx;
  }
  auto unit_len = [&](int u) {
    const int seg = (u / cols) % p.zs;
    const int left = p.tz - seg * p.zper;
    return left < p.zper ? left : p.zper;
  };
  int niter = 0;
  for (int u = u_begin; u < u_end; u += u_stride) niter += unit_len(u);

  // The 16 x 16 stride-1 layers use the row-split consumer below, everything else the tap-split one
  constexpr bool ROWS = KS == 3 && S == 1 && CTO == 1 && CTI == 1 && TW == 16 && TH == 8;
  if constexpr (ROWS) {
    if (wave < 4) {
      // ========================================================= consumers, row split
      // Wave w owns the output rows y = 2w, 2w + 1 of the tile (all 16 x, all TD planes) and ALL 27
      // taps, so that fragments are reused from registers instead of being re-read per tap:
      //   A (dY): one operand per output plane z = [row 2w | row 2w + 1] (k-slots 0-15 | 16-31),
      //           read once per tile: 2 TD transposing reads;
      //   B (X):  for every input plane c and x-shift dx a STRIP of the 4 rows 2w .. 2w + 3 in 8
      //           consecutive VGPRs; the operand of kh is the strip's rows kh, kh + 1 = registers
      //           2kh .. 2kh + 3 (no copies), and it serves kd = 0, 1, 2 for the output planes c - kd.
      // 4 reads feed up to 9 MFMAs: 80 transposing reads per 108 MFMAs per wave and tile, against 256
      // per 112 for the tap split (which re-reads the X fragment of every tap: the loop ran at the LDS
      // instruction rate, not at the MFMA rate).  The 27 accumulators of the four waves are summed
      // through LDS once, in fixed order, when the last tile is done.
      typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
      const int lane = tid & 63;
      const int g = lane >> 4, i16 = lane & 15;
      const int q = i16 >> 2, pp = i16 & 3;
      const int lane_line = (4 * g + q) * 32 + 8 * pp;           // a line = 16 voxels x 32 B
      const int ybase = lane_line + (2 * wave * TW) * G::YROWB;
      const int xbase = G::YBYTES + lane_line + (2 * wave * G::HW) * G::XROWB;
      f32x4 acc[27];
#pragma unroll
      for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int NSTRIP = G::HD * 3;                            // (plane c, shift dx)
      constexpr int PD = 2;                                        // strips in flight ahead of the MFMAs
      auto read_strip = [&](const char* buf, int sidx) {
        const int c = sidx / 3, dx = sidx % 3;
        u32x8 st;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4*)(buf + xbase + ((c * G::HH + j) * G::HW + dx) * G::XROWB));
          const u32x2 r2 = __builtin_bit_cast(u32x2, r);
          st[2 * j] = r2[0]; st[2 * j + 1] = r2[1];
        }
        return st;
      };
      ws_barrier();                                   // tile 0 is in buffer 0
      for (int it = 0; it < niter; ++it) {
        const char* buf = smem + (it & 1) * BUF;
        WS_STAMP(it, 0);
        if (!WGRAD_DBG(p, 4)) {
          frag_t af[TD];
#pragma unroll
          for (int z = 0; z < TD; ++z) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4*)(buf + ybase + ((z * TH) * TW) * G::YROWB));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4*)(buf + ybase + ((z * TH + 1) * TW) * G::YROWB));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            af[z] = frag_t{l2[0], l2[1], h2[0], h2[1]};
          }
          u32x8 strip[PD + 1];
#pragma unroll
          for (int s0 = 0; s0 < PD; ++s0) strip[s0] = read_strip(buf, s0);
#pragma unroll
          for (int sidx = 0; sidx < NSTRIP; ++sidx) {
            if (sidx + PD < NSTRIP) strip[(sidx + PD) % (PD + 1)] = read_strip(buf, sidx + PD);
            __builtin_amdgcn_sched_barrier(0);
            const int c = sidx / 3, dx = sidx % 3;
            const u32x8 st = strip[sidx % (PD + 1)];
            const frag_t bfs[3] = {__builtin_shufflevector(st, st, 0, 1, 2, 3),
                                   __builtin_shufflevector(st, st, 2, 3, 4, 5),
                                   __builtin_shufflevector(st, st, 4, 5, 6, 7)};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const frag_t bf = bfs[kh];
#pragma unroll
              for (int kd = 0; kd < 3; ++kd) {
                const int z = c - kd;
                if (z >= 0 && z < TD) {
                  const int t = (kd * 3 + kh) * 3 + dx;
                  acc[t] = mma16<T>(af[z], bf, acc[t]);
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        WS_STAMP(it, 1);
        ws_barrier();                                 // done with buffer it & 1; tile it + 1 is ready
        WS_STAMP(it, 2);
      }
      // ---- sum the four waves' accumulators through LDS: (w0 + w2) + (w1 + w3), fixed order
      f32x4* red = reinterpret_cast<f32x4*>(smem);    // [2][27][64] f32x4 = 55 KB (buffers are free now)
      if (wave >= 2) {
#pragma unroll
        for (int t = 0; t < 27; ++t) red[((wave - 2) * 27 + t) * 64 + lane] = acc[t];
      }
      ws_barrier();
      if (wave < 2) {
#pragma unroll
        for (int t = 0; t < 27; ++t) acc[t] += red[(wave * 27 + t) * 64 + lane];
      }
      ws_barrier();
      if (wave == 1) {
#pragma unroll
        for (int t = 0; t < 27; ++t) red[t * 64 + lane] = acc[t];
      }
      ws_barrier();
      if (wave == 0) {
        float* slab = p.partials + (int64_t)blockIdx.x * p.Cout * p.Cin * 27;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
          const f32x4 v = acc[t] + red[t * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co = co0 + 4 * g + e, ci = ci0 + i16;
            slab[((int64_t)co * p.Cin + ci) * 27 + t] = v[e];
          }
        }
      }
      return;
    }
  }
  if (!ROWS && wave < 4) {
    // =========================================================== consumers, tap split
    const int lane = tid & 63;
    const int g = lane >> 4, i16 = lane & 15;
    f32x4 acc[G::NTW][CTO][CTI];
#pragma unroll
    for (int a = 0; a < G::NTW; ++a)
#pragma unroll
      for (int b = 0; b < CTO; ++b)
#pragma unroll
        for (int c = 0; c < CTI; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int tapoff[G::NTW];
#pragma unroll
    for (int ti = 0; ti < G::NTW; ++ti) {
      int tap = wave + 4 * ti;
      if (tap > G::NTAPS - 1) tap = G::NTAPS - 1;
      const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
      tapoff[ti] = ((kd * G::HH + kh) * G::HW + kw) * G::XROWB;
    }
    const int q = i16 >> 2, pp = i16 & 3;
    const int v = 4 * g + q;  // voxel within the line
    const int ylane = v * G::YROWB + 8 * pp;
    const int xlane = ((v / TW) * S * G::HW + (v % TW) * S) * G::XROWB + 8 * pp;

    // One wave per SIMD reads LDS with nothing to hide its latency but its own look-ahead: the
    // fragments of k-group lg + 1 (2 * (CTO + NTW * CTI) transposing reads) are issued before the
    // MFMAs of k-group lg and waited for with counted lgkmcnt, in two register sets selected by the
    // parity of lg (full unroll).  Left to the compiler's schedule (reads placed next to their
    // MFMA, 2-4 in flight) the loop ran at the LDS latency: 50 clk per MFMA, 306 us per launch on
    // the 16x16 @ 128^3 layer with the global loads switched off.
    frag_t af[2][CTO], bf[2][G::NTW][CTI];
    auto read_group = [&](const char* ysm, const char* xsm, int lg, frag_t (&a)[CTO], frag_t (&b)[G::NTW][CTI]) {
#pragma unroll
      for (int ct = 0; ct < CTO; ++ct) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(ysm + ylane + (2 * lg) * 16 * G::YROWB + ct * 32));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (lds_s16x4*)(ysm + ylane + (2 * lg + 1) * 16 * G::YROWB + ct * 32));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        a[ct] = frag_t{l2[0], l2[1], h2[0], h2[1]};
      }
      const int r0 = wg_line_row<S, TH, TW, G::HH, G::HW>(2 * lg) * G::XROWB;
      const int r1 = wg_line_row<S, TH, TW, G::HH, G::HW>(2 * lg + 1) * G::XROWB;
#pragma unroll
      for (int ti = 0; ti < G::NTW; ++ti)
#pragma unroll
        for (int c = 0; c < CTI; ++c) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4*)(xsm + xlane + tapoff[ti] + r0 + c * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4*)(xsm + xlane + tapoff[ti] + r1 + c * 32));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          b[ti][c] = frag_t{l2[0], l2[1], h2[0], h2[1]};
        }
    };
    constexpr int NKG = G::NL / 2;
    static_assert(NKG % 2 == 0, "k-groups come in pairs (register-set parity)");
    ws_barrier();                                   // tile 0 is in buffer 0
    for (int it = 0; it < niter; ++it) {
      char* ysm = smem + (it & 1) * BUF;
      char* xsm = ysm + G::YBYTES;
      if (!WGRAD_DBG(p, 4)) {
        read_group(ysm, xsm, 0, af[0], bf[0]);
#pragma unroll
        for (int lg = 0; lg < NKG; ++lg) {
          if (lg + 1 < NKG) read_group(ysm, xsm, lg + 1, af[(lg + 1) & 1], bf[(lg + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ti = 0; ti < G::NTW; ++ti)
#pragma unroll
            for (int ct = 0; ct < CTO; ++ct)
#pragma unroll
              for (int c = 0; c < CTI; ++c)
                acc[ti][ct][c] = mma16<T>(af[lg & 1][ct], bf[lg & 1][ti][c], acc[ti][ct][c]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      ws_barrier();                                 // done with buffer it & 1; tile it + 1 is ready
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
          for (int c = 0; c < CTI; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int co = co0 + ct * 16 + 4 * g + e, ci = ci0 + c * 16 + i16;
              slab[((int64_t)co * p.Cin + ci) * G::NTAPS + tap] = acc[ti][ct][c][e];
            }
      }
    }
    return;
  }

  // ============================================================= producers (waves 4 .. 11)
  constexpr int PT = 512;                          // producer threads: every tile is split over all of them
  const int ptid = tid - 256;
  constexpr int NLY = (G::NV * G::YCPR + PT - 1) / PT, NLX = (G::XROWS * G::XCPR + PT - 1) / PT;
  static_assert(NLX <= 32, "one validity bit per staged X chunk");
  static_assert(G::HD < 127 && G::HH < 127 && G::HW < 127 && TD < 127 && TH < 127 && TW < 127, "packed coordinates");
  // step-invariant per-lane descriptors: 32-bit byte offset inside a tile + byte-packed tile-local
  // coordinates (0x7f7f7f = this lane stages nothing for slot k: fails every upper bound)
  unsigned y_goff[NLY], x_goff[NLX], y_pk[NLY], x_pk[NLX];
  // z-marching: a tile that continues its unit (the next TD output planes of the same column) shares its first
  // OV = HD - TD * S input planes with the last OV planes of the tile before it, which sit -- already transformed --
  // in the OTHER LDS buffer while this one is written.  Those chunks are copied LDS -> LDS instead of being fetched
  // again (x_ov: bit k = this lane's chunk k lies in the shared planes): X goes through the CU's vector-memory pipe
  // 1.4 x instead of 2.1 x per launch for the 4 x 8 x 16 tile.  Measured (round 4, serial kernel times): 153 -> 150 us
  // for the 16 x 16 stride-1 layers, 4.70 -> 4.67 ms per step beside the main chain -- the producers are bound by
  // their instruction issue (the loads of skipped chunks are still issued, with an out-of-range offset), not by the
  // bytes; stride 2 shares one plane of five and got 6 % slower with the per-lane selects, so it keeps the plain fetch.
  constexpr int OV = S == 1 ? G::HD - TD * S : 0;
  constexpr int OV_SRC = TD * S * G::HH * G::HW * G::XROWB;   // byte distance of the same (y, x, chunk) OV planes up
  static_assert(OV >= 0 && OV < G::HD, "overlap planes");
  unsigned x_ov = 0u;
#pragma unroll
  for (int k = 0; k < NLY; ++k) {
    const int i = ptid + PT * k;
    const int v = i / G::YCPR, ch = i % G::YCPR;
    const int vz = v / (TW * TH), vy = (v / TW) % TH, vx = v % TW;
    const bool has = i < G::NV * G::YCPR;
    y_goff[k] = has ? (unsigned)(((vz * p.Hy + vy) * p.Wy + vx) * p.ldy * (int)sizeof(T) + ch * 16) : 0u;
    y_pk[k] = has ? (unsigned)(vz | (vy << 8) | (vx << 16)) : 0x7f7f7fu;
  }
  // the tile's origin voxel (output (0,0,0) -> input (PAD,PAD,PAD)): always inside the volume
  const unsigned x_center = (unsigned)(((G::PAD * p.Hx + G::PAD) * p.Wx + G::PAD) * p.ldx * (int)sizeof(T));
#pragma unroll
  for (int k = 0; k < NLX; ++k) {
    const int i = ptid + PT * k;
    const int v = i / G::XCPR, ch = i % G::XCPR;
    const int hx = v % G::HW, hy = (v / G::HW) % G::HH, hz = v / (G::HW * G::HH);
    const bool has = i < G::XROWS * G::XCPR;
    x_goff[k] = has ? (unsigned)(((hz * p.Hx + hy) * p.Wx + hx) * p.ldx * (int)sizeof(T) + ch * 16) : x_center;
    x_pk[k] = has ? (unsigned)(hz | (hy << 8) | (hx << 16)) : 0x7f7f7fu;
    x_ov |= (has && hz < OV) ? (1u << k) : 0u;
  }
  const bool in_tf = p.in_scale != nullptr;
  const bool in_act = in_tf && p.in_alpha != nullptr;
  float tsc[8], tsh[8];
  float in_alpha = in_act ? *p.in_alpha : 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = ci0 + (ptid % G::XCPR) * 8 + e;     // PT % XCPR == 0: the same channels for every k
    tsc[e] = in_tf ? p.in_scale[ch] : 1.f;
    tsh[e] = in_tf ? p.in_shift[ch] : 0.f;
  }
  static_assert(PT % G::XCPR == 0, "a thread's X chunks share their channel offset");
  const int mode = !in_tf ? 0 : (in_act ? (in_alpha >= 0.f && in_alpha <= 1.f ? 3 : 2) : 1);
  // raw buffers over the two tensors: an offset >= num_records reads 0 (the zero padding)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;

  // walk state (wave-uniform): the tile the next fetch() takes, as 32-bit byte offsets of its
  // origins (x's may be "negative": lanes inside the volume still sum to a valid unsigned offset)
  int cur_u = u_begin, cur_z = 0, cur_zend = 0, cur_t = 0, cur_zbeg = 0;
  unsigned ybase = 0u, xbase = 0u;
  bool xy_border = false;       // the unit's column touches the volume border in y or x
  unsigned ylim_xy = 0u, xlo_xy = 0u, xhi_xy = 0u;   // y / x bytes of the packed bounds (constant per unit)
  const unsigned ystep = (unsigned)((int64_t)TD * p.Hy * p.Wy * p.ldy * 2);
  const unsigned xstep = (unsigned)((int64_t)TD * S * p.Hx * p.Wx * p.ldx * 2);
  auto clamp7 = [](int v) { return (unsigned)(v < 0 ? 0 : (v > 126 ? 126 : v)); };
  auto open_unit = [&](int u) {
    int t = u;
    const int ux = t % p.tx; t /= p.tx;
    const int uy = t % p.ty; t /= p.ty;
    const int seg = t % p.zs;
    const int n = t / p.zs;
    cur_z = seg * p.zper;
    cur_zbeg = cur_z;
    cur_zend = cur_z + unit_len(u);
    const int oz0 = cur_z * TD, oy0 = uy * TH, ox0 = ux * TW;
    const int iz0 = oz0 * S - G::PAD, iy0 = oy0 * S - G::PAD, ix0 = ox0 * S - G::PAD;
    ybase = (unsigned)((((((int64_t)n * p.Dy + oz0) * p.Hy + oy0) * p.Wy + ox0) * p.ldy + co0) * 2);
    xbase = (unsigned)((((((int64_t)n * p.Dx + iz0) * p.Hx + iy0) * p.Wx + ix0) * p.ldx + ci0) * 2);
    xy_border = iy0 < 0 || ix0 < 0 || iy0 + G::HH > p.Hx || ix0 + G::HW > p.Wx || oy0 + TH > p.Hy || ox0 + TW > p.Wy;
    ylim_xy = (clamp7(p.Hy - oy0 - 1) << 8) | (clamp7(p.Wy - ox0 - 1) << 16);
    xlo_xy = (clamp7(-iy0) << 8) | (clamp7(-ix0) << 16);
    xhi_xy = (clamp7(p.Hx - iy0 - 1) << 8) | (clamp7(p.Wx - ix0 - 1) << 16);
  };
  auto advance = [&]() {                          // to the next tile of the workgroup's sequence
    ++cur_t;
    if (cur_t < niter) {
      if (++cur_z == cur_zend) { cur_u += u_stride; open_unit(cur_u); }
      else { ybase += ystep; xbase += xstep; }
    }
  };
  if (niter > 0) open_unit(cur_u);

  // tile `cur_t` -> registers (ry, rx); xmask bit k: rx[k] lies inside the volume (kept for border
  // tiles only: the fused input transform must leave the zero padding 0); returns the border flag
  // `cont` (out): the tile continues its unit -- its first OV input planes are not fetched (commit copies them)
  auto fetch = [&](frag_t (&ry)[NLY], frag_t (&rx)[NLX], unsigned& xmask, bool& cont) {
    const int oz0 = cur_z * TD, iz0 = oz0 * S - G::PAD;
    const bool border = (xy_border || iz0 < 0 || iz0 + G::HD > p.Dx || oz0 + TD > p.Dy) && !WGRAD_DBG(p, 64);
    cont = OV > 0 && p.zmarch && cur_z != cur_zbeg;
    const unsigned skip = cont ? x_ov : 0u;
    if (!border && !WGRAD_DBG(p, 1)) {
#pragma unroll
      for (int k = 0; k < NLY; ++k) ry[k] = __builtin_amdgcn_raw_buffer_load_b128(yrs, WGRAD_DBG(p, 32) ? OOB : y_goff[k], ybase, 0);
      if (!cont) {
#pragma unroll
        for (int k = 0; k < NLX; ++k) rx[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, WGRAD_DBG(p, 16) ? OOB : x_goff[k], xbase, 0);
      } else {
        // (per-lane out-of-range marker in the vector offset, no scalar offset: marker + base must not wrap; a slot
        // whose 512 chunks ALL lie in the shared planes issues no load at all)
#pragma unroll
        for (int k = 0; k < NLX; ++k) {
          if ((PT * (k + 1) - 1) / G::XCPR < OV * G::HH * G::HW) continue;
          rx[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ((skip >> k) & 1u) ? OOB : xbase + x_goff[k], 0, 0);
        }
      }
    } else {
      // valid iff lo <= coordinate <= hi in every dimension, three byte-packed coordinates at once:
      // bit 7 of a byte of (pk | 0x80..) - LO survives iff field >= lo, of (HI | 0x80..) - pk iff field <= hi
      const unsigned ylim = clamp7(p.Dy - oz0 - 1) | ylim_xy;
      const unsigned xlo = clamp7(-iz0) | xlo_xy;
      const unsigned xhi = clamp7(p.Dx - iz0 - 1) | xhi_xy;
      const bool dead = WGRAD_DBG(p, 1);
#pragma unroll
      for (int k = 0; k < NLY; ++k) {
        const bool ok = ((((ylim | 0x808080u) - y_pk[k]) & 0x808080u) == 0x808080u) && !dead;
        ry[k] = __builtin_amdgcn_raw_buffer_load_b128(yrs, ok ? ybase + y_goff[k] : OOB, 0, 0);
      }
      unsigned xm = 0u;
#pragma unroll
      for (int k = 0; k < NLX; ++k) {
        const unsigned t1 = (x_pk[k] | 0x808080u) - xlo, t2 = (xhi | 0x808080u) - x_pk[k];
        const bool ok = ((t1 & t2 & 0x808080u) == 0x808080u) && !dead;
        rx[k] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok && !((skip >> k) & 1u) ? xbase + x_goff[k] : OOB, 0, 0);
        xm |= ok ? (1u << k) : 0u;
      }
      xmask = xm;
    }
    advance();
    return border;
  };
  auto commit_with = [&](int buf, const frag_t (&ry)[NLY], const frag_t (&rx)[NLX], unsigned xmask, bool cont, auto tf,
                         auto masked) {
    char* const ysm = smem + buf * BUF;
    char* const xsm = ysm + G::YBYTES;
    const char* const xprev = smem + (buf ^ 1) * BUF + G::YBYTES + OV_SRC;   // the tile before this one, OV planes up
    const unsigned skip = cont ? x_ov : 0u;
#pragma unroll
    for (int k = 0; k < NLY; ++k) {
      const int i = ptid + PT * k;
      if (i < G::NV * G::YCPR && !(WGRAD_DBG(p, 8) && ry[k][0] != 0x12345u))
        *reinterpret_cast<frag_t*>(ysm + (i / G::YCPR) * G::YROWB + (i % G::YCPR) * 16) = ry[k];
    }
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      const int i = ptid + PT * k;
      if (i < G::XROWS * G::XCPR && !(WGRAD_DBG(p, 8) && rx[k][0] != 0x12345u)) {
        const int lo = (i / G::XCPR) * G::XROWB + (i % G::XCPR) * 16;
        frag_t val = rx[k];
        if ((skip >> k) & 1u) {
          val = *reinterpret_cast<const frag_t*>(xprev + lo);   // shared plane: transformed when it was first staged
        } else if constexpr (decltype(masked)::value) {
          if ((xmask >> k) & 1u) val = tf(val);            // zero padding stays zero
        } else {
          val = tf(val);
        }
        *reinterpret_cast<frag_t*>(xsm + lo) = val;
      }
    }
  };
  auto commit = [&](int buf, const frag_t (&ry)[NLY], const frag_t (&rx)[NLX], unsigned xmask, bool border, bool cont) {
    // copies of the loop behind wave-uniform switches: no per-element selects on runtime flags
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    if (mode == 0) { commit_with(buf, ry, rx, xmask, cont, [](frag_t v) { return v; }, No{}); return; }
    auto with_mask = [&](auto tf) {
      if (border) commit_with(buf, ry, rx, xmask, cont, tf, Yes{});
      else commit_with(buf, ry, rx, xmask, cont, tf, No{});
    };
    if (mode == 3) with_mask([&](frag_t v) { return bn_prelu01_bf16x8(v, tsc, tsh, in_alpha); });
    else if (mode == 2) with_mask([&](frag_t v) { return bn_prelu_bf16x8(v, tsc, tsh, in_alpha, true); });
    else with_mask([&](frag_t v) { return bn_prelu_bf16x8(v, tsc, tsh, 0.f, false); });
  };

  // Every producer wave takes its share of EVERY tile (an eighth of the chunks) and keeps two tiles
  // in flight in two register sets (A: even tiles, B: odd tiles).  In iteration `it` (consumers on
  // tile it) tile it + 1 is written to its buffer and the loads of tile it + 3 are issued into the
  // registers it leaves.  (With the two groups of four waves alternating whole tiles instead, the
  // working group's commit + fetch was the critical path of every iteration while the other idled.)
  frag_t ryA[NLY], rxA[NLX], ryB[NLY], rxB[NLX];
  unsigned xmA = 0u, xmB = 0u;
  bool bdA = false, bdB = false;
  bool ctA = false, ctB = false, ctC = false;          // "continues its unit" of the tiles in A, in B, of tile 2
  if (niter > 0) {
    bdA = fetch(ryA, rxA, xmA, ctA);                     // tile 0
    if (niter > 1) bdB = fetch(ryB, rxB, xmB, ctB);      // tile 1
    commit(0, ryA, rxA, xmA, bdA, ctA);
    if (niter > 2) bdA = fetch(ryA, rxA, xmA, ctC);      // tile 2
    ctA = ctC;
  }
  ws_barrier();
  for (int it = 0; it < niter; it += 2) {
    WS_STAMP(it, 0);
    if (it + 1 < niter) commit(1, ryB, rxB, xmB, bdB, ctB);   // tile it + 1 (copies from buffer 0 = tile it)
    WS_STAMP(it, 1);
    if (it + 3 < niter) bdB = fetch(ryB, rxB, xmB, ctB);      // tile it + 3
    WS_STAMP(it, 2);
    ws_barrier();
    WS_STAMP(it, 3);
    if (it + 1 >= niter) break;
    WS_STAMP(it + 1, 0);
    if (it + 2 < niter) commit(0, ryA, rxA, xmA, bdA, ctA);   // tile it + 2 (copies from buffer 1 = tile it + 1)
    WS_STAMP(it + 1, 1);
    if (it + 4 < niter) bdA = fetch(ryA, rxA, xmA, ctA);      // tile it + 4
    WS_STAMP(it + 1, 2);
    ws_barrier();
    WS_STAMP(it + 1, 3);
  }
  if constexpr (ROWS) { ws_barrier(); ws_barrier(); ws_barrier(); }   // the consumers' final reduction
}

// LDS of the two buffers; 0 when the configuration does not fit one CU
template <int KS, int S, int CTO, int CTI, int TD, int TH, int TW>
static constexpr int wgrad_ws_lds() {
  using G = WgradGeom<bf16_t, KS, S, CTO, CTI, TD, TH, TW>;
  return 2 * ((G::LDS_BYTES + 255) / 256 * 256);
}

template <int KS, int S, int CTO, int CTI, int TD, int TH, int TW>
static int launch_wgrad_ws_cfg(WgradParams p, int gx, hipStream_t st) {
  using G = WgradGeom<bf16_t, KS, S, CTO, CTI, TD, TH, TW>;
  constexpr int LDS = wgrad_ws_lds<KS, S, CTO, CTI, TD, TH, TW>();
  static_assert(LDS <= 160 * 1024, "two tile buffers must fit the CU's LDS");
  p.tz = cdiv(p.Dy, TD);
  p.ty = cdiv(p.Hy, TH);
  p.tx = cdiv(p.Wy, TW);
  p.ntiles = p.N * p.tz * p.ty * p.tx;
  // z-segments per column: enough units for every workgroup, at least 2 tiles per unit
  int zs = 1;
  while (p.N * p.ty * p.tx * zs < gx && 2 * (zs + 1) <= p.tz) ++zs;
  p.zper = cdiv(p.tz, zs);
  p.zs = cdiv(p.tz, p.zper);
  SEGMI_CHECK_ARG((int64_t)G::HD * p.Hx * p.Wx * p.ldx * 2 < (1ll << 31) &&
                      (int64_t)TD * p.Hy * p.Wy * p.ldy * 2 < (1ll << 31),
                  "conv3d_wgrad: plane too large for the MFMA kernel's 32-bit tile offsets");
  // buffer descriptors (32-bit num_records) over the whole tensors as addressed through their views
  const int64_t xb = (((int64_t)p.N * p.Dx * p.Hx * p.Wx - 1) * p.ldx + p.Cin) * 2;
  const int64_t yb = (((int64_t)p.N * p.Dy * p.Hy * p.Wy - 1) * p.ldy + p.Cout) * 2;
  SEGMI_CHECK_ARG(xb < 0xfff00000ll && yb < 0xfff00000ll, "conv3d_wgrad(ws): tensor beyond the 4 GB buffer range");
  p.x_bytes = (unsigned)xb; p.y_bytes = (unsigned)yb;
  p.ci_chunks = p.Cin / (16 * CTI);
  static const int dbg = getenv("SEGMI_WGRAD_DBG") ? atoi(getenv("SEGMI_WGRAD_DBG")) : 0;
  p.dbg = dbg;
  static const bool zmarch = !(getenv("SEGMI_WGRAD_ZMARCH") && atoi(getenv("SEGMI_WGRAD_ZMARCH")) == 0);   // A/B
  p.zmarch = zmarch ? 1 : 0;
#ifdef SEGMI_WGRAD_DIAG
  // diag build: SEGMI_WGRAD_STAMPS = device address (decimal) of a 128 * 12 * 4 u64 buffer
  static const char* stamps_env = getenv("SEGMI_WGRAD_STAMPS");
  p.stamps = stamps_env ? (unsigned long long*)strtoull(stamps_env, nullptr, 10) : nullptr;
#endif
  const int co_chunks = p.Cout / (16 * CTO);
  dim3 grid((unsigned)gx, (unsigned)(co_chunks * p.ci_chunks));
  auto kern = wgrad_ws_kernel<KS, S, CTO, CTI, TD, TH, TW>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, 768, LDS, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_wgrad(ws)");
  return SEGMI_OK;
}

}  // namespace segmi
