// conv_ring3_impl.h -- third formulation of the z-marching ring (conv_ring_impl.h, conv_ring2_impl.h) for the
// bf16 16 -> 16 k3 s1 layers: the full-resolution convolution of the decoder (forward, input gradient, input
// gradient + BatchNorm-backward sums) and the 16-channel layers at 64^3.  Same tiling and MFMA loop as ring2 (a
// workgroup = an (8 x 16)-voxel column marching z, 4 planes per step; a wave = 2 rows x 4 planes, weights in
// registers, one voxel fragment feeding the kd = 0, 1, 2 taps of three output planes).  What changed against ring2:
//
//  * Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`, 16 B per lane, gfx950): the planes of step k + 1 go from
//    HBM straight into the ring while step k computes.  No staging registers (ring2 held 32 VGPRs across the MFMA
//    loop), no commit phase, and the zero padding comes from the buffer range check: a lane whose voxel is outside
//    the tensor (x / y: a per-lane out-of-range offset, computed once; z: voffset + soffset >= num_records) makes
//    the DMA write ZEROS to its LDS slot (scripts/probes/ldsdma_oob_probe.hip: out-of-range lanes write 0, and
//    soffset takes part in the range check).  The DMA is issued from inline asm: hipcc orders every later ds_read
//    behind a pending LDS-DMA it can see (vmcnt(0) in front of the MFMA loop); hidden from it, the DMA is ordered
//    by hand -- vmcnt counts a wave's vector-memory operations in issue order, so "my pieces have landed" is
//    `s_waitcnt vmcnt(8)` (8 = the output stores the wave issued after them), then the step's barrier.
//  * The ring is 3 GROUPS of 4 contiguous planes (+ a 512-byte pad that takes the tail lanes of the 23rd 1-KiB
//    piece): the 1440 16-byte chunks of a step are 22.5 wave-pieces, 6 per wave, instead of ring2's 8 loads per
//    thread of which 30 % were idle slots.  Step k reads the last two planes of group k - 1 and the four of group
//    k while group k + 1 is in flight.
//  * LDS fragment addresses are (per-lane base of the k-step, 5 VGPRs) + (group base, 10 v_add per step) + an
//    immediate: ring2 added a scalar ring-slot base to every one of its 120 reads per step.
//  * The input transform (segmi_in_affine: the producer's BatchNorm-apply + PReLU) runs IN PLACE on the landed
//    group: every thread reads back the chunks its own lanes fetched (its own vmcnt orders that, no barrier),
//    transforms and writes them back; padding chunks are redirected to a dump slot (border workgroups only).
//  * Stores address with a per-lane offset computed once + the plane offset as the instruction's scalar soffset.
//
// What it is bound by (round 4, scripts/ring3_diag.py with SEGMI_RING3_DBG; 8 x 128^3 x 16, cold, alone): full 303 us;
// MFMA loop off 288; staging DMA alone 176 (3.1 TB/s of input), stores alone 125 (4.3 TB/s), both 289: the memory
// phases ADD and the MFMA loop (183 us alone) hides under them -- the launch is bound by what each CU pulls through its
// vector-memory pipe (~17-20 GB/s per CU here, 23 in an element-wise kernel), L2-side bytes, the 1.41x halo of the
// 8 x 16 tile included: 1.29 GB, i.e. ~220 us at the element-wise rate.  DESIGN.md section 6 lists the variants that
// were built and measured on the way (all correct, none faster): a 6-pair ring of 2-plane steps with four pairs in
// flight, an L2 warm-up two steps ahead, lane-contiguous 16-byte stores through an LDS transpose, a 512-thread
// producer / consumer form on 16 x 16 columns, and this very kernel with 8 waves on 16 x 16 columns (one workgroup per
// CU, y halo 1.27 x instead of 1.41 x: 290 vs 277 us plain, 341 vs 342 with the input transform, 349 vs 343 / 376 vs
// 376 as input gradient without / with the sums; step 4.64 vs 4.61 ms) -- fewer halo bytes do not pay for the lockstep
// of a single workgroup per CU.
// (A hazard met on the way: `buffer_store_dwordx4` with an SGPR soffset followed by a VALU write of its data
// registers stored the NEW values in some lanes; hipcc pads that hazard only for stores without a register soffset.)
//
// Numerics: taps in ring2's (= every MFMA conv kernel's) k order, then + bias, then the residual: a layer gives the
// same bits whichever kernel family its shape selects.
#pragma once
#include "conv_ring_impl.h"

namespace segmi {

struct Ring3Geom {
  static constexpr int TD = 4, TH = 8, TW = 16, HH = TH + 2, HW = TW + 2;
  static constexpr int ROWB = 32;                          // 16 channels x 2 bytes
  static constexpr int PLANE_ROWS = HH * HW;               // 180
  static constexpr int PLANE_B = PLANE_ROWS * ROWB;        // 5760
  static constexpr int PLANE_CH = PLANE_ROWS * 2;          // 360 16-byte chunks
  static constexpr int GROUP_CH = TD * PLANE_CH;           // 1440
  static constexpr int GROUP_B = GROUP_CH * 16 + 512;      // 23552: + the tail lanes of the 23rd DMA piece
  static constexpr int NG = 3;
  static constexpr int RING_B = NG * GROUP_B;              // 70656
  static constexpr int NPIECE = (GROUP_CH + 63) / 64;      // 23 wave-pieces of 1 KiB per group
  static constexpr int NDMA = (NPIECE + 3) / 4;            // 6 per wave (wave 3: 5)
  // behind the ring: input-transform scale / shift [2][16] f32, MODE-4 parameters [4][16] f32, 256 dump slots
  static constexpr int TFS_OFF = RING_B, BPRM_OFF = TFS_OFF + 128, DUMP_OFF = BPRM_OFF + 256;
  static constexpr int LDS_BYTES = DUMP_OFF + 256 * 16;
};

// one 1-KiB piece: lane l's 16 bytes from (rs base + voff + soff) land at lds_addr + 16 l; a lane whose
// voff + soff is out of range writes zeros.  M0 carries the LDS address (saved / restored: the compiler owns M0).
__device__ __forceinline__ void ring3_dma16(const __amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned voff,
                                            unsigned soff) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
      : "memory");
}

// MODE as ring2: bit 0 = PReLU, bit 1 = BatchNorm statistics; 4 = plain + the BatchNorm-backward sums of the layer
// the output gradient flows into (ConvParams::bpart).
template <int MODE>
__global__ __launch_bounds__(256, 2) void conv_ring3_kernel(ConvParams p) {
  using G = Ring3Geom;
  using T = bf16_t;
  constexpr int J = 5;                       // k-steps per kd (two taps per k-step, the 10th half-k-step is a zero weight)
  constexpr int NIT = 6 * J;
  constexpr unsigned kOob = 0x80000000u;     // + any soffset < 2^31: out of range, no 32-bit wrap
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;

  // XCD-aware workgroup -> column map (see ring2)
  int t = blockIdx.x;
  if (p.xcd) t = (t & 7) * (gridDim.x >> 3) + (t >> 3);
  const int seg = t % p.tz; t /= p.tz;
  const int txi = t % p.tx; t /= p.tx;
  const int tyi = t % p.ty;
  const int n = t / p.ty;
  const int oy0 = tyi * G::TH, ox0 = txi * G::TW;
  const int nt0 = blockIdx.y;
  const int total_steps = (p.Do + G::TD - 1) / G::TD;
  const int seg_steps = (total_steps + p.tz - 1) / p.tz;
  const int z0 = seg * seg_steps * G::TD;
  const int nsteps_z = total_steps - seg * seg_steps < seg_steps ? total_steps - seg * seg_steps : seg_steps;

  constexpr bool PLAIN = MODE == 0;
  constexpr bool BSUM = MODE == 4;
  constexpr bool has_alpha = (MODE & 1) != 0;
  constexpr bool want_stats = (MODE & 2) != 0;
  const T* resp = (const T*)p.res;
  // identity residual (out = conv(x) + x): the rows are the centre plane of the ring
  const bool res_in = resp && p.res == p.in && p.ldr == p.ldi && p.Cin == p.Cout;
  const bool res_ext = resp && !res_in;

  // ---- weights -> registers (gathered from the standard pack: tap T sits in k-step T/2 at lane group
  // (T&1)*2 + (g&1)); the 10th half-k-step is a zero weight
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
        wreg[kd][j] = *reinterpret_cast<const frag_t*>(
            (const char*)p.wfrag + (((int64_t)sp * p.ntiles_total + nt0) * 64 + gp * 16 + r) * 16);
      }
    }
  // per-lane part of the voxel-fragment address of k-step j (inside a plane, first row of the wave); the spare
  // half-k-step points at the centre tap (kh = kw = 1)
  int lb[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    int t9 = 2 * j + (g >> 1);
    if (t9 > 8) t9 = 8;
    lb[j] = ((t9 / 3) * G::HW + t9 % 3 + r) * G::ROWB + (g & 1) * 16 + wave * 2 * G::HW * G::ROWB;
  }

  // ---- staging descriptors: this wave's NDMA pieces of a group; chunk i = 64 (wave + 4 mi) + lane of the group's
  // 1440: plane i / 360, row (i % 360) / 2, channel half i & 1.  goff = plane * plane bytes + in-plane offset
  // (or out of range); the group's first plane is the instruction's scalar soffset.
  const int64_t plane_stride = (int64_t)p.Hi * p.Wi * p.ldi * (int64_t)sizeof(T);
  const unsigned plane_b32 = (unsigned)plane_stride;
  const char* img = (const char*)p.in + (int64_t)n * p.Di * plane_stride;
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (int)((unsigned)p.Di * plane_b32), 0x00020000);
  unsigned goff[G::NDMA];
  unsigned gpl = 0;          // plane index (2 bits) of each of this thread's chunks, bit 3 of the nibble: padding in x / y
  bool xy_inside = true;     // no chunk of this thread is x / y padding
#pragma unroll
  for (int mi = 0; mi < G::NDMA; ++mi) {
    const int i = 64 * (wave + 4 * mi) + lane;
    const int pl = i / G::PLANE_CH, rem = i % G::PLANE_CH;
    const int row = rem >> 1, ch = rem & 1;
    const int y = oy0 - 1 + row / G::HW, x = ox0 - 1 + row % G::HW;
    const bool ok = i < G::GROUP_CH && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    goff[mi] = ok ? (unsigned)pl * plane_b32 + (unsigned)((y * p.Wi + x) * p.ldi * (int)sizeof(T) + ch * 16) : kOob;
    gpl |= (unsigned)((pl & 3) | (ok ? 0 : 8)) << (4 * mi);
    xy_inside = xy_inside && (ok || i >= G::GROUP_CH);
  }
  const bool wave_inside = __builtin_amdgcn_readfirstlane(__all(xy_inside) ? 1 : 0) != 0;   // no padding chunk in this wave
  // LDS byte address of the dynamic segment (static __shared__ words of the finalisation tail precede it): the DMA
  // takes absolute LDS addresses in M0
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned lds_wave = (unsigned)(1024 * wave);        // + 4096 mi + group base: LDS address of a piece
  const bool last_piece = wave + 4 * (G::NDMA - 1) < G::NPIECE;   // wave 3 has no 6th piece
  auto dma_group = [&](unsigned gbase, int zfirst) {
    const unsigned soff = (unsigned)zfirst * plane_b32;
#pragma unroll
    for (int mi = 0; mi < G::NDMA; ++mi)
      if (mi < G::NDMA - 1 || last_piece) ring3_dma16(rs_in, lds0 + gbase + lds_wave + 4096u * mi, goff[mi], soff);
  };

  // optional input transform: scale / shift in LDS behind the ring, re-read at every transform phase
  const bool in_tf = p.in_scale != nullptr;
  const bool in_act = in_tf && p.in_alpha != nullptr;
  float* tfs = reinterpret_cast<float*>(smem + G::TFS_OFF);   // [2][16]
  if (in_tf && tid < 32) tfs[tid] = tid < 16 ? p.in_scale[tid] : p.in_shift[tid - 16];
  float in_alpha = in_act ? *p.in_alpha : 0.f;
  touch_s(in_alpha);
  const bool in_act01 = in_act && in_alpha >= 0.f && in_alpha <= 1.f;
  // transform of this thread's chunks of a landed group, in place.  zrem = planes of the group inside the volume
  // (>= 4: all); padding chunks (x / y outside, or plane >= zrem) stay zero: their write goes to a dump slot.
  auto transform_group = [&](unsigned gbase, int zrem) {
    const int tfo = (lane & 1) * 32;
    float sc[8], sh[8];
    const char* tb = reinterpret_cast<const char*>(tfs);
#pragma unroll
    for (int e = 0; e < 8; e += 4) {
      *reinterpret_cast<f32x4*>(&sc[e]) = *reinterpret_cast<const f32x4*>(tb + tfo + e * 4);
      *reinterpret_cast<f32x4*>(&sh[e]) = *reinterpret_cast<const f32x4*>(tb + 64 + tfo + e * 4);
    }
    const unsigned mine = gbase + lds_wave + 16u * lane;
    const bool fast = zrem >= G::TD && wave_inside;
    // pieces 0 .. NDMA - 2 of every wave, then the last piece (waves 0 - 2 only): loads first, then transform + store
    auto run = [&](auto tf, auto dst) {
      frag_t raw[G::NDMA - 1];
#pragma unroll
      for (int mi = 0; mi < G::NDMA - 1; ++mi) raw[mi] = *reinterpret_cast<const frag_t*>(smem + mine + 4096u * mi);
      frag_t rawl = frag_t{0u, 0u, 0u, 0u};
      if (last_piece) rawl = *reinterpret_cast<const frag_t*>(smem + mine + 4096u * (G::NDMA - 1));
#pragma unroll
      for (int mi = 0; mi < G::NDMA - 1; ++mi)
        *reinterpret_cast<frag_t*>(smem + dst(mi, mine + 4096u * mi)) = tf(raw[mi]);
      // waves without a last piece write their (transformed zero) to their dump slot
      const unsigned al = last_piece ? dst(G::NDMA - 1, mine + 4096u * (G::NDMA - 1)) : (unsigned)(G::DUMP_OFF + tid * 16);
      *reinterpret_cast<frag_t*>(smem + al) = tf(rawl);
    };
    auto with_dst = [&](auto tf) {
      if (fast) run(tf, [&](int, unsigned a) { return a; });
      else run(tf, [&](int mi, unsigned a) {
        const unsigned nib = (gpl >> (4 * mi)) & 15u;
        const bool keep = nib < 8u && (int)nib < zrem;
        return keep ? a : (unsigned)(G::DUMP_OFF + tid * 16);
      });
    };
    if (in_act01) with_dst([&](frag_t v) { return bn_prelu01_bf16x8(v, sc, sh, in_alpha); });
    else with_dst([&](frag_t v) { return bn_prelu_bf16x8(v, sc, sh, in_alpha, in_act); });
  };

  // ---- prologue: planes z0 - 1, z0 -> positions 2, 3 of group 2; planes z0 + 1 .. z0 + 4 -> group 0
  {
    // the two planes below the first group: per-lane plane validity (z0 - 1 may be -1), exec-masked tail
    const unsigned dst0 = 2u * G::GROUP_B + 2u * G::PLANE_B;
#pragma unroll
    for (int mi = 0; mi < 3; ++mi) {
      const int i = 64 * (wave + 4 * mi) + lane;
      const int pl = i / G::PLANE_CH, rem = i % G::PLANE_CH;
      const int row = rem >> 1, ch = rem & 1;
      const int z = z0 - 1 + pl, y = oy0 - 1 + row / G::HW, x = ox0 - 1 + row % G::HW;
      const bool ok = (unsigned)z < (unsigned)p.Di && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
      const unsigned vo = ok ? (unsigned)z * plane_b32 + (unsigned)((y * p.Wi + x) * p.ldi * (int)sizeof(T) + ch * 16) : kOob;
      if (i < 2 * G::PLANE_CH) ring3_dma16(rs_in, lds0 + dst0 + 1024u * (wave + 4 * mi), vo, 0u);
    }
    dma_group(0u, z0 + 1);
    if (in_tf) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // tfs visible; every thread transforms its own chunks
      const int tfo = (lane & 1) * 32;
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = tfs[tfo / 4 + e]; sh[e] = tfs[16 + tfo / 4 + e]; }
#pragma unroll
      for (int mi = 0; mi < 3; ++mi) {
        const int i = 64 * (wave + 4 * mi) + lane;
        const int pl = i / G::PLANE_CH, row = (i % G::PLANE_CH) >> 1;
        const int z = z0 - 1 + pl, y = oy0 - 1 + row / G::HW, x = ox0 - 1 + row % G::HW;
        const bool ok = i < 2 * G::PLANE_CH && (unsigned)z < (unsigned)p.Di && (unsigned)y < (unsigned)p.Hi &&
                        (unsigned)x < (unsigned)p.Wi;
        if (ok) {
          frag_t* q = reinterpret_cast<frag_t*>(smem + dst0 + 16u * i);
          *q = bn_prelu_bf16x8(*q, sc, sh, in_alpha, in_act);
        }
      }
      transform_group(0u, p.Di - (z0 + 1));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }

  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nt0 * 16 + 4 * g);
  touch_v(bias4);
  float alpha = has_alpha ? *p.alpha : 0.f;
  touch_s(alpha);
  const bool alpha01 = has_alpha && alpha >= 0.f && alpha <= 1.f;
  f32x4 ssum = f32x4{0.f, 0.f, 0.f, 0.f}, ssq = ssum;
  // MODE 4 (see ring2): z = x*sc + sh; sum dz*xhat accumulated as sum dz*(x - mean), scaled per workgroup row
  f32x4 bs0 = ssum, bs1 = ssum, bs2 = ssum;
  float* bprm = reinterpret_cast<float*>(smem + G::BPRM_OFF);    // [4][16]: mean, invstd, sc, sh
  float balpha = 1.f;
  if constexpr (BSUM) {
    if (tid < 64) {
      const int which = tid / 16, ch = nt0 * 16 + tid % 16;
      const float mean = p.bmean[ch], istd = p.binvstd[ch];
      const float sc = istd * (p.bgamma ? p.bgamma[ch] : 1.f);
      bprm[tid] = which == 0 ? mean : which == 1 ? istd : which == 2 ? sc : fmaf(-mean, sc, p.bbeta ? p.bbeta[ch] : 0.f);
    }
    balpha = p.balpha ? *p.balpha : 1.f;
    touch_s(balpha);
    __syncthreads();
  }
  const bool b_has_alpha = BSUM && p.balpha != nullptr;
  const T* bxp = (const T*)p.bx;
  T* outp = (T*)p.out;
  const int co = nt0 * 16 + 4 * g;
  // per-lane byte offsets of the wave's two output rows inside an output plane (out of range where the lane has no
  // voxel); the plane offset is the store's / load's scalar soffset, planes beyond the volume fall out of range
  unsigned o_off[2], r_off[2], b_off[2];
  bool row_ok[2];
#pragma unroll
  for (int ro = 0; ro < 2; ++ro) {
    const int oy = oy0 + 2 * wave + ro, ox = ox0 + r;
    row_ok[ro] = oy < p.Ho && ox < p.Wo;
    o_off[ro] = row_ok[ro] ? (unsigned)((oy * p.Wo + ox) * p.ldo + co) * (unsigned)sizeof(T) : kOob;
    r_off[ro] = row_ok[ro] ? (unsigned)((oy * p.Wo + ox) * p.ldr + co) * (unsigned)sizeof(T) : kOob;
    b_off[ro] = row_ok[ro] ? (unsigned)((oy * p.Wo + ox) * p.ldbx + co) * (unsigned)sizeof(T) : kOob;
  }
  const int64_t oplane = (int64_t)p.Ho * p.Wo * p.ldo, rplane = (int64_t)p.Ho * p.Wo * p.ldr;
  const int64_t bplane = (int64_t)p.Ho * p.Wo * p.ldbx;
  const unsigned oplane_b = (unsigned)(oplane * (int64_t)sizeof(T)), rplane_b = (unsigned)(rplane * (int64_t)sizeof(T)),
                 bplane_b = (unsigned)(bplane * (int64_t)sizeof(T));
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(outp + (int64_t)n * p.Do * oplane), 0, (int)((unsigned)p.Do * oplane_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(res_ext ? resp + (int64_t)n * p.Do * rplane : (const T*)p.out), 0, res_ext ? (int)((unsigned)p.Do * rplane_b) : 0,
      0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(BSUM ? bxp + (int64_t)n * p.Do * bplane : (const T*)p.out), 0, BSUM ? (int)((unsigned)p.Do * bplane_b) : 0,
      0x00020000);

  // group bases: step k reads group (k - 1) % 3 (planes 2, 3) and group k % 3, group (k + 1) % 3 is in flight
  unsigned g_prev = 2u * G::GROUP_B, g_cur = 0u, g_next = (unsigned)G::GROUP_B;
  for (int step = 0; step < nsteps_z; ++step) {
    const int zb = step * G::TD;
    const bool more = step + 1 < nsteps_z;
    // ---- the next group: planes z0 + zb + 5 .. + 8, straight into the ring
    if (more && !(p.dbg & 2)) dma_group(g_next, z0 + zb + 5);
    // residual rows of this step's outputs (external residual: the gradient sums of the backward chain)
    typedef Raw4<T>::type raw4_t;
    raw4_t resv[4][2];
    if (res_ext) {
#pragma unroll
      for (int zi = 0; zi < 4; ++zi)
#pragma unroll
        for (int ro = 0; ro < 2; ++ro)
          resv[zi][ro] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, r_off[ro], (unsigned)(z0 + zb + zi) * rplane_b, 0);
    }
    raw4_t bxv[BSUM ? 4 : 1][2];
    auto fetch_bx = [&](int zi) {
#pragma unroll
      for (int ro = 0; ro < 2; ++ro)
        bxv[BSUM ? zi : 0][ro] =
            __builtin_amdgcn_raw_buffer_load_b64(rs_bx, b_off[ro], (unsigned)(z0 + zb + zi) * bplane_b, 0);
    };
    if constexpr (BSUM) { fetch_bx(0); fetch_bx(1); fetch_bx(2); fetch_bx(3); }

    // ---- compute: input plane c (z = z0 + zb - 1 + c): c = 0, 1 -> planes 2, 3 of the previous group; c = 2 .. 5 ->
    // the current group.  Addresses = per-lane base of (group, k-step) + an immediate.
    int bp[J], bc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { bp[j] = lb[j] + (int)g_prev; bc[j] = lb[j] + (int)g_cur; }
    f32x4 acc[4][2];
    constexpr int PD = MODE == 4 ? 3 : 4;
    frag_t a[PD + 1][2];
    auto issue = [&](int it, frag_t (&dst)[2]) {
      const int c = it / J, j = it % J;
      const char* base = smem + (c < 2 ? bp[j] + (c + 2) * G::PLANE_B : bc[j] + (c - 2) * G::PLANE_B);
      dst[0] = *reinterpret_cast<const frag_t*>(base);
      dst[1] = *reinterpret_cast<const frag_t*>(base + G::HW * G::ROWB);
    };
#pragma unroll
    for (int q = 0; q < PD; ++q) issue(q, a[q]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it + PD < NIT) issue(it + PD, a[(it + PD) % (PD + 1)]);
      __builtin_amdgcn_sched_barrier(0);
      const int c = it / J, j = it % J;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int zi = c - kd;
        if (zi >= 0 && zi < 4) {
          const bool first = kd == 0 && j == 0;     // plane c = zi, k-step 0 opens the sum (literal 0 as C)
          const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[zi][0] = mma16<T>(wreg[kd][j], a[it % (PD + 1)][0], first ? zero : acc[zi][0]);
          acc[zi][1] = mma16<T>(wreg[kd][j], a[it % (PD + 1)][1], first ? zero : acc[zi][1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the landed group's input transform, in place (each thread its own chunks: its own vmcnt orders it)
    if (in_tf && more && !(p.dbg & 16)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      transform_group(g_next, p.Di - (z0 + zb + 5));
    }
    // identity residual outside the chain (PReLU epilogues): the centre plane of the ring
    if (res_in) {
#pragma unroll
      for (int zi = 0; zi < 4; ++zi)
#pragma unroll
        for (int ro = 0; ro < 2; ++ro) {
          const int c = zi + 1;
          const unsigned pb = c < 2 ? g_prev + (c + 2) * G::PLANE_B : g_cur + (c - 2) * G::PLANE_B;
          resv[zi][ro] = *reinterpret_cast<const raw4_t*>(
              smem + pb + ((2 * wave + ro + 1) * G::HW + r + 1) * G::ROWB + (4 * g) * (int)sizeof(T));
        }
    }
    if (resp) {
#pragma unroll
      for (int zi = 0; zi < 4; ++zi)
#pragma unroll
        for (int ro = 0; ro < 2; ++ro) touch_v(resv[zi][ro]);
    }
    f32x4 bsc4, bsh4, bmean4;
    if constexpr (BSUM) {
#pragma unroll
      for (int zi = 0; zi < 4; ++zi)
#pragma unroll
        for (int ro = 0; ro < 2; ++ro) touch_v(bxv[zi][ro]);
      bsc4 = *reinterpret_cast<const f32x4*>(bprm + 2 * 16 + 4 * g);
      bsh4 = *reinterpret_cast<const f32x4*>(bprm + 3 * 16 + 4 * g);
      bmean4 = *reinterpret_cast<const f32x4*>(bprm + 4 * g);
    }
    // ---- epilogue of this step: 8 stores per wave, always issued (out-of-range ones are dropped): the end-of-step
    // wait counts on exactly these being the wave's youngest memory operations
#pragma unroll
    for (int zi = 0; zi < 4; ++zi) {
      const int oz = z0 + zb + zi;
      const unsigned opoff = (unsigned)oz * oplane_b;        // wave-uniform: the store's soffset
      const bool zin_out = oz < p.Do;
#pragma unroll
      for (int ro = 0; ro < 2; ++ro) {
        f32x4 v = acc[zi][ro] + bias4;
        const bool valid = zin_out & row_ok[ro];
        if (want_stats) {
          const f32x4 vm = valid ? v : f32x4{0.f, 0.f, 0.f, 0.f};
          ssum += vm;
          ssq += vm * vm;
        }
        if (has_alpha) {
          if (alpha01) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], alpha * v[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
          }
        }
        if (resp) v += Raw4<T>::cvt(resv[zi][ro]);
        u32x2 o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, (p.dbg & 4) ? kOob : o_off[ro], opoff, 0);
        if constexpr (BSUM) {
          // the sums are taken of the STORED gradient (bf16-rounded), as the separate pass reads it
          f32x4 d = Raw4<T>::cvt(o);
          if (!valid) d = f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x4 xr = Raw4<T>::cvt(bxv[zi][ro]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(xr[e], bsc4[e], bsh4[e]);
            float dz = d[e];
            if (b_has_alpha && !(z > 0.f)) { bs2[e] = fmaf(d[e], z, bs2[e]); dz = balpha * d[e]; }
            bs0[e] += dz;
            bs1[e] = fmaf(dz, xr[e] - bmean4[e], bs1[e]);
          }
        }
      }
    }
    // rotate the groups; every wave's DMA pieces of the next group have landed (all but its 8 youngest memory
    // operations -- the stores above -- are complete), its transform writes are done, then the barrier
    const unsigned gp = g_prev;
    g_prev = g_cur; g_cur = g_next; g_next = gp;
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  if constexpr (BSUM) {
    float* red = reinterpret_cast<float*>(smem);  // [wave][3][16]  (ring no longer needed)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a0 = row16_sum(bs0[e]);
      const float a1 = row16_sum(bs1[e]);
      const float a2 = row16_sum(bs2[e]);
      if (r == 0) {
        red[(wave * 3 + 0) * 16 + 4 * g + e] = a0;
        red[(wave * 3 + 1) * 16 + 4 * g + e] = a1;
        red[(wave * 3 + 2) * 16 + 4 * g + e] = a2;
      }
    }
    __syncthreads();
    if (tid < 48) {
      const int which = tid / 16, ch = tid % 16;
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sacc += red[(w * 3 + which) * 16 + ch];
      if (which == 1) sacc *= bprm[16 + ch];   // sum dz*(x - mean) -> sum dz*xhat
      fin_store(&p.bpart[((int64_t)blockIdx.x * 3 + which) * p.Cout + nt0 * 16 + ch], sacc);
    }
    fin_tail_run<BnBwdFin, 256, offsetof(ConvParams, ft), offsetof(ConvParams, bbfin)>(p.bpart, smem);
  }
  if (want_stats) {
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
      fin_store(&p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + nt0 * 16 + ch], sacc);
    }
    fin_tail_run<BnFin, 256, offsetof(ConvParams, ft), offsetof(ConvParams, bfin)>(p.stats, smem);
  }
}

static inline bool conv_ring3_ok(const ConvParams& p) {
  return conv_ring3_shape_ok(p.Cin, p.Cout, p.in, p.out, p.Di, p.Hi, p.Wi, p.ldi, p.Do, p.Ho, p.Wo, p.ldo, p.ldr, p.ldbx);
}

template <int MODE>
static int launch_conv_ring3_k(ConvParams p, hipStream_t st) {
  using G = Ring3Geom;
  p.tz = conv_ring_zsplit(SEGMI_BF16, p.Cin, 3, 1, p.N, p.Do, p.Ho, p.Wo);
  // diagnostics (timing probes, WRONG results): SEGMI_RING3_DBG bits: 2 = no staging DMA in the step loop, 4 = stores
  // dropped, 8 = no MFMA loop, 16 = no input transform
  static const int dbg3 = getenv("SEGMI_RING3_DBG") ? atoi(getenv("SEGMI_RING3_DBG")) : 0;
  p.dbg = dbg3;
  // ring2's XCD-aware column map (XCD k walks the k-th eighth of the columns: neighbours' halos meet in one L2) is used
  // for launches of at most one round of workgroups (<= 512: the 64^3 layers -- inference 41.4 vs 42.3 ms per volume
  // with / without) and NOT for larger ones: on the full-resolution 8 x 128^3 launch (1024 workgroups, two rounds) it
  // costs 10 % (0.300 -> 0.270 ms inside the training step, step 5.11 -> 5.02 ms, alternating runs; memory-only
  // diagnostic 289 -> 228 us).  The halo re-reads it saves there were Infinity-Cache hits, and a CU's vector-memory
  // pipe -- the bound of this kernel -- does not care where a line comes from.  SEGMI_RING3_XCD = 0 / 1 forces it.
  static const int xcd_env = getenv("SEGMI_RING3_XCD") ? atoi(getenv("SEGMI_RING3_XCD")) : -1;
  p.ty = cdiv(p.Ho, G::TH);
  p.tx = cdiv(p.Wo, G::TW);
  dim3 grid((unsigned)(p.N * p.ty * p.tx * p.tz), (unsigned)(p.Cout / 16));
  const int xcd = xcd_env >= 0 ? xcd_env : (grid.x <= 512 ? 1 : 0);
  p.xcd = xcd != 0 && grid.x % 8 == 0;
  constexpr bool kStats = (MODE & 2) != 0, kBsum = MODE == 4;
  p.fin_on = p.fin_on && (kStats || kBsum);
  (void)fin_tail_arm(p, grid, 256, (kBsum ? 3 : 2) * p.Cout, G::LDS_BYTES);   // LDS: the ring is larger than the tail's need
  auto kern = conv_ring3_kernel<MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              G::LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, 256, G::LDS_BYTES, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_fwd(ring3)");
  return SEGMI_OK;
}

static int launch_conv_ring3(const ConvParams& p, hipStream_t st) {
  if (p.bpart && !p.alpha && !p.stats && p.Cout == 16) return launch_conv_ring3_k<4>(p, st);
  switch ((p.alpha ? 1 : 0) | (p.stats ? 2 : 0)) {
    case 0: return launch_conv_ring3_k<0>(p, st);
    case 1: return launch_conv_ring3_k<1>(p, st);
    case 2: return launch_conv_ring3_k<2>(p, st);
    default: return launch_conv_ring3_k<3>(p, st);
  }
}

}  // namespace segmi
