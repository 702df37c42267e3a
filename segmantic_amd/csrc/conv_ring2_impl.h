// conv_ring2_impl.h -- second formulation of the z-marching ring (conv_ring_impl.h) for bf16
// single-chunk k3 s1 layers: the network's wide 16 -> 16 and 32 -> 32 convolutions.
//
// conv_ring_mfma_kernel gives a wave one output plane x 8 rows and reads one 1-KiB LDS fragment per
// MFMA and output tile: for Cout = 16 that is 288 B/clk/CU of LDS reads against the 128 B/clk the
// LDS delivers (<= 44 % MFMA utilisation, 27 % measured), and its 32-channel variants spill.
// Here a wave owns 2 y-rows x ALL 4 output planes of the step, so a voxel fragment of input plane
// p feeds the taps kd = 0,1,2 of output planes p+1, p, p-1 from registers (60 fragment reads per
// 120 MFMAs for 16 channels instead of 126 per 112), and every weight fragment lives in registers
// for the whole column: 15 fragments (60 VGPRs) for 16 -> 16, 54 fragments (216 VGPRs, one
// workgroup per CU -- the 144 KB ring allows no more anyway) for 32 -> 32.
//
// k-steps are organised per kd so that a voxel fragment is independent of kd:
//   CK = 16 (two taps per k-step): 5 k-steps per kd pairing (kh,kw) taps (0,1)(2,3)(4,5)(6,7)(8,-);
//            the fragments are gathered once from the standard pack (tap T sits in k-step T/2 at
//            lane group (T&1)*2 + (g&1)), the 10th slot is a zero weight;
//   CK = 32 (one tap per k-step): 9 k-steps per kd, the standard pack order.
// Index arithmetic: every per-lane quantity is a 32-bit offset inside a plane computed once, every
// per-step quantity a wave-uniform plane pointer -- no integer multiplies in the step loop (the
// first version spent more issue cycles on 64-bit voxel addresses than on its MFMAs).
#pragma once
#include "conv_ring_impl.h"

// Time-split diagnostic (build with -DSEGMI_RING2_DIAG, run with SEGMI_RING2_DBG=bits: 1 = no staging
// loads, 2 = no stores, 4 = no MFMA loop, 8 = no commit, 16 = no epilogue, 32 = no end-of-step barrier
// (only meaningful together with 4 + 8); scripts/ring2_diag.py).  Compiled out of the product build:
// the extra live flag costs the 32 -> 64 variant its last free registers.
#ifdef SEGMI_RING2_DIAG
#define RING2_DBG(p, bit) (((p).dbg & (bit)) != 0)
#else
#define RING2_DBG(p, bit) false
#endif

namespace segmi {

// MODE 4 = PLAIN + the BatchNorm-backward sums of the layer the output gradient flows into
// (ConvParams::bpart; see there).
// MODE: bit 0 = PReLU, bit 1 = BatchNorm statistics -- compile-time, because every epilogue
// instruction is an issue turn of the wave.  0 ("PLAIN": the two full-resolution launches of a
// training step and every input-gradient launch) is bias + residual + convert only, 1 the
// inference layers (folded BatchNorm + PReLU), 2 the training forward of a conv in front of a
// BatchNorm, 3 both.
template <typename T, int CK, int NT, int MODE>
__global__ __launch_bounds__(256, CK == 16 ? 2 : 1) void conv_ring2_kernel(ConvParams p) {
  static_assert(sizeof(T) == 2, "bf16 only");
  using G = RingGeom<T, CK>;
  constexpr int J = G::SPT == 2 ? 5 : 9;     // k-steps per kd
  constexpr int NIT = 6 * J;                 // (input plane, k-step) iterations per step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  constexpr int kDumpOff = G::LDS_BYTES + 6 * CK * 4;     // 256 x 16 bytes nobody reads (see the commit loop)

  // XCD-aware workgroup -> column map: XCD k (= blockIdx % 8) takes the k-th eighth of the columns
  // in raster order, so that x/y-neighbours share one L2 and their halos are fetched once.  L2 fetch
  // of the 8 x 128^3 x 16 launch: 1.41x -> 1.02x of the input bytes.  Alone the launch is 1-2 % slower
  // (361 -> 369 us: the re-fetches were MALL hits), inside the training step, where the weight-gradient
  // stream competes for the fabric, it is 3 % faster (step 5.83 -> 5.78 ms).  SEGMI_RING2_XCD=0: off.
  int t = blockIdx.x;
  if (p.xcd) t = (t & 7) * (gridDim.x >> 3) + (t >> 3);
  const int seg = t % p.tz; t /= p.tz;
  const int txi = t % p.tx; t /= p.tx;
  const int tyi = t % p.ty;
  const int n = t / p.ty;
  const int oy0 = tyi * G::TH, ox0 = txi * G::TW;
  const int nt0 = blockIdx.y * NT;
  const int total_steps = (p.Do + G::TD - 1) / G::TD;
  const int seg_steps = (total_steps + p.tz - 1) / p.tz;
  const int z0 = seg * seg_steps * G::TD;
  const int nsteps_z = total_steps - seg * seg_steps < seg_steps ? total_steps - seg * seg_steps
                                                                : seg_steps;

  // ---- weights -> registers
  frag_t wreg[3][J][NT];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        wreg[kd][j][jt] = frag_t{0u, 0u, 0u, 0u};
        if constexpr (G::SPT == 2) {
          const int t9 = 2 * j + (g >> 1);
          if (t9 <= 8) {
            const int tap = kd * 9 + t9;
            const int sp = tap >> 1, gp = (tap & 1) * 2 + (g & 1);
            wreg[kd][j][jt] = *reinterpret_cast<const frag_t*>(
                (const char*)p.wfrag + (((int64_t)sp * p.ntiles_total + nt0 + jt) * 64 + gp * 16 + r) * 16);
          }
        } else {
          wreg[kd][j][jt] = *reinterpret_cast<const frag_t*>(
              (const char*)p.wfrag + (((int64_t)(kd * 9 + j) * p.ntiles_total + nt0 + jt) * 64 + lane) * 16);
        }
      }
  // per-lane part of the voxel-fragment address of k-step j (inside a plane, row 0 of the wave)
  int laneoff[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    if constexpr (G::SPT == 2) {
      int t9 = 2 * j + (g >> 1);
      if (t9 > 8) t9 = 8;
      laneoff[j] = ((t9 / 3) * G::HW + t9 % 3 + r) * G::ROWB + (g & 1) * 16;
    } else {
      laneoff[j] = ((j / 3) * G::HW + j % 3 + r) * G::ROWB + g * 16;
    }
  }
  const int wrow = wave * 2 * G::HW * G::ROWB;   // the wave's first y-row

  // ---- staging descriptors: 32-bit offsets inside a plane, computed once
  constexpr int NLP = (G::PLANE_CHUNKS + 255) / 256;   // 16-byte chunks per thread per plane
  const char* inb = (const char*)p.in;
  const int64_t plane_stride = (int64_t)p.Hi * p.Wi * p.ldi * (int64_t)sizeof(T);
  const char* img = inb + (int64_t)n * p.Di * plane_stride;
  int g_off[NLP], l_off[NLP];
#pragma unroll
  for (int q = 0; q < NLP; ++q) {
    const int i = tid + 256 * q;
    const int row = i / G::CPR, ch = i % G::CPR;
    const int hy = row / G::HW, hx = row % G::HW;
    const int y = oy0 - 1 + hy, x = ox0 - 1 + hx;
    const bool ok = i < G::PLANE_CHUNKS && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    l_off[q] = i < G::PLANE_CHUNKS ? row * G::ROWB + ch * 16 : -1;
    g_off[q] = ok ? (y * p.Wi + x) * p.ldi * (int)sizeof(T) + ch * 16 : -1;
  }
  // optional input transform: scale / shift of this thread's channel chunk (every 16-byte chunk a
  // thread stages has the same channel offset: 256 % CPR == 0).  Kept in LDS behind the ring and
  // re-read at the start of every commit phase -- no registers held across the MFMA loop.
  const bool in_tf = p.in_scale != nullptr;
  const bool in_act = in_tf && p.in_alpha != nullptr;
  float* tfs = reinterpret_cast<float*>(smem + G::LDS_BYTES);   // [2][CK]
  if (in_tf && tid < 2 * CK) tfs[tid] = tid < CK ? p.in_scale[tid] : p.in_shift[tid - CK];
  float in_alpha = in_act ? *p.in_alpha : 0.f;
  touch_s(in_alpha);
  const int tf_ch = (tid % G::CPR) * 8;
  auto transform = [&](frag_t v) {
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = tfs[tf_ch + e]; sh[e] = tfs[CK + tf_ch + e]; }
    return bn_prelu_bf16x8(v, sc, sh, in_alpha, in_act);
  };
  const bool in_act01 = in_act && in_alpha >= 0.f && in_alpha <= 1.f;
  if (in_tf) __syncthreads();
  // prologue: planes z0-1 .. z0+4 -> ring slots 0 .. 5
#pragma unroll
  for (int pl = 0; pl < 6; ++pl) {
    const int z = z0 + pl - 1;
    const char* pp = img + (int64_t)z * plane_stride;
#pragma unroll
    for (int q = 0; q < NLP; ++q) {
      frag_t val = frag_t{0u, 0u, 0u, 0u};
      const bool inside = (unsigned)z < (unsigned)p.Di && g_off[q] >= 0;
      if (inside) val = *reinterpret_cast<const frag_t*>(pp + (unsigned)g_off[q]);
      if (in_tf && inside) val = transform(val);        // zero padding stays zero
      if (l_off[q] >= 0) *reinterpret_cast<frag_t*>(smem + pl * G::PLANE_B + l_off[q]) = val;
    }
  }
  __syncthreads();

  f32x4 bias4[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    bias4[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4[jt] = *reinterpret_cast<const f32x4*>(p.bias + (nt0 + jt) * 16 + 4 * g);
    touch_v(bias4[jt]);
  }
  constexpr bool PLAIN = MODE == 0;
  constexpr bool BSUM = MODE == 4;
  constexpr bool has_alpha = (MODE & 1) != 0;
  float alpha = has_alpha ? *p.alpha : 0.f;
  touch_s(alpha);
  constexpr bool want_stats = (MODE & 2) != 0;
  f32x4 ssum[NT], ssq[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) { ssum[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // MODE 4: sums of the BatchNorm backward of the layer behind the output; its parameters for the
  // lane's 4 channels sit in LDS behind the input-transform slots and are re-read every step
  f32x4 bs0[NT], bs1[NT], bs2[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) { bs0[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; bs1[jt] = bs0[jt]; bs2[jt] = bs0[jt]; }
  // z = xhat*gamma + beta = x*sc + sh with sc = invstd*gamma, sh = beta - mean*sc; sum dz*xhat is
  // accumulated as sum dz*(x - mean) and scaled by invstd per workgroup row
  float* bprm = tfs + 2 * CK;                       // [4][NT*16]: mean, invstd, sc, sh
  float balpha = 1.f;
  if constexpr (BSUM) {
    if (tid < 4 * NT * 16) {
      const int which = tid / (NT * 16), ch = nt0 * 16 + tid % (NT * 16);
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
  const T* resp = (const T*)p.res;
  // identity residual (out = conv(x) + x): the rows are the centre plane of the ring
  const bool res_in = resp && p.res == p.in && p.ldr == p.ldi && p.Cin == p.Cout;
  const int co = nt0 * 16 + 4 * g;
  // per-lane element offsets of the wave's two output rows inside an output plane
  unsigned o_off[2], r_off[2];
  bool row_ok[2];
#pragma unroll
  for (int ro = 0; ro < 2; ++ro) {
    const int oy = oy0 + 2 * wave + ro, ox = ox0 + r;
    row_ok[ro] = oy < p.Ho && ox < p.Wo;
    o_off[ro] = (unsigned)((oy * p.Wo + ox) * p.ldo + co) * (unsigned)sizeof(T);     // bytes inside a plane
    r_off[ro] = (unsigned)((oy * p.Wo + ox) * p.ldr + co) * (unsigned)sizeof(T);
  }
  const int64_t oplane = (int64_t)p.Ho * p.Wo * p.ldo, rplane = (int64_t)p.Ho * p.Wo * p.ldr;
  unsigned b_off[2];
#pragma unroll
  for (int ro = 0; ro < 2; ++ro)
    b_off[ro] = (unsigned)(((oy0 + 2 * wave + ro) * p.Wo + ox0 + r) * p.ldbx + co) * (unsigned)sizeof(T);
  const int64_t bplane = (int64_t)p.Ho * p.Wo * p.ldbx;

  // Plane addresses are a per-workgroup 64-bit base (loop-invariant) + a 32-bit product (the launcher
  // checks that one sample's tensors stay below 2^31 elements): the 64-bit products
  // ((n * Do + oz) * plane) and the ring slot's "% R" were dependent scalar chains of 8-14
  // instructions per plane in front of every load / store group.
  // Every global access of the step loop is a raw buffer operation over ONE SAMPLE of its tensor (the
  // launcher checks that a sample stays below 2^32 bytes): wave-uniform descriptor, 32-bit per-lane byte
  // offset = plane offset (scalar) + in-plane offset (per lane, computed once), and an out-of-range offset
  // where the lane or the plane is outside the tensor -- the load then returns zeros (the zero padding), the
  // store is dropped.  The first version guarded each access with an exec-mask branch: 79 s_and_saveexec /
  // 56 s_cbranch_execz / 82 s_or_b64 per step, a third of the loop's 750 scalar instructions.
  constexpr unsigned kOob = 0xFFFFFFFFu;
  const unsigned plane_b32 = (unsigned)plane_stride, oplane_b = (unsigned)(oplane * (int64_t)sizeof(T)),
                 rplane_b = (unsigned)(rplane * (int64_t)sizeof(T)), bplane_b = (unsigned)(bplane * (int64_t)sizeof(T));
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (int)((unsigned)p.Di * plane_b32), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(outp + (int64_t)n * p.Do * oplane), 0, (int)((unsigned)p.Do * oplane_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(resp ? resp + (int64_t)n * p.Do * rplane : (const T*)p.out), 0, resp ? (int)((unsigned)p.Do * rplane_b) : 0,
      0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(BSUM ? bxp + (int64_t)n * p.Do * bplane : (const T*)p.out), 0, BSUM ? (int)((unsigned)p.Do * bplane_b) : 0,
      0x00020000);
  int slot0 = 0;                                                               // zb % R
  auto ring_slot = [&](int c) { const int v = slot0 + c; return v >= G::R ? v - G::R : v; };
  // Staging loads: the 4 new planes step k + 1 computes from (z = z0 + 4k + 5 .. + 8) are committed to the
  // ring at the end of step k and issued at its start.  PF2 = issue them a whole step earlier into a second
  // set of 32 staging registers (the step body then exists twice, for static register indexing): tried in
  // round 3 for the 16 -> 16 forward variants (MODE 0 / 1: 252 / 256 VGPRs, no spill) because time splits of
  // the diagnostic build show the launch ADDING its memory time to its compute time (compute only 209 us,
  // + loads 104, + stores 66 of 361) -- and measured SLOWER: 382-389 vs 361 us alone, 0.338 vs 0.317 ms inside
  // the training step.  The loads are not late; left in the source, off.
  constexpr bool PF2 = false;
  auto issue_loads = [&](int k, frag_t (&dst)[G::TD][NLP]) {
    const int zbk = k * G::TD;
    const bool morek = k + 1 < nsteps_z;
#pragma unroll
    for (int pl = 0; pl < G::TD; ++pl) {
      const int z = z0 + zbk + 5 + pl;
      const unsigned poff = (unsigned)z * plane_b32;
      const bool zok = morek && z < p.Di && !RING2_DBG(p, 1);
#pragma unroll
      for (int q = 0; q < NLP; ++q)
        dst[pl][q] = __builtin_amdgcn_raw_buffer_load_b128(
            rs_in, (zok & (g_off[q] >= 0)) ? poff + (unsigned)g_off[q] : kOob, 0, 0);
    }
  };
  auto do_step = [&](const int step, frag_t (&stg)[G::TD][NLP], frag_t (&nxt)[G::TD][NLP]) {
    const int zb = step * G::TD;
    const bool more = step + 1 < nsteps_z;
    if constexpr (PF2) issue_loads(step + 1, nxt);     // committed at the end of the NEXT step
    else issue_loads(step, stg);                       // committed at the end of this step
    // residual rows of this step's outputs (4 planes x 2 rows), unless they are the input itself
    // (32-channel variant: 216 weight VGPRs leave no room for 32 more; it fetches them per plane
    // in the epilogue instead)
    constexpr bool PRE_RES = CK == 16;
    typename Raw4<T>::type resv[PRE_RES ? 4 : 1][2][NT];
    if (PRE_RES && resp && !res_in) {
#pragma unroll
      for (int zi = 0; zi < 4; ++zi) {
        const int oz = z0 + zb + zi;
        const unsigned poff = (unsigned)oz * rplane_b;
#pragma unroll
        for (int ro = 0; ro < 2; ++ro)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
            resv[zi][ro][jt] = __builtin_amdgcn_raw_buffer_load_b64(
                rs_res, ((oz < p.Do) & row_ok[ro]) ? poff + r_off[ro] + jt * 32 : kOob, 0, 0);
      }
    }
    // MODE 4: the BatchNorm's forward input at this step's output voxels.  Planes 0, 1 are fetched
    // here (under the MFMA loop), planes 2, 3 once the staging registers are free (under the
    // epilogue of planes 0, 1): all four up front do not fit the 256-register budget
    typename Raw4<T>::type bxv[BSUM ? 4 : 1][2][NT];
    auto fetch_bx = [&](int zi) {
      const int oz = z0 + zb + zi;
      const unsigned poff = (unsigned)oz * bplane_b;
#pragma unroll
      for (int ro = 0; ro < 2; ++ro)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
          bxv[BSUM ? zi : 0][ro][jt] = __builtin_amdgcn_raw_buffer_load_b64(
              rs_bx, ((oz < p.Do) & row_ok[ro]) ? poff + b_off[ro] + jt * 32 : kOob, 0, 0);
    };
    if constexpr (BSUM) { fetch_bx(0); fetch_bx(1); }
    // ---- compute: input plane c (z = zb - 1 + c) lives in ring slot (zb + c) % R
    // (no zero-initialisation: the first MFMA into an accumulator takes the literal 0 as C)
    f32x4 acc[4][2][NT];
    int pofs[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) pofs[c] = ring_slot(c) * G::PLANE_B + wrow;
    // software pipeline over the (plane, k-step) iterations, fragments PD iterations ahead (an
    // iteration is 2 - 6 MFMAs = 32 - 96 clk against >= 128 clk of loaded LDS latency; the depth is
    // what the variant's register budget allows)
    constexpr int PD = CK == 16 ? (MODE == 3 ? 3 : (MODE == 4 ? 2 : 4)) : (NT == 1 ? 4 : 2);
    frag_t a[PD + 1][2];
    auto issue = [&](int it, frag_t (&dst)[2]) {
      const int c = it / J, j = it % J;
      dst[0] = *reinterpret_cast<const frag_t*>(smem + pofs[c] + laneoff[j]);
      dst[1] = *reinterpret_cast<const frag_t*>(smem + pofs[c] + laneoff[j] + G::HW * G::ROWB);
    };
#pragma unroll
    for (int q = 0; q < PD; ++q) issue(q, a[q]);
    if (!RING2_DBG(p, 4))
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it + PD < NIT) issue(it + PD, a[(it + PD) % (PD + 1)]);
      __builtin_amdgcn_sched_barrier(0);
      const int c = it / J, j = it % J;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int zi = c - kd;
        if (zi >= 0 && zi < 4) {
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            const bool first = kd == 0 && j == 0;     // plane c = zi, k-step 0 opens the sum
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[zi][0][jt] = mma16<T>(wreg[kd][j][jt], a[it % (PD + 1)][0], first ? zero : acc[zi][0][jt]);
            acc[zi][1][jt] = mma16<T>(wreg[kd][j][jt], a[it % (PD + 1)][1], first ? zero : acc[zi][1][jt]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- write the prefetched planes into the free ring slots (zb+6 .. zb+9 mod R)
    if (more && !RING2_DBG(p, 8)) {
      // three copies of the commit loop behind wave-uniform branches (none / affine / affine +
      // PReLU) instead of per-element selects on the two runtime flags
      auto commit = [&](auto tf) {
#pragma unroll
        for (int pl = 0; pl < G::TD; ++pl) {
          const int slot = ring_slot(6 + pl);
          const bool zin = z0 + zb + 5 + pl < p.Di;
#pragma unroll
          for (int q = 0; q < NLP; ++q) {
            // no exec-mask branches: the transform runs on every lane and is selected away where the lane is
            // zero padding (the out-of-range load returned zeros); threads beyond the plane's chunks write a
            // dump slot behind the ring
            const frag_t raw = stg[pl][q];
            const frag_t t4 = tf(raw);
            const bool keep = zin & (g_off[q] >= 0);
            frag_t val;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) val[c4] = keep ? t4[c4] : raw[c4];
            const int la = l_off[q] >= 0 ? slot * G::PLANE_B + l_off[q] : kDumpOff + tid * 16;
            *reinterpret_cast<frag_t*>(smem + la) = val;
          }
        }
      };
      if (!in_tf) commit([](frag_t v) { return v; });
      else {
        // scale / shift of the thread's channel chunk: read from LDS once per commit phase (the voxel
        // fragments of the MFMA loop are dead here), not once per chunk -- 4 reads and one wait instead
        // of 32 reads in 8 dependent groups.  The step-dependent zero keeps the reads inside the loop.
        int tfo = tf_ch * 4;
        asm volatile("" : "+v"(tfo));
        float sc[8], sh[8];
        const char* tb = reinterpret_cast<const char*>(tfs);
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          *reinterpret_cast<f32x4*>(&sc[e]) = *reinterpret_cast<const f32x4*>(tb + tfo + e * 4);
          *reinterpret_cast<f32x4*>(&sh[e]) = *reinterpret_cast<const f32x4*>(tb + CK * 4 + tfo + e * 4);
        }
        if (in_act01) commit([&](frag_t v) { return bn_prelu01_bf16x8(v, sc, sh, in_alpha); });
        else if (in_act) commit([&](frag_t v) { return bn_prelu_bf16x8(v, sc, sh, in_alpha, true); });
        else commit([&](frag_t v) { return bn_prelu_bf16x8(v, sc, sh, 0.f, false); });
      }
    }
    // every staging register is dead from here on; say so on ALL control-flow paths (touch_v)
#pragma unroll
    for (int pl = 0; pl < G::TD; ++pl)
#pragma unroll
      for (int q = 0; q < NLP; ++q) touch_v(stg[pl][q]);
    auto lds_res = [&](int zi, typename Raw4<T>::type (&dst)[2][NT]) {
      // centre plane of output plane zi is input plane c = zi + 1
#pragma unroll
      for (int ro = 0; ro < 2; ++ro)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
          dst[ro][jt] = *reinterpret_cast<const typename Raw4<T>::type*>(
              smem + pofs[zi + 1] + ((ro + 1) * G::HW + r + 1) * G::ROWB +
              (co + jt * 16) * (int)sizeof(T));
    };
    if constexpr (PRE_RES) {
      if (res_in) {
#pragma unroll
        for (int zi = 0; zi < 4; ++zi) lds_res(zi, resv[zi]);
      }
      if (resp) {
#pragma unroll
        for (int zi = 0; zi < 4; ++zi)
#pragma unroll
          for (int ro = 0; ro < 2; ++ro)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) touch_v(resv[zi][ro][jt]);
      }
    }
    f32x4 bsc4[NT], bsh4[NT], bmean4[NT];
    if constexpr (BSUM) {
      fetch_bx(2); fetch_bx(3);
#pragma unroll
      for (int zi = 0; zi < 2; ++zi)
#pragma unroll
        for (int ro = 0; ro < 2; ++ro)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) touch_v(bxv[zi][ro][jt]);
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        bsc4[jt] = *reinterpret_cast<const f32x4*>(bprm + 2 * NT * 16 + jt * 16 + 4 * g);
        bsh4[jt] = *reinterpret_cast<const f32x4*>(bprm + 3 * NT * 16 + jt * 16 + 4 * g);
        bmean4[jt] = *reinterpret_cast<const f32x4*>(bprm + jt * 16 + 4 * g);
      }
    }
    // ---- epilogue of this step
    if (!RING2_DBG(p, 16))
#pragma unroll
    for (int zi = 0; zi < 4; ++zi) {
      const int oz = z0 + zb + zi;
      const unsigned opoff = (unsigned)oz * oplane_b;        // wave-uniform plane offset
      const bool zin_out = oz < p.Do && !RING2_DBG(p, 2);
      const int rz = PRE_RES ? zi : 0;
      if constexpr (!PRE_RES) {
        if (res_in) {
          lds_res(zi, resv[0]);
        } else if (resp) {
          const unsigned rpoff = (unsigned)oz * rplane_b;
#pragma unroll
          for (int ro = 0; ro < 2; ++ro)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
              resv[0][ro][jt] = __builtin_amdgcn_raw_buffer_load_b64(
                  rs_res, ((oz < p.Do) & row_ok[ro]) ? rpoff + r_off[ro] + jt * 32 : kOob, 0, 0);
        }
        if (resp) {
#pragma unroll
          for (int ro = 0; ro < 2; ++ro)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) touch_v(resv[0][ro][jt]);
        }
      }
#pragma unroll
      for (int ro = 0; ro < 2; ++ro)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          f32x4 v = acc[zi][ro][jt] + bias4[jt];
          const bool valid = zin_out & row_ok[ro];          // per lane; no branch: invalid lanes compute and drop
          if (want_stats) {
            const f32x4 vm = valid ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            ssum[jt] += vm;
            ssq[jt] += vm * vm;
          }
          if (has_alpha) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
          }
          if (resp) v += Raw4<T>::cvt(resv[rz][ro][jt]);
          u32x2 o;
          o[0] = pack_bf16x2(v[0], v[1]);
          o[1] = pack_bf16x2(v[2], v[3]);
          __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, valid ? opoff + o_off[ro] + jt * 32 : kOob, 0, 0);
          if constexpr (BSUM) {
            // the sums are taken of the STORED gradient (bf16-rounded), as the separate pass reads it; lanes
            // outside the tensor contribute d = 0 to all three
            f32x4 d = Raw4<T>::cvt(o);
            if (!valid) d = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 xr = Raw4<T>::cvt(bxv[zi][ro][jt]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = fmaf(xr[e], bsc4[jt][e], bsh4[jt][e]);
              float dz = d[e];
              if (b_has_alpha && !(z > 0.f)) { bs2[jt][e] = fmaf(d[e], z, bs2[jt][e]); dz = balpha * d[e]; }
              bs0[jt][e] += dz;
              // sum dz * (x - mean), not sum dz * x corrected by mean * sum dz afterwards: the latter cancels
              // badly when |mean| >> std; one subtraction per element
              bs1[jt][e] = fmaf(dz, xr[e] - bmean4[jt][e], bs1[jt][e]);
            }
          }
        }
    }
    slot0 = slot0 + G::TD >= G::R ? slot0 + G::TD - G::R : slot0 + G::TD;
    if (!RING2_DBG(p, 32)) __syncthreads();
  };
  if constexpr (PF2) {
    // two staging sets, alternating roles (static register indexing: the step body exists twice)
    frag_t sA[G::TD][NLP], sB[G::TD][NLP];
    issue_loads(0, sA);
    for (int step = 0; step < nsteps_z; step += 2) {
      do_step(step, sA, sB);
      if (step + 1 < nsteps_z) do_step(step + 1, sB, sA);
    }
  } else {
    for (int step = 0; step < nsteps_z; ++step) {
      frag_t stg[G::TD][NLP];
      do_step(step, stg, stg);
    }
  }

  if constexpr (BSUM) {
    float* red = reinterpret_cast<float*>(smem);  // [wave][3][NT*16]  (ring no longer needed)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = row16_sum(bs0[jt][e]);
        const float a1 = row16_sum(bs1[jt][e]);
        const float a2 = row16_sum(bs2[jt][e]);
        if (r == 0) {
          red[(wave * 3 + 0) * NT * 16 + jt * 16 + 4 * g + e] = a0;
          red[(wave * 3 + 1) * NT * 16 + jt * 16 + 4 * g + e] = a1;
          red[(wave * 3 + 2) * NT * 16 + jt * 16 + 4 * g + e] = a2;
        }
      }
    __syncthreads();
    if (tid < 3 * NT * 16) {
      const int which = tid / (NT * 16), ch = tid % (NT * 16);
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sacc += red[(w * 3 + which) * NT * 16 + ch];
      if (which == 1) sacc *= bprm[1 * NT * 16 + ch];   // sum dz*(x - mean) -> sum dz*xhat
      fin_store(&p.bpart[((int64_t)blockIdx.x * 3 + which) * p.Cout + nt0 * 16 + ch], sacc);
    }
    fin_tail_run<BnBwdFin, 256, offsetof(ConvParams, ft), offsetof(ConvParams, bbfin)>(p.bpart, smem);
  }
  if (want_stats) {
    float* red = reinterpret_cast<float*>(smem);  // [wave][2][NT*16]  (ring no longer needed)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = row16_sum(ssum[jt][e]);
        const float b0 = row16_sum(ssq[jt][e]);
        if (r == 0) {
          red[(wave * 2 + 0) * NT * 16 + jt * 16 + 4 * g + e] = a0;
          red[(wave * 2 + 1) * NT * 16 + jt * 16 + 4 * g + e] = b0;
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
    fin_tail_run<BnFin, 256, offsetof(ConvParams, ft), offsetof(ConvParams, bfin)>(p.stats, smem);
  }
}

template <typename T, int CK, int NT, int MODE>
static int launch_conv_ring2_k(ConvParams p, hipStream_t st) {
  using G = RingGeom<T, CK>;
  constexpr int dt = SEGMI_BF16;
  p.tz = conv_ring_zsplit(dt, p.Cin, 3, 1, p.N, p.Do, p.Ho, p.Wo);
  static const int dbg = getenv("SEGMI_RING2_DBG") ? atoi(getenv("SEGMI_RING2_DBG")) : 0;
  p.dbg = dbg;
  static const int xcd = getenv("SEGMI_RING2_XCD") ? atoi(getenv("SEGMI_RING2_XCD")) : 1;
  p.ty = cdiv(p.Ho, G::TH);
  p.tx = cdiv(p.Wo, G::TW);
  // 32-bit byte offsets inside one input / output plane
  SEGMI_CHECK_ARG((int64_t)p.Hi * p.Wi * p.ldi * 2 < (1ll << 31) && (int64_t)p.Ho * p.Wo * p.ldo < (1ll << 31) &&
                      (int64_t)p.Ho * p.Wo * (p.ldr > 0 ? p.ldr : 1) < (1ll << 31),
                  "conv3d: plane too large for the ring kernel's 32-bit offsets");
  // plane index x plane size as a 32-bit product: one sample of every operand below 2^31 (bytes for
  // the input, elements for the others)
  SEGMI_CHECK_ARG((int64_t)(p.Di + 8) * p.Hi * p.Wi * p.ldi * 2 < (1ll << 31) &&
                      (int64_t)(p.Do + 4) * p.Ho * p.Wo * p.ldo < (1ll << 31) &&
                      (int64_t)(p.Do + 4) * p.Ho * p.Wo * (p.ldr > 0 ? p.ldr : 1) < (1ll << 31) &&
                      (int64_t)(p.Do + 4) * p.Ho * p.Wo * (p.ldbx > 0 ? p.ldbx : 1) < (1ll << 31),
                  "conv3d: sample too large for the ring kernel's 32-bit plane offsets");
  dim3 grid((unsigned)(p.N * p.ty * p.tx * p.tz), (unsigned)(p.Cout / (16 * NT)));
  p.xcd = xcd != 0 && grid.x % 8 == 0;
  constexpr bool kStats = (MODE & 2) != 0, kBsum = MODE == 4;
  p.fin_on = p.fin_on && (kStats || kBsum);
  (void)fin_tail_arm(p, grid, 256, (kBsum ? 3 : 2) * p.Cout, G::LDS_BYTES + 6 * CK * 4 + 4096);   // LDS: the ring is larger
  auto kern = conv_ring2_kernel<T, CK, NT, MODE>;
  static bool attr_done = false;
  if (!attr_done && G::LDS_BYTES + 6 * CK * 4 + 4096 > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES + 6 * CK * 4 + 4096);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, 256, G::LDS_BYTES + 6 * CK * 4 + 4096, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_fwd(ring2)");
  return SEGMI_OK;
}

template <typename T, int CK, int NT>
static int launch_conv_ring2_cfg(const ConvParams& p, hipStream_t st) {
  if constexpr (NT == 1) {
    // MODE 4 (BatchNorm-backward sums in the input-gradient launch): 16 -> 16, and 32 -> 32 (BASELINE config 4's
    // full-resolution layers, round 4) as two 16-channel output tiles on grid.y -- with both tiles in one workgroup
    // (NT = 2: 216 weight registers + the sums) the variant spills 96 registers and the step got slower
    if (p.bpart && !p.alpha && !p.stats) return launch_conv_ring2_k<T, CK, NT, 4>(p, st);
  }
  switch ((p.alpha ? 1 : 0) | (p.stats ? 2 : 0)) {
    case 0: return launch_conv_ring2_k<T, CK, NT, 0>(p, st);
    case 1: return launch_conv_ring2_k<T, CK, NT, 1>(p, st);
    case 2: return launch_conv_ring2_k<T, CK, NT, 2>(p, st);
    default: return launch_conv_ring2_k<T, CK, NT, 3>(p, st);
  }
}

// bf16 ring layers: 16 -> 16*m and 32 -> 32*m
static int launch_conv_ring2(const ConvParams& p, hipStream_t st) {
  const int ck = pick_ck(SEGMI_BF16, p.Cin);
  const int nt = p.Cout / 16;
  if (ck == 32) {
    const bool sums = p.bpart && !p.alpha && !p.stats;
    if (nt % 2 == 0 && !sums) return launch_conv_ring2_cfg<bf16_t, 32, 2>(p, st);
    return launch_conv_ring2_cfg<bf16_t, 32, 1>(p, st);
  }
  return launch_conv_ring2_cfg<bf16_t, 16, 1>(p, st);
}

}  // namespace segmi
