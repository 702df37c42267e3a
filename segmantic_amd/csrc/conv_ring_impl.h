// conv_ring_impl.h -- persistent "z-marching" variant of the MFMA Conv3d forward for the
// full-resolution layers (k3, stride 1, Cin = one chunk).
//
// A workgroup owns a column of output tiles (fixed n, y-tile, x-tile) and marches along z in
// steps of TD = 4 planes.  The input halo lives in a ring of R = 10 LDS planes of (8+2)x(16+2)
// voxel rows: step i computes from ring slots 4i..4i+5 (mod 10) while the 4 NEW planes of step
// i+1 -- fetched from HBM before the compute phase -- are written to slots 4i+6..4i+9, which
// nobody reads during step i.  So, compared with the tile-at-a-time kernel:
//   * every input plane is read once per column instead of 1.5x (z halo re-reads are gone),
//   * the global-load latency hides under 112 MFMAs per wave (one s_barrier per step),
//   * weight fragments stay in registers for the whole column, residual rows are prefetched,
//   * BN statistics are reduced once per column (rows = columns, 32x fewer than tiles).
// Wave w computes plane z = 4i + w of the step (8 voxel tiles = 8 y-rows of 16 x), so the kd
// part of every LDS address is wave-uniform.
#pragma once
#include "conv_fwd_impl.h"

namespace segmi {

template <typename T, int CK>
struct RingGeom {
  static constexpr int KG = Elem<T>::KG;
  static constexpr int SPT = CK / KG;
  static constexpr int TD = 4, TH = 8, TW = 16;
  static constexpr int HH = TH + 2, HW = TW + 2;
  static constexpr int R = 10;
  static constexpr int NSLOT = 27 * SPT;
  static constexpr int NSTEP = (NSLOT + 3) / 4;
  static constexpr int RAWB = CK * (int)sizeof(T);
  static constexpr int ROWB = RAWB == 32 ? 32 : RAWB + 16;
  static constexpr int CPR = RAWB / 16;
  static constexpr int PLANE_ROWS = HH * HW;
  static constexpr int PLANE_B = PLANE_ROWS * ROWB;
  static constexpr int LDS_BYTES = R * PLANE_B;
  static constexpr int PLANE_CHUNKS = PLANE_ROWS * CPR;        // 16-byte chunks per plane
  static constexpr int NLD = (TD * PLANE_CHUNKS + 255) / 256;   // loads per thread per step
  static constexpr int NLD0 = (2 * PLANE_CHUNKS + 255) / 256;   // extra loads of the prologue
};

// 4 channels of one voxel in storage format
template <typename T> struct Raw4;
template <> struct Raw4<float> {
  typedef f32x4 type;
  __device__ static type ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  __device__ static f32x4 cvt(type v) { return v; }
};
template <> struct Raw4<bf16_t> {
  typedef u32x2 type;
  __device__ static type ld(const bf16_t* p) { return *reinterpret_cast<const u32x2*>(p); }
  __device__ static f32x4 cvt(type o) {
    return f32x4{__uint_as_float(o[0] << 16), __uint_as_float(o[0] & 0xffff0000u),
                 __uint_as_float(o[1] << 16), __uint_as_float(o[1] & 0xffff0000u)};
  }
};

template <typename T, int CK, int NT>
__global__ __launch_bounds__(256, 2) void conv_ring_mfma_kernel(ConvParams p) {
  using G = RingGeom<T, CK>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;

  // blockIdx.x = ((n * ty + tyi) * tx + txi) * zsplit + seg ; a segment marches `seg_steps` steps
  int t = blockIdx.x;
  const int seg = t % p.tz; t /= p.tz;
  const int txi = t % p.tx; t /= p.tx;
  const int tyi = t % p.ty;
  const int n = t / p.ty;
  const int oy0 = tyi * G::TH, ox0 = txi * G::TW;
  const int nt0 = blockIdx.y * NT;
  const int total_steps = (p.Do + G::TD - 1) / G::TD;
  const int seg_steps = (total_steps + p.tz - 1) / p.tz;
  const int z0 = seg * seg_steps * G::TD;                      // first output plane of the segment
  const int nsteps_z = total_steps - seg * seg_steps < seg_steps ? total_steps - seg * seg_steps
                                                                : seg_steps;

  // ---- weights: when the whole set is <= 16 KiB it lives in LDS behind the ring for the whole
  // column (one conflict-free lane-linear ds_read_b128 per k-step); otherwise fragments are
  // prefetched from L2 one k-step ahead.
  constexpr bool PRE = G::NSTEP * NT <= 16;
  const char* wb = (const char*)p.wfrag + (int64_t)nt0 * 1024 + lane * 16;
  char* wsm = smem + G::LDS_BYTES;
  if constexpr (PRE) {
    for (int i = tid; i < G::NSTEP * NT * 64; i += 256) {
      const int s = i / (NT * 64), j = (i / 64) % NT, l = i % 64;
      *reinterpret_cast<frag_t*>(wsm + (s * NT + j) * 1024 + l * 16) = *reinterpret_cast<const frag_t*>(
          (const char*)p.wfrag + (((int64_t)s * p.ntiles_total) + nt0 + j) * 1024 + l * 16);
    }
  }

  // ---- per-thread staging descriptors (plane-relative): same (y, x, chunk) every step
  const char* inb = (const char*)p.in;
  const int64_t plane_stride = (int64_t)p.Hi * p.Wi * p.ldi * (int64_t)sizeof(T);
  const int64_t img_base = (int64_t)n * p.Di * plane_stride;
  int s_goff[G::NLD];   // byte offset inside an input plane (or -1 when outside in y/x)
  int s_loff[G::NLD];   // byte offset inside an LDS plane
  int s_pl[G::NLD];     // plane index 0..TD-1 inside the step (TD = not a load)
#pragma unroll
  for (int k = 0; k < G::NLD; ++k) {
    const int i = tid + 256 * k;
    const int pl = i / G::PLANE_CHUNKS, rem = i % G::PLANE_CHUNKS;
    const int row = rem / G::CPR, ch = rem % G::CPR;
    const int hy = row / G::HW, hx = row % G::HW;
    const int y = oy0 - 1 + hy, x = ox0 - 1 + hx;
    const bool ok = (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    s_pl[k] = pl < G::TD ? pl : G::TD;
    s_loff[k] = row * G::ROWB + ch * 16;
    s_goff[k] = ok ? (int)((((int64_t)y * p.Wi + x) * p.ldi) * (int64_t)sizeof(T)) + ch * 16 : -1;
  }

  // ---- prologue: planes z = -1 .. 4  ->  ring slots 0 .. 5
  for (int i = tid; i < 6 * G::PLANE_CHUNKS; i += 256) {
    const int pl = i / G::PLANE_CHUNKS, rem = i % G::PLANE_CHUNKS;
    const int row = rem / G::CPR, ch = rem % G::CPR;
    const int hy = row / G::HW, hx = row % G::HW;
    const int z = z0 + pl - 1, y = oy0 - 1 + hy, x = ox0 - 1 + hx;
    frag_t val = frag_t{0u, 0u, 0u, 0u};
    if ((unsigned)z < (unsigned)p.Di && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi)
      val = *reinterpret_cast<const frag_t*>(inb + img_base + z * plane_stride +
                                             (((int64_t)y * p.Wi + x) * p.ldi) * (int64_t)sizeof(T) + ch * 16);
    *reinterpret_cast<frag_t*>(smem + pl * G::PLANE_B + row * G::ROWB + ch * 16) = val;
  }
  __syncthreads();

  // lane part of the operand address: wave w computes plane z = w of the step, voxel tile i = y row
  const int lane_addr = r * G::ROWB;   // x = r ; y added per tile; kh/kw per tap
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
  for (int j = 0; j < NT; ++j) { ssum[j] = f32x4{0.f, 0.f, 0.f, 0.f}; ssq[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  T* outp = (T*)p.out;
  const T* resp = (const T*)p.res;
  // identity residual (out = conv(x) + x, the top unit of the decoder and its dgrad): the rows are
  // the centre plane of the ring, already in LDS -- no second HBM read of x
  const bool res_in = resp && p.res == p.in && p.ldr == p.ldi && p.Cin == p.Cout;

  for (int step = 0; step < nsteps_z; ++step) {
    const int zb = step * G::TD;
    // ---- issue the global loads of the NEXT step's 4 new planes (z = zb+5 .. zb+8)
    frag_t stg[G::NLD];
    const bool more = step + 1 < nsteps_z;
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
      stg[k] = frag_t{0u, 0u, 0u, 0u};
      const int z = z0 + zb + 5 + s_pl[k];
      if (more && s_pl[k] < G::TD && s_goff[k] >= 0 && z < p.Di)
        stg[k] = *reinterpret_cast<const frag_t*>(inb + img_base + z * plane_stride + s_goff[k]);
    }
    // ---- prefetch the residual rows of this step's outputs
    const int oz = z0 + zb + wave;
    typename Raw4<T>::type resv[8][NT];   // kept in storage format (2 VGPRs per bf16 row)
    if (resp && !res_in) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int oy = oy0 + i, ox = ox0 + r;
        const bool valid = oz < p.Do && oy < p.Ho && ox < p.Wo;
        const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
        for (int j = 0; j < NT; ++j)
          resv[i][j] = valid ? Raw4<T>::ld(resp + vox * p.ldr + (nt0 + j) * 16 + 4 * g)
                             : typename Raw4<T>::type{};
      }
    }
    // ---- compute: ring slot of plane (zb - 1 + c), c = wave + kd
    int pofs[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) pofs[kd] = ((zb + wave + kd) % G::R) * G::PLANE_B;
    f32x4 acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag_t wnext[NT];
    if constexpr (!PRE) {
#pragma unroll
      for (int j = 0; j < NT; ++j) wnext[j] = *reinterpret_cast<const frag_t*>(wb + (int64_t)j * 1024);
    }
    // LDS operand pipeline: the 8 voxel-tile fragments of k-step s+1 are issued BEFORE the 8
    // MFMAs of k-step s (8 x 16 cycles cover the ~128-cycle LDS latency); without this the
    // compiler schedules each ds_read one MFMA ahead of its use and every MFMA pair stalls.
    auto step_loff = [&](int s) {
      int loff;
      if constexpr (G::SPT == 2) {
        const int t0 = 2 * s, t1 = 2 * s + 1 < 27 ? 2 * s + 1 : 0;
        const int o0 = pofs[t0 / 9] + (((t0 / 3) % 3) * G::HW + t0 % 3) * G::ROWB;
        const int o1 = pofs[t1 / 9] + (((t1 / 3) % 3) * G::HW + t1 % 3) * G::ROWB;
        loff = ((g >> 1) ? o1 : o0) + (g & 1) * 16;
      } else {
        const int tap = (4 * s) / G::SPT, sub0 = (4 * s) % G::SPT;
        loff = pofs[tap / 9] + (((tap / 3) % 3) * G::HW + tap % 3) * G::ROWB + (sub0 + g) * 16;
      }
      return loff + lane_addr;
    };
    // One-step-ahead operand pipeline: [8 fragment reads of k-step s+1] then [8 x NT MFMAs of
    // k-step s]; the sched_barriers keep hipcc from sinking each read next to its consumer.
    frag_t a_cur[8], a_nxt[8], wf[NT], wf_nxt[NT];
    {
      const int l0 = step_loff(0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        a_cur[i] = *reinterpret_cast<const frag_t*>(smem + l0 + i * G::HW * G::ROWB);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (PRE) wf[j] = *reinterpret_cast<const frag_t*>(wsm + j * 1024 + lane * 16);
        else wf[j] = wnext[j];
      }
    }
#pragma unroll
    for (int s = 0; s < G::NSTEP; ++s) {
      if (s + 1 < G::NSTEP) {   // everything read here is consumed in the NEXT k-step
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if constexpr (PRE)
            wf_nxt[j] = *reinterpret_cast<const frag_t*>(wsm + ((s + 1) * NT + j) * 1024 + lane * 16);
          else
            wf_nxt[j] = *reinterpret_cast<const frag_t*>(wb + ((int64_t)(s + 1) * p.ntiles_total + j) * 1024);
        }
        const int l1 = step_loff(s + 1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          a_nxt[i] = *reinterpret_cast<const frag_t*>(smem + l1 + i * G::HW * G::ROWB);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mma16<T>(wf[j], a_cur[i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = wf_nxt[j];
    }
    // ---- write the prefetched planes into the free ring slots (zb+6 .. zb+9 mod R)
#pragma unroll
    for (int k = 0; k < G::NLD; ++k) {
      if (more && s_pl[k] < G::TD) {
        const int slot = (zb + 6 + s_pl[k]) % G::R;
        *reinterpret_cast<frag_t*>(smem + slot * G::PLANE_B + s_loff[k]) = stg[k];
      }
    }
    if (res_in) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          resv[i][j] = *reinterpret_cast<const typename Raw4<T>::type*>(
              smem + pofs[1] + ((i + 1) * G::HW + r + 1) * G::ROWB +
              ((nt0 + j) * 16 + 4 * g) * (int)sizeof(T));
    }
    // ---- epilogue of this step
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int oy = oy0 + i, ox = ox0 + r;
      const bool valid = oz < p.Do && oy < p.Ho && ox < p.Wo;
      const int64_t vox = (((int64_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 v = acc[i][j] + bias4[j];
        if (valid) {
          if (p.stats) { ssum[j] += v; ssq[j] += v * v; }
          if (has_alpha) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : alpha * v[e];
          }
          if (resp) v += Raw4<T>::cvt(resv[i][j]);
          store4<T>(outp + vox * p.ldo + (nt0 + j) * 16 + 4 * g, v);
        }
      }
    }
    __syncthreads();
  }

  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem);  // [wave][2][NT*16]  (ring no longer needed)
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
      p.stats[((int64_t)blockIdx.x * 2 + which) * p.Cout + nt0 * 16 + ch] = sacc;
    }
  }
}

// Plan: columns = N * ceil(H/8) * ceil(W/16); each column is cut into `zsplit` z-segments so that
// >= ~512 workgroups exist, as long as every segment keeps >= 4 steps (else the 2-plane prologue
// overhead and the lost pipelining make the tile-at-a-time kernel the better choice).
// Returns zsplit, or 0 when the ring kernel should not be used.
static inline int conv_ring_zsplit(int dtype, int cin, int ksize, int stride, int N, int Do, int Ho,
                                   int Wo) {
  if (!(ksize == 3 && stride == 1 && cin == pick_ck(dtype, cin) && Wo > 8)) return 0;
  const int columns = N * cdiv(Ho, 8) * cdiv(Wo, 16);
  const int steps = cdiv(Do, 4);
  int zs = 512 / columns;
  if (zs < 1) zs = 1;
  if (zs > steps / 4) zs = steps / 4;
  if (zs < 1) return 0;
  if (columns * zs < 256) return 0;
  return zs;
}
static inline bool conv_ring_ok(int dtype, int cin, int ksize, int stride, const segmi_act* out) {
  return conv_ring_zsplit(dtype, cin, ksize, stride, out->n, out->d, out->h, out->w) > 0;
}
static inline int conv_ring_rows(int dtype, int cin, const segmi_act* out) {
  return out->n * cdiv(out->h, 8) * cdiv(out->w, 16) *
         conv_ring_zsplit(dtype, cin, 3, 1, out->n, out->d, out->h, out->w);
}

template <typename T, int CK, int NT>
static int launch_conv_ring_cfg(ConvParams p, hipStream_t st) {
  using G = RingGeom<T, CK>;
  constexpr int dt = sizeof(T) == 4 ? SEGMI_F32 : SEGMI_BF16;
  p.tz = conv_ring_zsplit(dt, p.Cin, 3, 1, p.N, p.Do, p.Ho, p.Wo);
  p.ty = cdiv(p.Ho, G::TH);
  p.tx = cdiv(p.Wo, G::TW);
  dim3 grid((unsigned)(p.N * p.ty * p.tx * p.tz), (unsigned)(p.Cout / (16 * NT)));
  auto kern = conv_ring_mfma_kernel<T, CK, NT>;
  constexpr int lds = G::LDS_BYTES + (G::NSTEP * NT <= 16 ? G::NSTEP * NT * 1024 : 0);
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, 256, lds, st, p);
  SEGMI_LAUNCH_CHECK("conv3d_fwd(ring)");
  return SEGMI_OK;
}

template <typename T>
static int launch_conv_ring_t(const ConvParams& p, hipStream_t st) {
  constexpr int dt = sizeof(T) == 4 ? SEGMI_F32 : SEGMI_BF16;
  const int ck = pick_ck(dt, p.Cin);
  const int nt = p.Cout / 16;
  if constexpr (sizeof(T) == 2) {
    if (ck == 32) {
      if (nt % 2 == 0) return launch_conv_ring_cfg<T, 32, 2>(p, st);
      return launch_conv_ring_cfg<T, 32, 1>(p, st);
    }
  }
  if (nt % 2 == 0) return launch_conv_ring_cfg<T, 16, 2>(p, st);
  return launch_conv_ring_cfg<T, 16, 1>(p, st);
}

}  // namespace segmi
