// conv_ring_impl.h -- geometry and plan of the persistent "z-marching" MFMA Conv3d forward for the
// full-resolution layers (k3, stride 1, Cin = one chunk); the kernel itself is conv_ring2_impl.h.
// (The first formulation -- one output plane x 8 rows per wave, described below -- needed one 1-KiB
// LDS fragment per MFMA and spilled in its f32 and 32-channel variants; bf16 layers run ring2, f32
// layers the K-split / tile kernels, which are 6 % faster there than the spilling ring was.)
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


// Plan: columns = N * ceil(H/8) * ceil(W/16); each column is cut into `zsplit` z-segments so that
// >= ~512 workgroups exist, as long as every segment keeps >= 4 steps (else the 2-plane prologue
// overhead and the lost pipelining make the tile-at-a-time kernel the better choice).
// Returns zsplit, or 0 when the ring kernel should not be used.
static inline int conv_ring_zsplit(int dtype, int cin, int ksize, int stride, int N, int Do, int Ho,
                                   int Wo) {
  if (!(ksize == 3 && stride == 1 && cin == pick_ck(dtype, cin) && Wo > 8)) return 0;
  if (dtype != SEGMI_BF16) return 0;
  // the kernel addresses one sample with 32-bit "plane index x plane size" products (bytes for the
  // input, elements for the others); rows may be strided views (ld up to 4 x cin in this engine: skip
  // buffers, padded class axis).  Larger samples take the tile / K-split kernels (64-bit addressing)
  // -- every eligibility predicate (in_affine_ok, bn_bwd_sums_ok, stats_rows, kernel_name) follows.
  if ((int64_t)(Do + 8) * Ho * Wo * (4 * cin) * 2 >= (1ll << 31)) return 0;
  const int columns = N * cdiv(Ho, 8) * cdiv(Wo, 16);
  const int steps = cdiv(Do, 4);
  int zs = 512 / columns;
  if (zs < 1) zs = 1;
  static const int zs_mul = getenv("SEGMI_RING_ZS") ? atoi(getenv("SEGMI_RING_ZS")) : 1;   // A/B: more, shorter segments
  if (zs_mul > 1) zs *= zs_mul;
  if (zs > steps / 4) zs = steps / 4;
  if (zs < 1) return 0;
  if (columns * zs < 256) return 0;
  return zs;
}
// ring3 (conv_ring3_impl.h: LDS-DMA staging) takes a ring layer when it has 16 input channels, 16-byte aligned
// input and output rows, and every operand sample stays below 2^31 BYTES (its out-of-range marker 2^31 + a plane offset must
// not wrap); ring2 takes the others.  SEGMI_RING3=0: ring2 everywhere (A/B).
static inline bool conv_ring3_shape_ok(int cin, int cout, const void* in, const void* out, int Di, int Hi, int Wi, int ldi, int Do,
                                       int Ho, int Wo, int ldo, int ldr, int ldbx) {
  static const bool on = !(getenv("SEGMI_RING3") && atoi(getenv("SEGMI_RING3")) == 0);
  if (!on || cin != 16 || cout % 16 != 0) return false;
  const int64_t lim = 1ll << 31;
  return (int64_t)(Di + 8) * Hi * Wi * ldi * 2 < lim && (int64_t)(Do + 4) * Ho * Wo * ldo * 2 < lim &&
         (int64_t)(Do + 4) * Ho * Wo * (ldr > 0 ? ldr : 1) * 2 < lim &&
         (int64_t)(Do + 4) * Ho * Wo * (ldbx > 0 ? ldbx : 1) * 2 < lim && ((uintptr_t)in % 16) == 0 && ldi % 8 == 0 &&
         ((uintptr_t)out % 16) == 0 && ldo % 8 == 0;          // 16-byte DMA pieces and 16-byte stores
}
static inline bool conv_ring_ok(int dtype, int cin, int ksize, int stride, const segmi_act* out) {
  return conv_ring_zsplit(dtype, cin, ksize, stride, out->n, out->d, out->h, out->w) > 0;
}
static inline int conv_ring_rows(int dtype, int cin, const segmi_act* out) {
  return out->n * cdiv(out->h, 8) * cdiv(out->w, 16) *
         conv_ring_zsplit(dtype, cin, 3, 1, out->n, out->d, out->h, out->w);
}


}  // namespace segmi
