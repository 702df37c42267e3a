// common.h -- shared device/host helpers for libsegmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/segmi.h"

namespace segmi {

// ------------------------------------------------------------------ errors
void set_error(const char* fmt, ...);

#define SEGMI_CHECK_ARG(cond, ...)                \
  do {                                            \
    if (!(cond)) {                                \
      ::segmi::set_error(__VA_ARGS__);            \
      return SEGMI_EINVAL;                        \
    }                                             \
  } while (0)

#define SEGMI_UNSUPPORTED(...)                    \
  do {                                            \
    ::segmi::set_error(__VA_ARGS__);              \
    return SEGMI_EUNSUPPORTED;                    \
  } while (0)

#define SEGMI_LAUNCH_CHECK(what)                                                     \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      ::segmi::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));     \
      return SEGMI_ELAUNCH;                                                          \
    }                                                                                \
  } while (0)

// ------------------------------------------------------------------ types
typedef unsigned short bf16_t;  // raw storage
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = static_cast<__bf16>(f);  // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(bf16_t, b);
}
typedef float f32x2_cv __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_cv __attribute__((ext_vector_type(2)));
// both halves in ONE v_cvt_pk_bf16_f32 (two scalar conversions cost 2 cvt + and + or_sdwa)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const f32x2_cv f{lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_cv));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int KG = 4;  // elements per 16-byte lane fragment
  __device__ static float ld(const float* p) { return *p; }
  __device__ static void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int KG = 8;
  __device__ static float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte fragment, generic over dtype
typedef u32x4 frag_t;

// D[m][n] += sum_k A[m][k] B[k][n] with the 16-byte-per-lane k-slot convention of DESIGN.md:
// bf16: one v_mfma_f32_16x16x32_bf16 (lane l holds k = 8*(l>>4)+j, j<8)
// f32 : four v_mfma_f32_16x16x4_f32, MFMA j consuming element j (k-slot 4*(l>>4)+j)
template <typename T> __device__ __forceinline__ f32x4 mma16(frag_t a, frag_t b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma16<bf16_t>(frag_t a, frag_t b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<float>(frag_t a, frag_t b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[0]), __uint_as_float(b[0]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[1]), __uint_as_float(b[1]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[2]), __uint_as_float(b[2]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[3]), __uint_as_float(b[3]), c, 0, 0, 0);
  return c;
}

// store 4 consecutive channels (f32 accumulators) as T
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) {
  *reinterpret_cast<f32x4*>(p) = v;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
  u32x2 o;
  o[0] = pack_bf16x2(v[0], v[1]);
  o[1] = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<u32x2*>(p) = o;
}
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
  u32x2 o = *reinterpret_cast<const u32x2*>(p);
  f32x4 v;
  v[0] = __uint_as_float(o[0] << 16);
  v[1] = __uint_as_float(o[0] & 0xffff0000u);
  v[2] = __uint_as_float(o[1] << 16);
  v[3] = __uint_as_float(o[1] & 0xffff0000u);
  return v;
}

// "Use" a value without emitting an instruction: the compiler must complete the load that
// produced it HERE.  hipcc's waitcnt insertion cannot prove that a load issued before a loop (bias,
// PReLU slope, prefetched residual rows) has landed when its first use sits in a conditionally
// executed block, and then puts s_waitcnt vmcnt(0) in front of EVERY such block -- on gfx9 that
// also waits for the previous block's global store, serialising an epilogue's stores on the
// full memory round trip.  Touching the value once, unconditionally, removes all those waits.
template <typename V> __device__ __forceinline__ void touch_v(V& v) { asm volatile("" : "+v"(v)); }
// scalar variant for wave-uniform values; "+v" because hipcc may already hold them in a VGPR
__device__ __forceinline__ void touch_s(float& v) { asm volatile("" : "+v"(v)); }

// One element of the BatchNorm (+ PReLU) backward apply, dx = gamma*invstd*(dz - c0 - xhat*c1), shared by
// bn_act_bwd_apply_kernel, the one-launch variant for small tensors and the stride-2 convolution that
// applies it while staging (conv_bnbwd_impl.h).  Contraction is spelled out (one fma, chosen here, not by
// the optimiser per call site): the three kernels must produce the same bits.  m: dropout multiplier (1 = none).
__device__ __forceinline__ float bn_bwd_apply_elem(float a, float d, float mean, float is, float gm, float beta,
                                                   float c0, float c1, bool has_alpha, float alpha, float m) {
#pragma clang fp contract(off)
  const float xh = (a - mean) * is;
  const float z = fmaf(xh, gm, beta) * m;
  float dz = d;
  if (has_alpha && !(z > 0.f)) dz = alpha * d;
  dz *= m;
  return (gm * is) * fmaf(-xh, c1, dz - c0);
}

// Two elements at once, no dropout: the same operations in the same order as bn_bwd_apply_elem with m = 1
// (x * 1 is exact), written on 2-vectors so that hipcc emits the packed f32 forms (v_pk_add / v_pk_mul /
// v_pk_fma_f32: two lanes' worth of IEEE results per issue).  These passes are VALU-bound, not memory-bound:
// ~15 scalar operations per element put the full-resolution apply at 205 us of vector issue out of its 306.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool HAS_ALPHA>
__device__ __forceinline__ f32x2 bn_bwd_apply_elem2(f32x2 a, f32x2 d, f32x2 mean, f32x2 is, f32x2 gm, f32x2 beta,
                                                    f32x2 c0, f32x2 c1, float alpha) {
#pragma clang fp contract(off)
  const f32x2 xh = (a - mean) * is;
  f32x2 dz = d;
  if constexpr (HAS_ALPHA) {
    const f32x2 z = __builtin_elementwise_fma(xh, gm, beta);
    const f32x2 ad = alpha * d;
    dz[0] = !(z[0] > 0.f) ? ad[0] : d[0];
    dz[1] = !(z[1] > 0.f) ? ad[1] : d[1];
  }
  return (gm * is) * __builtin_elementwise_fma(-xh, c1, dz - c0);
}

// the same with gamma * invstd supplied by the caller (a per-channel constant it keeps in registers)
template <bool HAS_ALPHA>
__device__ __forceinline__ f32x2 bn_bwd_apply_elem2g(f32x2 a, f32x2 d, f32x2 mean, f32x2 is, f32x2 gm, f32x2 beta,
                                                     f32x2 c0, f32x2 c1, f32x2 gm_is, float alpha) {
#pragma clang fp contract(off)
  const f32x2 xh = (a - mean) * is;
  f32x2 dz = d;
  if constexpr (HAS_ALPHA) {
    const f32x2 z = __builtin_elementwise_fma(xh, gm, beta);
    const f32x2 ad = alpha * d;
    dz[0] = !(z[0] > 0.f) ? ad[0] : d[0];
    dz[1] = !(z[1] > 0.f) ? ad[1] : d[1];
  }
  return gm_is * __builtin_elementwise_fma(-xh, c1, dz - c0);
}

// BatchNorm-apply + PReLU of the PRODUCER layer on 8 bf16 channels of one voxel, done while the
// consumer stages its input (segmi_in_affine): z = fma(x, scale, shift); z = z > 0 ? z : alpha * z,
// rounded to bf16 exactly as bn_act_fwd stores it, so a consumer that transforms on the fly sees
// the very bits the separate pass would have written.  sc / sh: the 8 channels of this 16-byte chunk.
__device__ __forceinline__ frag_t bn_prelu_bf16x8(const frag_t raw, const float (&sc)[8], const float (&sh)[8],
                                                  const float alpha, const bool has_alpha) {
  frag_t o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float z0 = fmaf(__uint_as_float(raw[q] << 16), sc[2 * q], sh[2 * q]);
    float z1 = fmaf(__uint_as_float(raw[q] & 0xffff0000u), sc[2 * q + 1], sh[2 * q + 1]);
    if (has_alpha) {
      z0 = z0 > 0.f ? z0 : alpha * z0;
      z1 = z1 > 0.f ? z1 : alpha * z1;
    }
    o[q] = pack_bf16x2(z0, z1);
  }
  return o;
}

// The same transform for 0 <= alpha <= 1 (every PReLU in practice, ReLU, LeakyReLU): there
// z > 0 ? z : alpha*z == max(z, alpha*z) bit for bit (signed zeros and NaN included), which is a
// packed multiply and one max per element instead of multiply + compare + select.  Callers pick it
// behind a wave-uniform branch.
__device__ __forceinline__ frag_t bn_prelu01_bf16x8(const frag_t raw, const float (&sc)[8], const float (&sh)[8],
                                                    const float alpha) {
  frag_t o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2_cv x = {__uint_as_float(raw[q] << 16), __uint_as_float(raw[q] & 0xffff0000u)};
    const f32x2_cv s = {sc[2 * q], sc[2 * q + 1]}, h = {sh[2 * q], sh[2 * q + 1]};
    const f32x2_cv z = __builtin_elementwise_fma(x, s, h);       // one rounding per element, as fmaf
    const f32x2_cv az = z * alpha;
    o[q] = pack_bf16x2(fmaxf(z[0], az[0]), fmaxf(z[1], az[1]));
  }
  return o;
}

// sum over the 16 lanes that share (lane>>4): result valid in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t act_voxels(const segmi_act* a) {
  return (int64_t)a->n * a->d * a->h * a->w;
}
static inline bool act_ok(const segmi_act* a) {
  return a && a->data && a->n > 0 && a->d > 0 && a->h > 0 && a->w > 0 && a->c > 0 &&
         a->ld >= a->c;
}
static inline int dtype_size(int dtype) { return dtype == SEGMI_F32 ? 4 : 2; }

// ------------------------------------------------------------------ fragment-pack geometry
// k-slot enumeration shared by the pack kernels and the MFMA kernels.
//  KG   = elements per 16-byte fragment (8 bf16 / 4 f32)
//  CK   = channels staged per chunk (16 or 32); SPT = CK / KG slots per tap
//  slot q = tap * SPT + sub ; k-step s covers slots 4s..4s+3, lane group g = lane>>4 owns 4s+g
struct PackGeom {
  int KG, CK, SPT, ntaps, nslots, nsteps, nchunks, ntiles;
};
// f32 always stages 16 channels per chunk (64-B rows keep the halo tile within LDS);
// bf16 stages 32 when the channel count allows it.
static inline int pick_ck(int dtype, int cin) {
  return (dtype == SEGMI_BF16 && cin % 32 == 0) ? 32 : 16;
}
static inline PackGeom pack_geom(int dtype, int cin, int cout, int ntaps) {
  PackGeom g;
  g.KG = dtype == SEGMI_F32 ? 4 : 8;
  g.CK = pick_ck(dtype, cin);
  g.SPT = g.CK / g.KG;
  g.ntaps = ntaps;
  g.nslots = ntaps * g.SPT;
  g.nsteps = (g.nslots + 3) / 4;
  g.nchunks = cin / g.CK;
  g.ntiles = cout / 16;
  return g;
}

// transposed-conv (k3 s2 p1) parity classes: class p = rd*4 + rh*2 + rw, r = output parity.
// per dim: r=0 -> {(k=1, di=0)} ; r=1 -> {(k=2, di=0), (k=0, di=+1)}
__host__ __device__ inline int ct_ntaps(int p) {
  return (1 + ((p >> 2) & 1)) * (1 + ((p >> 1) & 1)) * (1 + (p & 1));
}
// tap t of class p -> kernel index (kd,kh,kw) and input offset (dd,dh,dw)
__host__ __device__ inline void ct_tap(int p, int t, int& kd, int& kh, int& kw, int& dd,
                                       int& dh, int& dw) {
  int rw = p & 1, rh = (p >> 1) & 1, rd = (p >> 2) & 1;
  int nw = 1 + rw, nh = 1 + rh;
  int tw = t % nw, th = (t / nw) % nh, td = t / (nw * nh);
  kw = rw ? (tw ? 0 : 2) : 1; dw = rw ? tw : 0;
  kh = rh ? (th ? 0 : 2) : 1; dh = rh ? th : 0;
  kd = rd ? (td ? 0 : 2) : 1; dd = rd ? td : 0;
}

}  // namespace segmi
