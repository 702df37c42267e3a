#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 elem2(f32x2 a, f32x2 d, f32x2 mean, f32x2 is, f32x2 gm, f32x2 beta, f32x2 c0, f32x2 c1,
                                       bool has_alpha, float alpha) {
#pragma clang fp contract(off)
  const f32x2 xh = (a - mean) * is;
  const f32x2 z = __builtin_elementwise_fma(xh, gm, beta);
  f32x2 dz = d;
  if (has_alpha) {
    const f32x2 ad = alpha * d;
    dz[0] = !(z[0] > 0.f) ? ad[0] : d[0];
    dz[1] = !(z[1] > 0.f) ? ad[1] : d[1];
  }
  return (gm * is) * __builtin_elementwise_fma(-xh, c1, dz - c0);
}
__global__ void k(const f32x2* a, const f32x2* d, f32x2* o, const f32x2* prm, float alpha, int ha) {
  const int i = threadIdx.x + blockIdx.x * 256;
  o[i] = elem2(a[i], d[i], prm[0], prm[1], prm[2], prm[3], prm[4], prm[5], ha != 0, alpha);
}
