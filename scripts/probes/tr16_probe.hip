// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds u16 value = its own element index; every lane
// passes the address of 4 consecutive elements starting at lane*4; prints what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[256];
  for (int i = threadIdx.x; i < 256; i += 64) sm[i] = (unsigned short)i;
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sm + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
