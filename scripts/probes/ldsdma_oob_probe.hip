#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// mode 0: voffset OOB for odd lanes (num_records = n bytes); mode 1: soffset pushes past num_records; mode 2: num_records = 0
__global__ void probe(const unsigned* src, unsigned* out, int nbytes, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  unsigned* l = (unsigned*)smem;
  for (int i = lane; i < 1024; i += 64) l[i] = 0xDEAD0000u + i;
  __syncthreads();
  int nr = mode == 2 ? 0 : nbytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nr, 0x00020000);
  unsigned voff = lane * 16;
  unsigned soff = 0;
  if (mode == 0 && (lane & 1)) voff = 0xFFFFFFF0u;
  if (mode == 1) soff = nbytes - 512;     // lanes >= 32 are beyond the end only if soffset counts
  if (mode == 3 && (lane & 1)) voff = nbytes + lane * 16;  // just past the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 1024), 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = l[i];
}
int main() {
  const int n = 1024;  // bytes
  unsigned h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, 4096 * 4); hipMalloc(&o, 1024 * 4);
  hipMemcpy(d, h, 4096 * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, 1, 64, 8192, 0, d, o, n, mode);
    unsigned r[1024];
    hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    printf("mode %d:", mode);
    for (int lane = 0; lane < 64; lane += (mode == 1 ? 8 : 1)) { if (mode != 1 && lane >= 6 && lane < 60) continue; printf(" L%d=%x,%x", lane, r[256 + lane * 4], r[256 + lane * 4 + 3]); }
    printf("  before=%x after=%x\n", r[255], r[256 + 256]);
  }
  return 0;
}
