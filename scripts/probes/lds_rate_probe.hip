// LDS read throughput on gfx950 per instruction kind: every wave of a 256- or 512-thread workgroup
// issues N reads of one kind back to back (conflict-free addresses: lane-linear), cycles from
// s_memtime.  Prints bytes/clk/CU for 1 workgroup per CU (grid = #CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void probe(unsigned long long* cyc, unsigned* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)sm)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  const char* base = sm + wave * 4096;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int g = lane >> 4, i16 = lane & 15;
  const unsigned a_lin16 = (unsigned)(size_t)(base + lane * 16) & 0xffff;
  const unsigned a_lin8 = (unsigned)(size_t)(base + lane * 8) & 0xffff;
  const unsigned a_lin4 = (unsigned)(size_t)(base + lane * 4) & 0xffff;
  const unsigned a_tr = (unsigned)(size_t)(base + (4 * g + (i16 >> 2)) * 32 + 8 * (i16 & 3)) & 0xffff;
  for (int it = 0; it < iters; ++it) {
    // 8 independent reads, results never consumed by VALU (only the final wait)
    if constexpr (KIND == 0) {
      u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n"
                   "ds_read_b128 %3, %8 offset:3072\n ds_read_b128 %4, %8\n ds_read_b128 %5, %8 offset:1024\n"
                   "ds_read_b128 %6, %8 offset:2048\n ds_read_b128 %7, %8 offset:3072\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a_lin16));
      acc += v0[0] + v7[3];
    } else if constexpr (KIND == 1) {
      u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n"
                   "ds_read_b64 %3, %8 offset:1536\n ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n"
                   "ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a_lin8));
      acc += v0[0] + v7[1];
    } else if constexpr (KIND == 2) {
      u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:512\n ds_read_b64_tr_b16 %2, %8 offset:1024\n"
                   "ds_read_b64_tr_b16 %3, %8 offset:1536\n ds_read_b64_tr_b16 %4, %8 offset:2048\n ds_read_b64_tr_b16 %5, %8 offset:2560\n"
                   "ds_read_b64_tr_b16 %6, %8 offset:3072\n ds_read_b64_tr_b16 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a_tr));
      acc += v0[0] + v7[1];
    } else {
      unsigned v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n"
                   "ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n"
                   "ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a_lin4));
      acc += v0 + v7;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  unsigned long long* cyc; unsigned* sink;
  hipMalloc(&cyc, 8 * 4096); hipMalloc(&sink, 4 * 512 * 512);
  const int iters = 2000;
  const int bytes[4] = {1024, 512, 512, 256};
  const char* names[4] = {"ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16", "ds_read_b32"};
  for (int threads = 256; threads <= 1024; threads += 256)
    for (int kind = 0; kind < 4; ++kind) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (kind == 0) hipLaunchKernelGGL(probe<0>, 256, threads, 65536, 0, cyc, sink, iters);
        if (kind == 1) hipLaunchKernelGGL(probe<1>, 256, threads, 65536, 0, cyc, sink, iters);
        if (kind == 2) hipLaunchKernelGGL(probe<2>, 256, threads, 65536, 0, cyc, sink, iters);
        if (kind == 3) hipLaunchKernelGGL(probe<3>, 256, threads, 65536, 0, cyc, sink, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
      }
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[16];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      const int waves = threads / 64;
      double c = 0; for (int w = 0; w < waves; ++w) c += (double)h[w];
      c /= waves;
      // readcyclecounter ticks at 100 MHz on gfx9 (s_memtime); convert with the 2.4 GHz shader clock
      printf("%-20s %2d waves/CU: %.0f ticks, kernel %.3f ms (%.0f ticks/us) -> %.1f B/tick/CU, %.1f B/ns/CU\n",
             names[kind], waves, c, ms, c / (ms * 1e3), (double)bytes[kind] * iters * 8 * waves / c,
             (double)bytes[kind] * iters * 8 * waves / (ms * 1e6));
    }
  return 0;
}
