// reduce_fin.h -- one-launch, fixed-order reduction of a table of f32 partial rows followed by a
// caller-supplied finalisation.
//
//   stage 1: `nblk` (<= 64) workgroups each fold a fixed subset of the [rows][width] f32 table into
//            one f64 row of `scratch` ([nblk][width] doubles).
//   stage 2: the workgroup that takes the last ticket folds the nblk f64 rows (fixed order) into
//            `width` column sums held in LDS and runs `fin(sums)` with all 256 threads.
//
// The result does not depend on which workgroup arrives last: both stages add in an order fixed by
// (rows, width, nblk) only, so the reduction is deterministic without float atomics.  The ticket
// counters are integers, self-resetting, and handed out round-robin by the host so two launches in
// flight (main + side stream) never share one.
#pragma once
#include <atomic>

#include "common.h"

namespace segmi {

constexpr int kFinBlocks = 64;      // stage-1 rows; the scratch behind a partial table holds these
constexpr int kFinTickets = 2048;
constexpr int kFinLdsWidth = 4096;  // up to 32 KB of f64 column sums live in LDS; wider tables
                                    // keep them in one more scratch row
constexpr int kFinScratchRows = kFinBlocks + 1;   // f64 rows of `width` behind a partial table

// column sums [2][c] = {sum dy, sum dy^2} -> db[c]: the bias gradient is the first half
struct BiasFin {
  int c;
  float* db;
  __device__ void operator()(const double* sums, double*) const {
    for (int ch = threadIdx.x; ch < c; ch += 256) db[ch] = (float)sums[ch];
  }
};

static __device__ unsigned int g_fin_tickets[kFinTickets];
static std::atomic<unsigned> g_fin_next{0};

// Small tables (<= 64 K floats) are folded by ONE workgroup: the multi-block path costs four
// dependent device-scope round trips (rows -> scratch -> fence + ticket -> scratch -> result, ~16 us
// measured) where a single block needs one; its column sums use 16 independent accumulators per
// thread so the loads of a thread are in flight together.
static inline int fin_blocks(int rows, int width) {
  if ((int64_t)rows * width <= 65536) return 1;
  int b = rows / 8;
  return b < 1 ? 1 : (b > kFinBlocks ? kFinBlocks : b);
}

template <class Fin>
__global__ __launch_bounds__(256) void collapse_fin_kernel(const float* __restrict__ in, int rows,
                                                           int width, double* scratch,
                                                           unsigned ticket, Fin fin) {
  extern __shared__ double fin_lds[];   // [width] when width <= kFinLdsWidth
  double* fin_sums = width <= kFinLdsWidth ? fin_lds : scratch + (int64_t)kFinBlocks * width;
  __shared__ double red[256];
  __shared__ int s_last;
  const int wl = width < 256 ? width : 256;
  const int rl = 256 / wl;
  const int tid = threadIdx.x;
  const int col = tid % wl, lane = tid / wl;
  const int nblk = gridDim.x;
  // 16-byte columns: a thread owns 4 adjacent columns and every (256 / (width/4))-th row, 8 row
  // loads in flight -> 32 KB per round trip and workgroup instead of 16 KB of 4-byte loads (the
  // single-workgroup collapse of a 1024 x 48 table: 12 -> 5 us).  Same fixed order for a given
  // (rows, width, nblk), so still deterministic.
  const bool vec4 = width % 4 == 0 && width / 4 <= 256 && ((uintptr_t)in & 15) == 0;
  if (vec4) {
    __shared__ double red4[256][4];
    const int wl4 = width / 4, rl4 = 256 / wl4;
    const int col4 = tid % wl4, lane4 = tid / wl4;
    double s[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s[a][b] = 0.0;
    if (lane4 < rl4) {
      const int step = nblk * rl4;
      int r = blockIdx.x * rl4 + lane4;
      const f32x4* src = reinterpret_cast<const f32x4*>(in) + col4;
      for (; r + 7 * step < rows; r += 8 * step) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(r + u * step) * wl4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int b = 0; b < 4; ++b) s[u & 3][b] += (double)v[u][b];
      }
      for (; r < rows; r += step) {
        const f32x4 v = src[(int64_t)r * wl4];
#pragma unroll
        for (int b = 0; b < 4; ++b) s[0][b] += (double)v[b];
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) red4[tid][b] = (s[0][b] + s[1][b]) + (s[2][b] + s[3][b]);
    __syncthreads();
    for (int e = tid; e < width; e += 256) {
      const int c4 = e / 4, b = e % 4;
      double t = 0.0;
      for (int l = 0; l < rl4; ++l) t += red4[l * wl4 + c4][b];
      if (nblk == 1) fin_sums[e] = t;
      else scratch[(int64_t)blockIdx.x * width + e] = t;
    }
    __syncthreads();
  } else
  for (int w0 = 0; w0 < width; w0 += wl) {
    const int e = w0 + col;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (lane < rl && e < width) {
      const int step = nblk * rl;
      int r = blockIdx.x * rl + lane;
      for (; r + 15 * step < rows; r += 16 * step) {      // 16 loads in flight per thread
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = in[(int64_t)(r + u * step) * width + e];
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
          s0 += (double)v[u]; s1 += (double)v[u + 1]; s2 += (double)v[u + 2]; s3 += (double)v[u + 3];
        }
      }
      for (; r + 3 * step < rows; r += 4 * step) {
        s0 += (double)in[(int64_t)r * width + e];
        s1 += (double)in[(int64_t)(r + step) * width + e];
        s2 += (double)in[(int64_t)(r + 2 * step) * width + e];
        s3 += (double)in[(int64_t)(r + 3 * step) * width + e];
      }
      for (; r < rows; r += step) s0 += (double)in[(int64_t)r * width + e];
    }
    red[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tid < wl && w0 + tid < width) {
      double t = 0.0;
      for (int l = 0; l < rl; ++l) t += red[l * wl + tid];
      if (nblk == 1) fin_sums[w0 + tid] = t;
      else scratch[(int64_t)blockIdx.x * width + w0 + tid] = t;
    }
    __syncthreads();
  }
  if (nblk > 1) {
    __threadfence();   // release: this block's scratch row is visible device-wide (all XCDs)
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = atomicAdd(&g_fin_tickets[ticket], 1u);
      s_last = prev == (unsigned)nblk - 1;
      if (s_last) g_fin_tickets[ticket] = 0;   // ready for the next launch that draws this ticket
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();   // acquire: the other blocks' rows
    const double* t = scratch;
    for (int w0 = 0; w0 < width; w0 += wl) {
      const int e = w0 + col;
      double s0 = 0.0, s1 = 0.0;
      if (lane < rl && e < width) {
        int r = lane;
        for (; r + rl < nblk; r += 2 * rl) {
          s0 += t[(int64_t)r * width + e];
          s1 += t[(int64_t)(r + rl) * width + e];
        }
        for (; r < nblk; r += rl) s0 += t[(int64_t)r * width + e];
      }
      red[tid] = s0 + s1;
      __syncthreads();
      if (tid < wl && w0 + tid < width) {
        double a = 0.0;
        for (int l = 0; l < rl; ++l) a += red[l * wl + tid];
        fin_sums[w0 + tid] = a;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  fin(fin_sums, red);
}

// scratch: >= kFinScratchRows * width doubles, 8-byte aligned
template <class Fin>
static inline int collapse_fin_launch(const float* partials, int rows, int width, double* scratch,
                                      hipStream_t st, const Fin& fin, const char* what) {
  if (width <= 0 || rows <= 0) {
    set_error("%s: empty reduction table %d x %d", what, rows, width);
    return SEGMI_EINVAL;
  }
  const unsigned ticket = g_fin_next.fetch_add(1) % kFinTickets;
  hipLaunchKernelGGL(collapse_fin_kernel<Fin>, fin_blocks(rows, width), 256,
                     width <= kFinLdsWidth ? (size_t)width * sizeof(double) : 0,
                     st, partials, rows, width, scratch, ticket, fin);
  SEGMI_LAUNCH_CHECK(what);
  return SEGMI_OK;
}

// scratch tail behind a caller-visible partial table (the *_rows() queries reserve it)
static inline double* fin_scratch(const float* partials, int real_rows, int width) {
  uintptr_t p = (uintptr_t)(partials + (int64_t)real_rows * width);
  return (double*)((p + 7) & ~(uintptr_t)7);
}

}  // namespace segmi
