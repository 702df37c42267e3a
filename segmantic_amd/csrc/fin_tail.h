// fin_tail.h -- finalisation of a partial-row reduction INSIDE the launch that produces the rows.
//
// Every producer of a [rows][width] f32 partial table (convolutions with fused BatchNorm statistics,
// the BatchNorm-backward reduce, Dice sums, ...) used to be followed by a one-workgroup
// collapse_fin_kernel launch (reduce_fin.h): ~40 launches per training step that move no bytes but
// sit on the dependent chain of the main stream (8 us each + the dispatch gap).  With a FinTail the
// producer's workgroups take a ticket after their row is written; the workgroup that draws the LAST
// ticket folds the whole table in a fixed order (f64) and runs the finalisation functor -- in the
// same launch.  The result does not depend on which workgroup arrives last: the fold order is a
// function of (rows, width, threads) only, so the reduction stays bitwise reproducible without float
// atomics.  (The values can differ in the last bit from the two-stage collapse_fin_kernel, which
// folds 64 f64 rows: both are exact-to-f64 sums of the same f32 rows in different orders.)
//
// Ticket counters: reduce_fin.h's per-translation-unit g_fin_tickets (integer, self-resetting,
// handed out round-robin by the host).
#pragma once
#include "reduce_fin.h"

namespace segmi {

struct FinTail {
  int on;            // 0 = no finalisation in this launch
  unsigned ticket;   // index into g_fin_tickets of this translation unit
  unsigned nwg;      // workgroups of the launch: every one of them arrives exactly once
  int rows, width;   // the partial table [rows][width] f32
};

static inline FinTail fin_tail_make(int rows, int width, unsigned nwg) {
  FinTail t;
  t.on = 1;
  t.ticket = g_fin_next.fetch_add(1) % kFinTickets;
  t.nwg = nwg;
  t.rows = rows;
  t.width = width;
  return t;
}

// LDS bytes fin_tail_run needs behind `lds` for a workgroup of `threads` threads
static inline size_t fin_tail_lds(int width, int threads) {
  return ((size_t)width + 4 * (size_t)threads) * sizeof(double);
}

// Launcher side: arm the tail of a kernel whose workgroup blockIdx.x writes row blockIdx.x of a
// [grid.x][width] table (grid.y / grid.z split the columns).  Returns the dynamic LDS size to
// launch with (the kernel's own, raised to what the tail needs).  P has `int fin_on; FinTail ft;`.
template <class P>
static inline size_t fin_tail_arm(P& p, dim3 grid, int threads, int width, size_t lds) {
  if (!p.fin_on) { p.ft.on = 0; return lds; }
  p.ft = fin_tail_make((int)grid.x, width, grid.x * grid.y * grid.z);
  const size_t need = fin_tail_lds(width, threads);
  return lds > need ? lds : need;
}

// To be called by ALL threads of the workgroup, after the workgroup's own rows have been stored and
// when its LDS is free.  Returns true in the (single) workgroup that ran the finalisation.
template <class Fin>
__device__ __forceinline__ bool fin_tail_run(const FinTail& ft, const float* __restrict__ partials,
                                             void* lds, const Fin& fin) {
  __shared__ int s_fin_last;
  const int nt = blockDim.x, tid = threadIdx.x;
  __threadfence();   // release: this workgroup's rows are visible device-wide (all XCDs)
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = atomicAdd(&g_fin_tickets[ft.ticket], 1u);
    s_fin_last = prev == ft.nwg - 1;
    if (s_fin_last) g_fin_tickets[ft.ticket] = 0;   // ready for the next launch that draws this ticket
  }
  __syncthreads();
  if (!s_fin_last) return false;
  __threadfence();   // acquire: the other workgroups' rows
  double* sums = reinterpret_cast<double*>(lds);            // [width]
  double* red = sums + ft.width;                            // [4 * nt]
  const int rows = ft.rows, width = ft.width;
  const bool vec4 = width % 4 == 0 && width / 4 <= nt && ((uintptr_t)partials & 15) == 0;
  if (vec4) {
    const int wl4 = width / 4, rl4 = nt / wl4;
    const int col4 = tid % wl4, lane4 = tid / wl4;
    double s[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s[a][b] = 0.0;
    if (lane4 < rl4) {
      const f32x4* src = reinterpret_cast<const f32x4*>(partials) + col4;
      int r = lane4;
      for (; r + 7 * rl4 < rows; r += 8 * rl4) {             // 8 x 16-byte loads in flight per thread
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (int64_t)(r + u * rl4) * wl4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int b = 0; b < 4; ++b) s[u & 1][b] += (double)v[u][b];
      }
      for (; r < rows; r += rl4) {
        const f32x4 v = __builtin_nontemporal_load(src + (int64_t)r * wl4);
#pragma unroll
        for (int b = 0; b < 4; ++b) s[0][b] += (double)v[b];
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) red[tid * 4 + b] = s[0][b] + s[1][b];
    __syncthreads();
    for (int e = tid; e < width; e += nt) {
      const int c4 = e / 4, b = e % 4;
      double t = 0.0;
      for (int l = 0; l < rl4; ++l) t += red[(l * wl4 + c4) * 4 + b];
      sums[e] = t;
    }
  } else {
    const int wl = width < nt ? width : nt;
    const int rl = nt / wl;
    const int col = tid % wl, lane = tid / wl;
    for (int w0 = 0; w0 < width; w0 += wl) {
      const int e = w0 + col;
      double s0 = 0.0, s1 = 0.0;
      if (lane < rl && e < width) {
        int r = lane;
        for (; r + 7 * rl < rows; r += 8 * rl) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(partials + (int64_t)(r + u * rl) * width + e);
#pragma unroll
          for (int u = 0; u < 8; u += 2) { s0 += (double)v[u]; s1 += (double)v[u + 1]; }
        }
        for (; r < rows; r += rl) s0 += (double)__builtin_nontemporal_load(partials + (int64_t)r * width + e);
      }
      red[tid] = s0 + s1;
      __syncthreads();
      if (tid < wl && w0 + tid < width) {
        double t = 0.0;
        for (int l = 0; l < rl; ++l) t += red[l * wl + tid];
        sums[w0 + tid] = t;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  fin(sums, red);
  return true;
}

// ------------------------------------------------------------------ finalisation functors
// [2][c] column sums (sum, sum of squares) -> per-channel BatchNorm statistics (training mode):
// mean / invstd for the backward, scale / shift for the apply pass, running statistics update.
struct BnFin {
  int c;
  double count;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  float momentum, eps;
  float *mean, *invstd, *scale, *shift;
  __device__ void operator()(const double* sums, double*) const {
    for (int cc = threadIdx.x; cc < c; cc += blockDim.x) {
      const double m = sums[cc] / count;
      double var = sums[c + cc] / count - m * m;
      if (var < 0.0) var = 0.0;
      const float is = (float)(1.0 / sqrt(var + (double)eps));
      mean[cc] = (float)m;
      invstd[cc] = is;
      const float sc = (gamma ? gamma[cc] : 1.f) * is;
      scale[cc] = sc;
      shift[cc] = (beta ? beta[cc] : 0.f) - (float)m * sc;
      if (running_mean) running_mean[cc] = (1.f - momentum) * running_mean[cc] + momentum * (float)m;
      if (running_var) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[cc] = (1.f - momentum) * running_var[cc] + momentum * (float)unb;
      }
    }
  }
};

// [3][c] column sums -> dgamma, dbeta, dalpha (sum over channels), coef[2][c] = {mean dz, mean dz*xhat}
struct BnBwdFin {
  int c;
  double count;
  float *dgamma, *dbeta, *dalpha, *coef;
  __device__ void operator()(const double* sums, double* red) const {
    const int nt = blockDim.x;
    for (int cc = threadIdx.x; cc < c; cc += nt) {
      const double a0 = sums[cc], a1 = sums[c + cc];
      if (dbeta) dbeta[cc] = (float)a0;
      if (dgamma) dgamma[cc] = (float)a1;
      coef[cc] = (float)(a0 / count);
      coef[c + cc] = (float)(a1 / count);
    }
    if (dalpha) {   // fixed-order sum over channels: 256 strided partials, then a fixed-shape tree
      __syncthreads();
      double t = 0.0;
      if (threadIdx.x < 256)
        for (int cc = threadIdx.x; cc < c; cc += 256) t += sums[2 * c + cc];
      if (threadIdx.x < 256) red[threadIdx.x] = t;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
      }
      if (threadIdx.x == 0) *dalpha = (float)red[0];
    }
  }
};

static inline BnFin bn_fin_from(const segmi_bn_fin* f, int c) {
  return BnFin{c, f->count, f->gamma, f->beta, f->running_mean, f->running_var, f->momentum, f->eps,
               f->mean, f->invstd, f->scale, f->shift};
}
static inline bool bn_fin_ok(const segmi_bn_fin* f) {
  return f && f->count > 0 && f->mean && f->invstd && f->scale && f->shift;
}

}  // namespace segmi
