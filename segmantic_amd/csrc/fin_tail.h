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
// atomics.  All kernels that carry a tail run 256-thread workgroups.  (The values can differ in the last bit from the two-stage collapse_fin_kernel, which
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

// ------------------------------------------------------------------ device side
// Cross-XCD visibility WITHOUT cache-wide fences.  A __threadfence() (release at agent scope) makes
// gfx950 write back every dirty line of the XCD's L2 (buffer_wbl2): issued by each of the thousands
// of workgroups of a convolution that is streaming its output through that L2, it costs ~80 us per
// launch (measured: 9.3 vs 6.5 ms per training step).  Instead the few floats that have to cross
// XCDs are moved with agent-scope relaxed atomics -- write-through stores (sc1) and coherent loads
// -- and ordered by hand: every thread waits for its own stores with an EXPLICIT `s_waitcnt vmcnt(0)`
// (fin_drain_stores; a workgroup-scope release fence emits no VMEM wait on gfx950 -- rounds 2-3 had that wait
// only because kernarg_late() happened to compile to a waited global_load), the workgroup barrier
// collects the waves, one thread takes the ticket.
__device__ __forceinline__ void fin_store(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every storing wave, after its fin_store()s and before the barrier in front of the ticket / flag: inline asm, so
// that no compiler pass can drop or move it (MI355X_MICROARCH.md, "Compiler hazard")
__device__ __forceinline__ void fin_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ float fin_load1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A field of the kernel's by-value parameter struct (first kernel argument = offset 0 of the kernarg
// segment), read HERE and not at kernel entry: the finalisation descriptors (a dozen pointers) must
// not occupy SGPRs for the whole lifetime of a register-bound MFMA kernel (with the fields read
// through `p.` hipcc hoists the s_loads to the top: 83 -> 101 SGPRs, VGPR spills of SGPRs, one
// k-split variant beyond 256 VGPRs).
template <class T>
__device__ __forceinline__ T kernarg_late(unsigned offset) {
  const __attribute__((address_space(4))) char* ka =
      (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  T v;
  __builtin_memcpy(&v, (const char*)ka + offset, sizeof(T));
  return v;
}

// To be called by ALL NT threads of the workgroup, after the workgroup's own rows have been stored
// with fin_store() and when its LDS is free (needs fin_tail_lds(width, NT) bytes behind `lds`).
// Returns true in the (single) workgroup that ran the finalisation.  FT_OFF / FIN_OFF: offsetof the
// FinTail / functor members in the kernel's parameter struct (see kernarg_late).
template <class Fin, int NT, unsigned FT_OFF, unsigned FIN_OFF>
__device__ __forceinline__ bool fin_tail_run(const float* __restrict__ partials, void* lds) {
  const FinTail ft = kernarg_late<FinTail>(FT_OFF);
  if (!ft.on) return false;
  __shared__ int s_fin_last;
  const int tid = threadIdx.x;
  fin_drain_stores();                                        // this thread's row stores have completed
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(&g_fin_tickets[ft.ticket], 1u, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
    s_fin_last = prev == ft.nwg - 1;
    if (s_fin_last)     // ready for the next launch that draws this ticket
      __hip_atomic_store(&g_fin_tickets[ft.ticket], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_fin_last) return false;
  double* sums = reinterpret_cast<double*>(lds);            // [width]
  double* red = sums + ft.width;                            // [4 * NT]
  const int rows = ft.rows, width = ft.width;
  // 16-byte columns: agent-coherent buffer loads (sc1), 16 per thread in flight = 64 KB per round trip
  // of the workgroup (8-byte atomic loads with 16 in flight folded a 2048 x 64 table in 19 us)
  const bool vec4 = width % 4 == 0 && width / 4 <= NT && ((uintptr_t)partials & 15) == 0 &&
                    (int64_t)rows * width * 4 < (1ll << 31);
  if (vec4) {
    const int wl4 = width / 4, rl4 = NT / wl4;
    const int col4 = tid % wl4, lane4 = tid / wl4;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)partials, 0, rows * width * 4, 0x00020000);
    constexpr int kSc1 = 16;                                  // cache policy: sc1 = agent-scope coherent
    double s[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4) s[a][b4] = 0.0;
    if (lane4 < rl4) {
      const int rstep = rl4 * width * 4;                      // bytes between this thread's rows
      int off = (lane4 * width + 4 * col4) * 4, r = lane4;
      for (; r + 15 * rl4 < rows; r += 16 * rl4, off += 16 * rstep) {
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + u * rstep, 0, kSc1);
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int b4 = 0; b4 < 4; ++b4) s[u & 1][b4] += (double)__uint_as_float(v[u][b4]);
      }
      for (; r < rows; r += rl4, off += rstep) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kSc1);
#pragma unroll
        for (int b4 = 0; b4 < 4; ++b4) s[0][b4] += (double)__uint_as_float(v[b4]);
      }
    }
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) red[tid * 4 + b4] = s[0][b4] + s[1][b4];
    __syncthreads();
    for (int e = tid; e < width; e += NT) {
      const int c4 = e / 4, b4 = e % 4;
      double t = 0.0;
      for (int l = 0; l < rl4; ++l) t += red[(l * wl4 + c4) * 4 + b4];
      sums[e] = t;
    }
  } else {
    const int wl = width < NT ? width : NT;
    const int rl = NT / wl;
    const int col = tid % wl, lane = tid / wl;
    for (int w0 = 0; w0 < width; w0 += wl) {
      const int e = w0 + col;
      double s0 = 0.0, s1 = 0.0;
      if (lane < rl && e < width) {
        int r = lane;
        for (; r + 7 * rl < rows; r += 8 * rl) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = fin_load1(partials + (int64_t)(r + u * rl) * width + e);
#pragma unroll
          for (int u = 0; u < 8; u += 2) { s0 += (double)v[u]; s1 += (double)v[u + 1]; }
        }
        for (; r < rows; r += rl) s0 += (double)fin_load1(partials + (int64_t)r * width + e);
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
  const Fin fin = kernarg_late<Fin>(FIN_OFF);
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
    if (threadIdx.x >= 256) return;        // (workgroups of more than 256 threads: one thread per channel slot -- the
                                           // running statistics are read-modify-written)
    for (int cc = threadIdx.x; cc < c; cc += 256) {
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
    for (int cc = threadIdx.x; cc < c; cc += 256) {
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
