// loss_optim.hip -- fused softmax + one-hot + Dice loss (forward / backward) and the
// flat-arena optimisers (Adam, SGD, AdaBelief).  HBM-bound: logits are read once per pass in
// 16-byte vectors along the class axis (NDHWC keeps the K classes of a voxel contiguous), the
// softmax lives in registers, reductions are two-stage and deterministic.
#include "common.h"
#include "fin_tail.h"
#include <type_traits>

namespace segmi {

constexpr int kDiceVox = 8192;  // voxels per workgroup


// one block: per (n,k) sums of the collapsed rows (f64), loss + backward coefficients
struct DiceFin {
  int n, k;
  float smooth_nr, smooth_dr;
  float *coef, *loss;
  // sums: [n][3][k] = {intersection, sum p, sum t}
  __device__ void operator()(const double* sums, double* red) const {
    const int tid = threadIdx.x;
    double local = 0.0;
    const double nk = (double)n * k;
    for (int o = tid; o < n * k; o += 256) {
      const int b = o / k, j = o % k;
      const double* q = sums + ((int64_t)b * 3) * k + j;
      const double I = q[0], P = q[k], Tt = q[2 * k];
      // f32 arithmetic as the reference does on the reduced sums
      const float If = (float)I, Df = (float)Tt + (float)P;
      const float f = 1.0f - (2.0f * If + smooth_nr) / (Df + smooth_dr);
      local += (double)f;
      const double den = (double)Df + (double)smooth_dr;
      coef[((int64_t)b * 2 + 0) * k + j] = (float)(-2.0 / den / nk);
      coef[((int64_t)b * 2 + 1) * k + j] = (float)((2.0 * (double)If + (double)smooth_nr) / (den * den) / nk);
    }
    red[tid] = local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) *loss = (float)(red[0] / nk);
  }
};

struct ChanSumFin {   // sums: [k] channel sums
  int k;
  float* db;
  __device__ void operator()(const double* sums, double*) const {
    for (int ch = threadIdx.x; ch < k; ch += 256) db[ch] = (float)sums[ch];
  }
};

struct DiceParams {
  const void* logits;
  const float* labels;
  float* partials;   // [n][chunks][3][k]
  const float* coef; // [n][2][k]
  void* dlogits;
  int n, k, ld, ldd;
  int64_t vox;       // voxels per batch item
  int chunks;
  float grad_scale;
  float* bias_part;  // [n * chunks][k] per-workgroup channel sums of the written gradient (nullable)
  // finalisation by the last workgroup of the launch (fin_tail.h): forward -> DiceFin over `partials`,
  // backward -> ChanSumFin over `bias_part`
  FinTail ft;
  DiceFin dfin;
  ChanSumFin cfin;
};

// VECLD: the caller has checked ONCE per workgroup (logits_vec_ok) that every voxel's class row is
// 16-byte aligned -- a per-voxel alignment test is a divergent branch around every load and keeps the
// loads of an unrolled trip from being issued together
template <typename T>
__device__ __forceinline__ bool logits_vec_ok(const T* base, int k, int ld) {
  constexpr int VEC = 16 / sizeof(T);
  return k % VEC == 0 && ld % VEC == 0 && ((uintptr_t)base % 16) == 0;
}
template <typename T, int KMAX, bool VECLD>
__device__ __forceinline__ void load_logits(const T* p, int k, float (&v)[KMAX]) {
  constexpr int VEC = 16 / sizeof(T);
  if constexpr (VECLD) {
#pragma unroll
    for (int j = 0; j < KMAX; j += VEC) {
      if (j < k) {
        const frag_t f = *reinterpret_cast<const frag_t*>(p + j);
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (j + e < KMAX) v[j + e] = __uint_as_float(f[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (j + 2 * e < KMAX) v[j + 2 * e] = __uint_as_float(f[e] << 16);
            if (j + 2 * e + 1 < KMAX) v[j + 2 * e + 1] = __uint_as_float(f[e] & 0xffff0000u);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j < k) v[j] = Elem<T>::ld(p + j);
  }
}

// FAST: hardware exp2 (v_exp_f32, ~1 ulp in exp2 but ~2e-7 relative after the log2(e) scaling)
// for the bf16 path, where the logits carry 8 bits anyway; the f32 parity path keeps the
// correctly-rounded libm expf.
template <int KMAX, bool FAST>
__device__ __forceinline__ void softmax_inplace(int k, float (&v)[KMAX]) {
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) if (j < k) m = fmaxf(m, v[j]);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) if (j < k) {
    v[j] = FAST ? __builtin_amdgcn_exp2f((v[j] - m) * 1.4426950408889634f) : expf(v[j] - m);
    s += v[j];
  }
  const float inv = 1.f / s;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) if (j < k) v[j] *= inv;
}

// FULL: k == KMAX (16 / 32 / 64 labels): every per-channel predicate folds away at compile time
template <typename T, int KMAX, bool FULL>
__global__ __launch_bounds__(256) void dice_fwd_kernel(DiceParams p) {
  __shared__ float red[4][3 * KMAX];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t v0 = (int64_t)chunk * kDiceVox;
  const int64_t v1 = v0 + kDiceVox < p.vox ? v0 + kDiceVox : p.vox;
  const T* lg = (const T*)p.logits + (int64_t)n * p.vox * p.ld;
  const float* lb = p.labels + (int64_t)n * p.vox;
  float si[KMAX], sp[KMAX], stt[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) si[j] = sp[j] = stt[j] = 0.f;
  // U voxels per trip with every load issued before the first softmax (one voxel per trip behind a
  // per-voxel alignment branch left the 32-byte loads exposed: 3.4 TB/s); the sums keep their voxel
  // order, so the partials keep their bits
  auto sweep = [&](auto vec_tag) {
    constexpr bool VL = decltype(vec_tag)::value;
    constexpr int U = KMAX <= 16 ? 4 : (KMAX <= 32 ? 2 : 1);
    int64_t v = v0 + tid;
    for (; v + 256 * (U - 1) < v1; v += 256 * U) {
      float x[U][KMAX], lf[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        load_logits<T, KMAX, VL>(lg + (v + 256 * u) * p.ld, FULL ? KMAX : p.k, x[u]);
        lf[u] = __builtin_nontemporal_load(lb + v + 256 * u);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        softmax_inplace<KMAX, sizeof(T) == 2>(FULL ? KMAX : p.k, x[u]);
        const int lab = (int)lf[u];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          if (FULL || j < p.k) {
            sp[j] += x[u][j];
            if (j == lab) { si[j] += x[u][j]; stt[j] += 1.f; }
          }
        }
      }
    }
    for (; v < v1; v += 256) {
      float x[KMAX];
      load_logits<T, KMAX, VL>(lg + v * p.ld, FULL ? KMAX : p.k, x);
      softmax_inplace<KMAX, sizeof(T) == 2>(FULL ? KMAX : p.k, x);
      const int lab = (int)lb[v];
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        if (FULL || j < p.k) {
          sp[j] += x[j];
          if (j == lab) { si[j] += x[j]; stt[j] += 1.f; }
        }
      }
    }
  };
  if (logits_vec_ok<T>(lg, p.k, p.ld)) sweep(std::true_type{});
  else sweep(std::false_type{});
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (FULL || j < p.k) {
      const float a = wave_sum(si[j]), b = wave_sum(sp[j]), c = wave_sum(stt[j]);
      if (lane == 0) { red[wave][j] = a; red[wave][KMAX + j] = b; red[wave][2 * KMAX + j] = c; }
    }
  }
  __syncthreads();
  if (tid < 3 * p.k) {
    const int which = tid / p.k, j = tid % p.k;
    const float s = red[0][which * KMAX + j] + red[1][which * KMAX + j] +
                    red[2][which * KMAX + j] + red[3][which * KMAX + j];
    fin_store(&p.partials[(((int64_t)chunk * p.n + n) * 3 + which) * p.k + j], s);   // [chunk][n][3][k]
  }
  {
    extern __shared__ double dice_tail_lds[];
    fin_tail_run<DiceFin, 256, offsetof(DiceParams, ft), offsetof(DiceParams, dfin)>(p.partials, dice_tail_lds);
  }
}

template <typename T, int KMAX, bool FULL>
__global__ __launch_bounds__(256) void dice_bwd_kernel(DiceParams p) {
  __shared__ float cf[2 * KMAX];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid < 2 * p.k) cf[(tid / p.k) * KMAX + tid % p.k] = p.coef[(int64_t)n * 2 * p.k + tid];
  __syncthreads();
  const int64_t v0 = (int64_t)chunk * kDiceVox;
  const int64_t v1 = v0 + kDiceVox < p.vox ? v0 + kDiceVox : p.vox;
  const T* lg = (const T*)p.logits + (int64_t)n * p.vox * p.ld;
  T* dl = (T*)p.dlogits + (int64_t)n * p.vox * p.ldd;
  const float* lb = p.labels + (int64_t)n * p.vox;
  float gsum[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) gsum[j] = 0.f;
  const bool vec_in = logits_vec_ok<T>(lg, p.k, p.ld);
  const bool vec_out = (FULL || p.k % 4 == 0) && p.ldd % 4 == 0 && ((uintptr_t)dl % (4 * sizeof(T))) == 0;
  for (int64_t v = v0 + tid; v < v1; v += 256) {
    float x[KMAX];
    if (vec_in) load_logits<T, KMAX, true>(lg + v * p.ld, FULL ? KMAX : p.k, x);
    else load_logits<T, KMAX, false>(lg + v * p.ld, FULL ? KMAX : p.k, x);
    softmax_inplace<KMAX, sizeof(T) == 2>(FULL ? KMAX : p.k, x);
    const int lab = (int)lb[v];
    float dot = 0.f;
    float dp[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (FULL || j < p.k) {
        dp[j] = cf[KMAX + j] + (j == lab ? cf[j] : 0.f);
        dot = fmaf(x[j], dp[j], dot);
      }
    }
    T* o = dl + v * p.ldd;
    if (vec_out) {
#pragma unroll
      for (int j = 0; j < KMAX; j += 4) {
        if (FULL || j < p.k) {
          f32x4 g4;
#pragma unroll
          for (int e = 0; e < 4; ++e) g4[e] = (j + e < KMAX) ? p.grad_scale * x[j + e] * (dp[j + e] - dot) : 0.f;
          store4<T>(o + j, g4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < KMAX) gsum[j + e] += g4[e];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (FULL || j < p.k) {
          const float gv = p.grad_scale * x[j] * (dp[j] - dot);
          Elem<T>::st(o + j, gv);
          gsum[j] += gv;
        }
    }
  }
  // bias gradient of the layer that produced the logits = channel sums of dlogits: folded here so
  // the 537 MB tensor is not read once more for it (fixed-order: wave butterfly, then 4 waves)
  if (p.bias_part) {
    __shared__ float bsum[4][KMAX];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      const float s = wave_sum(gsum[j]);
      if (lane == 0) bsum[wave][j] = s;
    }
    __syncthreads();
    if (tid < p.k)
      fin_store(&p.bias_part[((int64_t)n * p.chunks + chunk) * p.k + tid],
                (bsum[0][tid] + bsum[1][tid]) + (bsum[2][tid] + bsum[3][tid]));
    {
      extern __shared__ double dice_tail_lds[];
      fin_tail_run<ChanSumFin, 256, offsetof(DiceParams, ft), offsetof(DiceParams, cfin)>(p.bias_part, dice_tail_lds);
    }
  }
}

// ---------------------------------------------------------------- optimisers
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ vmax, int64_t n, float omb1, float beta2,
                            float omb2, float eps, float wd, float step_size, float bc2_sqrt,
                            float grad_scale) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * grad_scale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    float mi = m[i];
    mi = mi + omb1 * (gi - mi);                   // exp_avg.lerp_(grad, 1 - beta1)
    float vi = v[i] * beta2;
    vi = fmaf(omb2 * gi, gi, vi);                 // mul_(beta2).addcmul_(g, g, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    float vv = vi;
    if (vmax) { vv = fmaxf(vmax[i], vi); vmax[i] = vv; }
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                           float* __restrict__ buf, int64_t n, float lr, float momentum,
                           float wd, int first, float grad_scale) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * grad_scale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    if (momentum != 0.f) {
      const float b = first ? gi : fmaf(buf[i], momentum, gi);
      buf[i] = b;
      gi = b;
    }
    p[i] = pi - lr * gi;
  }
}

__global__ void adabelief_kernel(float* __restrict__ p, const float* __restrict__ g,
                                 float* __restrict__ m, float* __restrict__ s, int64_t n,
                                 float decay, float beta1, float omb1, float beta2,
                                 float omb2, float eps, float wd, int decouple,
                                 float step_size, float bc2_sqrt, float grad_scale) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * grad_scale;
    float pi = p[i];
    if (wd != 0.f) {
      if (decouple) pi *= decay;             // p.mul_(1 - lr * weight_decay)
      else gi = fmaf(wd, pi, gi);
    }
    const float mi = fmaf(omb1, gi, m[i] * beta1);     // mul_(beta1).add_(g, alpha=1 - beta1)
    const float r = gi - mi;
    float si = fmaf(omb2 * r, r, s[i] * beta2);        // mul_(beta2).addcmul_(r, r, value=1 - beta2)
    si += eps;  // exp_avg_var.add_(eps) is in place in adabelief_pytorch
    m[i] = mi;
    s[i] = si;
    const float denom = sqrtf(si) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

static inline int opt_blocks(int64_t n) {
  const int64_t b = cdiv64(n, 256);
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

template <typename T>
static int dice_dispatch(bool fwd, const DiceParams& p, hipStream_t st) {
  dim3 grid(p.chunks, p.n);
  const size_t lds = p.ft.on ? fin_tail_lds(p.ft.width, 256) : 0;
#define DICE_K(KM)                                                                       \
  do {                                                                                   \
    if (p.k == KM) {                                                                     \
      if (fwd) hipLaunchKernelGGL((dice_fwd_kernel<T, KM, true>), grid, 256, lds, st, p);  \
      else hipLaunchKernelGGL((dice_bwd_kernel<T, KM, true>), grid, 256, lds, st, p);      \
    } else {                                                                             \
      if (fwd) hipLaunchKernelGGL((dice_fwd_kernel<T, KM, false>), grid, 256, lds, st, p); \
      else hipLaunchKernelGGL((dice_bwd_kernel<T, KM, false>), grid, 256, lds, st, p);     \
    }                                                                                    \
  } while (0)
  if (p.k <= 4) DICE_K(4);
  else if (p.k <= 16) DICE_K(16);
  else if (p.k <= 32) DICE_K(32);
  else if (p.k <= 64) DICE_K(64);
  else SEGMI_UNSUPPORTED("softmax_dice: at most 64 classes (got %d)", p.k);
#undef DICE_K
  SEGMI_LAUNCH_CHECK("softmax_dice");
  return SEGMI_OK;
}

}  // namespace segmi

using namespace segmi;

extern "C" {

static inline int dice_real_chunks(const segmi_act* logits) {
  return (int)cdiv64((int64_t)logits->d * logits->h * logits->w, kDiceVox);
}

// chunks the caller must allocate: the real ones + a tail that holds the f64 stage-1 reduction
// of reduce_fin.h (65 x [n][3][k] doubles), same scheme as the *_stats_rows() queries
int segmi_dice_chunks(const segmi_act* logits) {
  if (!logits) return 0;
  return dice_real_chunks(logits) + 2 * kFinScratchRows + 1;
}

int segmi_softmax_dice_fwd(int dtype, const segmi_act* logits, const float* labels,
                           float* partials, float* coef, float* loss, float smooth_nr,
                           float smooth_dr, void* stream) {
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "softmax_dice_fwd: bad dtype");
  SEGMI_CHECK_ARG(act_ok(logits) && labels && partials && coef && loss, "softmax_dice_fwd: bad arguments");
  DiceParams p{};
  p.logits = logits->data; p.labels = labels; p.partials = partials;
  p.n = logits->n; p.k = logits->c; p.ld = logits->ld;
  p.vox = (int64_t)logits->d * logits->h * logits->w;
  p.chunks = dice_real_chunks(logits);
  hipStream_t st = (hipStream_t)stream;
  // the loss and the backward coefficients are written by the last workgroup of the launch itself
  // (fin_tail.h): rows = chunks, one row = [n][3][k]
  p.ft = fin_tail_make(p.chunks, p.n * 3 * p.k, (unsigned)p.chunks * (unsigned)p.n);
  p.dfin = DiceFin{p.n, p.k, smooth_nr, smooth_dr, coef, loss};
  return dtype == SEGMI_F32 ? dice_dispatch<float>(true, p, st) : dice_dispatch<bf16_t>(true, p, st);
}

int segmi_softmax_dice_bwd(int dtype, const segmi_act* logits, const float* labels,
                           const float* coef, float grad_scale, const segmi_act* dlogits,
                           float* scratch, float* bias_grad, void* stream) {
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "softmax_dice_bwd: bad dtype");
  SEGMI_CHECK_ARG(act_ok(logits) && act_ok(dlogits) && labels && coef &&
                      logits->n == dlogits->n && logits->d == dlogits->d &&
                      logits->h == dlogits->h && logits->w == dlogits->w &&
                      logits->c == dlogits->c, "softmax_dice_bwd: bad arguments");
  DiceParams p{};
  p.logits = logits->data; p.labels = labels; p.coef = coef; p.dlogits = dlogits->data;
  p.n = logits->n; p.k = logits->c; p.ld = logits->ld; p.ldd = dlogits->ld;
  p.vox = (int64_t)logits->d * logits->h * logits->w;
  p.chunks = dice_real_chunks(logits);
  p.grad_scale = grad_scale;
  SEGMI_CHECK_ARG(!bias_grad || scratch, "softmax_dice_bwd: bias_grad needs the scratch buffer");
  p.bias_part = bias_grad ? scratch : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (bias_grad) {   // channel sums of the written gradient, folded by the last workgroup of the launch
    p.ft = fin_tail_make(p.n * p.chunks, p.k, (unsigned)p.chunks * (unsigned)p.n);
    p.cfin = ChanSumFin{p.k, bias_grad};
  }
  return dtype == SEGMI_F32 ? dice_dispatch<float>(false, p, st) : dice_dispatch<bf16_t>(false, p, st);
}

int segmi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    float* max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                    double eps, double weight_decay, int64_t step, float grad_scale,
                    void* stream) {
  SEGMI_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, "adam: bad arguments");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, opt_blocks(n), 256, 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, max_exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps, (float)weight_decay, step_size, bc2_sqrt,
                     grad_scale);
  SEGMI_LAUNCH_CHECK("adam");
  return SEGMI_OK;
}

int segmi_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, double lr,
                   double momentum, double weight_decay, int first_step, float grad_scale,
                   void* stream) {
  SEGMI_CHECK_ARG(param && grad && n > 0 && (momentum == 0.0 || momentum_buf), "sgd: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, opt_blocks(n), 256, 0, (hipStream_t)stream, param, grad,
                     momentum_buf, n, (float)lr, (float)momentum, (float)weight_decay, first_step,
                     grad_scale);
  SEGMI_LAUNCH_CHECK("sgd");
  return SEGMI_OK;
}

int segmi_adabelief_step(float* param, const float* grad, float* exp_avg, float* exp_avg_var,
                         int64_t n, double lr, double beta1, double beta2, double eps,
                         double weight_decay, int weight_decouple, int64_t step,
                         float grad_scale, void* stream) {
  SEGMI_CHECK_ARG(param && grad && exp_avg && exp_avg_var && n > 0 && step > 0, "adabelief: bad arguments");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adabelief_kernel, opt_blocks(n), 256, 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_var, n, (float)(1.0 - lr * weight_decay), (float)beta1,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay, weight_decouple, (float)(lr / bc1), (float)sqrt(bc2),
                     grad_scale);
  SEGMI_LAUNCH_CHECK("adabelief");
  return SEGMI_OK;
}

}  // extern "C"
