// wpack.hip -- fragment-packed weight layouts for the MFMA kernels + error plumbing.
#include "common.h"

namespace segmi {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// kind 0/1: [chunk][step][ntile][lane][KG]
template <typename T>
__global__ void wpack_conv_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                  T* __restrict__ out, int kind, int cin, int cout, int ntaps,
                                  int CK, int SPT, int nsteps, int ntiles, int64_t total) {
  constexpr int KG = Elem<T>::KG;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int j = e % KG;
    int64_t r = e / KG;
    int lane = r % 64; r /= 64;
    int nt = r % ntiles; r /= ntiles;
    int s = r % nsteps;
    int c = r / nsteps;
    int g = lane >> 4;
    int q = 4 * s + g;
    int tap = q / SPT, sub = q % SPT;
    int ch = c * CK + sub * KG + j;
    int co = nt * 16 + (lane & 15);
    float v = 0.f;
    if (tap < ntaps) {
      if (kind == 0) v = w[((int64_t)co * cin + ch) * ntaps + tap];
      else v = w[((int64_t)ch * cout + co) * ntaps + (ntaps - 1 - tap)];
      if (scale) v *= scale[co];
    }
    Elem<T>::st(out + e, v);
  }
}

// kind 2: [class][chunk][step][ntile][lane][KG]; src [cin][cout][27]
template <typename T>
__global__ void wpack_convT_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                   T* __restrict__ out, int cin, int cout, int CK, int SPT,
                                   int nchunks, int ntiles) {
  constexpr int KG = Elem<T>::KG;
  int p = blockIdx.y;
  int nt_p = ct_ntaps(p);
  int nsteps = (nt_p * SPT + 3) / 4;
  int64_t base = 0;
  for (int pp = 0; pp < p; ++pp)
    base += (int64_t)nchunks * ((ct_ntaps(pp) * SPT + 3) / 4) * ntiles * 64 * KG;
  int64_t total = (int64_t)nchunks * nsteps * ntiles * 64 * KG;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int j = e % KG;
    int64_t r = e / KG;
    int lane = r % 64; r /= 64;
    int nt = r % ntiles; r /= ntiles;
    int s = r % nsteps;
    int c = r / nsteps;
    int g = lane >> 4;
    int q = 4 * s + g;
    int tap = q / SPT, sub = q % SPT;
    int ch = c * CK + sub * KG + j;
    int co = nt * 16 + (lane & 15);
    float v = 0.f;
    if (tap < nt_p) {
      int kd, kh, kw, dd, dh, dw;
      ct_tap(p, tap, kd, kh, kw, dd, dh, dw);
      v = w[((int64_t)ch * cout + co) * 27 + (kd * 3 + kh) * 3 + kw];
      if (scale) v *= scale[co];
    }
    Elem<T>::st(out + base + e, v);
  }
}

static int64_t wpack_elems(int dtype, int kind, int cin, int cout, int ksize) {
  if (kind == 2) {
    PackGeom g = pack_geom(dtype, cin, cout, 1);
    int64_t steps = 0;
    for (int p = 0; p < 8; ++p) steps += (ct_ntaps(p) * g.SPT + 3) / 4;
    return (int64_t)g.nchunks * steps * g.ntiles * 64 * g.KG;
  }
  PackGeom g = pack_geom(dtype, cin, cout, ksize * ksize * ksize);
  return (int64_t)g.nchunks * g.nsteps * g.ntiles * 64 * g.KG;
}

}  // namespace segmi

using namespace segmi;

extern "C" {

int segmi_version(void) { return SEGMI_VERSION; }
const char* segmi_last_error(void) { return g_err; }

int64_t segmi_wpack_bytes(int dtype, int kind, int cin_k, int cout_k, int ksize) {
  if (cin_k % 16 || cout_k % 16 || cin_k <= 0 || cout_k <= 0) return 0;
  return wpack_elems(dtype, kind, cin_k, cout_k, ksize) * dtype_size(dtype);
}

int segmi_wpack(int dtype, int kind, const float* w_src, const float* scale, int cin_k,
                int cout_k, int ksize, void* packed, void* stream) {
  SEGMI_CHECK_ARG(w_src && packed, "wpack: null pointer");
  SEGMI_CHECK_ARG(cin_k > 0 && cout_k > 0 && cin_k % 16 == 0 && cout_k % 16 == 0,
                  "wpack: MFMA packs need cin %% 16 == 0 and cout %% 16 == 0 (got %d, %d)",
                  cin_k, cout_k);
  SEGMI_CHECK_ARG(kind >= 0 && kind <= 2, "wpack: bad kind %d", kind);
  SEGMI_CHECK_ARG(ksize == 1 || ksize == 3, "wpack: ksize must be 1 or 3");
  SEGMI_CHECK_ARG(kind != 2 || ksize == 3, "wpack: transposed conv is k3 only");
  hipStream_t st = (hipStream_t)stream;
  if (kind == 2) {
    PackGeom g = pack_geom(dtype, cin_k, cout_k, 1);
    dim3 grid(64, 8);
    if (dtype == SEGMI_F32)
      hipLaunchKernelGGL(wpack_convT_kernel<float>, grid, 256, 0, st, w_src, scale,
                         (float*)packed, cin_k, cout_k, g.CK, g.SPT, g.nchunks, g.ntiles);
    else
      hipLaunchKernelGGL(wpack_convT_kernel<bf16_t>, grid, 256, 0, st, w_src, scale,
                         (bf16_t*)packed, cin_k, cout_k, g.CK, g.SPT, g.nchunks, g.ntiles);
  } else {
    PackGeom g = pack_geom(dtype, cin_k, cout_k, ksize * ksize * ksize);
    int64_t total = wpack_elems(dtype, kind, cin_k, cout_k, ksize);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (dtype == SEGMI_F32)
      hipLaunchKernelGGL(wpack_conv_kernel<float>, blocks, 256, 0, st, w_src, scale,
                         (float*)packed, kind, cin_k, cout_k, g.ntaps, g.CK, g.SPT, g.nsteps,
                         g.ntiles, total);
    else
      hipLaunchKernelGGL(wpack_conv_kernel<bf16_t>, blocks, 256, 0, st, w_src, scale,
                         (bf16_t*)packed, kind, cin_k, cout_k, g.ntaps, g.CK, g.SPT, g.nsteps,
                         g.ntiles, total);
  }
  SEGMI_LAUNCH_CHECK("wpack");
  return SEGMI_OK;
}

}  // extern "C"
