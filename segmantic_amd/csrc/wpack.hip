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

// Batched packing: blockIdx.y selects a descriptor (device table), one launch re-packs every
// convolution of the network after an optimiser step (46 launches -> 1).
template <typename T>
__global__ void wpack_batch_kernel(const segmi_wpack_desc* __restrict__ descs) {
  constexpr int KG = Elem<T>::KG;
  const segmi_wpack_desc d = descs[blockIdx.y];
  const int CK = (sizeof(T) == 2 && d.cin_k % 32 == 0) ? 32 : 16;
  const int SPT = CK / KG;
  const int nchunks = d.cin_k / CK, ntiles = d.cout_k / 16;
  const float* w = d.w_src;
  const float* scale = d.scale;
  T* out = (T*)d.packed;
  if (d.kind == 2) {
    int64_t base = 0;
    for (int p = 0; p < 8; ++p) {
      const int nt_p = ct_ntaps(p);
      const int nsteps = (nt_p * SPT + 3) / 4;
      const int64_t total = (int64_t)nchunks * nsteps * ntiles * 64 * KG;
      for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
           e += (int64_t)gridDim.x * blockDim.x) {
        int j = e % KG;
        int64_t r = e / KG;
        int lane = r % 64; r /= 64;
        int nt = r % ntiles; r /= ntiles;
        int s = r % nsteps;
        int c = r / nsteps;
        int q = 4 * s + (lane >> 4);
        int tap = q / SPT, sub = q % SPT;
        int ch = c * CK + sub * KG + j;
        int co = nt * 16 + (lane & 15);
        float v = 0.f;
        if (tap < nt_p) {
          int kd, kh, kw, dd, dh, dw;
          ct_tap(p, tap, kd, kh, kw, dd, dh, dw);
          // (two sources: the INPUT channels of the transposed convolution from cout_split on come from w_src2 --
          // the paired input gradient of two stride-2 convolutions of one input, whose forward weights
          // [c][cin][27] stack along this axis)
          if (d.w_src2 && ch >= d.cout_split)
            v = d.w_src2[((int64_t)(ch - d.cout_split) * d.cout_k + co) * 27 + (kd * 3 + kh) * 3 + kw];
          else
            v = w[((int64_t)ch * d.cout_k + co) * 27 + (kd * 3 + kh) * 3 + kw];
          if (scale) v *= scale[co];
        }
        Elem<T>::st(out + base + e, v);
      }
      base += total;
    }
  } else {
    const int ntaps = d.ksize * d.ksize * d.ksize;
    const int nsteps = (ntaps * SPT + 3) / 4;
    const int64_t total = (int64_t)nchunks * nsteps * ntiles * 64 * KG;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
      int j = e % KG;
      int64_t r = e / KG;
      int lane = r % 64; r /= 64;
      int nt = r % ntiles; r /= ntiles;
      int s = r % nsteps;
      int c = r / nsteps;
      int q = 4 * s + (lane >> 4);
      int tap = q / SPT, sub = q % SPT;
      int ch = c * CK + sub * KG + j;
      int co = nt * 16 + (lane & 15);
      float v = 0.f;
      if (tap < ntaps) {
        if (d.kind == 0) {
          if (d.w_src2 && co >= d.cout_split) v = d.w_src2[((int64_t)(co - d.cout_split) * d.cin_k + ch) * ntaps + tap];
          else v = w[((int64_t)co * d.cin_k + ch) * ntaps + tap];
        } else v = w[((int64_t)ch * d.cout_k + co) * ntaps + (ntaps - 1 - tap)];
        if (scale) v *= scale[co];
      }
      Elem<T>::st(out + e, v);
    }
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

// A stream whose kernels run on a subset of the compute units.  MI355X: 256 CUs = 8 XCDs x 32; KFD
// deals the bits of a CU mask round-robin over the XCDs (bit b -> XCD b % 8, local CU b / 8), so
// enabling bits [0, 8 k) gives the first k CUs of EVERY XCD -- the remaining 32 - k per XCD stay free
// for the other streams (which may still use all of them).
int segmi_stream_create_cumask(int cus_enabled, void** stream_out) {
  SEGMI_CHECK_ARG(stream_out && cus_enabled >= 8 && cus_enabled <= 256 && cus_enabled % 8 == 0,
                  "stream_create_cumask: cus_enabled must be a multiple of 8 in [8, 256]");
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < cus_enabled; ++b) mask[b >> 5] |= 1u << (b & 31);
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
  if (e != hipSuccess) {
    set_error("stream_create_cumask: %s", hipGetErrorString(e));
    return SEGMI_ELAUNCH;
  }
  *stream_out = (void*)st;
  return SEGMI_OK;
}
int segmi_stream_destroy(void* stream) {
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
    set_error("stream_destroy failed");
    return SEGMI_ELAUNCH;
  }
  return SEGMI_OK;
}

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

int segmi_wpack_batch(int dtype, const segmi_wpack_desc* descs_host, int ndesc,
                      segmi_wpack_desc* descs_dev, int upload, void* stream) {
  SEGMI_CHECK_ARG(descs_host && descs_dev, "wpack_batch: null descriptor table");
  SEGMI_CHECK_ARG(ndesc > 0 && ndesc <= 65535, "wpack_batch: ndesc %d out of range", ndesc);
  SEGMI_CHECK_ARG(dtype == SEGMI_F32 || dtype == SEGMI_BF16, "wpack_batch: bad dtype %d", dtype);
  for (int i = 0; i < ndesc; ++i) {
    const segmi_wpack_desc& d = descs_host[i];
    SEGMI_CHECK_ARG(d.w_src && d.packed, "wpack_batch[%d]: null pointer", i);
    SEGMI_CHECK_ARG(d.cin_k > 0 && d.cout_k > 0 && d.cin_k % 16 == 0 && d.cout_k % 16 == 0,
                    "wpack_batch[%d]: MFMA packs need cin %% 16 == 0 and cout %% 16 == 0", i);
    SEGMI_CHECK_ARG(d.kind >= 0 && d.kind <= 2, "wpack_batch[%d]: bad kind %d", i, d.kind);
    SEGMI_CHECK_ARG(d.ksize == 1 || d.ksize == 3, "wpack_batch[%d]: ksize must be 1 or 3", i);
    SEGMI_CHECK_ARG(d.kind != 2 || d.ksize == 3, "wpack_batch[%d]: transposed conv is k3 only", i);
    SEGMI_CHECK_ARG(!d.w_src2 || (d.kind == 0 && d.cout_split > 0 && d.cout_split < d.cout_k) ||
                        (d.kind == 2 && d.cout_split > 0 && d.cout_split < d.cin_k),
                    "wpack_batch[%d]: a second source needs kind 0 with 0 < cout_split < cout_k or kind 2 with 0 < "
                    "cout_split < cin_k", i);
  }
  hipStream_t st = (hipStream_t)stream;
  if (upload) {
    hipError_t e = hipMemcpyAsync(descs_dev, descs_host, sizeof(segmi_wpack_desc) * ndesc,
                                  hipMemcpyHostToDevice, st);
    if (e != hipSuccess) {
      set_error("wpack_batch: descriptor upload failed: %s", hipGetErrorString(e));
      return SEGMI_ELAUNCH;
    }
  }
  dim3 grid(64, ndesc);
  if (dtype == SEGMI_F32)
    hipLaunchKernelGGL(wpack_batch_kernel<float>, grid, 256, 0, st, descs_dev);
  else
    hipLaunchKernelGGL(wpack_batch_kernel<bf16_t>, grid, 256, 0, st, descs_dev);
  SEGMI_LAUNCH_CHECK("wpack_batch");
  return SEGMI_OK;
}

}  // extern "C"
