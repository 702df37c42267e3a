// ensemble.hip -- combination of several models' predictions (reference
// src/segmantic/seg/monai_unet.py:848-1004: MeanEnsembled with validation-score weights,
// VoteEnsembled, and segmantic's own SelectBestEnsemble, src/segmantic/seg/transforms.py:15-88).
// Elementwise over the voxel grid, HBM-bound.
#include "common.h"

namespace segmi {

constexpr int kMaxModels = 16;
constexpr int kMaxTissues = 256;

struct EnsPtrs {
  int e;
  const void* p[kMaxModels];
  float w[kMaxModels];      // mean: weight_e / mean(weights) / E
};

// MONAI MeanEnsemble: stack * w / mean(w), then mean over the model axis
__global__ void ensemble_mean_kernel(EnsPtrs ep, int64_t n, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int m = 0; m < ep.e; ++m) s += ((const float*)ep.p[m])[i] * ep.w[m];
    out[i] = s;
  }
}

// MONAI VoteEnsemble on discrete labels: the most frequent label; ties -> the smallest label
// (argmax over the mean one-hot takes the first maximum)
__global__ void ensemble_vote_kernel(EnsPtrs ep, int64_t n, int32_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int lab[kMaxModels];
    for (int m = 0; m < ep.e; ++m) lab[m] = ((const int32_t*)ep.p[m])[i];
    int best = lab[0], bestc = 0;
    for (int a = 0; a < ep.e; ++a) {
      int c = 0;
      for (int b = 0; b < ep.e; ++b) c += lab[b] == lab[a];
      if (c > bestc || (c == bestc && lab[a] < best)) { best = lab[a]; bestc = c; }
    }
    out[i] = best;
  }
}

struct SelectMap {
  int n;
  int tissue[kMaxTissues];
  unsigned char model[kMaxTissues];
};

// SelectBestEnsemble: for (tissue, model) in dictionary order: out[pred_model == tissue] = tissue.
// Voxels no pair claims are background 0 (the reference leaves them uninitialised).
__global__ void ensemble_select_kernel(EnsPtrs ep, SelectMap sm, int64_t n, int32_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int o = 0;
    for (int t = 0; t < sm.n; ++t)
      if (((const int32_t*)ep.p[sm.model[t]])[i] == sm.tissue[t]) o = sm.tissue[t];
    out[i] = o;
  }
}

static inline int ens_grid(int64_t n) {
  const int64_t b = cdiv64(n, 256);
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace segmi

using namespace segmi;

extern "C" {

int segmi_ensemble_mean(const float* const* logits_host, const float* weights_host, int models,
                        int64_t n, float* out, void* stream) {
  SEGMI_CHECK_ARG(logits_host && out && models > 0 && models <= kMaxModels && n > 0,
                  "ensemble_mean: bad arguments (1..%d models)", kMaxModels);
  EnsPtrs ep{};
  ep.e = models;
  double wm = 0.0;
  for (int m = 0; m < models; ++m) wm += weights_host ? (double)weights_host[m] : 1.0;
  wm /= models;
  SEGMI_CHECK_ARG(wm != 0.0, "ensemble_mean: weights sum to zero");
  for (int m = 0; m < models; ++m) {
    SEGMI_CHECK_ARG(logits_host[m], "ensemble_mean: null model output");
    ep.p[m] = logits_host[m];
    ep.w[m] = (float)((weights_host ? (double)weights_host[m] : 1.0) / wm / models);
  }
  hipLaunchKernelGGL(ensemble_mean_kernel, ens_grid(n), 256, 0, (hipStream_t)stream, ep, n, out);
  SEGMI_LAUNCH_CHECK("ensemble_mean");
  return SEGMI_OK;
}

int segmi_ensemble_vote(const int32_t* const* labels_host, int models, int64_t n, int32_t* out,
                        void* stream) {
  SEGMI_CHECK_ARG(labels_host && out && models > 0 && models <= kMaxModels && n > 0,
                  "ensemble_vote: bad arguments (1..%d models)", kMaxModels);
  EnsPtrs ep{};
  ep.e = models;
  for (int m = 0; m < models; ++m) {
    SEGMI_CHECK_ARG(labels_host[m], "ensemble_vote: null model output");
    ep.p[m] = labels_host[m];
  }
  hipLaunchKernelGGL(ensemble_vote_kernel, ens_grid(n), 256, 0, (hipStream_t)stream, ep, n, out);
  SEGMI_LAUNCH_CHECK("ensemble_vote");
  return SEGMI_OK;
}

int segmi_ensemble_select(const int32_t* const* labels_host, int models, const int32_t* tissue_host,
                          const int32_t* model_host, int pairs, int64_t n, int32_t* out,
                          void* stream) {
  SEGMI_CHECK_ARG(labels_host && out && tissue_host && model_host && models > 0 &&
                      models <= kMaxModels && pairs > 0 && pairs <= kMaxTissues && n > 0,
                  "ensemble_select: bad arguments (1..%d models, 1..%d tissues)", kMaxModels, kMaxTissues);
  EnsPtrs ep{};
  ep.e = models;
  for (int m = 0; m < models; ++m) {
    SEGMI_CHECK_ARG(labels_host[m], "ensemble_select: null model output");
    ep.p[m] = labels_host[m];
  }
  SelectMap sm{};
  sm.n = pairs;
  for (int t = 0; t < pairs; ++t) {
    SEGMI_CHECK_ARG(model_host[t] >= 0 && model_host[t] < models,
                    "ensemble_select: model index %d out of range", model_host[t]);
    sm.tissue[t] = tissue_host[t];
    sm.model[t] = (unsigned char)model_host[t];
  }
  hipLaunchKernelGGL(ensemble_select_kernel, ens_grid(n), 256, 0, (hipStream_t)stream, ep, sm, n, out);
  SEGMI_LAUNCH_CHECK("ensemble_select");
  return SEGMI_OK;
}

}  // extern "C"
