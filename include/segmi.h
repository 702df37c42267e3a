/* segmi.h -- C ABI of libsegmi.so: the MI355X (gfx950) hot path behind segmantic's
 * `segmantic-unet train / train-config / predict` surface.
 *
 * The reference (dyollb/segmantic) is pure Python and has no FFI of its own: its arithmetic is
 * delegated to torch.nn / MONAI / SimpleITK.  Each entry point
 * below therefore cites the reference *call site* whose third-party operator it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - every function returns 0 on success, a negative SEGMI_E* code on failure;
 *    segmi_last_error() returns a thread-local message for the last failure.
 *  - all pointers are DEVICE pointers unless the name ends in _host; nothing is allocated or
 *    freed behind the caller's back, no ownership is transferred.
 *  - `stream` is a hipStream_t (0 = default stream); every call is asynchronous on it.
 *  - activations are NDHWC ("channels last"): element (n,z,y,x,c) of a view lives at
 *    data[(((n*d + z)*h + y)*w + x)*ld + c]; ld >= c allows concat-by-offset views.
 *  - dtype: SEGMI_F32 (exact-f32 MFMA path, parity mode) or SEGMI_BF16 (bf16 storage,
 *    f32 accumulation).
 */
#ifndef SEGMI_H_
#define SEGMI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGMI_VERSION 1

enum { SEGMI_F32 = 0, SEGMI_BF16 = 1 };
enum { SEGMI_OK = 0, SEGMI_EINVAL = -1, SEGMI_EUNSUPPORTED = -2, SEGMI_ELAUNCH = -3 };

/* NDHWC activation view */
typedef struct segmi_act {
  void* data;
  int32_t n, d, h, w; /* batch and spatial extents            */
  int32_t c;          /* channels of this view                */
  int32_t ld;         /* elements between consecutive voxels  */
} segmi_act;

int segmi_version(void);
const char* segmi_last_error(void);
/* A HIP stream restricted to `cus_enabled` compute units (a multiple of 8: the first cus_enabled / 8 CUs
 * of every XCD; hipExtStreamCreateWithCUMask).  For the weight-gradient stream of a training step, which
 * the reference leaves to autograd's single stream (monai_unet.py:345): the persistent weight-gradient
 * kernels then cannot take every CU away from the dependent chain of the main stream. */
int segmi_stream_create_cumask(int cus_enabled, void** stream_out);
int segmi_stream_destroy(void* stream);

/* ---------------------------------------------------------------- weights -------------- */
/* Fragment-packed weights: the layout the MFMA kernels stream as B/A operands.
 * kind: 0 = Conv3d forward            (src = torch Conv3d weight [Cout][Cin][k][k][k])
 *       1 = Conv3d stride-1 dgrad     (same src; flipped taps, channels transposed)
 *       2 = transposed-conv kernel    (src = torch ConvTranspose3d weight [Cin][Cout][3][3][3],
 *                                      or a stride-2 Conv3d weight for its dgrad)
 * cin/cout are those of the *source* tensor's role in the kernel that will consume the pack
 * (see DESIGN.md "weight packs").  scale (nullable, f32[cout_k]) folds a per-output-channel
 * factor (eval-mode BatchNorm) into the weights.
 * Replaces: weight handling inside torch.nn.Conv3d / ConvTranspose3d reached from
 * src/segmantic/seg/monai_unet.py:114-124,221-222. */
int64_t segmi_wpack_bytes(int dtype, int kind, int cin_k, int cout_k, int ksize);
int segmi_wpack(int dtype, int kind, const float* w_src, const float* scale, int cin_k,
                int cout_k, int ksize, void* packed, void* stream);

/* Batched form: one launch re-packs every convolution of a network after an optimiser step
 * (the per-step equivalent of torch re-reading .weight in each Conv3d.forward/backward).
 * `descs_host[ndesc]` is validated on the host on every call; when `upload` != 0 it is copied
 * (stream-ordered) into `descs_dev` (device, ndesc * sizeof(segmi_wpack_desc) bytes) first,
 * otherwise `descs_dev` must already hold the same table from an earlier call. */
typedef struct segmi_wpack_desc {
  const float* w_src;  /* device f32 source weight                                   */
  const float* scale;  /* device f32[cout_k] or NULL                                 */
  void* packed;        /* device destination, segmi_wpack_bytes(...) bytes            */
  int32_t kind, cin_k, cout_k, ksize;
  /* kind 0: output channels >= cout_split come from a second source [cout_k - cout_split][cin_k][taps] (the pair
   * pack of segmi_conv3d_fwd_split_act out of two separate parameter tensors); kind 2: INPUT channels >= cout_split
   * come from the second source [cin_k - cout_split][cout_k][27] (the paired input gradient of those two
   * convolutions: one transposed convolution over their concatenated output gradients); NULL = one source */
  const float* w_src2;
  int32_t cout_split, reserved;
} segmi_wpack_desc;
int segmi_wpack_batch(int dtype, const segmi_wpack_desc* descs_host, int ndesc,
                      segmi_wpack_desc* descs_dev, int upload, void* stream);

/* ---------------------------------------------------------------- convolution ---------- */
/* Conv3d k in {1,3}, stride in {1,2}, pad (k-1)/2, fused epilogue:
 *   v = conv(in) + bias ; stats += (v, v^2) ; v = prelu(v) ; v += residual ; out = v
 * bias f32[cout] nullable; prelu_alpha device f32* nullable; residual nullable;
 * stats_partials nullable: f32[segmi_conv3d_stats_rows(...)][2][cout] per-workgroup partial
 * sums of v and v^2 over valid voxels (deterministic; reduced by segmi_bn_finalize).
 * MFMA path needs cin % 16 == 0 and cout % 16 == 0 with a kind-0/1 pack; otherwise pass
 * packed = NULL and w_src = torch-layout f32 weights for the direct kernel.
 * Replaces torch.nn.Conv3d under monai UNet, src/segmantic/seg/monai_unet.py:114-124,341. */
int segmi_conv3d_stats_rows(int dtype, const segmi_act* in, const segmi_act* out, int ksize,
                            int stride);
/* Optional input transform of a convolution (forward and weight gradient): the consumer applies the
 * PRODUCER layer's BatchNorm-apply + PReLU, in' = prelu(in * scale[c] + shift[c]) rounded to the
 * storage type, while it stages its input, so the normalised activation is never written to HBM
 * (zero padding stays zero; an identity residual `residual == in` adds in').  scale / shift: device
 * f32[cin] as written by segmi_bn_finalize; prelu_alpha: device f32* or NULL.  Bit-identical to
 * running segmi_bn_act_fwd first.  segmi_conv3d_in_affine_ok() says whether the layer's kernels
 * implement it (bf16 z-marching ring forward + MFMA weight gradient); pass NULL otherwise.
 * Replaces the separate ADN pass between two torch modules of monai UNet, monai_unet.py:114-124. */
typedef struct segmi_in_affine {
  const float* scale;
  const float* shift;
  const float* prelu_alpha;
} segmi_in_affine;
/* Finalisation of a BatchNorm reduction carried out BY THE LAUNCH THAT PRODUCES THE PARTIAL ROWS: the
 * workgroup that finishes last folds the rows in a fixed order (f64) and writes the per-channel
 * results, so that no separate segmi_bn_finalize / segmi_bn_act_bwd_finalize launch sits on the
 * dependent chain of the stream (csrc/fin_tail.h; ~40 launches per training step).  Deterministic:
 * the fold order depends on the table shape only.  Same argument meaning as the separate calls.
 * Replaces the statistics half of torch.nn.BatchNorm3d forward / backward under monai ADN,
 * monai_unet.py:114-124, 345. */
typedef struct segmi_bn_fin {       /* = the arguments of segmi_bn_finalize */
  double count;                     /* voxels per channel (N*D*H*W of the conv output) */
  const float* gamma;               /* nullable = 1 */
  const float* beta;                /* nullable = 0 */
  float* running_mean;              /* nullable */
  float* running_var;               /* nullable */
  float momentum, eps;
  float* mean;
  float* invstd;
  float* scale;
  float* shift;
} segmi_bn_fin;
typedef struct segmi_bn_bwd_fin {   /* = the arguments of segmi_bn_act_bwd_finalize */
  double count;
  float* dgamma;                    /* nullable */
  float* dbeta;                     /* nullable */
  float* dalpha;                    /* nullable */
  float* coef;                      /* f32[2][c] for segmi_bn_act_bwd_apply */
} segmi_bn_bwd_fin;
/* Optional epilogue of an INPUT-GRADIENT convolution (segmi_conv3d_fwd with a kind-1 pack) whose
 * output `out` is the gradient g flowing into a training-mode BatchNorm + PReLU: with x = that
 * layer's forward input (the raw output of its producer conv, same extents as `out`) the kernel
 * also accumulates the three per-channel sums that segmi_bn_act_bwd_reduce would compute from
 * (g, x) in a separate 2-tensor pass -- sum dz, sum dz*xhat, sum g*z[z<=0] -- taken of the STORED
 * (rounded) gradient, one partial row per workgroup: partials f32 [rows][3][c] with rows =
 * segmi_conv3d_stats_rows(in, out, ksize, stride); feed them to segmi_bn_act_bwd_finalize.
 * segmi_conv3d_bn_bwd_sums_ok() says whether the layer's kernel implements it (bf16 z-marching
 * ring, 16 -> 16 channels); pass NULL otherwise.  Replaces autograd's separate BatchNorm / PReLU
 * backward reductions under monai ADN, monai_unet.py:114-124, 345. */
typedef struct segmi_bn_bwd_sums {
  const segmi_act* x;
  const float* mean;
  const float* invstd;
  const float* gamma;        /* nullable = 1 */
  const float* beta;         /* nullable = 0 */
  const float* prelu_alpha;  /* nullable = no activation */
  float* partials;
  const segmi_bn_bwd_fin* fin;   /* nullable: also finalise in this launch (then no segmi_bn_act_bwd_finalize) */
} segmi_bn_bwd_sums;
int segmi_conv3d_bn_bwd_sums_ok(int dtype, const segmi_act* in, const segmi_act* out, int ksize,
                                int stride);
/* Which forward kernel family segmi_conv3d_fwd() runs for this layer shape: a static string such as
 * "conv_ring2_kernel<bf16, CK=16, NT=1>" (reports / benchmarks label their roofline line with it). */
const char* segmi_conv3d_fwd_kernel_name(int dtype, const segmi_act* in, const segmi_act* out,
                                         int ksize, int stride);
int segmi_conv3d_in_affine_ok(int dtype, const segmi_act* in, const segmi_act* out, int ksize,
                              int stride);
int segmi_conv3d_fwd(int dtype, const segmi_act* in, const segmi_act* out, const void* packed,
                     const float* w_src, int w_kind, const float* bias,
                     const float* prelu_alpha, const segmi_act* residual,
                     float* stats_partials, int ksize, int stride, const segmi_in_affine* in_tf,
                     const segmi_bn_bwd_sums* bn_bwd /* nullable */,
                     const segmi_bn_fin* stats_fin /* nullable; needs stats_partials */, void* stream);
/* Inference, full-resolution decoder of monai UNet as ONE launch (csrc/dectop.hip):
 *   h = PReLU(ConvTranspose3d(k3, s2, p1, op1; 32 -> 16)(in) with BatchNorm folded), out = Conv3d(k3; 16 -> 16)(h) + bias + h
 * i.e. `up` layer "model.2.0" followed by the conv-only ResidualUnit "model.2.1" (monai_unet.py:114-124), whose
 * 16-channel intermediate then never reaches HBM.  Bit-identical to segmi_convT3d_fwd + segmi_conv3d_fwd.
 * up_frag: bf16 [27][64][8]: tap (kd*3+kh)*3+kw, lane (g, co), W_T[ci = 8g .. 8g+7][co][tap] * bn_scale[co];
 * up_bias: f32[16] folded bias; up_alpha: PReLU slope; conv_packed: kind-0 segmi_wpack of the 16 -> 16 conv.
 * segmi_dectop_ok(): bf16, in [N,D,H,W,32], out [N,2D,2H,2W,16] with 2D % 4 == 0, 2H % 16 == 0, 2W % 16 == 0. */
int segmi_dectop_ok(int dtype, const segmi_act* in, const segmi_act* out);
int segmi_dectop_fwd(int dtype, const segmi_act* in, const segmi_act* out, const void* up_frag,
                     const float* up_bias, const float* up_alpha,
                     int up_alpha_in_unit_range /* caller asserts 0 <= *up_alpha <= 1: PReLU as max(v, slope v), same bits */,
                     const void* conv_packed, const float* conv_bias, void* stream);
/* The first ResidualUnit of the network convolves its (<= 4 channel) input twice with the same
 * geometry: subunit 0 (k3, stride s) and the residual convolution (k3, stride s).  One launch
 * stages the input once and produces both:  out_a = prelu_a(conv_a(in) + bias_a) with optional
 * statistics of the pre-activation values (rows as segmi_conv3d_stats_rows(in, out_a)),
 * out_b = conv_b(in) + bias_b.  w_a / w_b are torch-layout f32 [cout][cin][27].
 * segmi_conv3d_pair_ok() says whether the layer qualifies (1) or needs two segmi_conv3d_fwd (0).
 * Replaces the two torch.nn.Conv3d of monai ResidualUnit, monai_unet.py:114-124. */
/* Window views (inference): sample n of `in` is not a slice of a dense batch but the (d, h, w) block that starts
 * offset[n] elements into a larger single-channel volume (`in->data` = the volume, row / plane strides in
 * elements), with ZERO padding at the block's own borders -- the windows of MONAI's sliding_window_inference
 * (monai_unet.py:354-356, 637-639, 665) read in place, instead of being gathered into a batch first.  <= 32
 * windows per call; offsets and strides multiples of 4 elements. */
typedef struct segmi_windows {
  int32_t count;
  int32_t row_stride;
  int64_t plane_stride;
  int64_t offset[32];
} segmi_windows;
int segmi_conv3d_pair_ok(int dtype, const segmi_act* in, const segmi_act* out_a,
                         const segmi_act* out_b);
/* The same pairing for MFMA layers (the stride-2 first subunit + residual convolution of the deeper
 * ResidualUnits, inference): ONE convolution whose fragment pack holds both weight sets
 * (segmi_wpack of the concatenated [2c][cin][27] weight, per-channel scale = folded BatchNorm for the
 * first c outputs, 1 for the rest) writes 2c channels, PReLU only on the first `act_channels`:
 *   out[..., :act_channels] = prelu(conv_a(in) + bias[:c]) ; out[..., act_channels:] = conv_b(in) + bias[c:]
 * Consumers read the two halves as channel-slice views (ld = 2c).  Same bits as the two calls.
 * Training (round 4; replaces the two torch convolutions of a ResidualUnit's first subunit and residual path,
 * monai_unet.py:341 through monai.networks.blocks.ResidualUnit): prelu_alpha = NULL, `bias_b` = the second
 * convolution's bias (nullable: then bias holds all 2c values), `stats_partials` / `stats_fin` = BatchNorm statistics
 * rows [segmi_conv3d_stats_rows][2][act_channels] of the FIRST act_channels outputs and their finalisation in the
 * same launch; the pack comes from a segmi_wpack_desc with two sources (w_src2 / cout_split). */
int segmi_conv3d_split_act_ok(int dtype, const segmi_act* in, const segmi_act* out, int ksize,
                              int stride);
int segmi_conv3d_fwd_split_act(int dtype, const segmi_act* in, const segmi_act* out,
                               const void* packed, const float* bias, const float* prelu_alpha,
                               int act_channels, int ksize, int stride,
                               const float* bias_b /* nullable */, float* stats_partials /* nullable */,
                               const segmi_bn_fin* stats_fin /* nullable; needs stats_partials */, void* stream);
int segmi_conv3d_fwd_pair(int dtype, const segmi_act* in, const segmi_act* out_a, const float* w_a,
                          const float* bias_a, const float* prelu_alpha_a, float* stats_partials_a,
                          const segmi_act* out_b, const float* w_b, const float* bias_b, int stride,
                          const segmi_bn_fin* stats_fin_a /* nullable; needs stats_partials_a */,
                          const segmi_windows* windows /* nullable: `in` samples are window views */,
                          void* stream);

/* ConvTranspose3d k3 s2 p1 (output extent 2*in or 2*in-1 per dim, taken from `out`),
 * same fused epilogue.  Also the dgrad of a stride-2 Conv3d.
 * Replaces torch.nn.ConvTranspose3d, src/segmantic/seg/monai_unet.py:114-124. */
int segmi_convT3d_stats_rows(int dtype, const segmi_act* in, const segmi_act* out);
int segmi_convT3d_fwd(int dtype, const segmi_act* in, const segmi_act* out, const void* packed,
                      const float* w_src, const float* bias, const float* prelu_alpha,
                      const segmi_act* residual, float* stats_partials,
                      const segmi_bn_fin* stats_fin /* nullable; needs stats_partials */, void* stream);

/* Weight gradient of a Conv3d (k, stride as forward): dw[co][ci][tap] (torch layout, f32) and
 * db[co] (nullable) from x (forward input) and dy (grad of the conv output).  Two-stage
 * deterministic reduction through `workspace` (>= segmi_conv3d_wgrad_workspace bytes).
 * The ConvTranspose3d weight gradient is the same call with x := dy_T, dy := x_T (stride 2).
 * Replaces autograd's conv backward reached from manual_backward,
 * src/segmantic/seg/monai_unet.py:345. */
int64_t segmi_conv3d_wgrad_workspace(int dtype, const segmi_act* x, const segmi_act* dy,
                                     int ksize, int stride, int cus);
int segmi_conv3d_wgrad(int dtype, const segmi_act* x, const segmi_act* dy, float* dw,
                       float* db, int ksize, int stride, void* workspace,
                       const segmi_in_affine* in_tf /* transform of x, nullable */, int cus, void* stream);
/* `cus`: how many compute units the weight-gradient kernels size their grid (and their partial slabs: pass the same
 * value to the workspace query) for -- a multiple of 8 in [8, 256]; <= 0 = the whole chip.  Their workgroups hold a
 * CU exclusively, so a caller that runs them on a side stream beside its dependent chain -- what autograd's single
 * stream cannot do, monai_unet.py:345 -- sizes them for part of the chip.  A per-call argument since round 4 (no
 * process-wide state); the environment variable SEGMI_WGRAD_CUS overrides it for A/B runs.  segmi_wgrad_cus(cus)
 * returns the count a call with that argument would use. */
int segmi_wgrad_cus(int cus);
/* bias gradient only: db[c] = sum over voxels of dy */
int segmi_bias_grad(int dtype, const segmi_act* dy, float* db, void* workspace, void* stream);

/* ---------------------------------------------------------------- norm + activation ---- */
/* BatchNorm3d (training statistics) + PReLU, MONAI ADN "NDA" ordering.
 * Replaces torch.nn.BatchNorm3d / PReLU under monai ADN, monai_unet.py:114-124. */
int segmi_bn_stats_rows(const segmi_act* x);
int segmi_bn_stats(int dtype, const segmi_act* x, float* stats_partials, void* stream);
/* partials f32[rows][2][c] -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale;
 * running_mean/var (nullable) updated with `momentum` and the unbiased variance. */
int segmi_bn_finalize(const float* stats_partials, int rows, int c, double count,
                      const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, float* mean,
                      float* invstd, float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running statistics */
int segmi_bn_eval_affine(int c, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, float eps,
                         float* scale, float* shift, void* stream);
/* y = prelu(drop(x*scale + shift)) + residual   (alpha nullable -> identity, residual nullable).
 * MONAI ADN "NDA" dropout (the `dropout` key of the config, monai_unet.py:107,119): with
 * dropout_p > 0 an element is kept iff hash(dropout_seed, logical NDHWC element index) >> 8 >=
 * dropout_p * 2^24 and kept values are scaled by 1 / (1 - p); the mask is never stored, the two
 * backward passes recompute it from the same seed.  dropout_p = 0 (eval, default): identity. */
int segmi_bn_act_fwd(int dtype, const segmi_act* x, const segmi_act* y, const float* scale,
                     const float* shift, const float* prelu_alpha, const segmi_act* residual,
                     float dropout_p, uint32_t dropout_seed, void* stream);
/* backward of y = prelu(drop(bn(x))):  pass 1 reduces, finalize, pass 2 writes dx.
 * red_partials f32[rows][3][c]: sum dz, sum dz*xhat, sum dy*z*[z<=0]  (dz = dy*prelu'(z)*mask) */
int segmi_bn_act_bwd_rows(const segmi_act* x);
int segmi_bn_act_bwd_reduce(int dtype, const segmi_act* dy, const segmi_act* x,
                            const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const float* prelu_alpha, float* red_partials,
                            float dropout_p, uint32_t dropout_seed,
                            const segmi_bn_bwd_fin* fin /* nullable: finalise in this launch */, void* stream);
int segmi_bn_act_bwd_finalize(const float* red_partials, int rows, int c, double count,
                              const float* gamma, const float* invstd, float* dgamma,
                              float* dbeta, float* dalpha, float* coef, void* stream);
/* The three calls above as ONE launch for small tensors (<= 32 MB, <= 256 channels in multiples of 4, no
 * dropout): the workgroups reduce, the last one finalises and publishes the coefficients, all apply
 * (a grid-wide hand-off through device-scope atomics; csrc/norm_act.hip).  red_partials: f32
 * [segmi_bn_act_bwd_fused_rows(x)][3][c].  Same results as the three calls up to the f32 summation order of the
 * partial rows.  Replaces autograd's BatchNorm / PReLU backward under monai ADN, monai_unet.py:114-124, 345.
 * Residency: a workgroup of this launch holds a whole CU and waits on it for the last one, so the launch uses at
 * most as many workgroups as the device holds at once (occupancy query x compute units; _ok returns 0 on a device
 * that cannot hold one) and at most `max_wgs` (> 0): a caller that runs CU-exclusive kernels on another stream
 * (segmi_conv3d_wgrad with `cus`) passes the CUs they leave free; <= 0 = the device's capacity.
 * segmi_bn_act_bwd_fused_wgs: the workgroup count a call would launch.  A waiting workgroup gives up after a
 * bounded number of polls (about a second), writes NaN gradients and counts the expiry; segmi_fused_timeouts(reset)
 * returns the count from host-visible memory WITHOUT a device sync (expiries of launches that have executed so far)
 * -- a non-zero value means gradients of this process are poisoned (the reference stops on a non-finite
 * loss, monai_unet.py:512-518 `check_finite`); the Python engine raises.  segmi_fused_test_hook: tests only --
 * a poll bound (0 = default) and `no_publish` (the finalising workgroup withholds the flag: every waiter expires). */
int segmi_bn_act_bwd_fused_ok(int dtype, const segmi_act* dy, const segmi_act* x, const segmi_act* dx);
int segmi_bn_act_bwd_fused_rows(const segmi_act* x);
int segmi_bn_act_bwd_fused_wgs(int dtype, const segmi_act* x, int max_wgs);
int segmi_bn_act_bwd_fused(int dtype, const segmi_act* dy, const segmi_act* x, const segmi_act* dx,
                           const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* prelu_alpha, float* red_partials, const segmi_bn_bwd_fin* fin,
                           int max_wgs, void* stream);
unsigned segmi_fused_timeouts(int reset);
int segmi_fused_test_hook(unsigned poll_limit, int no_publish);
int segmi_bn_act_bwd_apply(int dtype, const segmi_act* dy, const segmi_act* x,
                           const segmi_act* dx, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, const float* prelu_alpha,
                           const float* coef, float dropout_p, uint32_t dropout_seed, void* stream);
/* segmi_bn_act_bwd_apply fused into the k3 stride-2 convolution that consumes its result -- the input gradient
 * of a decoder level's ConvTranspose3d (monai UNet up layer, monai_unet.py:114-124; backward of :345): the
 * convolution computes dx = apply(dy, x) while it stages its halo tile, feeds the bf16-rounded values to its
 * MFMAs and writes dx once (for the weight gradient): same bits as the two calls, one pass over dx less.
 * bf16, 16 channels, out->c in {16, 32, 64}, out->w >= 16, no dropout; `packed`: segmi_wpack of the
 * convolution (kind 0, as segmi_conv3d_fwd takes it).  dx must not alias dy or x. */
int segmi_bn_act_bwd_apply_conv_ok(int dtype, const segmi_act* dy, const segmi_act* x, const segmi_act* dx,
                                   const segmi_act* out);
int segmi_bn_act_bwd_apply_conv(int dtype, const segmi_act* dy, const segmi_act* x, const segmi_act* dx,
                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const float* prelu_alpha, const float* coef, const segmi_act* out,
                                const void* packed, void* stream);

/* elementwise helpers on NDHWC views */
int segmi_add(int dtype, const segmi_act* a, const segmi_act* b, const segmi_act* out,
              void* stream);                       /* out = a + b (b nullable -> copy)        */
int segmi_cast_copy(int src_dtype, const segmi_act* src, int dst_dtype, const segmi_act* dst,
                    void* stream);                 /* dtype / ld converting copy              */
int segmi_nchw_to_ndhwc(const float* src, int dst_dtype, const segmi_act* dst, void* stream);
int segmi_ndhwc_to_nchw(int src_dtype, const segmi_act* src, float* dst, void* stream);

/* ---------------------------------------------------------------- loss + optimiser ----- */
/* MONAI DiceLoss(to_onehot_y=True, softmax=True), monai_unet.py:128,344.
 * labels: f32[n*d*h*w] integer-valued class ids.  partials f32[segmi_dice_chunks()][n][3][k]
 * (the chunk count includes a scratch tail); coef f32[n][2][k] receives the backward
 * coefficients; loss f32[1]. */
int segmi_dice_chunks(const segmi_act* logits);
int segmi_softmax_dice_fwd(int dtype, const segmi_act* logits, const float* labels,
                           float* partials, float* coef, float* loss, float smooth_nr,
                           float smooth_dr, void* stream);
/* backward: dlogits = grad_scale * dLoss/dlogits.  bias_grad (nullable, f32[k]): also the channel
 * sums of dlogits -- the bias gradient of the layer that produced the logits -- folded into the same
 * pass; it needs `scratch` = the forward's `partials` buffer (free again after the forward). */
int segmi_softmax_dice_bwd(int dtype, const segmi_act* logits, const float* labels,
                           const float* coef, float grad_scale, const segmi_act* dlogits,
                           float* scratch, float* bias_grad, void* stream);

/* torch.optim.Adam / SGD semantics over one flat f32 arena, monai_unet.py:292-304,346, and
 * adabelief_pytorch.AdaBelief(rectify=False, fixed_decay=False), monai_unet.py:305-314.
 * Hyper-parameters are doubles, as the Python optimisers hold them: derived scalars (1 - beta,
 * lr / bias_correction, ...) are formed in double and rounded to f32 once, like torch's scalar
 * arguments.  grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). */
int segmi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    float* max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                    double eps, double weight_decay, int64_t step, float grad_scale,
                    void* stream);
int segmi_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, double lr,
                   double momentum, double weight_decay, int first_step, float grad_scale,
                   void* stream);
int segmi_adabelief_step(float* param, const float* grad, float* exp_avg, float* exp_avg_var,
                         int64_t n, double lr, double beta1, double beta2, double eps,
                         double weight_decay, int weight_decouple, int64_t step,
                         float grad_scale, void* stream);

/* ---------------------------------------------------------------- sliding window ------- */
/* MONAI sliding_window_inference, monai_unet.py:354-356,637-639,665.
 * starts_host: int32[nwin][3] (z,y,x) window origins in the (padded) image. */
int segmi_sw_gather(int dtype_src, const segmi_act* image, int img_index,
                    const int32_t* starts_host, int nwin, int dst_dtype,
                    const segmi_act* windows, void* stream);
/* acc[z,y,x,k] += w * pred ; cnt[z,y,x] += w  for each window in order (deterministic).
 * importance nullable (constant 1) else f32[roi_d*roi_h*roi_w]. cnt nullable. */
int segmi_sw_scatter_add(int dtype, const segmi_act* pred, const int32_t* starts_host,
                         int nwin, const float* importance, const segmi_act* acc, float* cnt,
                         void* stream);
/* logits = acc / cnt (in place, nullable skip) and labels = argmax_k (first max wins).
 * label_bytes in {1,2,4}. */
int segmi_sw_finalize(const segmi_act* acc, const float* cnt, int write_logits, void* labels,
                      int label_bytes, void* stream);
/* Deferred form of the same blend (same reference call sites): the caller keeps EVERY window
 * prediction of the dense schedule, `cache` = [win_hi - win_lo][rd][rh][rw][ldp >= k] of `dtype`
 * (slot = window index - win_lo; window index = (iz*ny + iy)*nx + ix over the per-dimension origin
 * lists, first dimension slowest -- MONAI's dense_patch_slices order).  One pass sums, per output
 * voxel, the covering windows in ascending window index (the f32 addition order of the
 * reference's sequential `out[slice] += w * pred`), then
 *   normalize != 0: out_logits = sum / count (nullable), labels = argmax_k (nullable)
 *   normalize == 0: out_logits = sum, out_count = count (partial result of a window shard).
 * Origins are host arrays in un-padded image coordinates (may be negative); at most 64 per
 * dimension (SEGMI_EUNSUPPORTED beyond: use the streaming segmi_sw_scatter_add).
 * out_logits: f32 [d][h][w][ldo >= k]; out_count: f32 [d][h][w]; label_bytes in {1,2,4}. */
int segmi_sw_blend(int dtype, const void* cache, int k, int ldp, const int32_t* starts_z, int nz,
                   const int32_t* starts_y, int ny, const int32_t* starts_x, int nx, int win_lo,
                   int win_hi, int rd, int rh, int rw, const float* importance, int d, int h, int w,
                   float* out_logits, int ldo, float* out_count, void* labels, int label_bytes,
                   int normalize, void* stream);
/* AsDiscrete(argmax=True), monai_unet.py:129-134,622,673 */
int segmi_argmax(int dtype, const segmi_act* logits, void* labels, int label_bytes,
                 void* stream);
/* per-class overlap counts for DiceMetric (monai_unet.py:136-138): counts i64[k][3] =
 * |pred==c & true==c|, |pred==c|, |true==c| ; labels are int32 */
int segmi_label_counts(const int32_t* pred, const int32_t* truth, int64_t n, int k,
                       int64_t* counts, void* stream);

/* ---------------------------------------------------------------- image ops ------------ */
/* ITK ResampleImageFilter replacement, src/segmantic/image/processing.py:49-120.
 * index_map_host: 12 doubles, row-major 3x4 affine taking an output index (x,y,z,1) to the
 * continuous input index (x,y,z).  pixel: 0=f32 1=u8 2=i16 3=i32 4=u16.  Arrays are [z][y][x].
 * interp: 0 = linear, 1 = nearest (outside the input buffer -> default_value, as ITK); +2 = border
 * padding: the continuous index is clamped to the buffer first (MONAI Spacingd's padding_mode=
 * "border", monai_unet.py:173-174 and its inverse under Invertd, :615-621); +4 (with 1 only) =
 * nearest rounds x.5 to the even index (torch grid_sample / MONAI mode="nearest") instead of up (ITK). */
int segmi_resample3d(int pixel, const void* src, int sx, int sy, int sz, void* dst, int dx,
                     int dy, int dz, const double* index_map_host, int interp,
                     double default_value, void* stream);
/* NormalizeIntensityd(channel_wise=True), monai_unet.py:164: in-place (x-mean)/std per channel
 * of a [c][nvox] f32 array.  workspace >= segmi_normalize_workspace(c, nvox) bytes. */
int64_t segmi_normalize_workspace(int c, int64_t nvox);
int segmi_normalize_intensity(float* x, int c, int64_t nvox, void* workspace, void* stream);
/* on-device patch sampler: copies `count` roi-sized crops (origins in starts_host, int32
 * [count][4] = n,z,y,x) of image (f32 -> dst dtype) and label (f32) volumes; per-axis flips
 * from flips_host (uint8[count], bit0=z bit1=y bit2=x).  monai_unet.py:193-217. */
int segmi_crop_patches(const segmi_act* image, const float* label, const int32_t* starts_host,
                       const uint8_t* flips_host, int count, int dst_dtype,
                       const segmi_act* out_image, float* out_label, void* stream);
/* The same sampler with the spatial augmentation of monai_unet.py:181-191 (RandRotated about the
 * three axes, RandZoomd keep_size) composed into the gather: a patch voxel goes (flip, crop origin)
 * -> index in the augmented volume -> index_map_host (12 doubles, row-major 3x4, (x,y,z,1) ->
 * continuous source index) -> image trilinear with border clamping, label nearest; positions
 * outside the augmented volume's extent (the SpatialPadd region) are 0. */
int segmi_warp_crop_patches(const segmi_act* image, const float* label, const int32_t* starts_host,
                            const uint8_t* flips_host, int count, const double* index_map_host,
                            int dst_dtype, const segmi_act* out_image, float* out_label,
                            void* stream);
/* Intensity augmentation of monai_unet.py:205-208 on `count` dense f32 NDHWC patches
 * [count][rd][rh][rw][c], in place, in the reference's order: RandAdjustContrastd (gamma),
 * RandHistogramShiftd (nctrl floating control points in [0,1] per patch, ascending),
 * RandBiasFieldd (20 degree-3 Legendre coefficients per patch).  Each *_on_host (uint8[count],
 * nullable = skip the transform) selects the patches a transform applies to; the random draws
 * are the caller's.  workspace >= segmi_intensity_workspace(count) bytes. */
int64_t segmi_intensity_workspace(int count);
int segmi_intensity_augment(float* patches, int count, int rd, int rh, int rw, int c,
                            const uint8_t* contrast_on_host, const float* gamma_host,
                            const uint8_t* hist_on_host, const float* ctrl_host, int nctrl,
                            const uint8_t* bias_on_host, const float* coef_host, void* workspace,
                            void* stream);
/* k-space augmentation of monai_unet.py:209-210 on the same patch layout, channel-wise, in place:
 * RandGibbsNoised (spectrum outside radius (1-alpha)*max(shape)*sqrt(2)/2 of the centred k-space
 * zeroed) then RandKSpaceSpikeNoised (one bin at spike_loc_host int32[count][3] = (z,y,x) of the
 * centred k-space set to magnitude exp(2.5 * mean log|K| * (0.95 + 0.15 * spike_u)), phase kept).
 * 3-D DFT of any extents <= 512 (direct per-axis transform in LDS; only selected patches are
 * transformed).  workspace >= segmi_kspace_workspace(count, rd, rh, rw) bytes. */
int64_t segmi_kspace_workspace(int count, int rd, int rh, int rw);
int segmi_kspace_augment(float* patches, int count, int rd, int rh, int rw, int c,
                         const uint8_t* gibbs_on_host, const float* gibbs_alpha_host,
                         const uint8_t* spike_on_host, const int32_t* spike_loc_host,
                         const float* spike_u_host, void* workspace, void* stream);

/* ---------------------------------------------------------------- model ensembles ------ */
/* Combination of `models` (<= 16) predictions over n elements, monai_unet.py:848-1004.
 * *_host arguments are HOST arrays (of device pointers / scalars).
 *  mean  : MeanEnsembled with weights: out = mean_e(logits_e * w_e / mean(w)) (f32, n = K * voxels)
 *  vote  : VoteEnsembled on int32 label volumes: most frequent label, ties -> smallest label
 *  select: segmantic SelectBestEnsemble (seg/transforms.py:15-88): for each (tissue, model) pair in
 *          order, out[labels[model] == tissue] = tissue; unclaimed voxels are 0. */
int segmi_ensemble_mean(const float* const* logits_host, const float* weights_host, int models,
                        int64_t n, float* out, void* stream);
int segmi_ensemble_vote(const int32_t* const* labels_host, int models, int64_t n, int32_t* out,
                        void* stream);
int segmi_ensemble_select(const int32_t* const* labels_host, int models, const int32_t* tissue_host,
                          const int32_t* model_host, int pairs, int64_t n, int32_t* out,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEGMI_H_ */
