/*
 * afan_hip.h — C-ABI of libafan_hip.so: the MI355X (gfx950) kernels of the A-FAN hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes + an explicit HIP stream,
 * allocates nothing, never synchronises, is re-entrant and hipGraph-capturable.  Return value:
 * 0 on success, a positive hipError_t if the launch failed, or a negative AFAN_E* argument error.
 *
 * The reference (VITA-Group/CV_A-FAN) is Python on PyTorch; the interfaces these symbols replace are
 * cited per function as `path:line` relative to the reference root.  The ctypes binding a reference
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Layout contract: all activation tensors are dense, contiguous NCHW (the reference's layout,
 * Classification/resnet_s.py:119-121); `hw` = H*W, one (n,c) plane is `hw` consecutive elements.
 */
#ifndef AFAN_HIP_H
#define AFAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* stream handle: a hipStream_t passed as an opaque pointer (0 = the null stream) */
typedef void* afan_stream_t;

/* element types of activation / gradient buffers */
enum { AFAN_F32 = 0, AFAN_BF16 = 1 };

/* memory layout of an activation tensor of logical shape [N, C, H, W]:
 * AFAN_NCHW = dense planes (the reference's layout), AFAN_NHWC = channels-last, one dense [N*H*W][C] matrix
 * (what the bf16 MFMA backbone uses internally; torch calls it memory_format=channels_last). */
enum { AFAN_NCHW = 0, AFAN_NHWC = 1 };

/* argument errors (negative so they never collide with hipError_t) */
enum {
    AFAN_OK = 0,
    AFAN_EDTYPE = -1,  /* unknown dtype code */
    AFAN_EALIGN = -2,  /* pointer not aligned to its element type */
    AFAN_ESHAPE = -3,  /* non-positive / inconsistent sizes */
    AFAN_ENULL = -4,   /* required pointer is NULL */
    AFAN_ELAYOUT = -5  /* unknown layout code */
};

/* Library identification: version = 10000*major + 100*minor + patch; arch string is "gfx950". */
int afan_version(void);
const char* afan_arch(void);

/* ------------------------------------------------------------------------------------------------
 * PGD ascent step.  Replaces Classification/attack_algo.py:53 (+ :55-56 → :35-36 → :9-19 when clip).
 *   x_adv[i] <- fl32(x_adv[i] + fl32(gamma) * sign(grad[i]))            sign(0)=0, sign(NaN)=NaN
 *   if clip:  lo = fl32(x_clean[i]-eps), hi = fl32(x_clean[i]+eps); t<lo -> lo; then t>hi -> hi
 * x_adv: fp32, updated in place.  grad: `grad_dtype` (fp32 or bf16 — only its sign is used).
 * x_clean: fp32, required iff clip != 0.  shadow_bf16: optional (may be NULL) bf16 copy of the new
 * x_adv written in the same pass (the tail's conv input when the backbone runs in bf16).
 * Algorithmic HBM bytes: 12 B/elt (fp32 grad), 16 B/elt with clip; +2 with the bf16 shadow.
 */
int afan_pgd_step(float* x_adv, const void* grad, int grad_dtype, const float* x_clean,
                  uint16_t* shadow_bf16, int64_t n, float gamma, float eps, int clip,
                  afan_stream_t stream);

/* tensor_clamp with arbitrary bound tensors.  Replaces Classification/attack_algo.py:9-19 (in place:
 * t<lo -> lo, then t>hi -> hi; NaN passes through).  linfball_proj (:35-36) is afan_pgd_step with gamma=0. */
int afan_tensor_clamp(float* t, const float* lo, const float* hi, int64_t n, afan_stream_t stream);

/* Last PGD step fused with the perturbation norms.  Replaces attack_algo.py:53-56 plus
 * Classification/main_perturb.py:188-192 (per-sample ||x_adv-x||_2 and ||.||_inf over C*H*W, which
 * the reference computes on the host after a full-tensor D2H copy).
 * batch*per_sample elements; x_clean is always required (it is the norm's centre).
 * partial: workspace of afan_norms_workspace_floats(batch, per_sample) floats.
 * l2_out/linf_out: [batch] fp32.  Deterministic (no float atomics): two launches on `stream`.
 */
int64_t afan_norms_workspace_floats(int64_t batch, int64_t per_sample);
int afan_pgd_step_norms(float* x_adv, const void* grad, int grad_dtype, const float* x_clean,
                        uint16_t* shadow_bf16, int64_t batch, int64_t per_sample, float gamma,
                        float eps, int clip, float* partial, float* l2_out, float* linf_out,
                        afan_stream_t stream);

/* Stand-alone perturbation norms (main_perturb.py:188-192) for callers that run 0 PGD steps. */
int afan_perturb_norms(const float* x_adv, const float* x_clean, int64_t batch, int64_t per_sample,
                       float* partial, float* l2_out, float* linf_out, afan_stream_t stream);

/* Random start.  Replaces attack_algo.py:42-44:  x_adv += (2.0*u - 1.0) * eps  with u ~ U[0,1) drawn by
 * the caller on the HOST generator (the reference draws it on the CPU, so parity needs the host's
 * stream of numbers) and uploaded to `u`.  Rounding: fl(fl(fl(2u)-1)*eps) then one add, no FMA.
 */
int afan_axpy_noise(float* x_adv, const float* u, int64_t n, float eps, uint16_t* shadow_bf16,
                    afan_stream_t stream);

/* mix_feature.  Replaces Segmentation/attack_algo.py:121-130 == Detection/attack_algo.py:254-265:
 * per (n,h,w): mean and unbiased variance over the C channels of clean and adv;
 *   out = (clean - mean_c) / sqrt(var_c + 1e-5) * sqrt(var_a + 1e-5) + mean_a
 * clean/adv/out: [N, C, HW] of `dtype`.  eps is the 1e-5 above.  Algorithmic bytes 12 B/elt (fp32).
 */
int afan_mix_feature(const void* clean, const void* adv, void* out, int64_t n, int64_t c, int64_t hw,
                     float eps, int dtype, afan_stream_t stream);
/* The same transform for channels-last tensors ([N,H,W,C] memory: a pixel's channels contiguous — the layout of the bf16
 * backbone): one wave per pixel, moments merged across lanes. */
int afan_mix_feature_nhwc(const void* clean, const void* adv, void* out, int64_t n, int64_t c, int64_t hw, float eps,
                          int dtype, afan_stream_t stream);

/* SAT sample points.  Replaces Segmentation/attack_algo.py:108-118 (torch.lerp semantics:
 * w<0.5: x + w*(y-x); else y - (y-x)*(1-w)).  Writes the n_points-2 interior points, point k (1-based)
 * at out + (k-1)*n elements, weight k*(1/(n_points-1)) computed as the reference does in double and
 * rounded to fp32 by the caller into `weights`[n_points-2] (host array, read at launch time).
 */
int afan_lerp_points(const float* x, const float* y, float* out, int64_t n, const float* weights,
                     int n_interior, afan_stream_t stream);

/* Fused sample points + mix_feature (SURVEY 8(a) a11 "fuse with a10"; Segmentation/main_aug_final.py:186-192 runs
 * get_sample_points(clean, adv, 3) and then mix_feature(clean, point) on the points its --mix_layer flags name).
 * Points j = 1 .. n_points-1 (interior points lerp(clean, adv, weights[j-1]); the last one is adv) go to
 * out + (j-1)*n*c*hw; bit j-1 of mix_mask: the stored point is mix_feature(clean, point_j, eps).  An end point without its
 * bit is not written.  fp32, clean/adv/out in `layout` (AFAN_NCHW / AFAN_NHWC), n_points <= 5.  Results are bit-identical
 * to afan_lerp_points followed by afan_mix_feature[_nhwc].  Algorithmic bytes: 4*(2 + n_points-1) per element. */
int afan_lerp_mix(const float* clean, const float* adv, float* out, int64_t n, int64_t c, int64_t hw, const float* weights,
                  int n_points, unsigned mix_mask, float eps, int layout, afan_stream_t stream);

/* Learnable feature mixing of the multi-layer A-FAN (Classification/main_learnable.py:226):
 *   out[i] = clean[i] + w * (adv[i] - clean[i])      three fp32 roundings, like the eager expression
 * clean/adv fp32 [n] (same memory layout), w = one fp32 on the DEVICE (an entry of the model's `w` parameter), out in
 * `out_dtype` (the backbone's compute dtype).  Backward: d(loss)/dw = sum_i grad_out[i] * (adv[i] - clean[i]) written
 * (accumulate = 0) or added (1) to *dw; clean is detached and adv's gradient is never used (:205-215), so nothing else
 * is produced.  Deterministic: block partials in workspace[afan_mix_w_workspace_floats()], folded by one wave. */
int64_t afan_mix_w_workspace_floats(void);
int afan_mix_w(const float* clean, const float* adv, const float* w, void* out, int out_dtype, int64_t n,
               afan_stream_t stream);
int afan_mix_w_backward(const void* grad_out, int grad_dtype, const float* clean, const float* adv, int64_t n,
                        float* workspace, float* dw, int accumulate, afan_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm, training mode (per-channel moments over N*H*W) — the "feature-norm stats" the backbone runs
 * K+2 times per iteration in the tail and twice in the head (main_perturb.py:173,195-196 through
 * torch.nn.BatchNorm2d in resnet_s.py:52-55,88).  Both layouts; fp32 or bf16 activations; fp32 statistics.
 *
 * A per-layer statistics block `stats` is 4*C floats: [mean | invstd | alpha | beta] with
 *   invstd = 1/sqrt(var_biased + eps), alpha = invstd*weight, beta = bias - mean*alpha   (y = x*alpha + beta).
 * (The NCHW kernels fill and read only mean/invstd; the NHWC kernels use all four, 16-byte aligned.)
 * workspace: afan_bn_workspace_floats(c) floats, private to the stream.  No float atomics: bitwise reproducible.
 *
 * afan_bn_stats: statistics only (+ running-stat update  r <- (1-momentum)*r + momentum*batch  with the UNBIASED
 *   variance, and ++(*num_batches), when running_mean != NULL).
 */
int64_t afan_bn_workspace_floats(int64_t c);
int afan_bn_stats(const void* x, int dtype, int layout, int64_t n, int64_t c, int64_t hw, float eps,
                  float momentum, float* workspace, float* stats, float* running_mean, float* running_var,
                  int64_t* num_batches, afan_stream_t stream);

/* Fused training forward:  y = [relu]( (x-mean)*invstd*weight + bias [+ residual] ), statistics saved in
 * save_stats for the backward, running stats updated.  x,y,residual: `dtype`, `layout`, [N,C,HW];
 * weight/bias nullable (= 1/0). */
int afan_bn_train_forward(const void* x, const void* residual, void* y, int dtype, int layout, int64_t n,
                          int64_t c, int64_t hw, float eps, float momentum, const float* weight,
                          const float* bias, int relu, float* workspace, float* save_stats,
                          float* running_mean, float* running_var, int64_t* num_batches,
                          afan_stream_t stream);

/* afan_bn_train_forward for a channels-last x whose per-tile moments were already written by the producing
 * convolution (see afan_conv_fwd_nhwc_bf16): fold of the partials_g partials per channel + normalise; no stats pass. */
int afan_bn_train_forward_partials(const void* x, const void* residual, void* y, int dtype, int64_t n, int64_t c,
                                   int64_t hw, float eps, float momentum, const float* weight, const float* bias,
                                   int relu, const float* partials, int64_t partials_g, const float* partials_shift,
                                   float* save_stats, float* running_mean, float* running_var,
                                   int64_t* num_batches, afan_stream_t stream);

/* Same transform with GIVEN statistics (eval mode: mean = running_mean, invstd = rsqrt(running_var+eps)).
 * workspace is required for AFAN_NHWC (may be NULL for AFAN_NCHW). */
int afan_bn_apply(const void* x, const void* residual, void* y, int dtype, int layout, int64_t n, int64_t c,
                  int64_t hw, const float* mean, const float* invstd, const float* weight, const float* bias,
                  int relu, float* workspace, afan_stream_t stream);

/* Backward of the fused training forward.
 *   g   = dy * (act > 0) when relu, else dy;  act = y if y != NULL, else recomputed from x (no residual)
 *   sum_g[c] = sum g ;  sum_gx[c] = sum g * xhat ;  M = N*HW
 *   dx  = weight*invstd * ( g - sum_g/M - xhat*sum_gx/M )
 *   d_residual = g (written iff d_residual != NULL; may alias dy)
 * dweight[c] = sum_gx[c], dbias[c] = sum_g[c] (added into them iff accumulate != 0; NULL skips them —
 * the PGD inner loop only needs dx: attack_algo.py:52 `only_inputs=True`).
 * weight/bias: the forward's (read by the NCHW kernels; the NHWC kernels take alpha/beta from save_stats). */
int afan_bn_backward(const void* dy, const void* x, const void* y, void* dx, void* d_residual, int dtype,
                     int layout, int64_t n, int64_t c, int64_t hw, const float* save_stats,
                     const float* weight, const float* bias, int relu, float* workspace, float* dweight,
                     float* dbias, int accumulate, const float* partials, int64_t partials_g,
                     afan_stream_t stream);
/* partials != NULL (AFAN_NHWC only): the reduction sums were already taken by the producing dgrad's epilogue
 * ([2][C][partials_g], see afan_conv_dgrad_nhwc_bf16): only finalize + apply run. */

/* Accumulator variants (channels-last only): the two per-channel sums of a BatchNorm pass live in a caller-provided
 * block acc = double[afan_bn_acc_doubles(c)] (16-byte aligned), ZEROED by the caller before its producer runs:
 *   NS copies of [2][c]      f64 sums, added with native f64 atomics (order-independent to ~1e-16 relative); NS =
 *                            min(16, 1024/c) copies spread same-address atomics, the consumer sums the copies
 *   then c floats            the shift the forward moments were taken around (written by the producer)
 * acc_ready != 0: a convolution epilogue already filled acc (afan_conv_fwd_nhwc_bf16 stats_acc / afan_conv_dgrad_nhwc_bf16
 * bn_acc) and ONE launch remains (the normalise / dx pass derives its coefficients from acc in its prologue; its first
 * block publishes save_stats, running statistics, dweight/dbias).  acc_ready == 0: the sums are taken here first (two
 * launches).  No partial slabs, no finalize launch.  afan_bn_acc_supported(): c / (8 bf16 | 4 fp32) must divide 256. */
int64_t afan_bn_acc_doubles(int64_t c);
int afan_bn_acc_supported(int dtype, int64_t c);

/* main_perturb.py:173 and :196 run the head of the network twice per iteration on the same images with the same weights
 * (once detached for PGD, once inside the clean forward): same values, and each train-mode BatchNorm updates its running
 * statistics twice from the same batch moments.  afan_bn_set_running_updates(n) makes the channels-last
 * afan_bn_train_forward* launches issued by THIS host thread from now on stand for n such identical passes: the
 * running-statistics update is applied n times in sequence (bit-identical to n passes) and num_batches_tracked advances by
 * n.  Returns the previous value; 1 restores the default.  Ungrouped launches only; the NCHW kernels refuse n != 1. */
int afan_bn_set_running_updates(int n);

/* One more running-statistics update from the saved statistics (mean | invstd rows of save_stats) of an earlier train-mode
 * forward over m_count values per channel — the update the reference's value-identical second clean tail pass
 * (attack_algo.py:50 at step 0, then main_perturb.py:196) would apply last.  running_var uses 1/invstd^2 - eps. */
int afan_bn_running_update(const float* stats, int64_t c, double m_count, float eps, float momentum, float* running_mean,
                           float* running_var, int64_t* num_batches, afan_stream_t stream);
/* The same for n layers (host arrays of n device pointers / sizes) in ceil(n / 64) launches. */
int afan_bn_running_update_batched(const float* const* stats, float* const* running_mean, float* const* running_var,
                                   int64_t* const* num_batches, const int64_t* c, const double* m_count, const float* eps,
                                   const float* momentum, int n, afan_stream_t stream);
int afan_bn_train_forward_acc(const void* x, const void* residual, void* y, int dtype, int64_t n, int64_t c,
                              int64_t hw, float eps, float momentum, const float* weight, const float* bias,
                              int relu, double* acc, int acc_ready, float* save_stats, float* running_mean,
                              float* running_var, int64_t* num_batches, int groups, afan_stream_t stream);

/* The end of a residual block with a projection shortcut (Classification/resnet_s.py:72-77, option B), one launch:
 *   y = relu(bn_a(x_a) + bn_b(x_b)),  x_a = the block's last convolution output, x_b = the 1x1 projection's output,
 * both train-mode BatchNorms with their moments already in accumulator blocks acc_a / acc_b (filled by the producing
 * convolutions' epilogues, shift snapshot behind them).  Writes both save_stats blocks, applies both running-statistics
 * updates (afan_bn_set_running_updates times) — the same values as afan_bn_train_forward_acc(x_b) followed by
 * afan_bn_train_forward_acc(x_a, residual = its result, relu), without the bf16 rounding of the intermediate tensor. */
int afan_bn_train_forward_acc_dual(const void* x_a, const void* x_b, void* y, int dtype, int64_t n, int64_t c, int64_t hw,
                                   float eps_a, float momentum_a, const float* weight_a, const float* bias_a, double* acc_a,
                                   float* save_stats_a, float* running_mean_a, float* running_var_a, int64_t* num_batches_a,
                                   float eps_b, float momentum_b, const float* weight_b, const float* bias_b, double* acc_b,
                                   float* save_stats_b, float* running_mean_b, float* running_var_b, int64_t* num_batches_b,
                                   afan_stream_t stream);
int afan_bn_backward_acc(const void* dy, const void* x, const void* y, void* dx, void* d_residual, int dtype,
                         int64_t n, int64_t c, int64_t hw, const float* save_stats, int relu, double* acc,
                         int acc_ready, float* dweight, float* dbias, int accumulate, int groups,
                         afan_stream_t stream);
/* groups = 2 (acc_ready only): x is two concatenated half-batches (n/2 images each) normalised separately in ONE launch:
 * acc holds one accumulator block per half (stride afan_bn_acc_doubles(c) rounded up to even), save_stats is [2][4*c],
 * the running statistics receive the two updates in order (first half, then second), dweight/dbias the sum. */

/* ------------------------------------------------------------------------------------------------
 * Backbone convolutions (bf16, channels-last, fp32 accumulate on MFMA) — what torch.nn.Conv2d runs inside
 * `model(x_adv, end_point, start_point)` (attack_algo.py:50; resnet_s.py:52-54,66,72-73) and its input-gradient
 * (attack_algo.py:52), implicit-GEMM kernels of this library.  k in {1,3}, padding dilation*(k/2), stride in {1,2},
 * dilation >= 1 (> 1: the atrous 3x3 stride-1 convolutions of Segmentation/network/backbone/resnet.py:29-32 and
 * _deeplab.py:146-153).  Channels: Ci and Co multiples of 8, both >= 40 (the tiled kernel; counts that are not multiples of
 * its 64-channel tiles — DeepLab's 48 and 304, _deeplab.py:33,41 — are masked in the last tile), or — the reference's own
 * 16-32-64-channel CIFAR ResNets — Ci, Co in {16, 32, 64} (afan_conv_supported() tells).  Layers with a 16/32-channel side take the small-channel kernel, whose statistics
 * fusions exist in the accumulator form (stats_acc / bn_acc) only.  The image stem (resnet_s.py:88: Ci == 3, k == 3,
 * stride 1, Co in {16, 32, 64}, image width % 32 == 0) has its own forward kernel (moments: accumulator form only, no
 * image groups) and weight-gradient kernel; it has no input gradient (the images carry none).
 *   fwd  : y[N,Ho,Wo,Co]  = conv(x[N,Hi,Wi,Ci], w[Co,k,k,Ci])
 *   dgrad: dx[N,Hi,Wi,Ci] = conv_transpose(dy[N,Ho,Wo,Co], w)  given  wt[Ci,k,k,Co] = w transposed
 */
int afan_conv_supported(int64_t ci, int64_t co, int k, int stride);
int afan_conv_fwd_nhwc_bf16(const void* x, const void* w, void* y, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                            int64_t co, int k, int stride, int dilation, float* stats_partials, const float* stats_shift,
                            double* stats_acc, int groups, afan_stream_t stream);

/* afan_conv_fwd_affine_nhwc_bf16 — a convolution of Detection's frozen-BatchNorm backbone (backbone/resnet101_ori.py:97-119 under
 * model.py:27-35,46-47: every BatchNorm in eval mode) with that BatchNorm (+ the block's residual) (+ ReLU) applied in the epilogue:
 * y = [relu](bf16(conv(x, w)) * alpha + beta [+ residual]); coefs = an afan_affine_coefs block [4][co].  Bit for bit
 * afan_conv_fwd_nhwc_bf16 followed by afan_affine_apply.  AFAN_ESHAPE for shapes the stem / small-channel / 64 -> 64 kernels take. */
int afan_conv_fwd_affine_nhwc_bf16(const void* x, const void* w, void* y, int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co,
                                   int k, int stride, const float* coefs, const void* residual, int relu, afan_stream_t stream);
/* nb (1..4) forward convolutions on the SAME input x with the SAME output shape in ONE launch: w[b] / y[b] / ksize[b] (1 or 3)
 * / dilation[b] per problem (host arrays; device pointers inside), BatchNorm moments of y[b] around stats_shift[b] into the
 * f64 accumulator block stats_acc[b] (both arrays NULL: no moments).  (1) The atrous branches of ASPP,
 * Segmentation/network/_deeplab.py:143-150,173-176 (three 3x3 2048 -> 256 convolutions of one feature map at dilations
 * 6/12/18 or 12/24/36): at 2 images per GPU one branch fills a quarter of the chip, the three together cost the time of
 * one.  (2) A BasicBlock's first 3x3 / stride-2 convolution and its 1x1 / stride-2 projection shortcut
 * (Classification/resnet_s.py:52-77, option B). */
int afan_conv_fwd_multi_nhwc_bf16(const void* x, const void* const* w, void* const* y, int nb, int64_t n, int64_t hi,
                                  int64_t wi, int64_t ci, int64_t co, const int* ksize, int stride, const int* dilation,
                                  const float* const* stats_shift, double* const* stats_acc, int groups, afan_stream_t stream);
/* Fusion of the following train-mode BatchNorm's moments into the convolution epilogue: when stats_partials != NULL
 * the forward also writes, per row tile g and output channel c, sum(y - shift[c]) at [(0*Co + c)*G + g] and
 * sum((y - shift[c])^2) at [(1*Co + c)*G + g], G = afan_conv_fwd_tiles(...), over the bf16 values it stores
 * (shift = the BN layer's running mean, NULL = 0).  afan_bn_train_forward_partials() consumes them.
 * stats_acc != NULL (instead of stats_partials): the same sums are added into the zeroed accumulator block
 * stats_acc[afan_bn_acc_doubles(Co)] and the shift is copied behind them; afan_bn_train_forward_acc() consumes it.
 * groups = 2 (with stats_acc): the batch is two concatenated half-batches normalised SEPARATELY (the adversarial and
 * the clean pass of main_perturb.py:195-196 run as one launch): images n >= N/2 add into a second accumulator block at
 * stats_acc + afan_bn_acc_doubles(Co) rounded up to even.  (N/2)*Ho*Wo must be a multiple of 128.  groups = 1 otherwise. */
int64_t afan_conv_fwd_tiles(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride);
int afan_conv_dgrad_nhwc_bf16(const void* dy, const void* wt, void* dx, int64_t n, int64_t hi, int64_t wi,
                              int64_t ci, int64_t co, int k, int stride, int dilation, const void* addend, const void* bn_x,
                              const float* bn_stats, int bn_relu, const void* bn_y, float* bn_partials,
                              double* bn_acc, int groups, afan_stream_t stream);

/* afan_conv_dgrad_affine_nhwc_bf16 — the input gradient of a convolution whose input came out of a frozen BatchNorm + ReLU
 * (backbone/resnet101_ori.py:97-119 under model.py:27-35,46-47), with that layer's backward applied in the epilogue:
 * dx = bf16((act > 0 ? bf16(dgrad(dy)) : 0) * alpha[c]); alpha = the layer's alpha row [ci], act = its stored output.  Bit for bit
 * afan_conv_dgrad_nhwc_bf16 followed by afan_affine_relu_bwd(relu = 1).  AFAN_ESHAPE where another kernel owns the shape. */
int afan_conv_dgrad_affine_nhwc_bf16(const void* dy, const void* wt, void* dx, int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co,
                                     int k, int stride, const float* alpha, const void* act, afan_stream_t stream);
/* The gradient arriving at the OUTPUT of a frozen-BatchNorm residual block with that block's first backward step applied on the way
 * out: g = bf16(bf16(dgrad(dy)) + addend) (addend optional), m = act > 0 ? g : 0, dres = m, d3 = bf16(m * alpha[c]) — bit for bit
 * afan_conv_dgrad_nhwc_bf16(addend) followed by afan_affine_relu_bwd(relu = 1) with both outputs (Detection/backbone/
 * resnet101_ori.py:118-125 backwards).  AFAN_ESHAPE where another kernel owns the shape. */
int afan_conv_dgrad_dual_nhwc_bf16(const void* dy, const void* wt, void* d3, void* dres, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                                   int64_t co, int k, int stride, const void* addend, const float* alpha, const void* act,
                                   afan_stream_t stream);
/* The input gradient of a residual block's two stride-2 branches in ONE launch (Classification/resnet_s.py:52-77 with the
 * option-B projection): dx = conv_transpose(dy, w1: 3x3 / 2, pad 1) + conv_transpose(dy_sc, w_sc: 1x1 / 2) — what autograd
 * computes as two input gradients and a sum at main_perturb.py:200 / attack_algo.py:52.  dy, dy_sc: [N, hi/2, wi/2, co] bf16
 * in ONE allocation, dy_sc behind dy; wt10: [ci][10][co] (nine transposed taps of w1, then w_sc transposed:
 * afan_transpose_weights); dx [N, hi, wi, ci].  bn_*: the same BatchNorm-backward epilogue fusion as
 * afan_conv_dgrad_nhwc_bf16. */
int afan_conv_dgrad_sc_nhwc_bf16(const void* dy, const void* dy_sc, const void* wt10, void* dx, int64_t n, int64_t hi,
                                 int64_t wi, int64_t ci, int64_t co, const void* bn_x, const float* bn_stats, int bn_relu,
                                 const void* bn_y, float* bn_partials, double* bn_acc, afan_stream_t stream);
/* ---- convolution + train-mode BatchNorm in ONE launch (round 5) --------------------------------------------------------------
 * Replaces, for the chain conv - bn - relu - conv - bn - (+shortcut) - relu of Classification/resnet_s.py:72-77 run K + 2 times per
 * iteration by attack_algo.py:48-56 / main_perturb.py:195-196, the pair (afan_conv_fwd_nhwc_bf16 with stats_acc,
 * afan_bn_train_forward_acc[_dual]) resp. (afan_conv_dgrad_nhwc_bf16 with bn_acc, afan_bn_backward_acc) by one launch each, bit
 * for bit: the epilogue adds its tile's column sums to the f64 accumulators, meets every other workgroup of the launch at a grid-
 * wide barrier (relaxed agent-scope atomics behind drained accumulator atomics), reads the totals back and finishes its tile from
 * LDS.  Only launches whose workgroups are all resident take it (3x3 or 1x1 at stride 1, whole 64-channel chunks, output channels
 * a multiple of 128; the projection shortcut's backward, sc_x, only on the LDS-resident halo form of the 3x3 ones); AFAN_ESHAPE =
 * "not this launch": NOTHING has run, issue the two launches instead.
 * barrier: afan_grid_barrier_bytes() of 64-byte aligned device memory, zeroed once, shared by all such launches of a stream (one
 * launch at a time may use it).  Its word afan_grid_barrier_error_word() becomes non-zero when a bounded barrier spin (0.2 s) gave
 * up — the launch's workgroups were not co-resident (another process's kernels on the same GPU): results are invalid, zero the
 * buffer before the next use.  Never issue such launches concurrently on two streams of one device. */
int afan_grid_barrier_bytes(void);
int afan_grid_barrier_error_word(void);
/* on != 0 while kernels of ANOTHER stream of this process may run beside these launches (e.g. weight gradients on a side stream):
 * only launches of at most one workgroup per CU then take the in-launch form — two per CU need two contiguous LDS ranges, and beside
 * a long-lived workgroup of another kernel the second may never find its range while the first spins for it.  Returns the previous
 * setting.  (Kernels that themselves wait for something — a collective — must not run beside these launches at all.) */
int afan_grid_barrier_shared_gpu(int on);
/* y_raw = conv(x, w) (ksize 3 with padding = dilation, or 1; stride 1) and y_act = [relu](bn(y_raw) [+ residual]) with y_raw's batch statistics (acc: zeroed accumulator block,
 * moments taken around shift = the running mean); sc_raw != NULL: y_act = relu(bn(y_raw) + bn_sc(sc_raw)) with the projection
 * shortcut's BatchNorm derived from sc_acc (filled by the launch that produced sc_raw).  stats / sc_stats [4][co] out; running
 * statistics updated afan_bn_set_running_updates() times. */
int afan_conv_fwd_bn_nhwc_bf16(const void* x, const void* w, void* y_raw, void* y_act, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                               int64_t co, int ksize, int dilation, double* acc, const float* shift, const float* bn_weight, const float* bn_bias, float eps,
                               float momentum, float* stats, float* running_mean, float* running_var, int64_t* num_batches,
                               const void* residual, int relu, const void* sc_raw, const double* sc_acc, const float* sc_weight,
                               const float* sc_bias, float sc_eps, float sc_momentum, float* sc_stats, float* sc_running_mean,
                               float* sc_running_var, int64_t* sc_num_batches, void* barrier, afan_stream_t stream);
/* dx = the gradient entering the INPUT of the BatchNorm (+ ReLU) in front of the 3x3 (dilation >= 1) or 1x1 (ksize) stride-1 convolution whose output gradient
 * is dy (bn_x, bn_stats, bn_relu, bn_y, addend as in afan_conv_dgrad_nhwc_bf16; bn_acc zeroed), dres (optional) = the masked
 * gradient (afan_bn_backward_acc's second output); dweight / dbias [ci] optional.
 * sc_x != NULL (only with bn_y: the gradient at a residual block's output): that block's projection shortcut's BatchNorm (no ReLU)
 * receives the masked gradient too and its backward runs here as well: sc_x its input (the 1x1 projection's raw output), sc_stats its
 * [4][ci] block, sc_acc a second zeroed accumulator block, d_sc = the gradient entering its input, sc_dweight / sc_dbias optional
 * (afan_bn_backward(dres, sc_x, relu = 0) up to the summation order of its two channel sums). */
int afan_conv_dgrad_bn_nhwc_bf16(const void* dy, const void* wt, void* dx, void* dres, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                                 int64_t co, int ksize, int dilation, const void* addend, const void* bn_x, const float* bn_stats, int bn_relu, const void* bn_y,
                                 double* bn_acc, float* dweight, float* dbias, int accumulate, const void* sc_x, const float* sc_stats,
                                 double* sc_acc, void* d_sc, float* sc_dweight, float* sc_dbias, void* barrier, afan_stream_t stream);
/* afan_conv_fwd_multi_nhwc_bf16 with problem 0's train-mode BatchNorm (+ ReLU) applied inside the launch: y_act0 = [relu](bn(y[0])),
 * stats0 [4][co] out, running buffers updated like afan_bn_train_forward_acc — a residual block's first 3x3 / stride-2 convolution
 * (Classification/resnet_s.py:72) with its 1x1 projection as problem 1 (whose BatchNorm the block's last launch applies).  Only
 * problem 0's workgroups meet at the grid barrier; stats_acc / stats_shift required for every problem. */
int afan_conv_fwd_multi_bn_nhwc_bf16(const void* x, const void* const* w, void* const* y, int nb, int64_t n, int64_t hi, int64_t wi,
                                     int64_t ci, int64_t co, const int* ksize, int stride, const int* dilation,
                                     const float* const* stats_shift, double* const* stats_acc, void* y_act0, const float* bn_weight,
                                     const float* bn_bias, float eps, float momentum, float* stats0, float* running_mean,
                                     float* running_var, int64_t* num_batches, int relu, void* barrier, afan_stream_t stream);
/* The same for the stride-2 pair form (afan_conv_dgrad_sc_nhwc_bf16): the backward of the PREVIOUS block's last BatchNorm (block-output
 * form: bn_y its stored output, dres the masked gradient for its shortcut) inside the launch that computes the gradient leaving
 * that block — one set of sums over the four output-parity classes.  hi, wi even. */
int afan_conv_dgrad_sc_bn_nhwc_bf16(const void* dy, const void* dy_sc, const void* wt10, void* dx, void* dres, int64_t n, int64_t hi,
                                    int64_t wi, int64_t ci, int64_t co, const void* bn_x, const float* bn_stats, int bn_relu,
                                    const void* bn_y, double* bn_acc, float* dweight, float* dbias, int accumulate, void* barrier,
                                    afan_stream_t stream);
/* dgrad epilogue fusions (all optional, NULL = off):
 *   addend      [N,Hi,Wi,Ci] bf16: dx = bf16(dgrad + addend) — the sum autograd would launch where a block input feeds
 *               both the main branch and the shortcut;
 *   bn_x, bn_stats, bn_relu, bn_partials: dx is the gradient arriving at the output of the BatchNorm(+ReLU) whose INPUT
 *               was bn_x and whose saved statistics block is bn_stats[4*Ci]; the reduction pass of that BN's backward
 *               (sum g, sum g*(x-mean), g = ReLU-masked dx) is taken here: bn_partials[2][Ci][G],
 *               G = afan_conv_dgrad_tiles(...), consumed by afan_bn_backward(..., partials, G);
 *   bn_y        optional [N,Hi,Wi,Ci] bf16: the OUTPUT of that BatchNorm after its residual add and ReLU (the next
 *               block's input): the ReLU mask is (bn_y > 0) instead of the recomputed bn_x*alpha+beta > 0 — the form a
 *               BasicBlock's second BatchNorm needs (resnet_s.py:54-55: relu(bn2(conv2(.)) + shortcut(x)));
 *   groups      2 (with bn_acc): two concatenated half-batches with separate statistics — bn_stats is [2][4*Ci] and
 *               bn_acc two accumulator blocks, as for the forward; 1 otherwise;
 *   bn_acc      (instead of bn_partials) the same sums added into the zeroed accumulator block bn_acc[afan_bn_acc_doubles(Ci)], consumed by
 *               afan_bn_backward_acc(..., acc_ready = 1). */
int64_t afan_conv_dgrad_tiles(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride);

/* Weight gradient: grad[Co,k,k,Ci] (fp32, KRSC — the layout of the parameter arena) (+)= sum over output pixels of
 * dy[p][co] * x[pixel(p) + tap][ci]; bf16 channels-last operands, the `loss.backward()` of main_perturb.py:200 for the
 * conv weights.  Deterministic two-launch split over pixel slices (partials in `workspace`, summed in order);
 * accumulate != 0 adds into grad (the two branches of the joint loss both contribute to tail weights). */
int64_t afan_conv_wgrad_workspace_floats(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride);
int afan_conv_wgrad_nhwc_bf16(const void* x, const void* dy, float* grad, int64_t n, int64_t hi, int64_t wi,
                              int64_t ci, int64_t co, int k, int stride, int dilation, float* workspace, int accumulate,
                              afan_stream_t stream);
/* The same over TWO operand pairs of one layer in one launch and one slab reduction:
 * grad (+)= wgrad(x[n], dy[n]) + wgrad(x2[n2], dy2[n2]) — a tail layer's operands of the clean and of the adversarial pass of
 * one iteration (main_perturb.py:195-200: both contribute to the same .grad).  Tiled kernel (ci, co multiples of 8, >= 40) or the small-channel one;
 * n * Ho * Wo must be a multiple of 64; workspace = afan_conv_wgrad_workspace_floats(n + n2, ...); n2 = 0: one pair. */
int afan_conv_wgrad2_nhwc_bf16(const void* x, const void* dy, int64_t n, const void* x2, const void* dy2, int64_t n2,
                               float* grad, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride, int dilation,
                               float* workspace, int accumulate, afan_stream_t stream);

/* Which tile configuration the tiled weight-gradient kernel would use for this problem: 0 = not its problem (image stem,
 * small-channel layers, unsupported shapes), else an opaque code; problems with EQUAL codes can share one
 * afan_conv_wgrad_multi_nhwc_bf16 launch. */
int afan_conv_wgrad_plan(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride);
/* nb (1..4) weight gradients of DIFFERENT layers — e.g. the three convolutions of a bottleneck
 * (Segmentation/network/backbone/resnet.py:97-119) once all three output gradients exist — in ONE launch and one slab
 * reduction: grad[b] (+)= wgrad(x[b], dy[b]), shapes per problem in the arrays (host arrays; device pointers inside).
 * workspace = the sum of the problems' afan_conv_wgrad_workspace_floats.  Bit-identical to nb afan_conv_wgrad_nhwc_bf16
 * calls (same slices, same order); at 33 x 33 pixels and 2 images each of those is a 35-step reduction that costs mostly
 * its own launch. */
int afan_conv_wgrad_multi_nhwc_bf16(int nb, const void* const* x, const void* const* dy, float* const* grad, const int64_t* n,
                                    const int64_t* hi, const int64_t* wi, const int64_t* ci, const int64_t* co, const int* k,
                                    const int* stride, const int* dilation, float* workspace, int accumulate,
                                    afan_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The GENERAL convolution, fp32-accurate (afan_conv_f32.hip): what torch.nn.Conv2d computes in the reference's fp32 training
 * (Classification/main_perturb.py:173-201, attack_algo.py:50-52; Segmentation/network/backbone/resnet.py:143,
 * _deeplab.py:33-45,146-155) for ANY channel counts (the 3-channel image stems included), k <= 7, padding 0..127, stride 1 or 2,
 * dilation >= 1, optional bias — forward, input gradient and weight gradient on v_mfma_f32_32x32x2_f32 (f32 products, f32
 * accumulate: a k-ordered fmaf chain).  `dtype` AFAN_F32: fp32 storage (parity mode — north_star's 1e-4 bar is held on these
 * kernels) or AFAN_BF16: bf16 storage, the same fp32 arithmetic (shapes the tuned bf16 kernels above decline).  `layout` of the
 * activations AFAN_NHWC / AFAN_NCHW, `w_layout` of the weight tensor [Co,Ci,k,k]: AFAN_NHWC = KRSC memory (the parameter
 * arena's), AFAN_NCHW = KCRS (torch's default).  Ho = (Hi + 2 pad - dilation (k-1) - 1) / stride + 1.
 *   fwd  : y[N,Co,Ho,Wo]  = conv(x[N,Ci,Hi,Wi], w) (+ bias[Co], fp32, nullable)
 *   dgrad: dx[N,Ci,Hi,Wi] = conv_transpose(dy[N,Co,Ho,Wo], w) — reads the UNTRANSPOSED weight; stride 2 as four output-parity
 *          classes (no products with inserted zeros); input positions no window reaches receive 0
 *   wgrad: grad[Co,Ci,k,k] (fp32, `w_layout`) (+)= sum_p dy[p][co] * x[pixel(p) + tap][ci]; pixel slices reduced in fixed order
 *          (deterministic); workspace = afan_conv_wgrad_f32_workspace_floats(...) floats, 16-byte aligned.
 * n == 0: no launch (wgrad without accumulate zero-fills grad).  No vendor library is involved anywhere in this package. */
int afan_conv_fwd(const void* x, const void* w, const float* bias, void* y, int dtype, int layout, int w_layout, int64_t n,
                  int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride, int pad, int dilation, afan_stream_t stream);
int afan_conv_dgrad(const void* dy, const void* w, void* dx, int dtype, int layout, int w_layout, int64_t n, int64_t hi,
                    int64_t wi, int64_t ci, int64_t co, int k, int stride, int pad, int dilation, afan_stream_t stream);
int64_t afan_conv_wgrad_f32_workspace_floats(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride,
                                             int pad, int dilation);
int afan_conv_wgrad(const void* x, const void* dy, float* grad, int dtype, int layout, int w_layout, int64_t n, int64_t hi,
                    int64_t wi, int64_t ci, int64_t co, int k, int stride, int pad, int dilation, float* workspace,
                    int accumulate, afan_stream_t stream);

/* The ImageNet-style image stem of the DeepLabv3+ / ResNet-50/101 backbones (Segmentation/network/backbone/resnet.py:143-144
 * `conv1`: 7x7, stride 2, padding 3, 3 -> 64 channels): forward and weight gradient, bf16 channels-last.
 * x[N,Hi,Wi,3], w[64,7,7,3] (KRSC), y / dy[N,Ho,Wo,64], Ho = (Hi-1)/2+1; grad fp32 [64,7,7,3] written or added into.
 * No input gradient: the images carry none on the feature-perturbation path. */
int afan_conv_stem7_supported(int64_t ci, int64_t co, int k, int stride);
int afan_conv_stem7_fwd_nhwc_bf16(const void* x, const void* w, void* y, int64_t n, int64_t hi, int64_t wi,
                                  afan_stream_t stream);
/* im2col form (the product path): cols[N*Ho*Wo][K] bf16 with K = afan_conv_stem7_im2col_k() = 152 (147 taps in the
 * weights' KRSC order + 5 zeros); forward and weight gradient are then 1x1 problems for afan_conv_fwd_nhwc_bf16 /
 * afan_conv_wgrad_nhwc_bf16 (ci = 152) on the MFMA kernels. */
int afan_conv_stem7_im2col_k(void);
int afan_conv_stem7_im2col(const void* x, void* cols, int64_t n, int64_t hi, int64_t wi, afan_stream_t stream);
/* The adjoint: dx[N,Hi,Wi,3] bf16 = sum of the columns of dcols[N*Ho*Wo][K] over the taps that cover each input pixel.  With
 * dcols = dy x W (afan_conv_fwd_nhwc_bf16 as a 1x1 problem, 64 -> 152) it is the stem's input gradient — the image-level
 * perturbation of Detection/train_aug_sat_muti_advt.py:82-95 differentiates through conv1. */
int afan_conv_stem7_col2im(const void* dcols, void* dx, int64_t n, int64_t hi, int64_t wi, afan_stream_t stream);
int64_t afan_conv_stem7_wgrad_workspace_floats(int64_t n, int64_t hi, int64_t wi);
int afan_conv_stem7_wgrad_nhwc_bf16(const void* x, const void* dy, float* grad, int64_t n, int64_t hi, int64_t wi,
                                    float* workspace, int accumulate, afan_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The non-convolution layers of the DeepLabv3+ split-forward network.  `dtype` AFAN_F32 / AFAN_BF16,
 * `layout` AFAN_NCHW / AFAN_NHWC; tensors dense.  Backward passes are gathers (no float atomics: reproducible).
 *
 * Bilinear resize with align_corners=False — F.interpolate(..., mode='bilinear', align_corners=False) at
 * Segmentation/network/utils.py:30,45 and _deeplab.py:54,66,75,141 (ATen's source-index rule in fp32:
 * src = max(0, (in/out)*(dst+0.5)-0.5)).  fwd: y[n,c,ho,wo] from x[n,c,hi,wi]; bwd: dx[n,c,hi,wi] from dy[n,c,ho,wo]. */
int afan_upsample_bilinear_fwd(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                               int64_t ho, int64_t wo, afan_stream_t stream);
int afan_upsample_bilinear_bwd(const void* dy, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                               int64_t ho, int64_t wo, afan_stream_t stream);
/* The same resize where the HIGH-resolution side is a channel slice of a wider channels-last tensor: y (forward) / dy (backward)
 * points at the slice's first channel and consecutive pixels are `ld` elements apart (ld >= c).  The decoder's concat,
 * torch.cat([low_level_feature, F.interpolate(aspp_out, ...)], dim=1) at Segmentation/network/_deeplab.py:54-56, without the
 * concat's own pass: the resized tensor is written into channels 48..303 of the 304-channel tensor, and its gradient is read
 * from there.  Channels-last only. */
int afan_upsample_bilinear_fwd_slice(const void* x, void* y, int dtype, int64_t n, int64_t c, int64_t hi, int64_t wi, int64_t ho,
                                     int64_t wo, int64_t ld, afan_stream_t stream);
/* bwd: ws = afan_upsample_bilinear_bwd_workspace_floats(n, c, wi, ho) floats of scratch (16-byte aligned) selects the two-pass
 * separable form (sum over output columns into ws, then over output rows: ~scale loads per thread twice instead of ~scale^2
 * once on a small grid); NULL (or a shape whose accesses are not 16-byte vectors): the one-pass gather of the dense entry. */
int64_t afan_upsample_bilinear_bwd_workspace_floats(int64_t n, int64_t c, int64_t wi, int64_t ho);
int afan_upsample_bilinear_bwd_slice(const void* dy, void* dx, int dtype, int64_t n, int64_t c, int64_t hi, int64_t wi, int64_t ho,
                                     int64_t wo, int64_t ld, float* ws, afan_stream_t stream);
/* Per-pixel cross-entropy, nn.CrossEntropyLoss(ignore_index=I, reduction='mean') of Segmentation/main_aug_final.py:95:
 * logits fp32 [n,c,hw] (NCHW) or [n,hw,c] (NHWC), c <= 32; target int64 [n,hw].  loss[0] = sum over pixels with
 * target != I of (logsumexp - logit[target]) / count; dlogits (nullable, same layout) = grad_scale * d(loss)/d(logits)
 * (0 at ignored pixels).  A target outside [0,c) that is not I makes the loss NaN.  workspace: afan_ce2d_workspace_floats(n*hw). */
int64_t afan_ce2d_workspace_floats(int64_t pixels);
int afan_ce2d(const float* logits, const int64_t* target, int layout, int64_t n, int64_t c, int64_t hw, int64_t ignore_index,
              float grad_scale, float* workspace, float* loss, float* dlogits, afan_stream_t stream);
/* The same loss taken on logits that Segmentation/network/utils.py:30,45 first resizes bilinearly (align_corners=False) from
 * [n,h,w,c] (channels-last fp32, c <= 32) to the labels' [n,ho,wo] — resize, cross-entropy and both backward passes in ONE
 * pass over 16 x 16 output tiles (+ a small gather of per-tile partials), the gradient returned at the LOW resolution:
 * dlogits [n,h,w,c] (nullable) = grad_scale * d(loss)/d(logits).  The arithmetic is afan_upsample_bilinear_fwd + afan_ce2d +
 * afan_upsample_bilinear_bwd operation for operation (sums grouped by tile: equal to rounding) without the four passes over
 * the [n,ho,wo,c] tensor.  ho >= h, wo >= w (up-scaling); deterministic; workspace: afan_ce2d_upsampled_workspace_floats(...). */
int64_t afan_ce2d_upsampled_workspace_floats(int64_t n, int64_t c, int64_t h, int64_t w, int64_t ho, int64_t wo);
int afan_ce2d_upsampled(const float* logits, const int64_t* target, int64_t n, int64_t c, int64_t h, int64_t w, int64_t ho,
                        int64_t wo, int64_t ignore_index, float grad_scale, float* workspace, float* loss, float* dlogits,
                        afan_stream_t stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (backbone/resnet.py:146): ho = (hi-1)/2+1.  The backward routes each
 * output gradient to the FIRST maximum of its window in (h, w) scan order, NaN winning — ATen's CPU rule. */
int afan_maxpool3x3s2_fwd(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                          afan_stream_t stream);
int afan_maxpool3x3s2_bwd(const void* dy, const void* x, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hi,
                          int64_t wi, afan_stream_t stream);
/* The general form: k x k windows (k <= 15), any stride, padding <= k/2; ho = (hi + 2 pad - k) / stride + 1.  Same winner rule.
 * Detection/roi/pooler.py:43 (2 x 2 / 2 after ROIAlign), Detection/model.py:285,336 (adaptive_max_pool2d(., 1) = one window).
 * idx (nullable): uint8 [output elements, the output's layout] — the winner's position inside its window (r * k + s), written
 * by the forward; the backward given idx reads (gradient, position) of the windows over an input pixel instead of
 * re-scanning them (x may then be NULL). */
int afan_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                       int k, int stride, int pad, afan_stream_t stream);
int afan_maxpool2d_bwd(const void* dy, const void* x, const uint8_t* idx, void* dx, int dtype, int layout, int64_t n, int64_t c,
                       int64_t hi, int64_t wi, int k, int stride, int pad, afan_stream_t stream);
/* Backward of y = [relu](x * alpha[c] + beta[c] [+ res]) with CONSTANT per-channel coefficients — a frozen (eval-mode,
 * gradient-free) BatchNorm, Detection/model.py:27-35,46-47, whose forward is afan_bn_apply; or a convolution bias (+ ReLU),
 * Detection/rpn/region_proposal_network.py:19-22: g = dy masked by (y > 0) when relu; dx (nullable) = g * alpha[c] (alpha
 * NULL = 1); d_res (nullable) = g.  Tensors [n, c, hw] in `layout`. */
int afan_affine_relu_bwd(const void* dy, const void* y, const float* alpha, void* dx, void* dres, int dtype, int layout,
                         int64_t n, int64_t c, int64_t hw, int relu, afan_stream_t stream);
/* The same layer's forward with its coefficients computed ONCE: afan_affine_coefs writes coefs[4][c] = mean | invstd |
 * alpha | beta (alpha = invstd * weight, beta = bias - mean * alpha; weight / bias NULL = 1 / 0) — they are constants of a
 * frozen BatchNorm (Detection/model.py:31-35: no gradient, eval mode) — and afan_affine_apply is the one streaming launch
 * y = [relu](x * alpha[c] + beta[c] [+ residual]) on channels-last tensors [n, hw, c] (afan_bn_apply = both, per call). */
int afan_affine_coefs(const float* mean, const float* invstd, const float* weight, const float* bias, int64_t c,
                      float* coefs, afan_stream_t stream);
int afan_affine_apply(const void* x, const void* residual, void* y, int dtype, int64_t n, int64_t c, int64_t hw,
                      const float* coefs, int relu, afan_stream_t stream);

/* A frozen-BatchNorm bottleneck (Detection/backbone/resnet101_ori.py:78-127 under Detection/model.py:27-35,46-47) as ONE host
 * call each way — no new kernel, the block's launches issued from native code (afan_block.hip): the eager Detection iteration
 * is bound by per-launch Python dispatch.  Shapes: x [n, cin, h, w] -> out [n, 4 planes, ho, wo], ho = (h - 1) / stride + 1;
 * all activations bf16 channels-last; w1 / w2 / w3 / wd KRSC bf16 (wd NULL: identity shortcut), wt*: their CRSK transposes;
 * k*: afan_affine_coefs blocks, al*: their alpha rows.  fwd scratch: n * (planes*h*w + 2 * 4planes*ho*wo) bf16 elements;
 * a1 [n, planes, h, w], a2 [n, planes, ho, wo] are what the backward needs besides x and out.  bwd: gw* (nullable) fp32 KRSC
 * gradients ADDED into; wgrad_ws: the sum of the wanted layers' afan_conv_wgrad_workspace_floats; dx nullable. */
int afan_frozen_bottleneck_fwd(const void* x, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t planes, int stride,
                               const void* w1, const void* w2, const void* w3, const void* wd, const float* k1, const float* k2,
                               const float* k3, const float* kd, void* scratch, void* a1, void* a2, void* out,
                               afan_stream_t stream);
int64_t afan_frozen_bottleneck_bwd_scratch(int64_t n, int64_t h, int64_t w, int64_t cin, int64_t planes, int stride);
int afan_frozen_bottleneck_bwd(const void* g, const void* x, const void* a1, const void* a2, const void* out, int64_t n,
                               int64_t h, int64_t w, int64_t cin, int64_t planes, int stride, const void* wt1, const void* wt2,
                               const void* wt3, const void* wtd, const float* al1, const float* al2, const float* al3,
                               const float* ald, float* gw1, float* gw2, float* gw3, float* gwd, float* wgrad_ws, void* scratch,
                               void* dx, afan_stream_t stream);
/* The same inside a chain of blocks (a stage: block i + 1's input is block i's output): g NULL -> the first step of this block's
 * backward was done by the block behind it (pre_d3 = bf16(m * al3), pre_dres = m, m = the output's ReLU mask applied to the
 * gradient; pre_dres may be overwritten); dx NULL and prev_al3 given -> that step of the block in front is done HERE, in the
 * epilogue of the last input-gradient launch (afan_conv_dgrad_dual_nhwc_bf16): prev_d3 / prev_dres [n, cin, h, w]. */
int afan_frozen_bottleneck_bwd_chain(const void* g, void* pre_d3, void* pre_dres, const void* x, const void* a1, const void* a2,
                                     const void* out, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t planes, int stride,
                                     const void* wt1, const void* wt2, const void* wt3, const void* wtd, const float* al1, const float* al2,
                                     const float* al3, const float* ald, float* gw1, float* gw2, float* gw3, float* gwd, float* wgrad_ws,
                                     void* scratch, void* dx, const float* prev_al3, void* prev_d3, void* prev_dres, afan_stream_t stream);
/* nn.AdaptiveAvgPool2d(1) (_deeplab.py:133): y[n,c] = mean over hw (fp32 accumulate); dx = dy / hw broadcast.
 * pooled_f32 != 0: the pooled side (y / dy) is fp32 whatever `dtype` the map has. */
int afan_avgpool_fwd(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hw, int pooled_f32,
                     afan_stream_t stream);
int afan_avgpool_bwd(const void* dy, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hw, int pooled_f32,
                     afan_stream_t stream);
/* The 1x1 convolution of the ASPP pooling branch (_deeplab.py:155: one 2048-vector per image) as an fp32 linear layer on
 * n <= afan_linear_small_max_rows() rows with the fp32 master weights: y[n,co] = sum_ci x[n,ci]*w[co,ci].  That branch's
 * BatchNorm normalises over the n images only; bf16 storage of the pooled vectors rounds their differences away.
 * bwd: dx (nullable) [n,ci], dw (nullable) [co,ci] written or added into. */
int afan_linear_small_max_rows(void);
int afan_linear_small_fwd(const float* x, const float* w, float* y, int64_t n, int64_t ci, int64_t co, afan_stream_t stream);
int afan_linear_small_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, int64_t n, int64_t ci,
                          int64_t co, int accumulate, afan_stream_t stream);
/* 1x1 convolution WITH bias to a few output channels and fp32 logits — the classifier nn.Conv2d(256, num_classes, 1) of
 * _deeplab.py:45 on channels-last pixels: y[m,co] = b[co] + sum_ci x[m,ci]*w[co,ci]; x `x_dtype` [m,ci] (ci % 8 == 0),
 * w fp32 [co,ci], co <= afan_pointwise_max_co().  dx[m,ci] = dy[m,:] @ w; dw[co,ci] (+)= sum_m dy[m,co]*x[m,ci],
 * db[co] (+)= sum_m dy[m,co] (two launches, slices of 128 pixels summed in order). */
int afan_pointwise_max_co(void);
int afan_pointwise_fwd(const void* x, int x_dtype, const float* w, const float* b, float* y, int64_t m, int64_t ci,
                       int64_t co, afan_stream_t stream);
int afan_pointwise_bwd_dx(const float* dy, const float* w, void* dx, int dx_dtype, int64_t m, int64_t ci, int64_t co,
                          afan_stream_t stream);
int64_t afan_pointwise_workspace_floats(int64_t m, int64_t ci, int64_t co);
int afan_pointwise_bwd_dw(const float* dy, const void* x, int x_dtype, float* dw, float* db, int64_t m, int64_t ci,
                          int64_t co, float* workspace, int accumulate, afan_stream_t stream);
/* nn.Dropout(p) (_deeplab.py:185): y = keep ? x / (1 - p) : 0.  mask != NULL: keep = mask[i] != 0 (host-supplied, parity
 * tests).  Otherwise keep(i) is a counter-based hash of (seed, i): forward passes `state` (device uint64, the generator:
 * seed = state[0]; copied to used[0]; advanced afterwards when `advance`), backward passes state = NULL and the `used`
 * the forward filled — the mask is re-derived, never stored, and a replayed hipGraph draws fresh masks. */
int afan_dropout(const void* x, void* y, int dtype, int64_t n, float p, const uint8_t* mask, uint64_t* state, uint64_t* used,
                 int advance, afan_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Detection operators: the reference's only native code, Detection/support/src/cuda/{nms,ROIAlign_cuda}.cu.
 *
 * afan_nms — `support._C.nms(dets, scores, threshold)` (nms.h:10-28 -> nms.cu:70-130).  boxes [n,4] fp32 (left, top, right,
 * bottom, corners inclusive: areas and intersections use +1), order [n] int64 = box indices by DESCENDING score (the caller's
 * sort, as nms.cu:73-75 sorts with ATen).  A box is dropped when its IoU with a kept box of higher score is > threshold
 * (inclusive = 0: nms.cu:49) or >= threshold (inclusive = 1: the CPU path, nms_cpu.cpp:62).  keep_out [n] int64 receives the
 * kept boxes' ORIGINAL indices in ascending order (what nms_cuda returns after its final sort), count_out[0] how many.  The
 * greedy scan runs on the device (the reference copies the n x n/64 mask to the host and syncs).  workspace:
 * afan_nms_workspace_bytes(n) bytes, 8-byte aligned (the mask, a flag per box, and the 13 tiles at and right of the diagonal
 * of every row block once more in column form).  n <= 524 288. */
int64_t afan_nms_workspace_bytes(int64_t n);
int afan_nms(const float* boxes, const int64_t* order, int64_t n, float threshold, int inclusive, void* workspace,
             int64_t* keep_out, int64_t* count_out, afan_stream_t stream);
/* afan_nms_top — the same for a caller that looks at the first max_keep survivors only, the proposal layer's
 * `nms(...)` followed by `[:post_nms_top_n]` (rpn/region_proposal_network.py:88-93 with boxes pre-sorted by score): the scan
 * stops after the 64-box block in which the kept count reaches max_keep (> 0), since what greedy NMS keeps first never depends
 * on later boxes.  count_out[0] may exceed max_keep by up to 63; keep_out is ascending by ORIGINAL index as above, so "first
 * max_keep" means first by score only when the boxes were passed in score order (then the result's first max_keep entries
 * equal afan_nms's).  max_keep = 0: afan_nms. */
int afan_nms_top(const float* boxes, const int64_t* order, int64_t n, float threshold, int inclusive, void* workspace,
                 int64_t* keep_out, int64_t* count_out, int64_t max_keep, afan_stream_t stream);

/* Training targets and per-image losses of the Faster-RCNN step as single launches (the reference composes each from dozens of
 * tensor operations: bbox.py:41-92, rpn/region_proposal_network.py:58-105,163-185, model.py:256-282,343-367).  fp32, boxes as
 * (left, top, right, bottom), indices int64, all pointers device memory.
 *
 * afan_box_decode_clip — bbox.py:54-64 `apply_transformer` then :89-92 `clip` to [0, right] x [0, bottom]; n boxes.
 * afan_box_assign — IoU (bbox.py:66-82) of boxes [B,N,4] with gt [B,G,4]: assign [B,N] = the best ground truth (first maximum),
 *   labels [B,N] by mode 0 (anchors, region_proposal_network.py:66-82: -1; 0 below lo; 1 at hi and above, or tying a ground truth's
 *   best positive IoU) or mode 1 (proposals, model.py:256-264: -1; 0 below lo; gt_classes[b, assign] at lo and above).  workspace:
 *   B*G*4 bytes (mode 0).
 * afan_sample_lists — fg / bg [M]: the ascending positions of labels > 0 / == 0 (`nonzero()` order), counts[2] their lengths:
 *   what the host's `randperm` draws (region_proposal_network.py:84-90, model.py:277-282) need to know.
 * afan_sample_gather — the S drawn rows: pos[i] >= 0 names fg[pos[i]], pos[i] < 0 names bg[-pos[i]-1]; out: sel (flat position),
 *   the row's box, label, image index and regression target bbox.py:41-52 `calc_transformer`(box, gt[b, assign]).
 * afan_det_loss_fwd / _bwd — region_proposal_network.py:163-185 == model.py:343-367: per image the mean cross-entropy of its
 *   samples and the beta-smooth-L1 (extension/functional.py:6-10) of its foreground samples over 4 x foreground + 1e-8.  Sample s
 *   reads row rows[s] (or s) of logits [R,C] and deltas [R,K,4], K = 1 or C (the sample's own class); norm = NULL or mean[4],
 *   std[4] for the targets.  save: S*4 + S*C + 2*B floats, kept for _bwd, which writes DENSE d_logits [R,C] / d_deltas [R,K,4]. */
int afan_box_decode_clip(const float* src, const float* t, float* out, int64_t n, float right, float bottom, afan_stream_t stream);
int afan_box_assign(const float* boxes, const float* gt, int64_t B, int64_t N, int64_t G, int mode, float lo, float hi,
                    const int64_t* gt_classes, int64_t* labels, int64_t* assign, void* workspace, afan_stream_t stream);
int afan_sample_lists(const int64_t* labels, int64_t M, int64_t* fg, int64_t* bg, int64_t* counts, afan_stream_t stream);
int afan_sample_gather(const int64_t* fg, const int64_t* bg, const int64_t* pos, int64_t S, const float* boxes, const float* gt,
                       const int64_t* assign, const int64_t* labels, int64_t N, int64_t G, int64_t* sel, float* out_boxes,
                       int64_t* out_labels, float* out_deltas, int64_t* out_batch, afan_stream_t stream);
int afan_det_loss_fwd(const float* logits, const float* deltas, const int64_t* rows, const int64_t* gt_labels, const float* gt_deltas,
                      const int64_t* batch, int64_t S, int64_t B, int64_t C, int64_t K, float beta, const float* norm, float* ce, float* sl1,
                      float* save, afan_stream_t stream);
/* out[0] = ((a[0] + b[0]) + c[0]) + d[0] (c, d optional): `loss1.mean() + loss2.mean() + loss3.mean() + loss4.mean()`
 * (Detection/train_aug_sat_muti_advt.py:21-27, Detection/attack_algo.py:62) for per-image loss vectors of one image. */
int afan_sum_scalars_f32(const float* a, const float* b, const float* c, const float* d, float* out, afan_stream_t stream);
/* afan_linear_pair_* — two small `nn.Linear` / 1x1 `nn.Conv2d` layers on ONE input, fp32, N1 + N2 <= 128 output features: the ROI head's
 * `_proposal_class` | `_proposal_transformer` (Detection/model.py:235-236 <- :255-256, :290-291, :343-344) and the RPN's `_anchor_objectness` |
 * `_anchor_transformer` (Detection/rpn/region_proposal_network.py:35-36 <- :53-54, :120-121; x = the channels-last trunk read as [pixels, 512]).
 * x [M,K], w1 [N1,K], w2 [N2,K] row-major, K % 4 == 0; n2 == 0: one layer.  Deterministic (partials added in order, no float atomics).
 *   fwd:   y1 [M,N1] = x w1^T + b1, y2 [M,N2] = x w2^T + b2 (biases optional)
 *   dgrad: gx [M,K] = g1 w1 + g2 w2
 *   wgrad: gw1 [N1,K] (+)= g1^T x, gb1 [N1] (+)= column sums of g1 (optional), the same for layer 2; accumulate != 0 adds into the outputs
 * ws: afan_linear_pair_workspace_floats(op, ...) floats (op 0 forward — 0 when none is needed —, 1 parameter gradients), 16-byte aligned. */
int64_t afan_linear_pair_workspace_floats(int op, int64_t M, int64_t n1, int64_t n2, int64_t K);
int afan_linear_pair_fwd_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y1, float* y2, int64_t M,
                             int64_t n1, int64_t n2, int64_t K, float* ws, afan_stream_t stream);
int afan_linear_pair_dgrad_f32(const float* g1, const float* g2, const float* w1, const float* w2, float* gx, int64_t M, int64_t n1, int64_t n2,
                               int64_t K, afan_stream_t stream);
int afan_linear_pair_wgrad_f32(const float* g1, const float* g2, const float* x, float* gw1, float* gb1, float* gw2, float* gb2, int accumulate,
                               int64_t M, int64_t n1, int64_t n2, int64_t K, float* ws, afan_stream_t stream);
/* afan_proposal_rows — `sorted_bboxes[kept_indices][:post_nms_top_n]` and the zero padding of shorter images
 * (Detection/rpn/region_proposal_network.py:255-270) for ONE image with the survivor count still on the device: rows [P,4] =
 * cand[keep[j]] for j < min(count[0], P, n_keep), zero rows behind; kept[0] = that number.
 * afan_labels_limit — labels [B,N]: columns >= max(kept[0..n_kept)) become -1, so that the ROI head's sampling
 * (Detection/model.py:256-282) on the padded rows sees exactly the reference's candidates. */
int afan_proposal_rows(const float* cand, int64_t n_cand, const int64_t* keep, int64_t n_keep, const int64_t* count, int64_t P,
                       float* rows, int64_t* kept, afan_stream_t stream);
int afan_labels_limit(int64_t* labels, int64_t B, int64_t N, const int64_t* kept, int64_t n_kept, afan_stream_t stream);
int afan_det_loss_bwd(const float* g_ce, const float* g_sl1, const float* save, const int64_t* rows, const int64_t* gt_labels,
                      const int64_t* batch, int64_t S, int64_t B, int64_t C, int64_t K, int64_t R, float* d_logits, float* d_deltas,
                      afan_stream_t stream);
/* afan_roi_align_{fwd,bwd} — `support._C.roi_align_forward / roi_align_backward` (ROIAlign.h:12-47 -> ROIAlign_cuda.cu:256-346).
 * x [N,C,H,W], rois [num_rois,5] fp32 = (batch index, x1, y1, x2, y2) in image coordinates (scaled by spatial_scale, NOT
 * rounded), y [num_rois,C,PH,PW]; sampling_ratio <= 0: ceil(roi extent / pooled extent) sample points per bin and axis.
 * layout AFAN_NCHW (the reference's) or AFAN_NHWC (x [N,H,W,C], y [num_rois,PH,PW,C]: lanes along channels).
 * bwd: dx fp32 [N,C,H,W] in `layout` is zeroed and receives the scatter of dy through hardware fp32 atomics, like the
 * reference's atomicAdd (ROIAlign_cuda.cu:238-241): bitwise run-to-run reproducibility is not promised by either. */
int afan_roi_align_fwd(const void* x, const float* rois, void* y, int dtype, int layout, int64_t num_rois, int64_t c, int64_t h,
                       int64_t w, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, afan_stream_t stream);
int afan_roi_align_bwd(const void* dy, const float* rois, float* dx, int dtype, int layout, int64_t num_rois, int64_t n, int64_t c,
                       int64_t h, int64_t w, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                       afan_stream_t stream);
/* The same with caller-owned scratch for the channels-last form's bilinear weight tables (per (roi, map row) and (roi, map
 * column): 16 weights + the non-zero bin range), filled by one small launch: afan_roi_align_bwd_workspace_bytes(num_rois, h, w)
 * bytes, 16-byte aligned; NULL = afan_roi_align_bwd. */
int64_t afan_roi_align_bwd_workspace_bytes(int64_t num_rois, int64_t h, int64_t w);
int afan_roi_align_bwd_ws(const void* dy, const float* rois, float* dx, int dtype, int layout, int64_t num_rois, int64_t n, int64_t c,
                          int64_t h, int64_t w, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, void* workspace,
                          afan_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Classifier head of the slice protocol: AdaptiveAvgPool2d((1,1)) -> Flatten -> Linear (resnet_s.py:108-110), run at the
 * end of every tail pass.  x: channels-last [N,HW,C] (`dtype`); weight fp32 [K,C], bias fp32 [K] (nullable), K <=
 * afan_head_max_classes(); pooled fp32 [N,C] is kept for the backward; logits fp32 [N,K].
 * Backward: dx (nullable, `dx_dtype`, [N,HW,C]) = (dlogits @ weight) / HW broadcast over HW; dweight (nullable) and dbias
 * written (accumulate = 0) or added into (1): the batch is summed in index order (deterministic). */
int afan_head_max_classes(void);
int afan_head_forward(const void* x, int dtype, int64_t n, int64_t c, int64_t hw, const float* weight, const float* bias,
                      int64_t k, float* pooled, float* logits, afan_stream_t stream);
int afan_head_backward(const float* dlogits, const float* weight, const float* pooled, int64_t n, int64_t c, int64_t hw,
                       int64_t k, void* dx, int dx_dtype, float* dweight, float* dbias, int accumulate,
                       afan_stream_t stream);

/* The loss at the end of every tail pass: mean cross-entropy of logits[n][k] (fp32) against target[n] (int64 class ids;
 * nn.CrossEntropyLoss with its defaults, main_perturb.py:71 / attack_algo.py:51) AND its gradient in one launch:
 * loss[0] = mean_r (logsumexp(logits[r]) - logits[r][target[r]]),  dlogits[r][j] = (softmax(logits[r])[j] - [j == target[r]]) / n.
 * n * k <= 65536 (one workgroup; rows summed in fixed order). */
int afan_cross_entropy(const float* logits, const int64_t* target, int64_t n, int64_t k, float* loss, float* dlogits,
                       afan_stream_t stream);

/* Batched KRSC -> CRSK transpose of every convolution weight of the parameter arena (the dgrad operands `wt`), one
 * launch per SGD step.  desc_dev: device array of n_desc x 8 int64 {src_off, dst_off, K, RS, C, first_tile, dst_RS, rs0}
 * (element offsets into src_arena / dst_arena, K % 8 == 0, C % 8 == 0, first_tile = running sum of
 * ceil(K/64)*RS*ceil(C/64)); the destination is [C][dst_RS][K] and the tensor's RS taps fill slots rs0 .. rs0+RS-1
 * (dst_RS = RS, rs0 = 0: the plain CRSK copy; a 3x3 weight with dst_RS = 10 plus its block's 1x1 projection at rs0 = 9:
 * the operand of afan_conv_dgrad_sc_nhwc_bf16). */
int afan_transpose_weights(const void* src_arena, void* dst_arena, const int64_t* desc_dev, int n_desc,
                           int64_t total_tiles, afan_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SGD with momentum over ONE flat parameter arena.  Replaces torch.optim.SGD.step as configured at
 * main_perturb.py:72-74 (momentum, weight_decay, no nesterov/dampening) on every tensor at once:
 *   g' = g + wd*p ;  buf = first ? g' : momentum*buf + g' ;  p -= lr*buf
 * lr is read from DEVICE memory (warm-up changes it every step, main_perturb.py:288-293, and the
 * step may be replayed from a hipGraph).  grad_scale multiplies g first (1/world_size after a sum
 * all-reduce).  shadow_bf16 (nullable): bf16 copy of the updated parameters for the bf16 backbone.
 */
int afan_sgd_step(float* param, const float* grad, float* momentum_buf, uint16_t* shadow_bf16,
                  int64_t n, const float* lr_dev, float momentum, float weight_decay,
                  float grad_scale, int first_step, afan_stream_t stream);

/* The same update, skipped ON THE DEVICE when *guard != 0.  guard = the grid barrier's error word (barrier + 4 *
 * afan_grid_barrier_error_word()): an in-launch BatchNorm (afan_conv_*_bn_*) whose bounded spin gave up went on with partial batch
 * totals, so the step's gradients are invalid — this launch then touches neither parameters nor momentum nor the bf16 shadow, and
 * (the word is sticky until the host clears it) neither does any later one: no corrupted update is ever applied, without a host
 * read inside the step.  The host notices at its next look (train_step.GridGuard) and re-runs the lost steps on the two-launch forms. */
int afan_sgd_step_guarded(float* param, const float* grad, float* momentum_buf, uint16_t* shadow_bf16,
                          int64_t n, const float* lr_dev, float momentum, float weight_decay,
                          float grad_scale, int first_step, const void* guard, afan_stream_t stream);
/* dst = src (bytes a multiple of 16; 16-byte aligned) unless *guard != 0; *counter (nullable, device u32) += 1 when the copy
 * happened.  The pre-step snapshot of the BatchNorm buffers (running statistics are updated inside the launches that may give up):
 * once the word is set the snapshot keeps the buffers as they were at the START of the step that failed. */
int afan_guarded_copy(void* dst, const void* src, int64_t bytes, const void* guard, void* counter, afan_stream_t stream);
/* Test / probe aid (no reference counterpart): `workgroups` one-wave workgroups holding `lds_bytes` (256 .. 163840) of LDS each spin
 * on `stream` for `microseconds` (<= 2 s) — stands for another stream's long-lived kernels (an RCCL channel kernel, a side-stream
 * weight gradient) beside the step's launches: tests/test_grid_guard_gpu.py, tools/probe/cu_sharing.py. */
int afan_occupy_cus(int workgroups, int lds_bytes, int microseconds, afan_stream_t stream);

/* fp32 -> bf16 (round-to-nearest-even, NaN-preserving) and per-channel input normalisation
 * (resnet_s.py:87: (x - mean[c]) / std[c]); the input image is NCHW fp32, the output may be NHWC and/or bf16. */
int afan_cast_bf16(const float* src, uint16_t* dst, int64_t n, afan_stream_t stream);
/* PGD's start, Classification/attack_algo.py:41 (`x_adv = x.clone()`), for a feature map stored in bf16 or fp32: writes the
 * fp32 copy x32 (centre of the L-inf ball / of the norms; may be NULL when x is fp32), the fp32 iterate x_adv and,
 * if shadow != NULL, x_adv's bf16 copy — one pass. */
int afan_pgd_init(const void* x, int dtype, float* x32, float* x_adv, uint16_t* shadow, int64_t n, afan_stream_t stream);
int afan_normalize_nchw(const float* x, void* y, int out_dtype, int out_layout, int64_t n, int64_t c, int64_t hw,
                        const float* mean, const float* std, afan_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): per-launch HIP-event timing of the kernels above, recorded
 * on the launch stream.  afan_profile_enable(1) starts recording (launches are then not graph-capturable),
 * afan_profile_collect() synchronises, aggregates per kernel name and clears: names_out holds max_kernels
 * 64-byte C strings; launches/total_ms/total_bytes/total_flops are per-kernel launch counts, summed durations, summed
 * ALGORITHMIC bytes (DESIGN.md lists the per-element figures) and, for the MFMA kernels, summed algorithmic FLOPs.  Returns the number of kernels written.
 */
int afan_profile_enable(int on);
int afan_profile_collect(char* names_out, int64_t* launches, double* total_ms, double* total_bytes,
                         double* total_flops, int max_kernels);
/* The bias of one (event, launch, event) bracket on `stream`: the elapsed time of n back-to-back event pairs with NOTHING between
 * them, averaged, in microseconds -> *us_out (host float).  bench.py subtracts it per launch from the instrumented durations
 * (they otherwise read ~5-8 % longer than rocprofv3's kernel durations on 10-20 us kernels).  Synchronises the stream. */
int afan_profile_event_overhead(int n, float* us_out, afan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AFAN_HIP_H */
