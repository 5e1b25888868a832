// Problem description shared by the implicit-GEMM convolution kernels (afan_conv.hip, afan_conv_small.hip).
#pragma once
#include "afan_common.h"

namespace afan_conv {

constexpr int MAX_TAPS = 9;

// One "class" = one set of output positions with its tap list.  A forward conv or a stride-1 dgrad has one class;
// a stride-2 dgrad has four (output-pixel parity), run as blockIdx.z of ONE launch.
struct ConvClass {
    int Hg, Wg;                  // grid of output positions of this class
    int out_h0, out_w0;          // output coordinate = g * out_s + out_0
    int T;                       // number of taps
    int dh[MAX_TAPS], dw[MAX_TAPS], wofs[MAX_TAPS];  // tap offsets and weight element offset of the tap inside a row
    int aofs[MAX_TAPS];          // element offset added to the gathered tensor's address for this tap: 0, or the distance to a
                                 //   SECOND gathered tensor of the same shape in the same allocation (ConvP::a_extra covers it)
};

struct ConvP {
    const uint16_t* x;
    const uint16_t* w;
    uint16_t* y;
    int N, Hi, Wi, Ci;           // input tensor [N, Hi, Wi, Ci]
    int Ho, Wo, Co;              // output tensor [N, Ho, Wo, Co]
    int in_s;                    // input coordinate = g * in_s + d
    int out_s;
    int w_row_stride;            // elements between consecutive weight rows (output channels of this GEMM)
    int n_classes;
    int max_pad;                 // largest |tap displacement| in pixels (1; the dilation for an atrous 3x3): sizes the DMA descriptor's bias
    float* stats;                // optional [2][Co][G] per-tile column sums, G = gridDim.y * gridDim.z; meaning by `bnx`:
    const float* shift;          //   bnx == NULL: sum (y - shift), sum (y - shift)^2      (moments for a following BN forward)
    const uint16_t* bnx;         //   bnx != NULL: y is d(loss)/d(BN output); bnx = that BN's INPUT (same shape as y):
    const float* bn_stats;       //     g = relu-masked y, sums g and g*(bnx - mean) from bn_stats[4][Co] = mean|invstd|alpha|beta
    int bn_relu;                 //     (the reduction pass of that BN's backward, fused here)
    const uint16_t* bny;         //     optional: that BN layer's OUTPUT after residual add + ReLU — the mask is (bny > 0)
                                 //     instead of the recomputed bnx*alpha+beta > 0 (a BN whose ReLU follows a residual add)
    double* acc;                 // alternative to `stats`: the same column sums added into f64 accumulators [NS][2][Co]
    int acc_ns;                  //   with native atomics, copy = row tile % NS (+ [Co] floats after them: the shift used)
    int groups;                  // 1, or 2: the batch is two concatenated half-batches with SEPARATE BatchNorm statistics
    int acc_stride;              //   (adv | clean): images >= N/2 use acc + acc_stride (doubles) and bn_stats + 4*Co
    const uint16_t* addend;      // optional tensor of y's shape added to y before it is stored (residual-gradient sum)
    const float* aff;            // optional [4][Co] coefficient block (mean | invstd | alpha | beta) of a FROZEN BatchNorm behind this
    const uint16_t* aff_res;     //   convolution: y = [relu](bf16(conv) * alpha + beta [+ aff_res]) — the arithmetic, rounding points
    int aff_relu;                //   included, of the convolution launch followed by afan_affine_apply (Detection's bottlenecks)
    int aff_bwd;                 // 1: the BACKWARD of such a layer applied to an input gradient on its way out: aff = that layer's alpha
                                 //   row [Co], aff_res = its stored OUTPUT: y = bf16((aff_res > 0 ? bf16(dgrad) : 0) * alpha) —
                                 //   afan_affine_relu_bwd's expression on the dgrad launch's rounded result
                                 // 2: the same at a residual block's OUTPUT, where the gradient splits: g = bf16(bf16(dgrad) + addend)
                                 //   (addend optional), m = aff_res > 0 ? g : 0; y2 = m (the shortcut's share), y = bf16(m * alpha) (the
                                 //   last convolution's) — afan_affine_relu_bwd with both outputs, on the dgrad-with-addend launch's result
    uint16_t* y2;                //   (aff_bwd == 2 only)
    int64_t a_extra;             // bytes beyond the gathered tensor x that taps with aofs != 0 may reach (descriptor range)
    int multi;                   // 1: the classes are INDEPENDENT forward problems on the same input (ASPP's atrous branches,
    int64_t w_off[4];            //   _deeplab.py:173-176): class z reads weights w + w_off[z], writes y + y_off[z] (elements),
    int64_t y_off[4];            //   sums into acc + acc_off[z] (doubles) around shift + shift_off[z] (floats).  All zero
    int64_t acc_off[4];          //   otherwise.
    int64_t shift_off[4];
    int w_rs[4];                 //   and its weight rows are w_rs[z] elements apart (problems may differ in kernel size)
    // filled by the launcher (launch_gs), not by callers: the epilogue's three per-channel coefficient rows (see coef_s in
    // afan_conv.hip) as ready addresses — slot k reads coef[k][channel (+ coef_gofs for the second image group)]; an unused slot points at
    // readable memory and is masked (coef_mask bit k); bit 3: slot 0 is the BatchNorm shift, displaced by shift_off[class]
    const float* coef[3];
    int coef_mask;
    int coef_gofs;
    ConvClass cls[4];
    // ---- round 5: the BatchNorm behind (forward) / in front of (backward) this convolution applied INSIDE the launch.  Batch
    // statistics need every tile of the launch, so the epilogue meets all other workgroups at a grid-wide barrier between its
    // sums and its stores: only for launches whose workgroups are all resident (dispatch_bnf checks), `acc` sums only.
    int bnf;                     // 0: off
                                 // 1 (forward):  y = the raw convolution output as always; after the barrier y2 = [relu](alpha*y + beta
                                 //    [+ bnf_res] [+ alpha_s*bnf_res + beta_s: the projection shortcut's BatchNorm, bnf_sc != NULL]) with the
                                 //    coefficients of apply_acc_kernel / apply_acc_dual_kernel, term for term
                                 // 2 (backward): y = the gradient entering the BatchNorm's INPUT (bwd_apply_acc_kernel's dx), y2 (optional) =
                                 //    the masked gradient (its dres); the unnormalised gradient is never written
    unsigned* bar;               // afan_grid_barrier_bytes() of zero-initialised device memory, reused by every launch of a stream
    const float* bnf_w;          // forward: BatchNorm weight / bias (NULL = 1 / 0), eps, momentum, running buffers, number of
    const float* bnf_b;          //   running-statistics updates this pass stands for, stats out [4][Co] = mean | invstd | alpha | beta
    float bnf_eps, bnf_mom;
    float* bnf_rmean;
    float* bnf_rvar;
    int64_t* bnf_nbt;
    int bnf_updates;
    float* bnf_stats;
    const uint16_t* bnf_res;     // forward: residual (y's shape) added after the affine, or the projection's raw output (bnf_sc)
    int bnf_relu;
    double bnf_inv_m;            // 1 / (N * Ho * Wo)
    float bnf_unbias;
    struct Sc {                  // forward, projection shortcut (Classification/resnet_s.py option B): its BatchNorm, sums complete in
        const double* acc;       //   acc (an EARLIER launch filled them, shift snapshot behind the slots)
        const float* w;
        const float* b;
        float eps, mom;
        float* rmean;
        float* rvar;
        int64_t* nbt;
        float* stats;
    } bnf_sc;
    float* bnf_dw;               // backward: BatchNorm weight / bias gradients [Co] (NULL: not wanted), accumulate flag
    float* bnf_db;
    int bnf_accum;
    struct BSc {                 // backward, block-output form (y2 = the masked gradient m): the producing block's PROJECTION shortcut's
        const uint16_t* x;       //   BatchNorm (no ReLU) receives m as well — its whole backward here too: x = that BatchNorm's input (the
        const float* stats;      //   projection's raw output, y's shape; staged in LDS by DMA), stats its [4][Co] block, acc a second zeroed
        double* acc;             //   accumulator block (sum m, sum m*(x - mean)), y3 = the gradient entering its input
        uint16_t* y3;            //   (bwd_apply_acc_kernel<RELU = false>'s expression), dw / db its parameter gradients
        float* dw;
        float* db;
    } bsc;
};

// afan_conv_bnf.hip: the tiled kernel's instantiations with the in-launch BatchNorm (ConvP::bnf != 0); AFAN_ESHAPE when no such
// instantiation takes the problem or its workgroups would not all be resident (nothing is launched then)
int dispatch_bnf(const ConvP& p, hipStream_t st, bool dgrad);
// 1: kernels of ANOTHER stream may run beside these launches (afan_grid_barrier_shared_gpu): a launch counts as resident only at
// one workgroup per CU.  Two per CU need two contiguous LDS ranges: beside a long-lived workgroup of another kernel the first
// lands between its range and the end, that kernel leaves, and the two free pieces either side never again hold the second — which
// the first, spinning at the barrier, waits for (seen: DeepLab at 8 images with weight gradients on the side stream, 276 workgroups
// of the two-stage tile beside a 288-workgroup weight-gradient launch: 200 ms, the spin bound)
extern int g_bnf_one_per_cu;


// small-channel kernel (afan_conv_small.hip): reduction channels in {16, 32, 64}, output channels a multiple of 16, at
// least one of the two below 64
bool small_eligible(const ConvP& p);
int small_launch(const ConvP& p, hipStream_t st);

}  // namespace afan_conv
