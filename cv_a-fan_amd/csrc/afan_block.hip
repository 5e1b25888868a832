// Host-side sequencing of a frozen-BatchNorm bottleneck (Detection/backbone/resnet101_ori.py:78-127 with the BatchNorms of
// Detection/model.py:27-35,46-47): ONE C call issues the block's launches — three (four) tuned convolutions and three (four)
// fused affine(+residual)(+ReLU) launches forward; per layer an affine backward, an input gradient and a weight gradient
// backward.  No new kernel: the eager Detection iteration (proposal counts change shape every forward: no hipGraph) is bound
// by Python dispatch — seven ctypes calls of ~20 us each per block and pass, 390 block passes per iteration — and this is
// the runtime answering that in native code.
#include "afan_common.h"
#include <stdlib.h>
#include "../../include/afan_hip.h"

using namespace afan;

// AFAN_BLOCK_FUSE_AFFINE=0: every frozen BatchNorm as its own launch again (A/B of the epilogue forms)
static bool fuse_affine() {
    static const int on = [] { const char* e = getenv("AFAN_BLOCK_FUSE_AFFINE"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

extern "C" {

// x [n, cin, h, w] -> out [n, 4 * planes, ho, wo] (bf16 channels-last; ho = (h - 1) / stride + 1: the 3x3 carries the stride).
// w1 [planes, cin, 1, 1], w2 [planes, planes, 3, 3], w3 [4 planes, planes, 1, 1], wd [4 planes, cin, 1, 1] or NULL (identity
// shortcut: cin == 4 planes, stride 1) — KRSC bf16; k1 .. kd: afan_affine_coefs blocks.  a1 [n, planes, h, w] and
// a2 [n, planes, ho, wo] are kept for the backward; scratch: bf16 elements for the raw convolution outputs,
// n * (planes * h * w + 2 * 4 planes * ho * wo) of them.
int afan_frozen_bottleneck_fwd(const void* x, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t planes, int stride,
                               const void* w1, const void* w2, const void* w3, const void* wd, const float* k1, const float* k2,
                               const float* k3, const float* kd, void* scratch, void* a1, void* a2, void* out,
                               afan_stream_t stream) {
    if (!x || !w1 || !w2 || !w3 || !k1 || !k2 || !k3 || !scratch || !a1 || !a2 || !out) return AFAN_ENULL;
    if (wd && !kd) return AFAN_ENULL;
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || planes <= 0 || !(stride == 1 || stride == 2)) return AFAN_ESHAPE;
    if (!wd && (cin != 4 * planes || stride != 1)) return AFAN_ESHAPE;
    const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1, co = 4 * planes;
    uint16_t* r1 = (uint16_t*)scratch;                       // [n, planes, h, w] (then [n, planes, ho, wo])
    uint16_t* r3 = r1 + n * planes * h * w;                  // [n, co, ho, wo]
    uint16_t* rd = r3 + n * co * ho * wo;                    // [n, co, ho, wo]: the projection branch, raw then normalised in place
    int e;
    // a convolution and its frozen BatchNorm (+ residual) (+ ReLU) as ONE launch where the tiled kernel takes the shape
    // (afan_conv_fwd_affine_nhwc_bf16: the same bits), else the two launches
    auto conv_bn = [&](const void* in, const void* wt, void* raw, void* dst, int64_t hi_, int64_t wi_, int64_t ci_, int64_t co_, int k, int st_,
                       const float* kc, const void* res, int relu) -> int {
        int rc = fuse_affine() ? afan_conv_fwd_affine_nhwc_bf16(in, wt, dst, n, hi_, wi_, ci_, co_, k, st_, kc, res, relu, stream) : AFAN_ESHAPE;
        if (rc != AFAN_ESHAPE) return rc;
        if ((rc = afan_conv_fwd_nhwc_bf16(in, wt, raw, n, hi_, wi_, ci_, co_, k, st_, 1, nullptr, nullptr, nullptr, 1, stream))) return rc;
        const int64_t ho_ = (hi_ - 1) / st_ + 1, wo_ = (wi_ - 1) / st_ + 1;
        return afan_affine_apply(raw, res, dst, AFAN_BF16, n, co_, ho_ * wo_, kc, relu, stream);
    };
    if ((e = conv_bn(x, w1, r1, a1, h, w, cin, planes, 1, 1, k1, nullptr, 1))) return e;
    if ((e = conv_bn(a1, w2, r1, a2, h, w, planes, planes, 3, stride, k2, nullptr, 1))) return e;
    const void* res = x;
    if (wd) {
        if ((e = conv_bn(x, wd, rd, rd, h, w, cin, co, 1, stride, kd, nullptr, 0))) return e;      // (raw == dst: the apply is elementwise)
        res = rd;
    }
    return conv_bn(a2, w3, r3, out, ho, wo, planes, co, 1, 1, k3, res, 1);
}

// bf16 elements of scratch the backward needs: two gradient buffers of the larger shapes
int64_t afan_frozen_bottleneck_bwd_scratch(int64_t n, int64_t h, int64_t w, int64_t cin, int64_t planes, int stride) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || planes <= 0 || !(stride == 1 || stride == 2)) return 0;
    const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1, co = 4 * planes;
    // d3, dres (dd in place), t (layer gradients before their affine backward), d2, d1, dxs
    return n * (2 * co * ho * wo + 2 * planes * h * w + 2 * planes * ho * wo + cin * h * w);
}

// g = d(loss)/d(out).  wt1 .. wtd: the transposed (CRSK) weights; al1 .. ald: the alpha rows of the coefficient blocks;
// gw1 .. gwd: fp32 KRSC gradient tensors the weight gradients are ADDED into (NULL: that layer's weight gradient is not
// wanted — frozen layers, input-gradient-only passes); wgrad_ws: the sum of the wanted layers'
// afan_conv_wgrad_workspace_floats; dx (nullable) [n, cin, h, w].
int afan_frozen_bottleneck_bwd(const void* g, const void* x, const void* a1, const void* a2, const void* out, int64_t n,
                               int64_t h, int64_t w, int64_t cin, int64_t planes, int stride, const void* wt1, const void* wt2,
                               const void* wt3, const void* wtd, const float* al1, const float* al2, const float* al3,
                               const float* ald, float* gw1, float* gw2, float* gw3, float* gwd, float* wgrad_ws, void* scratch,
                               void* dx, afan_stream_t stream) {
    return afan_frozen_bottleneck_bwd_chain(g, nullptr, nullptr, x, a1, a2, out, n, h, w, cin, planes, stride, wt1, wt2, wt3, wtd, al1, al2, al3,
                                            ald, gw1, gw2, gw3, gwd, wgrad_ws, scratch, dx, nullptr, nullptr, nullptr, stream);
}

// The same inside a chain of blocks (a stage: block i + 1's input IS block i's output).  Entering: g as above, or — g NULL — the
// first step of this block's backward already done by the block behind it: pre_d3 = bf16(m * al3), pre_dres = m with m = the
// output's ReLU mask applied to the gradient (pre_dres may be overwritten).  Leaving: dx as above, or — dx NULL, prev_al3 given —
// that first step of the block IN FRONT done here, in the epilogue of this block's last input-gradient launch
// (afan_conv_dgrad_dual_nhwc_bf16; where that kernel does not take the shape: the gradient into prev_dres, then the separate
// launch): prev_d3 / prev_dres [n, cin, h, w] with prev_al3 = the block in front's last alpha row, its stored output = x.
// One launch per block and pass less (the Detection iteration: ~410 of 4 800).  Same bits either way.
int afan_frozen_bottleneck_bwd_chain(const void* g, void* pre_d3, void* pre_dres, const void* x, const void* a1, const void* a2,
                                     const void* out, int64_t n, int64_t h, int64_t w, int64_t cin, int64_t planes, int stride,
                                     const void* wt1, const void* wt2, const void* wt3, const void* wtd, const float* al1, const float* al2,
                                     const float* al3, const float* ald, float* gw1, float* gw2, float* gw3, float* gwd, float* wgrad_ws,
                                     void* scratch, void* dx, const float* prev_al3, void* prev_d3, void* prev_dres, afan_stream_t stream) {
    if (!x || !a1 || !a2 || !out || !wt1 || !wt2 || !wt3 || !al1 || !al2 || !al3 || !scratch) return AFAN_ENULL;
    if (!g && (!pre_d3 || !pre_dres)) return AFAN_ENULL;
    if (prev_al3 && (dx || !prev_d3 || !prev_dres)) return AFAN_ENULL;
    if (wtd && !ald) return AFAN_ENULL;
    if ((gw1 || gw2 || gw3 || gwd) && !wgrad_ws) return AFAN_ENULL;
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || planes <= 0 || !(stride == 1 || stride == 2)) return AFAN_ESHAPE;
    if (!wtd && (cin != 4 * planes || stride != 1)) return AFAN_ESHAPE;
    const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1, co = 4 * planes;
    uint16_t* d3 = (uint16_t*)scratch;                       // [n, co, ho, wo]
    uint16_t* dres = d3 + n * co * ho * wo;                  // [n, co, ho, wo]
    uint16_t* t2 = dres + n * co * ho * wo;                  // [n, planes, ho, wo]
    if (!g) { d3 = (uint16_t*)pre_d3; dres = (uint16_t*)pre_dres; }
    uint16_t* d2 = t2 + n * planes * ho * wo;                // [n, planes, ho, wo]
    uint16_t* t1 = d2 + n * planes * ho * wo;                // [n, planes, h, w]
    uint16_t* d1 = t1 + n * planes * h * w;                  // [n, planes, h, w]
    uint16_t* dxs = d1 + n * planes * h * w;                 // [n, cin, h, w]
    int e;
    if (g && (e = afan_affine_relu_bwd(g, out, al3, d3, dres, AFAN_BF16, AFAN_NHWC, n, co, ho * wo, 1, stream))) return e;
    // an input gradient and the backward of the frozen BatchNorm + ReLU it runs into as ONE launch where the tiled kernel takes the
    // shape (afan_conv_dgrad_affine_nhwc_bf16: the same bits), else the two launches
    auto dgrad_bn = [&](const void* dyp, const void* wt, void* raw, void* dst, int64_t hi_, int64_t wi_, int64_t ci_, int64_t co_, int k, int st_,
                        const float* al, const void* act) -> int {
        int rc = fuse_affine() ? afan_conv_dgrad_affine_nhwc_bf16(dyp, wt, dst, n, hi_, wi_, ci_, co_, k, st_, al, act, stream) : AFAN_ESHAPE;
        if (rc != AFAN_ESHAPE) return rc;
        if ((rc = afan_conv_dgrad_nhwc_bf16(dyp, wt, raw, n, hi_, wi_, ci_, co_, k, st_, 1, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                            nullptr, 1, stream))) return rc;
        return afan_affine_relu_bwd(raw, act, al, dst, nullptr, AFAN_BF16, AFAN_NHWC, n, ci_, hi_ * wi_, 1, stream);
    };
    if ((e = dgrad_bn(d3, wt3, t2, d2, ho, wo, planes, co, 1, 1, al2, a2))) return e;
    if ((e = dgrad_bn(d2, wt2, t1, d1, h, w, planes, planes, 3, stride, al1, a1))) return e;
    const void* addend = dres;
    const bool want_dx = dx || prev_al3;
    if (wtd) {
        if ((e = afan_affine_relu_bwd(dres, nullptr, ald, dres, nullptr, AFAN_BF16, AFAN_NHWC, n, co, ho * wo, 0, stream))) return e;   // dd, in place
        if (want_dx) {
            if ((e = afan_conv_dgrad_nhwc_bf16(dres, wtd, dxs, n, h, w, cin, co, 1, stride, 1, nullptr, nullptr, nullptr, 0, nullptr,
                                               nullptr, nullptr, 1, stream))) return e;
            addend = dxs;
        }
    }
    if (dx && (e = afan_conv_dgrad_nhwc_bf16(d1, wt1, dx, n, h, w, cin, planes, 1, 1, 1, addend, nullptr, nullptr, 0, nullptr, nullptr,
                                             nullptr, 1, stream))) return e;
    if (prev_al3) {
        e = fuse_affine() ? afan_conv_dgrad_dual_nhwc_bf16(d1, wt1, prev_d3, prev_dres, n, h, w, cin, planes, 1, 1, addend, prev_al3, x, stream) : AFAN_ESHAPE;
        if (e == AFAN_ESHAPE) {
            if ((e = afan_conv_dgrad_nhwc_bf16(d1, wt1, prev_dres, n, h, w, cin, planes, 1, 1, 1, addend, nullptr, nullptr, 0, nullptr, nullptr,
                                               nullptr, 1, stream))) return e;
            e = afan_affine_relu_bwd(prev_dres, x, prev_al3, prev_d3, prev_dres, AFAN_BF16, AFAN_NHWC, n, cin, h * w, 1, stream);
        }
        if (e) return e;
    }
    // weight gradients: one multi launch when the tuned kernel would tile the wanted problems alike, else one by one
    const void* xs[4]; const void* dys[4]; float* gws[4];
    int64_t ns[4], hs[4], ws_[4], cis[4], cos[4];
    int ks[4], sts[4], dils[4], codes[4], m = 0;
    auto add = [&](float* gw, const void* xin, const void* dy, int64_t hh, int64_t ww, int64_t ci, int64_t co_, int k, int st) {
        if (!gw) return;
        xs[m] = xin; dys[m] = dy; gws[m] = gw; ns[m] = n; hs[m] = hh; ws_[m] = ww; cis[m] = ci; cos[m] = co_; ks[m] = k; sts[m] = st;
        dils[m] = 1;
        codes[m] = afan_conv_wgrad_plan(n, hh, ww, ci, co_, k, st);
        ++m;
    };
    add(gw3, a2, d3, ho, wo, planes, co, 1, 1);
    add(gw2, a1, d2, h, w, planes, planes, 3, stride);
    add(gw1, x, d1, h, w, cin, planes, 1, 1);
    if (wtd) add(gwd, x, dres, h, w, cin, co, 1, stride);
    bool same = m >= 2;
    for (int i = 0; i < m; ++i) same = same && codes[i] != 0 && codes[i] == codes[0];
    if (same) return afan_conv_wgrad_multi_nhwc_bf16(m, xs, dys, gws, ns, hs, ws_, cis, cos, ks, sts, dils, wgrad_ws, 1, stream);
    for (int i = 0; i < m; ++i)
        if ((e = afan_conv_wgrad_nhwc_bf16(xs[i], dys[i], gws[i], ns[i], hs[i], ws_[i], cis[i], cos[i], ks[i], sts[i], 1, wgrad_ws, 1,
                                           stream))) return e;
    return AFAN_OK;
}

}  // extern "C"
