/*
 * afan_oracle.c — plain-C CPU ORACLE for the element-wise / reduction kernels of the A-FAN hot path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle (built into oracle/_ref/ by oracle/Makefile).
 *
 * Each function restates the reference behaviour cited next to it (paths relative to the reference
 * root).  Parity status: PINNED through tests/test_oracle_golden.py, which checks these functions
 * against tests/golden/*.npz (vectors produced by the reference's own Python via oracle/gen_golden.py).
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: every op rounds on its own, like ATen's
 * element-wise CPU kernels for these expressions).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static float sign_f(float g) { return (g != g) ? g : (float)((g > 0.f) - (g < 0.f)); }

/* Classification/attack_algo.py:53 (+ :55-56 -> :35-36 -> :9-19 when clip) */
void oracle_pgd_step(float* x_adv, const float* grad, const float* x_clean, int64_t n, float gamma,
                     float eps, int clip) {
    for (int64_t i = 0; i < n; ++i) {
        float d = gamma * sign_f(grad[i]);
        float t = x_adv[i] + d;
        if (clip) {
            float lo = x_clean[i] - eps, hi = x_clean[i] + eps;
            if (t < lo) t = lo;
            if (t > hi) t = hi;
        }
        x_adv[i] = t;
    }
}

/* Classification/attack_algo.py:42-44 with the host-drawn uniform u */
void oracle_axpy_noise(float* x_adv, const float* u, int64_t n, float eps) {
    for (int64_t i = 0; i < n; ++i) {
        float t = 2.0f * u[i];
        t = t - 1.0f;
        t = t * eps;
        x_adv[i] = x_adv[i] + t;
    }
}

/* Classification/main_perturb.py:188-192 (double accumulation: a tighter reference than fp32 torch.norm) */
void oracle_perturb_norms(const float* x_adv, const float* x_clean, int64_t batch, int64_t per_sample,
                          float* l2, float* linf) {
    for (int64_t b = 0; b < batch; ++b) {
        double ss = 0.0;
        float mx = 0.f;
        int nan = 0;
        for (int64_t j = 0; j < per_sample; ++j) {
            float d = x_adv[b * per_sample + j] - x_clean[b * per_sample + j];
            ss += (double)d * (double)d;
            float a = fabsf(d);
            if (a != a) nan = 1;
            if (a > mx) mx = a;
        }
        l2[b] = (float)sqrt(ss);
        linf[b] = nan ? NAN : mx;
    }
}

/* Segmentation/attack_algo.py:121-130: statistics over the channel dim per pixel, unbiased variance */
void oracle_mix_feature(const float* clean, const float* adv, float* out, int64_t n, int64_t c, int64_t hw,
                        float eps) {
    for (int64_t b = 0; b < n; ++b)
        for (int64_t p = 0; p < hw; ++p) {
            const float* pc = clean + b * c * hw + p;
            const float* pa = adv + b * c * hw + p;
            double mc = 0, ma = 0;
            for (int64_t k = 0; k < c; ++k) { mc += pc[k * hw]; ma += pa[k * hw]; }
            mc /= (double)c; ma /= (double)c;
            double vc = 0, va = 0;
            for (int64_t k = 0; k < c; ++k) {
                double dc = pc[k * hw] - mc, da = pa[k * hw] - ma;
                vc += dc * dc; va += da * da;
            }
            vc /= (double)(c - 1); va /= (double)(c - 1);
            float fmc = (float)mc, fma_ = (float)ma;
            float sc = sqrtf((float)vc + eps), sa = sqrtf((float)va + eps);
            for (int64_t k = 0; k < c; ++k) {
                float t = (pc[k * hw] - fmc) / sc;
                t = t * sa;
                t = t + fma_;
                out[b * c * hw + k * hw + p] = t;
            }
        }
}

/* Segmentation/attack_algo.py:108-118 with ATen's lerp: |w|<0.5 ? x + w*(y-x) : y - (y-x)*(1-w), as one
 * fused multiply-add fma(coeff, y-x, base) — what ATen's vectorised CPU kernel executes. */
void oracle_lerp_points(const float* x, const float* y, float* out, int64_t n, const float* w, int k_int) {
    for (int k = 0; k < k_int; ++k) {
        int small = fabsf(w[k]) < 0.5f;
        float coeff = small ? w[k] : w[k] - 1.0f;
        for (int64_t i = 0; i < n; ++i)
            out[(int64_t)k * n + i] = fmaf(coeff, y[i] - x[i], small ? x[i] : y[i]);
    }
}

/* nn.BatchNorm2d training forward (+ residual, + ReLU) as the backbone applies it (resnet_s.py:72-77),
 * double accumulation.  Updates running stats with the unbiased variance (momentum m). */
void oracle_bn_train_forward(const float* x, const float* res, float* y, int64_t n, int64_t c, int64_t hw,
                             float eps, float momentum, const float* weight, const float* bias, int relu,
                             float* save_mean, float* save_invstd, float* rmean, float* rvar) {
    const double M = (double)n * (double)hw;
    for (int64_t k = 0; k < c; ++k) {
        double s = 0;
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) s += x[(b * c + k) * hw + p];
        const double mu = s / M;
        double v = 0;
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) { double d = x[(b * c + k) * hw + p] - mu; v += d * d; }
        const double var_b = v / M;
        const float is = (float)(1.0 / sqrt(var_b + (double)eps));
        save_mean[k] = (float)mu;
        save_invstd[k] = is;
        if (rmean) {
            rmean[k] = (1.0f - momentum) * rmean[k] + momentum * (float)mu;
            rvar[k] = (1.0f - momentum) * rvar[k] + momentum * (float)(v / (M - 1.0));
        }
        const float w = weight ? weight[k] : 1.f, bb = bias ? bias[k] : 0.f;
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) {
                int64_t i = (b * c + k) * hw + p;
                float t = (float)(((double)x[i] - mu) * (double)is * (double)w + (double)bb);
                if (res) t += res[i];
                if (relu && !(t != t)) t = t > 0.f ? t : 0.f;
                y[i] = t;
            }
    }
}

/* autograd of the op above: g = dy*(y>0); dx = w*invstd*(g - mean(g) - xhat*mean(g*xhat)) */
void oracle_bn_backward(const float* dy, const float* x, const float* y, float* dx, float* dres, int64_t n,
                        int64_t c, int64_t hw, const float* mean, const float* invstd, const float* weight,
                        int relu, float* dweight, float* dbias) {
    const double M = (double)n * (double)hw;
    for (int64_t k = 0; k < c; ++k) {
        double sg = 0, sgx = 0;
        const double mu = mean[k], is = invstd[k];
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) {
                int64_t i = (b * c + k) * hw + p;
                double g = (relu && !(y[i] > 0.f)) ? 0.0 : dy[i];
                sg += g;
                sgx += g * (x[i] - mu) * is;
            }
        if (dweight) dweight[k] = (float)sgx;
        if (dbias) dbias[k] = (float)sg;
        const double w = weight ? weight[k] : 1.0;
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) {
                int64_t i = (b * c + k) * hw + p;
                double g = (relu && !(y[i] > 0.f)) ? 0.0 : dy[i];
                double xh = (x[i] - mu) * is;
                dx[i] = (float)((g - sg / M - xh * sgx / M) * w * is);
                if (dres) dres[i] = (float)g;
            }
    }
}

/* torch.optim.SGD step as configured at Classification/main_perturb.py:72-74 */
void oracle_sgd_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float wd,
                     float gscale) {
    for (int64_t i = 0; i < n; ++i) {
        float gg = g[i] * gscale;
        gg = gg + wd * p[i];
        m[i] = m[i] * momentum + gg;
        p[i] = p[i] - lr * m[i];
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Detection operators.  NMS is pinned to the reference's own golden (Detection/test/nms/
 * nms-large-{input,output}.npy, copied as data into tests/golden/) and to the three small cases of
 * Detection/test/nms/test_nms.py:21-37.  ROIAlign: the reference holds no vector, and the at::Tensor wrapper of its CPU
 * source does not compile against this image's torch — but the kernel templates under it
 * (Detection/support/src/cpu/ROIAlign_cpu.cpp:4-219) are plain C++: oracle/Makefile compiles them unedited and
 * oracle/gen_golden.py stores their outputs (tests/golden/roi_align_fwd_*.npz).  PINNED: the forward below equals those
 * vectors bit for bit in fp32 and f64 (tests/test_det_oracle.py); the backward (no CPU reference, ROIAlign.h:44) is the
 * exact adjoint of that forward, checked in f64 and as a full transposed Jacobian.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Greedy NMS over boxes visited in `order` (descending score).  inclusive = 1: suppress at IoU >= thresh
 * (Detection/support/src/cpu/nms_cpu.cpp:36-66); inclusive = 0: IoU > thresh (Detection/support/src/cuda/nms.cu:49).  Areas and
 * intersections with +1 (nms_cpu.cpp:22,56-57; nms.cu:14-21).  keep_out: kept ORIGINAL indices, ascending; returns the count. */
int64_t oracle_nms(const float* boxes, const int64_t* order, int64_t n, float thresh, int inclusive, int64_t* keep_out,
                   unsigned char* suppressed /* scratch, n bytes */) {
    for (int64_t i = 0; i < n; ++i) suppressed[i] = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        const int64_t i = order[_i];
        if (suppressed[i]) continue;
        const float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        const float iarea = (ix2 - ix1 + 1) * (iy2 - iy1 + 1);
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            const int64_t j = order[_j];
            if (suppressed[j]) continue;
            const float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
            const float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
            const float w = fmaxf(0.f, xx2 - xx1 + 1), h = fmaxf(0.f, yy2 - yy1 + 1);
            const float inter = w * h;
            const float jarea = (boxes[4 * j + 2] - boxes[4 * j] + 1) * (boxes[4 * j + 3] - boxes[4 * j + 1] + 1);
            const float ovr = inter / (iarea + jarea - inter);
            if (inclusive ? (ovr >= thresh) : (ovr > thresh)) suppressed[j] = 1;
        }
    }
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i)
        if (!suppressed[i]) keep_out[k++] = i;
    return k;
}

/* ROIAlign, restated once and instantiated for float (what the reference's GPU path and its CPU path compute in) and
 * double (the adjoint identity <bwd(dy), x> == <dy, fwd(x)> is checked in f64, tests/test_det_oracle.py).
 * PIN: the forward is held bit for bit to the reference's own CPU kernel (Detection/support/src/cpu/ROIAlign_cpu.cpp:4-219,
 * compiled unedited by oracle/Makefile -> _ref/libref_roialign.so at fixture time: tests/golden/roi_align_fwd_*.npz); the
 * backward has no CPU reference (Detection/support/src/ROIAlign.h:44) and is pinned as the exact adjoint of that linear
 * forward (same samples, same weights, ROIAlign_cuda.cu:125-170,178-254). */
#define ORACLE_ROI_ALIGN(SUF, T, CEIL, FMAX)                                                                                      \
typedef struct { int yl, xl, yh, xh; T w1, w2, w3, w4; int empty; } oracle_bilin##SUF;                                            \
/* ROIAlign_cuda.cu:16-62 / :125-170 == ROIAlign_cpu.cpp:45-103 */                                                                \
static oracle_bilin##SUF oracle_bilin_prep##SUF(int height, int width, T y, T x) {                                                \
    oracle_bilin##SUF b;                                                                                                          \
    b.empty = (y < (T)-1.0 || y > (T)height || x < (T)-1.0 || x > (T)width);                                                      \
    b.yl = b.xl = b.yh = b.xh = -1; b.w1 = b.w2 = b.w3 = b.w4 = (T)0;                                                             \
    if (b.empty) return b;                                                                                                        \
    if (y <= 0) y = 0;                                                                                                            \
    if (x <= 0) x = 0;                                                                                                            \
    b.yl = (int)y; b.xl = (int)x;                                                                                                 \
    if (b.yl >= height - 1) { b.yh = b.yl = height - 1; y = (T)b.yl; } else b.yh = b.yl + 1;                                      \
    if (b.xl >= width - 1) { b.xh = b.xl = width - 1; x = (T)b.xl; } else b.xh = b.xl + 1;                                        \
    const T ly = y - b.yl, lx = x - b.xl, hy = (T)1 - ly, hx = (T)1 - lx;                                                         \
    b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;                                                               \
    return b;                                                                                                                     \
}                                                                                                                                 \
/* mode 0: forward (ROIAlign_cuda.cu:65-122 == ROIAlign_cpu.cpp:110-219): y[num_rois,C,PH,PW] from x[N,C,H,W] (NCHW);            \
 * mode 1: backward (:178-254): x is the zero-initialised gradient map that dy = y scatters into, in index order. */              \
void oracle_roi_align##SUF(T* x, const T* rois, T* y, int64_t num_rois, int64_t C, int64_t H, int64_t W, int PH, int PW,          \
                           T scale, int sampling_ratio, int mode) {                                                               \
    for (int64_t n = 0; n < num_rois; ++n) {                                                                                      \
        const T* r = rois + n * 5;                                                                                                \
        const int b = (int)r[0];                                                                                                  \
        const T sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale;                                       \
        const T rw = FMAX(ew - sw, (T)1), rh = FMAX(eh - sh, (T)1);                                                               \
        const T bh = rh / (T)PH, bw = rw / (T)PW;                                                                                 \
        const int gh = sampling_ratio > 0 ? sampling_ratio : (int)CEIL(rh / (T)PH);                                               \
        const int gw = sampling_ratio > 0 ? sampling_ratio : (int)CEIL(rw / (T)PW);                                               \
        const T count = (T)(gh * gw);                                                                                             \
        for (int64_t c = 0; c < C; ++c) {                                                                                         \
            T* plane = x + ((int64_t)b * C + c) * H * W;                                                                          \
            for (int ph = 0; ph < PH; ++ph)                                                                                       \
                for (int pw = 0; pw < PW; ++pw) {                                                                                 \
                    T* out = y + ((n * C + c) * PH + ph) * PW + pw;                                                               \
                    T acc = (T)0;                                                                                                 \
                    for (int iy = 0; iy < gh; ++iy) {                                                                             \
                        const T yy = sh + ph * bh + (T)((float)iy + .5f) * bh / (T)gh;                                            \
                        for (int ix = 0; ix < gw; ++ix) {                                                                         \
                            const T xx = sw + pw * bw + (T)((float)ix + .5f) * bw / (T)gw;                                        \
                            const oracle_bilin##SUF q = oracle_bilin_prep##SUF((int)H, (int)W, yy, xx);                           \
                            if (q.empty) continue;                                                                                \
                            if (mode == 0) {                                                                                      \
                                acc += q.w1 * plane[q.yl * W + q.xl] + q.w2 * plane[q.yl * W + q.xh] +                            \
                                       q.w3 * plane[q.yh * W + q.xl] + q.w4 * plane[q.yh * W + q.xh];                             \
                            } else {                                                                                              \
                                const T g = *out;                                                                                 \
                                plane[q.yl * W + q.xl] += g * q.w1 / count; plane[q.yl * W + q.xh] += g * q.w2 / count;           \
                                plane[q.yh * W + q.xl] += g * q.w3 / count; plane[q.yh * W + q.xh] += g * q.w4 / count;           \
                            }                                                                                                     \
                        }                                                                                                         \
                    }                                                                                                             \
                    if (mode == 0) *out = acc / count;                                                                            \
                }                                                                                                                 \
        }                                                                                                                         \
    }                                                                                                                             \
}
ORACLE_ROI_ALIGN(, float, ceilf, fmaxf)
ORACLE_ROI_ALIGN(_f64, double, ceil, fmax)
