// The non-convolution layers of the DeepLabv3+ split-forward network for gfx950:
//   * bilinear resize, align_corners=False    Segmentation/network/utils.py:30,45; _deeplab.py:54,66,75,141
//   * per-pixel cross-entropy, ignore_index   Segmentation/main_aug_final.py:95 (nn.CrossEntropyLoss(ignore_index=255))
//   * 3x3 / stride 2 / pad 1 max pooling      Segmentation/network/backbone/resnet.py:146
//   * global average pool                     _deeplab.py:133 (ASPPPooling)
//   * 1x1 convolution with bias to a few channels (the classifier, _deeplab.py:45) with fp32 logits
//   * dropout                                 _deeplab.py:185
// All are HBM-bound: one pass over the tensor they produce / consume, 16-byte accesses where the channel count allows,
// gather formulations for the backward passes (no float atomics: results are run-to-run reproducible).
#include "afan_common.h"

using namespace afan;

namespace {

constexpr int BLOCK = 256;

// ---- index math shared by forward and backward of the resize (ATen's area_pixel_compute_source_index, fp32) ----------
struct Src {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Src src_index(float scale, int dst, int in_size) {
    float s = fmaf(scale, (float)dst + 0.5f, -0.5f);    // (ATen's CPU kernels are built with FMA contraction: same rounding)
    if (s < 0.f) s = 0.f;
    Src r;
    r.i0 = (int)s;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = s - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// One thread = one output pixel x VEC consecutive channels (NHWC) or one output element (NCHW, VEC = 1).
template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int C, int Hi,
                                                             int Wi, int Ho, int Wo, float sh, float sw, int64_t total) {
    const int CV = C / VEC;
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLOCK) {
        int64_t t = v;
        int cv, ox, oy;
        int64_t n;
        if constexpr (NHWC) { cv = (int)(t % CV); t /= CV; ox = (int)(t % Wo); t /= Wo; oy = (int)(t % Ho); n = t / Ho; }
        else { ox = (int)(t % Wo); t /= Wo; oy = (int)(t % Ho); t /= Ho; cv = (int)(t % CV); n = t / CV; }
        const Src a = src_index(sh, oy, Hi), b = src_index(sw, ox, Wi);
        float o[VEC];
        if constexpr (NHWC) {
            const T* base = x + (n * Hi * (int64_t)Wi) * C + cv * VEC;
            float p00[VEC], p01[VEC], p10[VEC], p11[VEC];
            if constexpr (VEC == 1) {
                p00[0] = Elt<T>::ld(base + ((int64_t)a.i0 * Wi + b.i0) * C);
                p01[0] = Elt<T>::ld(base + ((int64_t)a.i0 * Wi + b.i1) * C);
                p10[0] = Elt<T>::ld(base + ((int64_t)a.i1 * Wi + b.i0) * C);
                p11[0] = Elt<T>::ld(base + ((int64_t)a.i1 * Wi + b.i1) * C);
            } else {
                Elt<T>::ldv(base + ((int64_t)a.i0 * Wi + b.i0) * C, reinterpret_cast<float(&)[Elt<T>::VEC]>(p00));
                Elt<T>::ldv(base + ((int64_t)a.i0 * Wi + b.i1) * C, reinterpret_cast<float(&)[Elt<T>::VEC]>(p01));
                Elt<T>::ldv(base + ((int64_t)a.i1 * Wi + b.i0) * C, reinterpret_cast<float(&)[Elt<T>::VEC]>(p10));
                Elt<T>::ldv(base + ((int64_t)a.i1 * Wi + b.i1) * C, reinterpret_cast<float(&)[Elt<T>::VEC]>(p11));
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k)     // ATen's association: h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11)
                o[k] = a.l0 * (b.l0 * p00[k] + b.l1 * p01[k]) + a.l1 * (b.l0 * p10[k] + b.l1 * p11[k]);
            T* dst = y + ((n * Ho + oy) * (int64_t)Wo + ox) * C + cv * VEC;
            if constexpr (VEC == 1) Elt<T>::st(dst, o[0]);
            else Elt<T>::stv(dst, reinterpret_cast<const float(&)[Elt<T>::VEC]>(o));
        } else {
            const T* base = x + (n * C + cv) * (int64_t)Hi * Wi;
            const float p00 = Elt<T>::ld(base + (int64_t)a.i0 * Wi + b.i0), p01 = Elt<T>::ld(base + (int64_t)a.i0 * Wi + b.i1);
            const float p10 = Elt<T>::ld(base + (int64_t)a.i1 * Wi + b.i0), p11 = Elt<T>::ld(base + (int64_t)a.i1 * Wi + b.i1);
            Elt<T>::st(y + v, a.l0 * (b.l0 * p00 + b.l1 * p01) + a.l1 * (b.l0 * p10 + b.l1 * p11));
        }
    }
}

// candidate output range of input index i along one axis (a superset: weights of non-contributors come out as 0)
__device__ __forceinline__ void out_range(float inv_scale, int i, int out_size, int& lo, int& hi) {
    float a = ((float)i - 0.5f) * inv_scale - 0.5f, b = ((float)i + 1.5f) * inv_scale - 0.5f;
    lo = (int)floorf(a) - 1;
    hi = (int)ceilf(b) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}
__device__ __forceinline__ float axis_weight(float scale, int o, int i, int in_size) {
    const Src s = src_index(scale, o, in_size);
    return (s.i0 == i ? s.l0 : 0.f) + (s.i1 == i ? s.l1 : 0.f);
}

// Backward as a gather: one thread = one INPUT pixel x VEC channels, summing the output gradients it fed, rows then
// columns in increasing order (deterministic).
template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void upsample_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int C, int Hi,
                                                             int Wi, int Ho, int Wo, float sh, float sw, float ish,
                                                             float isw, int64_t total) {
    const int CV = C / VEC;
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLOCK) {
        int64_t t = v;
        int cv, ix, iy;
        int64_t n;
        if constexpr (NHWC) { cv = (int)(t % CV); t /= CV; ix = (int)(t % Wi); t /= Wi; iy = (int)(t % Hi); n = t / Hi; }
        else { ix = (int)(t % Wi); t /= Wi; iy = (int)(t % Hi); t /= Hi; cv = (int)(t % CV); n = t / CV; }
        int ylo, yhi, xlo, xhi;
        out_range(ish, iy, Ho, ylo, yhi);
        out_range(isw, ix, Wo, xlo, xhi);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = axis_weight(sh, oy, iy, Hi);
            if (wy == 0.f) continue;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float wx = axis_weight(sw, ox, ix, Wi);
                if (wx == 0.f) continue;
                const float w = wy * wx;
                if constexpr (NHWC) {
                    const T* src = dy + ((n * Ho + oy) * (int64_t)Wo + ox) * C + cv * VEC;
                    if constexpr (VEC == 1) acc[0] += w * Elt<T>::ld(src);
                    else {
                        float g[VEC];
                        Elt<T>::ldv(src, reinterpret_cast<float(&)[Elt<T>::VEC]>(g));
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] += w * g[k];
                    }
                } else {
                    acc[0] += w * Elt<T>::ld(dy + ((n * C + cv) * (int64_t)Ho + oy) * Wo + ox);
                }
            }
        }
        if constexpr (NHWC) {
            T* dst = dx + ((n * Hi + iy) * (int64_t)Wi + ix) * C + cv * VEC;
            if constexpr (VEC == 1) Elt<T>::st(dst, acc[0]);
            else Elt<T>::stv(dst, reinterpret_cast<const float(&)[Elt<T>::VEC]>(acc));
        } else {
            Elt<T>::st(dx + v, acc[0]);
        }
    }
}

// ---- per-pixel cross entropy with ignore_index ------------------------------------------------------------------
constexpr int CE_MAX_C = 32;

__global__ __launch_bounds__(BLOCK) void ce2d_count_kernel(const int64_t* __restrict__ target, int64_t P, int64_t ignore,
                                                           float* __restrict__ partial) {
    float c = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x; p < P; p += (int64_t)gridDim.x * BLOCK)
        c += (target[p] != ignore) ? 1.f : 0.f;
    c = wave_sum(c);
    __shared__ float red[BLOCK / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ws: [0] = count (written by ce2d_fold_count_kernel), [1..1+G) count partials, [1+G..1+2G) loss partials
__global__ void ce2d_fold_count_kernel(float* ws, int G) {
    float c = 0.f;
    for (int i = 0; i < G; ++i) c += ws[1 + i];
    ws[0] = c;
}

template <bool NHWC>
__global__ __launch_bounds__(BLOCK) void ce2d_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     float* __restrict__ dlogits, int C, int64_t HW, int64_t P,
                                                     int64_t ignore, float grad_scale, float* __restrict__ ws, int G) {
    const float count = ws[0];
    const float gs = grad_scale / count;
    float loss = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x; p < P; p += (int64_t)gridDim.x * BLOCK) {
        const int64_t t = target[p];
        const int64_t n = p / HW, q = p - n * HW;
        const int64_t base = NHWC ? p * C : n * C * HW + q;
        const int64_t cs = NHWC ? 1 : HW;
        float l[CE_MAX_C];               // fully unrolled with a guard: stays in registers
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CE_MAX_C; ++c) {
            l[c] = c < C ? logits[base + c * cs] : -INFINITY;
            m = fmaxf(m, l[c]);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CE_MAX_C; ++c) {
            l[c] = c < C ? expf(l[c] - m) : 0.f;
            s += l[c];
        }
        const bool live = t != ignore;
        const bool bad = live && (t < 0 || t >= C);          // torch asserts here; poison the loss instead of reading out of range
        const float inv = 1.f / s;
        if (live && !bad) loss += logf(s) + m - logits[base + t * cs];
        if (bad) loss = NAN;
        if (dlogits) {
#pragma unroll
            for (int c = 0; c < CE_MAX_C; ++c) {
                float g = 0.f;
                if (live && !bad) g = (l[c] * inv - (c == (int)t ? 1.f : 0.f)) * gs;
                if (c < C) dlogits[base + c * cs] = g;
            }
        }
    }
    loss = wave_sum(loss);
    __shared__ float red[BLOCK / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) ws[1 + G + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void ce2d_finalize_kernel(const float* ws, int G, float* loss) {
    float s = 0.f;
    for (int i = 0; i < G; ++i) s += ws[1 + G + i];
    loss[0] = s / ws[0];
}

// ---- 3x3 / stride 2 / pad 1 max pooling --------------------------------------------------------------------------------
// first maximum in (h, w) scan order, NaN wins (ATen's CPU kernel: `val > maxval || isnan(val)`), so that the backward
// routes the gradient to the same element the reference's does — post-ReLU windows tie at 0 all the time.
template <typename T, int VEC, bool NHWC>
__device__ __forceinline__ void pool_window(const T* __restrict__ x, int64_t n, int cv, int C, int Hi, int Wi, int oy,
                                            int ox, float (&best)[VEC], int (&arg)[VEC]) {
    const int h0 = oy * 2 - 1, w0 = ox * 2 - 1;
    const int hs = h0 < 0 ? 0 : h0, ws = w0 < 0 ? 0 : w0;
    const int he = h0 + 3 > Hi ? Hi : h0 + 3, we = w0 + 3 > Wi ? Wi : w0 + 3;
#pragma unroll
    for (int k = 0; k < VEC; ++k) { best[k] = -INFINITY; arg[k] = hs * Wi + ws; }
    for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) {
            float v[VEC];
            if constexpr (NHWC) {
                const T* src = x + ((n * Hi + h) * (int64_t)Wi + w) * C + cv * VEC;
                if constexpr (VEC == 1) v[0] = Elt<T>::ld(src);
                else Elt<T>::ldv(src, reinterpret_cast<float(&)[Elt<T>::VEC]>(v));
            } else {
                v[0] = Elt<T>::ld(x + ((n * C + cv) * (int64_t)Hi + h) * Wi + w);
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; arg[k] = h * Wi + w; }
        }
}

template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int C, int Hi, int Wi,
                                                            int Ho, int Wo, int64_t total) {
    const int CV = C / VEC;
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLOCK) {
        int64_t t = v;
        int cv, ox, oy;
        int64_t n;
        if constexpr (NHWC) { cv = (int)(t % CV); t /= CV; ox = (int)(t % Wo); t /= Wo; oy = (int)(t % Ho); n = t / Ho; }
        else { ox = (int)(t % Wo); t /= Wo; oy = (int)(t % Ho); t /= Ho; cv = (int)(t % CV); n = t / CV; }
        float best[VEC];
        int arg[VEC];
        pool_window<T, VEC, NHWC>(x, n, cv, C, Hi, Wi, oy, ox, best, arg);
        if constexpr (NHWC) {
            T* dst = y + ((n * Ho + oy) * (int64_t)Wo + ox) * C + cv * VEC;
            if constexpr (VEC == 1) Elt<T>::st(dst, best[0]);
            else Elt<T>::stv(dst, reinterpret_cast<const float(&)[Elt<T>::VEC]>(best));
        } else {
            Elt<T>::st(y + v, best[0]);
        }
    }
}

// gather: one thread = one input pixel x VEC channels; the (at most 2 x 2) windows that contain it are re-scanned
template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void maxpool_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            T* __restrict__ dx, int C, int Hi, int Wi, int Ho, int Wo,
                                                            int64_t total) {
    const int CV = C / VEC;
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLOCK) {
        int64_t t = v;
        int cv, ix, iy;
        int64_t n;
        if constexpr (NHWC) { cv = (int)(t % CV); t /= CV; ix = (int)(t % Wi); t /= Wi; iy = (int)(t % Hi); n = t / Hi; }
        else { ix = (int)(t % Wi); t /= Wi; iy = (int)(t % Hi); t /= Hi; cv = (int)(t % CV); n = t / CV; }
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        // windows oy with 2*oy - 1 <= iy <= 2*oy + 1
        const int oy0 = iy / 2, oy1 = (iy + 1) / 2, ox0 = ix / 2, ox1 = (ix + 1) / 2;
        const int me = iy * Wi + ix;
        for (int oy = oy0; oy <= oy1; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                if (ox >= Wo) continue;
                float best[VEC];
                int arg[VEC];
                pool_window<T, VEC, NHWC>(x, n, cv, C, Hi, Wi, oy, ox, best, arg);
                float g[VEC];
                if constexpr (NHWC) {
                    const T* src = dy + ((n * Ho + oy) * (int64_t)Wo + ox) * C + cv * VEC;
                    if constexpr (VEC == 1) g[0] = Elt<T>::ld(src);
                    else Elt<T>::ldv(src, reinterpret_cast<float(&)[Elt<T>::VEC]>(g));
                } else {
                    g[0] = Elt<T>::ld(dy + ((n * C + cv) * (int64_t)Ho + oy) * Wo + ox);
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    if (arg[k] == me) acc[k] += g[k];
            }
        }
        if constexpr (NHWC) {
            T* dst = dx + ((n * Hi + iy) * (int64_t)Wi + ix) * C + cv * VEC;
            if constexpr (VEC == 1) Elt<T>::st(dst, acc[0]);
            else Elt<T>::stv(dst, reinterpret_cast<const float(&)[Elt<T>::VEC]>(acc));
        } else {
            Elt<T>::st(dx + v, acc[0]);
        }
    }
}

// ---- global average pool ----------------------------------------------------------------------------------------------
// NHWC: block = (sample, 64 channels); 4 waves walk the pixels 4 apart, lanes along the channels (coalesced 128 B rows)
template <typename T, typename TP>
__global__ __launch_bounds__(BLOCK) void avgpool_nhwc_kernel(const T* __restrict__ x, TP* __restrict__ y, int C, int64_t HW) {
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int64_t p = r; p < HW; p += 4) s += Elt<T>::ld(x + ((int64_t)n * HW + p) * C + c);
    __shared__ float red[4][64];
    red[r][threadIdx.x & 63] = s;
    __syncthreads();
    if (r == 0 && c < C) Elt<TP>::st(y + (int64_t)n * C + c, (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)HW);
}
// NCHW: one wave per (n, c) plane
template <typename T, typename TP>
__global__ __launch_bounds__(BLOCK) void avgpool_nchw_kernel(const T* __restrict__ x, TP* __restrict__ y, int64_t planes, int64_t HW) {
    const int64_t pl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pl >= planes) return;
    float s = 0.f;
    for (int64_t p = threadIdx.x & 63; p < HW; p += 64) s += Elt<T>::ld(x + pl * HW + p);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) Elt<TP>::st(y + pl, s / (float)HW);
}
template <typename T, typename TP, bool NHWC>
__global__ __launch_bounds__(BLOCK) void avgpool_bwd_kernel(const TP* __restrict__ dy, T* __restrict__ dx, int C, int64_t HW, int64_t total) {
    const float inv = 1.f / (float)HW;
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLOCK) {
        int64_t plane;
        if constexpr (NHWC) { const int c = (int)(v % C); plane = (v / C / HW) * C + c; }
        else plane = v / HW;
        Elt<T>::st(dx + v, Elt<TP>::ld(dy + plane) * inv);
    }
}

// ---- 1x1 convolution with bias to a few output channels, fp32 out ------------------------------------------------
constexpr int PW_MAX_CO = 32;

// y[m][co] = b[co] + sum_ci x[m][ci] * w[co][ci]; weights in LDS (read as broadcasts), one thread per pixel
template <typename T>
__global__ __launch_bounds__(BLOCK) void pointwise_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ y,
                                                              int64_t M, int Ci, int Co) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [Co][Ci]
    for (int i = threadIdx.x; i < Co * Ci; i += BLOCK) wl[i] = w[i];
    __syncthreads();
    constexpr int V = Elt<T>::VEC;
    const int64_t m = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (m >= M) return;
    float acc[PW_MAX_CO];
#pragma unroll
    for (int o = 0; o < PW_MAX_CO; ++o) acc[o] = (o < Co && b) ? b[o] : 0.f;
    const T* row = x + m * Ci;
    for (int c = 0; c < Ci; c += V) {
        float xv[V];
        Elt<T>::ldv(row + c, xv);
#pragma unroll
        for (int o = 0; o < PW_MAX_CO; ++o) {
            if (o < Co) {
#pragma unroll
                for (int k = 0; k < V; ++k) acc[o] = fmaf(xv[k], wl[o * Ci + c + k], acc[o]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < PW_MAX_CO; ++o)
        if (o < Co) y[m * Co + o] = acc[o];
}

// dx[m][ci] = sum_co dy[m][co] * w[co][ci]
template <typename T>
__global__ __launch_bounds__(BLOCK) void pointwise_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             T* __restrict__ dx, int64_t M, int Ci, int Co) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    for (int i = threadIdx.x; i < Co * Ci; i += BLOCK) wl[i] = w[i];
    __syncthreads();
    constexpr int V = Elt<T>::VEC;
    const int64_t m = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (m >= M) return;
    float g[PW_MAX_CO];
#pragma unroll
    for (int o = 0; o < PW_MAX_CO; ++o) g[o] = o < Co ? dy[m * Co + o] : 0.f;
    T* row = dx + m * Ci;
    for (int c = 0; c < Ci; c += V) {
        float o8[V];
#pragma unroll
        for (int k = 0; k < V; ++k) o8[k] = 0.f;
#pragma unroll
        for (int o = 0; o < PW_MAX_CO; ++o) {
            if (o < Co) {
#pragma unroll
                for (int k = 0; k < V; ++k) o8[k] = fmaf(g[o], wl[o * Ci + c + k], o8[k]);
            }
        }
        Elt<T>::stv(row + c, o8);
    }
}

// dw partials: block = PW_SLICE pixels; thread = input channel (strided if Ci > BLOCK); slab[blk][Co + 1][Ci]: row Co is unused
// padding so that db (summed by thread 0.. over dy only) lives in its own array
constexpr int PW_SLICE = 128;
template <typename T>
__global__ __launch_bounds__(BLOCK) void pointwise_dw_kernel(const float* __restrict__ dy, const T* __restrict__ x,
                                                             float* __restrict__ slab, float* __restrict__ bslab,
                                                             int64_t M, int Ci, int Co) {
    __shared__ float g[PW_SLICE][PW_MAX_CO];
    const int64_t m0 = (int64_t)blockIdx.x * PW_SLICE;
    const int rows = (int)((M - m0) < PW_SLICE ? (M - m0) : PW_SLICE);
    for (int i = threadIdx.x; i < PW_SLICE * Co; i += BLOCK) {
        const int r = i / Co, o = i - r * Co;
        g[r][o] = r < rows ? dy[(m0 + r) * Co + o] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Ci; c += BLOCK) {
        float acc[PW_MAX_CO];
#pragma unroll
        for (int o = 0; o < PW_MAX_CO; ++o) acc[o] = 0.f;
        for (int r = 0; r < rows; ++r) {
            const float xv = Elt<T>::ld(x + (m0 + r) * Ci + c);
#pragma unroll
            for (int o = 0; o < PW_MAX_CO; ++o)
                if (o < Co) acc[o] = fmaf(g[r][o], xv, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < PW_MAX_CO; ++o)
            if (o < Co) slab[((int64_t)blockIdx.x * Co + o) * Ci + c] = acc[o];
    }
    if (threadIdx.x < Co) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += g[r][threadIdx.x];
        bslab[(int64_t)blockIdx.x * Co + threadIdx.x] = s;
    }
}
__global__ __launch_bounds__(BLOCK) void pointwise_dw_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                                    float* __restrict__ dw, float* __restrict__ db, int G,
                                                                    int Ci, int Co, int accumulate) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < Co * Ci) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += slab[(int64_t)g * Co * Ci + i];
        dw[i] = accumulate ? dw[i] + s : s;
    }
    if (db && i < Co) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += bslab[(int64_t)g * Co + i];
        db[i] = accumulate ? db[i] + s : s;
    }
}

// ---- fp32 linear layer on a handful of rows -------------------------------------------------------------------------------
// The ASPP pooling branch (_deeplab.py:152-163) works on ONE vector per image: with 2 images per GPU its BatchNorm sees two
// samples per channel and normalises their DIFFERENCE, which bf16 storage of the pooled vector would round away (measured:
// backbone gradients off by 30-60 %).  The branch therefore stays in fp32 from the pooled vector to the broadcast:
//   y[n][co] = sum_ci x[n][ci] * w[co][ci]      (n <= LIN_MAX_N rows, fp32 master weights)
constexpr int LIN_MAX_N = 16;
__global__ __launch_bounds__(BLOCK) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 float* __restrict__ y, int N, int Ci, int Co) {
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (co >= Co) return;
    float acc[LIN_MAX_N];
#pragma unroll
    for (int n = 0; n < LIN_MAX_N; ++n) acc[n] = 0.f;
    for (int c = lane; c < Ci; c += 64) {
        const float wv = w[(int64_t)co * Ci + c];
#pragma unroll
        for (int n = 0; n < LIN_MAX_N; ++n)
            if (n < N) acc[n] = fmaf(x[(int64_t)n * Ci + c], wv, acc[n]);
    }
#pragma unroll
    for (int n = 0; n < LIN_MAX_N; ++n) {
        if (n < N) {
            const float s = wave_sum(acc[n]);
            if (lane == 0) y[(int64_t)n * Co + co] = s;
        }
    }
}
// dx[n][ci] = sum_co dy[n][co] * w[co][ci]
__global__ __launch_bounds__(BLOCK) void linear_small_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                float* __restrict__ dx, int N, int Ci, int Co) {
    extern __shared__ float g[];      // [N][Co]
    for (int i = threadIdx.x; i < N * Co; i += BLOCK) g[i] = dy[i];
    __syncthreads();
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= Ci) return;
    float acc[LIN_MAX_N];
#pragma unroll
    for (int n = 0; n < LIN_MAX_N; ++n) acc[n] = 0.f;
    for (int co = 0; co < Co; ++co) {
        const float wv = w[(int64_t)co * Ci + c];
#pragma unroll
        for (int n = 0; n < LIN_MAX_N; ++n)
            if (n < N) acc[n] = fmaf(g[n * Co + co], wv, acc[n]);
    }
#pragma unroll
    for (int n = 0; n < LIN_MAX_N; ++n)
        if (n < N) dx[(int64_t)n * Ci + c] = acc[n];
}
// dw[co][ci] (+)= sum_n dy[n][co] * x[n][ci]
__global__ __launch_bounds__(BLOCK) void linear_small_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                float* __restrict__ dw, int N, int Ci, int Co, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= (int64_t)Co * Ci) return;
    const int co = (int)(i / Ci), c = (int)(i - (int64_t)co * Ci);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(dy[(int64_t)n * Co + co], x[(int64_t)n * Ci + c], s);
    dw[i] = accumulate ? dw[i] + s : s;
}

// ---- dropout -----------------------------------------------------------------------------------------------------------
// keep(i) from a counter-based hash of (seed, i): the backward re-derives the mask from the seed the forward used, so no
// mask tensor is stored, and a captured step draws fresh masks at every replay (the seed lives in device memory).
__device__ __forceinline__ uint32_t mix64(uint64_t z) {   // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
template <typename T>
__global__ __launch_bounds__(BLOCK) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p,
                                                        const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_in,
                                                        uint64_t* __restrict__ used) {
    const uint64_t seed = seed_in ? seed_in[0] : 0;
    if (used && seed_in && blockIdx.x == 0 && threadIdx.x == 0 && used != seed_in) used[0] = seed;
    const float scale = 1.f / (1.f - p);
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const bool keep = mask ? (mask[i] != 0) : (p <= 0.f || mix64(seed ^ ((uint64_t)i * 0xD6E8FEB86659FD93ull)) >= thr);
        Elt<T>::st(y + i, keep ? Elt<T>::ld(x + i) * scale : 0.f);
    }
}
__global__ void dropout_advance_kernel(uint64_t* state) { state[0] = state[0] * 6364136223846793005ull + 1442695040888963407ull; }

template <typename T>
int vec_for(int64_t c, std::initializer_list<const void*> ptrs) {
    bool ok = c % Elt<T>::VEC == 0;
    for (const void* q : ptrs)
        if (q && !aligned(q, 16)) ok = false;
    return ok ? Elt<T>::VEC : 1;
}

}  // namespace

extern "C" {

#define AFAN_SEG_DISPATCH(KERNEL, ...)                                                                                   \
    do {                                                                                                                 \
        if (dtype == AFAN_F32) {                                                                                         \
            typedef float T;                                                                                             \
            if (layout == AFAN_NHWC) { if (vec > 1) KERNEL(T, 4, true, __VA_ARGS__); else KERNEL(T, 1, true, __VA_ARGS__); } \
            else KERNEL(T, 1, false, __VA_ARGS__);                                                                       \
        } else {                                                                                                         \
            typedef uint16_t T;                                                                                          \
            if (layout == AFAN_NHWC) { if (vec > 1) KERNEL(T, 8, true, __VA_ARGS__); else KERNEL(T, 1, true, __VA_ARGS__); } \
            else KERNEL(T, 1, false, __VA_ARGS__);                                                                       \
        }                                                                                                                \
    } while (0)

static int check_t(int dtype, int layout) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (layout != AFAN_NCHW && layout != AFAN_NHWC) return AFAN_ELAYOUT;
    return AFAN_OK;
}

int afan_upsample_bilinear_fwd(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                               int64_t ho, int64_t wo, afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x || !y) return AFAN_ENULL;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, es) || !aligned(y, es)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {x, y}) : vec_for<uint16_t>(c, {x, y})) : 1;
    const int64_t total = n * ho * wo * c / vec;
    const float sh = (float)hi / (float)ho, sw = (float)wi / (float)wo;
    AFAN_PROF("upsample_bilinear_fwd_kernel", (double)es * n * c * (ho * wo + hi * wi), st);
#define K_(T, V, L, ...) upsample_fwd_kernel<T, V, L><<<grid_for(total, BLOCK, 4096), BLOCK, 0, st>>>((const T*)x, (T*)y, (int)c, (int)hi, (int)wi, (int)ho, (int)wo, sh, sw, total)
    AFAN_SEG_DISPATCH(K_, 0);
#undef K_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_upsample_bilinear_bwd(const void* dy, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                               int64_t ho, int64_t wo, afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!dy || !dx) return AFAN_ENULL;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(dy, es) || !aligned(dx, es)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {dy, dx}) : vec_for<uint16_t>(c, {dy, dx})) : 1;
    const int64_t total = n * hi * wi * c / vec;
    const float sh = (float)hi / (float)ho, sw = (float)wi / (float)wo;
    const float ish = (float)ho / (float)hi, isw = (float)wo / (float)wi;
    AFAN_PROF("upsample_bilinear_bwd_kernel", (double)es * n * c * (ho * wo + hi * wi), st);
#define K_(T, V, L, ...) upsample_bwd_kernel<T, V, L><<<grid_for(total, BLOCK, 4096), BLOCK, 0, st>>>((const T*)dy, (T*)dx, (int)c, (int)hi, (int)wi, (int)ho, (int)wo, sh, sw, ish, isw, total)
    AFAN_SEG_DISPATCH(K_, 0);
#undef K_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

static int ce2d_blocks(int64_t pixels) { return grid_for(pixels, BLOCK, 2048); }

int64_t afan_ce2d_workspace_floats(int64_t pixels) { return pixels > 0 ? 1 + 2 * (int64_t)ce2d_blocks(pixels) : 0; }

int afan_ce2d(const float* logits, const int64_t* target, int layout, int64_t n, int64_t c, int64_t hw, int64_t ignore_index,
              float grad_scale, float* workspace, float* loss, float* dlogits, afan_stream_t stream) {
    if (layout != AFAN_NCHW && layout != AFAN_NHWC) return AFAN_ELAYOUT;
    if (n <= 0 || hw <= 0 || c <= 0 || c > CE_MAX_C) return AFAN_ESHAPE;
    if (!logits || !target || !workspace || !loss) return AFAN_ENULL;
    if (!aligned(logits, 4) || !aligned(target, 8) || !aligned(workspace, 4)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = n * hw;
    const int G = ce2d_blocks(P);
    AFAN_PROF("ce2d_kernel", (double)P * (8.0 + 8.0 + 4.0 * c * (dlogits ? 2 : 1)), st);
    ce2d_count_kernel<<<G, BLOCK, 0, st>>>(target, P, ignore_index, workspace + 1);
    AFAN_LAUNCH_CHECK();
    ce2d_fold_count_kernel<<<1, 1, 0, st>>>(workspace, G);
    AFAN_LAUNCH_CHECK();
    if (layout == AFAN_NHWC) ce2d_kernel<true><<<G, BLOCK, 0, st>>>(logits, target, dlogits, (int)c, hw, P, ignore_index, grad_scale, workspace, G);
    else ce2d_kernel<false><<<G, BLOCK, 0, st>>>(logits, target, dlogits, (int)c, hw, P, ignore_index, grad_scale, workspace, G);
    AFAN_LAUNCH_CHECK();
    ce2d_finalize_kernel<<<1, 1, 0, st>>>(workspace, G, loss);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_maxpool3x3s2_fwd(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                          afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hi <= 0 || wi <= 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x || !y) return AFAN_ENULL;
    const int64_t ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    hipStream_t st = (hipStream_t)stream;
    const int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {x, y}) : vec_for<uint16_t>(c, {x, y})) : 1;
    const int64_t total = n * ho * wo * c / vec;
    AFAN_PROF("maxpool_fwd_kernel", (double)es * n * c * (ho * wo + hi * wi), st);
#define K_(T, V, L, ...) maxpool_fwd_kernel<T, V, L><<<grid_for(total, BLOCK, 4096), BLOCK, 0, st>>>((const T*)x, (T*)y, (int)c, (int)hi, (int)wi, (int)ho, (int)wo, total)
    AFAN_SEG_DISPATCH(K_, 0);
#undef K_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_maxpool3x3s2_bwd(const void* dy, const void* x, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hi,
                          int64_t wi, afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hi <= 0 || wi <= 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!dy || !x || !dx) return AFAN_ENULL;
    const int64_t ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    hipStream_t st = (hipStream_t)stream;
    const int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {dy, x, dx}) : vec_for<uint16_t>(c, {dy, x, dx})) : 1;
    const int64_t total = n * hi * wi * c / vec;
    AFAN_PROF("maxpool_bwd_kernel", (double)es * n * c * (ho * wo + 2 * hi * wi), st);
#define K_(T, V, L, ...) maxpool_bwd_kernel<T, V, L><<<grid_for(total, BLOCK, 4096), BLOCK, 0, st>>>((const T*)dy, (const T*)x, (T*)dx, (int)c, (int)hi, (int)wi, (int)ho, (int)wo, total)
    AFAN_SEG_DISPATCH(K_, 0);
#undef K_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_avgpool_fwd(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hw, int pooled_f32,
                     afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hw <= 0 || n > 65535) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x || !y) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    AFAN_PROF("avgpool_fwd_kernel", (double)es * n * c * (hw + 1), st);
    const bool pf = pooled_f32 || dtype == AFAN_F32;
    if (layout == AFAN_NHWC) {
        dim3 grid((unsigned)((c + 63) / 64), (unsigned)n);
        if (dtype == AFAN_F32) avgpool_nhwc_kernel<float, float><<<grid, BLOCK, 0, st>>>((const float*)x, (float*)y, (int)c, hw);
        else if (pf) avgpool_nhwc_kernel<uint16_t, float><<<grid, BLOCK, 0, st>>>((const uint16_t*)x, (float*)y, (int)c, hw);
        else avgpool_nhwc_kernel<uint16_t, uint16_t><<<grid, BLOCK, 0, st>>>((const uint16_t*)x, (uint16_t*)y, (int)c, hw);
    } else {
        const int64_t planes = n * c;
        const unsigned g = (unsigned)((planes + 3) / 4);
        if (dtype == AFAN_F32) avgpool_nchw_kernel<float, float><<<g, BLOCK, 0, st>>>((const float*)x, (float*)y, planes, hw);
        else if (pf) avgpool_nchw_kernel<uint16_t, float><<<g, BLOCK, 0, st>>>((const uint16_t*)x, (float*)y, planes, hw);
        else avgpool_nchw_kernel<uint16_t, uint16_t><<<g, BLOCK, 0, st>>>((const uint16_t*)x, (uint16_t*)y, planes, hw);
    }
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_avgpool_bwd(const void* dy, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hw, int pooled_f32,
                     afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hw <= 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!dy || !dx) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = n * c * hw;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    AFAN_PROF("avgpool_bwd_kernel", (double)es * n * c * (hw + 1), st);
    const int g = grid_for(total, BLOCK, 4096);
    const bool pf = pooled_f32 || dtype == AFAN_F32;
#define AP_(T, TP) do { if (layout == AFAN_NHWC) avgpool_bwd_kernel<T, TP, true><<<g, BLOCK, 0, st>>>((const TP*)dy, (T*)dx, (int)c, hw, total); \
                        else avgpool_bwd_kernel<T, TP, false><<<g, BLOCK, 0, st>>>((const TP*)dy, (T*)dx, (int)c, hw, total); } while (0)
    if (dtype == AFAN_F32) AP_(float, float);
    else if (pf) AP_(uint16_t, float);
    else AP_(uint16_t, uint16_t);
#undef AP_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

static int pw_check(int dtype, int64_t m, int64_t ci, int64_t co) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (m <= 0 || ci <= 0 || co <= 0 || co > PW_MAX_CO || ci % 8 != 0 || ci * co * 4 > 64 * 1024) return AFAN_ESHAPE;
    return AFAN_OK;
}

int afan_pointwise_max_co(void) { return PW_MAX_CO; }

int afan_pointwise_fwd(const void* x, int x_dtype, const float* w, const float* b, float* y, int64_t m, int64_t ci,
                       int64_t co, afan_stream_t stream) {
    int e = pw_check(x_dtype, m, ci, co);
    if (e) return e;
    if (!x || !w || !y) return AFAN_ENULL;
    if (!aligned(x, 16) || !aligned(w, 4) || !aligned(y, 4)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int es = x_dtype == AFAN_F32 ? 4 : 2;
    AFAN_PROF("pointwise_fwd_kernel", (double)m * (es * ci + 4.0 * co), st);
    const unsigned g = (unsigned)((m + BLOCK - 1) / BLOCK);
    const size_t lds = (size_t)ci * co * 4;
    if (x_dtype == AFAN_F32) pointwise_fwd_kernel<float><<<g, BLOCK, lds, st>>>((const float*)x, w, b, y, m, (int)ci, (int)co);
    else pointwise_fwd_kernel<uint16_t><<<g, BLOCK, lds, st>>>((const uint16_t*)x, w, b, y, m, (int)ci, (int)co);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_pointwise_bwd_dx(const float* dy, const float* w, void* dx, int dx_dtype, int64_t m, int64_t ci, int64_t co,
                          afan_stream_t stream) {
    int e = pw_check(dx_dtype, m, ci, co);
    if (e) return e;
    if (!dy || !w || !dx) return AFAN_ENULL;
    if (!aligned(dx, 16) || !aligned(w, 4) || !aligned(dy, 4)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int es = dx_dtype == AFAN_F32 ? 4 : 2;
    AFAN_PROF("pointwise_dx_kernel", (double)m * (es * ci + 4.0 * co), st);
    const unsigned g = (unsigned)((m + BLOCK - 1) / BLOCK);
    const size_t lds = (size_t)ci * co * 4;
    if (dx_dtype == AFAN_F32) pointwise_dx_kernel<float><<<g, BLOCK, lds, st>>>(dy, w, (float*)dx, m, (int)ci, (int)co);
    else pointwise_dx_kernel<uint16_t><<<g, BLOCK, lds, st>>>(dy, w, (uint16_t*)dx, m, (int)ci, (int)co);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int64_t afan_pointwise_workspace_floats(int64_t m, int64_t ci, int64_t co) {
    if (m <= 0 || ci <= 0 || co <= 0) return 0;
    const int64_t g = (m + PW_SLICE - 1) / PW_SLICE;
    return g * co * (ci + 1);
}

int afan_pointwise_bwd_dw(const float* dy, const void* x, int x_dtype, float* dw, float* db, int64_t m, int64_t ci,
                          int64_t co, float* workspace, int accumulate, afan_stream_t stream) {
    int e = pw_check(x_dtype, m, ci, co);
    if (e) return e;
    if (!dy || !x || !dw || !workspace) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const int G = (int)((m + PW_SLICE - 1) / PW_SLICE);
    float* bslab = workspace + (int64_t)G * co * ci;
    const int es = x_dtype == AFAN_F32 ? 4 : 2;
    AFAN_PROF("pointwise_dw_kernel", (double)m * (es * ci + 4.0 * co) + 8.0 * G * co * ci, st);
    if (x_dtype == AFAN_F32) pointwise_dw_kernel<float><<<G, BLOCK, 0, st>>>(dy, (const float*)x, workspace, bslab, m, (int)ci, (int)co);
    else pointwise_dw_kernel<uint16_t><<<G, BLOCK, 0, st>>>(dy, (const uint16_t*)x, workspace, bslab, m, (int)ci, (int)co);
    AFAN_LAUNCH_CHECK();
    pointwise_dw_reduce_kernel<<<(unsigned)((co * ci + BLOCK - 1) / BLOCK), BLOCK, 0, st>>>(workspace, bslab, dw, db, G, (int)ci, (int)co, accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_linear_small_max_rows(void) { return LIN_MAX_N; }

int afan_linear_small_fwd(const float* x, const float* w, float* y, int64_t n, int64_t ci, int64_t co, afan_stream_t stream) {
    if (n <= 0 || n > LIN_MAX_N || ci <= 0 || co <= 0) return AFAN_ESHAPE;
    if (!x || !w || !y) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("linear_small_fwd_kernel", 4.0 * (ci * co + n * (ci + co)), st);
    linear_small_fwd_kernel<<<(unsigned)((co + 3) / 4), BLOCK, 0, st>>>(x, w, y, (int)n, (int)ci, (int)co);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_linear_small_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, int64_t n, int64_t ci,
                          int64_t co, int accumulate, afan_stream_t stream) {
    if (n <= 0 || n > LIN_MAX_N || ci <= 0 || co <= 0 || n * co * 4 > 64 * 1024) return AFAN_ESHAPE;
    if (!dy || !w || !x) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("linear_small_bwd_kernel", 4.0 * (2 * ci * co + 2 * n * (ci + co)), st);
    if (dx) {
        linear_small_dx_kernel<<<(unsigned)((ci + BLOCK - 1) / BLOCK), BLOCK, (size_t)n * co * 4, st>>>(dy, w, dx, (int)n, (int)ci, (int)co);
        AFAN_LAUNCH_CHECK();
    }
    if (dw) {
        linear_small_dw_kernel<<<(unsigned)((ci * co + BLOCK - 1) / BLOCK), BLOCK, 0, st>>>(dy, x, dw, (int)n, (int)ci, (int)co, accumulate);
        AFAN_LAUNCH_CHECK();
    }
    return AFAN_OK;
}

int afan_dropout(const void* x, void* y, int dtype, int64_t n, float p, const uint8_t* mask, uint64_t* state, uint64_t* used,
                 int advance, afan_stream_t stream) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n < 0 || !(p >= 0.f && p < 1.f)) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x || !y) return AFAN_ENULL;
    if (!mask && p > 0.f && !state && !used) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    AFAN_PROF("dropout_kernel", 2.0 * es * n, st);
    const uint64_t* seed = state ? state : used;
    const int g = grid_for(n, BLOCK, 2048);
    if (dtype == AFAN_F32) dropout_kernel<float><<<g, BLOCK, 0, st>>>((const float*)x, (float*)y, n, p, mask, seed, state ? used : nullptr);
    else dropout_kernel<uint16_t><<<g, BLOCK, 0, st>>>((const uint16_t*)x, (uint16_t*)y, n, p, mask, seed, state ? used : nullptr);
    AFAN_LAUNCH_CHECK();
    if (state && advance) {
        dropout_advance_kernel<<<1, 1, 0, st>>>(state);
        AFAN_LAUNCH_CHECK();
    }
    return AFAN_OK;
}

}  // extern "C"
