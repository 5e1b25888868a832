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

// Grid: blockIdx.x = (image, output row) [NHWC] or (image, channel, output row) [NCHW]; a thread = one (column, channel
// vector) of that row — one 32-bit division per thread, the row's source rows and weights are wave-uniform.
template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int C, int Hi,
                                                             int Wi, int Ho, int Wo, float sh, float sw, int ldy) {
    const uint32_t CV = NHWC ? C / VEC : 1;
    const uint32_t row = blockIdx.x, oy = row % Ho, plane = row / Ho;      // plane = n (NHWC) or n * C + c (NCHW)
    const uint32_t t = blockIdx.y * BLOCK + threadIdx.x;
    if (t >= (uint32_t)Wo * CV) return;
    const uint32_t ox = t / CV, cv = t - ox * CV;
    const Src a = src_index(sh, (int)oy, Hi), b = src_index(sw, (int)ox, Wi);
    if constexpr (NHWC) {
        const T* base = x + (int64_t)plane * Hi * Wi * C + cv * VEC;
        float p00[VEC], p01[VEC], p10[VEC], p11[VEC], o[VEC];
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
        T* dst = y + (((int64_t)plane * Ho + oy) * Wo + ox) * ldy + cv * VEC;     // (ldy = C, or the pixel stride of a wider tensor)
        if constexpr (VEC == 1) Elt<T>::st(dst, o[0]);
        else Elt<T>::stv(dst, reinterpret_cast<const float(&)[Elt<T>::VEC]>(o));
    } else {
        const T* base = x + (int64_t)plane * Hi * Wi;
        const float p00 = Elt<T>::ld(base + (int64_t)a.i0 * Wi + b.i0), p01 = Elt<T>::ld(base + (int64_t)a.i0 * Wi + b.i1);
        const float p10 = Elt<T>::ld(base + (int64_t)a.i1 * Wi + b.i0), p11 = Elt<T>::ld(base + (int64_t)a.i1 * Wi + b.i1);
        Elt<T>::st(y + ((int64_t)plane * Ho + oy) * Wo + ox, a.l0 * (b.l0 * p00 + b.l1 * p01) + a.l1 * (b.l0 * p10 + b.l1 * p11));
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

// Backward in two separable passes (channels-last, 16-byte vectors): the resize is a tensor product of two 1-D maps, so
//   tmp[n, oy, ix, c] = sum_ox wx(ox, ix) dy[n, oy, ox, c]        (rows pass: one thread = one (oy, ix, channel vector), ~scale loads)
//   dx [n, iy, ix, c] = sum_oy wy(oy, iy) tmp[n, oy, ix, c]       (columns pass)
// instead of ~scale^2 dependent loads per thread on a grid of only Hi x Wi x C/8 threads (33 x 33 x 32 per image for the
// decoder's 33 -> 129 resize: 40 us at 0.07 of the HBM rate).  tmp is fp32; sums in increasing index order (deterministic).
template <typename T, int VEC>
__global__ __launch_bounds__(BLOCK) void upsample_bwd_rows_kernel(const T* __restrict__ dy, float* __restrict__ tmp, int C, int Wi,
                                                                  int Ho, int Wo, float sw, float isw, int ldy) {
    const uint32_t CV = C / VEC;
    const uint32_t row = blockIdx.x;                                   // n * Ho + oy
    const uint32_t t = blockIdx.y * BLOCK + threadIdx.x;
    if (t >= (uint32_t)Wi * CV) return;
    const uint32_t ix = t / CV, cv = t - ix * CV;
    int xlo, xhi;
    out_range(isw, (int)ix, Wo, xlo, xhi);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    const T* src = dy + ((int64_t)row * Wo) * ldy + cv * VEC;
    for (int ox = xlo; ox <= xhi; ++ox) {
        const float wx = axis_weight(sw, ox, (int)ix, Wi);
        float g[VEC];
        Elt<T>::ldv(src + (int64_t)ox * ldy, reinterpret_cast<float(&)[Elt<T>::VEC]>(g));
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(wx, g[k], acc[k]);
    }
    float* dst = tmp + ((int64_t)row * Wi + ix) * C + cv * VEC;
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
}

template <typename T, int VEC>
__global__ __launch_bounds__(BLOCK) void upsample_bwd_cols_kernel(const float* __restrict__ tmp, T* __restrict__ dx, int C, int Hi,
                                                                  int Wi, int Ho, float sh, float ish) {
    const uint32_t CV = C / VEC;
    const uint32_t row = blockIdx.x, iy = row % Hi, plane = row / Hi;
    const uint32_t t = blockIdx.y * BLOCK + threadIdx.x;
    if (t >= (uint32_t)Wi * CV) return;
    const uint32_t ix = t / CV, cv = t - ix * CV;
    int ylo, yhi;
    out_range(ish, (int)iy, Ho, ylo, yhi);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    const float* src = tmp + (((int64_t)plane * Ho) * Wi + ix) * C + cv * VEC;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = axis_weight(sh, oy, (int)iy, Hi);
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(src + (int64_t)oy * Wi * C + 4 * q);
            acc[4 * q] = fmaf(wy, g.x, acc[4 * q]); acc[4 * q + 1] = fmaf(wy, g.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(wy, g.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(wy, g.w, acc[4 * q + 3]);
        }
    }
    Elt<T>::stv(dx + (((int64_t)plane * Hi + iy) * Wi + ix) * C + cv * VEC, reinterpret_cast<const float(&)[Elt<T>::VEC]>(acc));
}

// Backward as a gather over the same row-wise grid (blockIdx.x = input row): a thread = one INPUT (column, channel vector),
// summing the output gradients it fed, rows then columns in increasing order (deterministic).
template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void upsample_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int C, int Hi,
                                                             int Wi, int Ho, int Wo, float sh, float sw, float ish, float isw,
                                                             int ldy) {
    const uint32_t CV = NHWC ? C / VEC : 1;
    const uint32_t row = blockIdx.x, iy = row % Hi, plane = row / Hi;
    const uint32_t t = blockIdx.y * BLOCK + threadIdx.x;
    if (t >= (uint32_t)Wi * CV) return;
    const uint32_t ix = t / CV, cv = t - ix * CV;
    int ylo, yhi, xlo, xhi;
    out_range(ish, (int)iy, Ho, ylo, yhi);
    out_range(isw, (int)ix, Wo, xlo, xhi);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = axis_weight(sh, oy, (int)iy, Hi);
        if (wy == 0.f) continue;
        for (int ox = xlo; ox <= xhi; ++ox) {
            const float wx = axis_weight(sw, ox, (int)ix, Wi);
            if (wx == 0.f) continue;
            const float w = wy * wx;
            if constexpr (NHWC) {
                const T* src = dy + (((int64_t)plane * Ho + oy) * Wo + ox) * ldy + cv * VEC;
                if constexpr (VEC == 1) acc[0] += w * Elt<T>::ld(src);
                else {
                    float g[VEC];
                    Elt<T>::ldv(src, reinterpret_cast<float(&)[Elt<T>::VEC]>(g));
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += w * g[k];
                }
            } else {
                acc[0] += w * Elt<T>::ld(dy + ((int64_t)plane * Ho + oy) * Wo + ox);
            }
        }
    }
    if constexpr (NHWC) {
        T* dst = dx + (((int64_t)plane * Hi + iy) * Wi + ix) * C + cv * VEC;
        if constexpr (VEC == 1) Elt<T>::st(dst, acc[0]);
        else Elt<T>::stv(dst, reinterpret_cast<const float(&)[Elt<T>::VEC]>(acc));
    } else {
        Elt<T>::st(dx + ((int64_t)plane * Hi + iy) * Wi + ix, acc[0]);
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
__device__ __forceinline__ float block_sum256(float v) {      // fixed order: wave butterflies, then the 4 waves in index order
    v = wave_sum(v);
    __shared__ float red[BLOCK / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(BLOCK) void ce2d_fold_count_kernel(float* ws, int G) {
    float c = 0.f;
    for (int i = threadIdx.x; i < G; i += BLOCK) c += ws[1 + i];
    c = block_sum256(c);
    if (threadIdx.x == 0) ws[0] = c;
}

// One thread = one pixel.  NHWC: the block's 256 pixels x C logits are one contiguous run — copied through LDS (row stride C
// words: conflict-free for odd C, 2-way for C = 2 mod 4) so that global loads and stores are fully coalesced; NCHW: class
// planes are already coalesced along the pixels.
template <bool NHWC>
__global__ __launch_bounds__(BLOCK) void ce2d_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     float* __restrict__ dlogits, int C, int64_t HW, int64_t P,
                                                     int64_t ignore, float grad_scale, float* __restrict__ ws, int G) {
    extern __shared__ float tile[];          // NHWC: [BLOCK][C]
    const float count = ws[0];
    const float gs = grad_scale / count;
    float loss = 0.f;
    for (int64_t p0 = (int64_t)blockIdx.x * BLOCK; p0 < P; p0 += (int64_t)gridDim.x * BLOCK) {
        const int npx = (int)((P - p0) < BLOCK ? (P - p0) : BLOCK);
        if (NHWC) {
            __syncthreads();
            for (int i = threadIdx.x; i < npx * C; i += BLOCK) tile[i] = logits[p0 * C + i];
            __syncthreads();
        }
        const int64_t p = p0 + threadIdx.x;
        if ((int)threadIdx.x < npx) {
            const int64_t t = target[p];
            const int64_t n = p / HW, q = p - n * HW;
            const int64_t base = n * C * HW + q;          // NCHW
            float l[CE_MAX_C];               // fully unrolled with a guard: stays in registers
            float m = -INFINITY;
#pragma unroll
            for (int c = 0; c < CE_MAX_C; ++c) {
                l[c] = c < C ? (NHWC ? tile[threadIdx.x * C + c] : logits[base + c * HW]) : -INFINITY;
                m = fmaxf(m, l[c]);
            }
            const bool live = t != ignore;
            const bool bad = live && (t < 0 || t >= C);      // torch asserts here; poison the loss instead of reading out of range
            float lt = 0.f, s = 0.f;
#pragma unroll
            for (int c = 0; c < CE_MAX_C; ++c) {
                if (c == (int)t) lt = l[c];
                l[c] = c < C ? expf(l[c] - m) : 0.f;
                s += l[c];
            }
            const float inv = 1.f / s;
            if (live && !bad) loss += logf(s) + m - lt;
            if (bad) loss = NAN;
            if (dlogits) {
#pragma unroll
                for (int c = 0; c < CE_MAX_C; ++c) {
                    float g = 0.f;
                    if (live && !bad) g = (l[c] * inv - (c == (int)t ? 1.f : 0.f)) * gs;
                    if (c < C) {
                        if (NHWC) tile[threadIdx.x * C + c] = g;
                        else dlogits[base + c * HW] = g;
                    }
                }
            }
        }
        if (NHWC && dlogits) {
            __syncthreads();
            for (int i = threadIdx.x; i < npx * C; i += BLOCK) dlogits[p0 * C + i] = tile[i];
        }
    }
    loss = wave_sum(loss);
    __shared__ float red[BLOCK / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) ws[1 + G + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(BLOCK) void ce2d_finalize_kernel(const float* ws, int G, float* loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < G; i += BLOCK) s += ws[1 + G + i];
    s = block_sum256(s);
    if (threadIdx.x == 0) loss[0] = s / ws[0];
}

// ---- bilinear resize + per-pixel cross-entropy + its gradient back at the LOW resolution, fused --------------------------
// Segmentation/network/utils.py:30,45 resizes the classifier's [N, C, h, w] logits to the image size and main_aug_final.py:95's
// criterion takes the mean cross-entropy over the H x W pixels; the backward resizes the [N, C, H, W] gradient back.  Every
// PGD pass and every perturbed forward does that (9 of an iteration's 10 passes need no full-resolution logits at all): four
// trips of a 44 MB tensor through HBM (2 x 21 x 513 x 513 fp32) for 0.7 MB of information.  Here a workgroup owns a 16 x 16
// tile of OUTPUT pixels: it stages the <= 6 x 6 source pixels the tile reads, computes each pixel's interpolated logits,
// soft-max, loss and gradient ONCE (into LDS), and folds the tile's gradient onto its source window (per-axis weight tables
// in LDS; separably: rows, then columns).  A source pixel on a tile border receives parts from up to four tiles: they are
// written as per-tile partials and summed by a second small kernel in tile order — deterministic, no atomics.  Values:
// afan_upsample_bilinear_fwd + afan_ce2d to the bit; the gradient's resize-backward is the same linear map with its double sum
// associated separably and by tile (rounding-level differences against afan_upsample_bilinear_bwd).  Measured (rocprofv3,
// 2 x 21 x 129 x 129 -> 513 x 513): 70 us + 7 us gather against 35 + 23 + 51 us for the three kernels it replaces — the
// kernel is instruction-bound (~15 k instructions per wave: the soft-max of 21 classes per pixel and runtime-C index math),
// no longer HBM-bound; 1.5 % of a DeepLab iteration.
constexpr int UP_OT = 16;        // output tile side
constexpr int UP_SW = 8;         // upper bound of the source window side the kernel accepts (checked by the host)

__global__ __launch_bounds__(BLOCK) void ce2d_up_kernel(const float* __restrict__ lo, const int64_t* __restrict__ target,
                                                        float* __restrict__ part, int C, int h, int w, int H, int W, float sh,
                                                        float sw, int64_t ignore, float grad_scale,
                                                        const float* __restrict__ ws_count, float* __restrict__ loss_part,
                                                        int tiles_x, int tiles_y, int want_grad) {
    extern __shared__ float lds_up[];
    float* gt = lds_up;                           // [256][C]  gradients w.r.t. the interpolated logits
    float* st = gt + BLOCK * C;                   // [SH * SW][C] source logits of the tile's window
    __shared__ float wy[UP_SW][UP_OT], wx[UP_SW][UP_OT];
    const int tile = blockIdx.x % (tiles_x * tiles_y), n = blockIdx.x / (tiles_x * tiles_y);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * UP_OT, y1 = (y0 + UP_OT < H ? y0 + UP_OT : H) - 1;
    const int x0 = tx * UP_OT, x1 = (x0 + UP_OT < W ? x0 + UP_OT : W) - 1;
    const int s0 = src_index(sh, y0, h).i0, s1 = src_index(sh, y1, h).i1;
    const int t0 = src_index(sw, x0, w).i0, t1 = src_index(sw, x1, w).i1;
    const int SH_ = s1 - s0 + 1, SW_ = t1 - t0 + 1;
    const float* base = lo + (int64_t)n * h * w * C;
    for (int e = threadIdx.x; e < SH_ * SW_ * C; e += BLOCK) {
        const int c = e % C, p = e / C, r = p / SW_, q = p - r * SW_;
        st[e] = base[((int64_t)(s0 + r) * w + (t0 + q)) * C + c];
    }
    if (threadIdx.x < UP_SW * UP_OT) {            // the two weight tables: [source row of the window][output row of the tile]
        const int r = threadIdx.x / UP_OT, o = threadIdx.x % UP_OT;
        wy[r][o] = (r < SH_ && y0 + o <= y1) ? axis_weight(sh, y0 + o, s0 + r, h) : 0.f;
        wx[r][o] = (r < SW_ && x0 + o <= x1) ? axis_weight(sw, x0 + o, t0 + r, w) : 0.f;
    }
    __syncthreads();
    const float gs = grad_scale / ws_count[0];
    float loss = 0.f;
    const int oyl = threadIdx.x / UP_OT, oxl = threadIdx.x % UP_OT, oy = y0 + oyl, ox = x0 + oxl;
    const bool inside = oy <= y1 && ox <= x1;
    if (inside) {
        const Src a = src_index(sh, oy, h), b = src_index(sw, ox, w);
        const int64_t t = target[((int64_t)n * H + oy) * W + ox];
        const float* r0 = st + ((a.i0 - s0) * SW_ - t0) * C;
        const float* r1 = st + ((a.i1 - s0) * SW_ - t0) * C;
        float l[CE_MAX_C];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < CE_MAX_C; ++c) {
            if (c < C) {    // ATen's association, as upsample_fwd_kernel
                const float p00 = r0[b.i0 * C + c], p01 = r0[b.i1 * C + c], p10 = r1[b.i0 * C + c], p11 = r1[b.i1 * C + c];
                l[c] = a.l0 * (b.l0 * p00 + b.l1 * p01) + a.l1 * (b.l0 * p10 + b.l1 * p11);
            } else {
                l[c] = -INFINITY;
            }
            m = fmaxf(m, l[c]);
        }
        const bool live = t != ignore;
        const bool bad = live && (t < 0 || t >= C);
        float lt = 0.f, s = 0.f;
#pragma unroll
        for (int c = 0; c < CE_MAX_C; ++c) {
            if (c == (int)t) lt = l[c];
            l[c] = c < C ? expf(l[c] - m) : 0.f;
            s += l[c];
        }
        const float inv = 1.f / s;
        if (live && !bad) loss += logf(s) + m - lt;
        if (bad) loss = NAN;
        if (want_grad) {
#pragma unroll
            for (int c = 0; c < CE_MAX_C; ++c) {
                float g = 0.f;
                if (live && !bad) g = (l[c] * inv - (c == (int)t ? 1.f : 0.f)) * gs;
                if (c < C) gt[threadIdx.x * C + c] = g;
            }
        }
    } else if (want_grad) {
        for (int c = 0; c < C; ++c) gt[threadIdx.x * C + c] = 0.f;
    }
    __syncthreads();
    if (want_grad) {
        // the tile's gradient folded onto its source window, separably: rows first (T[r][b][c] = sum_a wy[r][a] g[a][b][c]),
        // then columns (P[r][q][c] = sum_b wx[q][b] T[r][b][c]); both sums ascending, every operand from LDS, no branches
        float* T = st;                              // (the staged source logits are dead: reuse their LDS; SH*16*C <= 8*16*C floats
        float* P = part + (int64_t)blockIdx.x * UP_SW * UP_SW * C;   //  are reserved behind gt by the host)
        for (int e = threadIdx.x; e < SH_ * UP_OT * C; e += BLOCK) {
            const int c = e % C, p = e / C, r = p / UP_OT, b = p - r * UP_OT;
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < UP_OT; ++a) acc += wy[r][a] * gt[(a * UP_OT + b) * C + c];
            T[e] = acc;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < SH_ * SW_ * C; e += BLOCK) {
            const int c = e % C, p = e / C, r = p / SW_, q = p - r * SW_;
            float acc = 0.f;
#pragma unroll
            for (int b = 0; b < UP_OT; ++b) acc += wx[q][b] * T[(r * UP_OT + b) * C + c];
            P[(r * UP_SW + q) * C + c] = acc;
        }
    }
    loss = block_sum256(loss);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = loss;
}

// dlo[n, si, sj, c] = sum over the output tiles whose source window holds (si, sj), tile rows then tile columns ascending
__global__ __launch_bounds__(BLOCK) void ce2d_up_gather_kernel(const float* __restrict__ part, float* __restrict__ dlo, int C, int h,
                                                               int w, int H, int W, float sh, float sw, float ish, float isw,
                                                               int tiles_x, int tiles_y, int64_t total) {
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLOCK) {
        const int c = (int)(v % C);
        int64_t t = v / C;
        const int sj = (int)(t % w);
        t /= w;
        const int si = (int)(t % h), n = (int)(t / h);
        int a0, a1, b0, b1;
        out_range(ish, si, H, a0, a1);
        out_range(isw, sj, W, b0, b1);
        float acc = 0.f;
        for (int ty = a0 / UP_OT; ty <= a1 / UP_OT && ty < tiles_y; ++ty) {
            const int y0 = ty * UP_OT, y1 = (y0 + UP_OT < H ? y0 + UP_OT : H) - 1;
            const int s0 = src_index(sh, y0, h).i0, s1 = src_index(sh, y1, h).i1;
            if (si < s0 || si > s1) continue;
            for (int tx = b0 / UP_OT; tx <= b1 / UP_OT && tx < tiles_x; ++tx) {
                const int x0 = tx * UP_OT, x1 = (x0 + UP_OT < W ? x0 + UP_OT : W) - 1;
                const int t0 = src_index(sw, x0, w).i0, t1 = src_index(sw, x1, w).i1;
                if (sj < t0 || sj > t1) continue;
                const int64_t blk = ((int64_t)n * tiles_y + ty) * tiles_x + tx;
                acc += part[(blk * UP_SW * UP_SW + (si - s0) * UP_SW + (sj - t0)) * C + c];
            }
        }
        dlo[v] = acc;
    }
}

__global__ __launch_bounds__(BLOCK) void ce2d_up_finalize_kernel(const float* __restrict__ ws_count, const float* __restrict__ part,
                                                                 int G, float* loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < G; i += BLOCK) s += part[i];
    s = block_sum256(s);
    if (threadIdx.x == 0) loss[0] = s / ws_count[0];
}

// ---- max pooling: k x k windows, stride s, padding p (3/2/1: backbone/resnet.py:146; 2/2/0 and the global one: Detection/
// roi/pooler.py:43, model.py:285) -------------------------------------------------------------------------------------------
// first maximum in (h, w) scan order, NaN wins (ATen's CPU kernel: `val > maxval || isnan(val)`), so that the backward
// routes the gradient to the same element the reference's does — post-ReLU windows tie at 0 all the time.  The forward can
// leave the winner's position inside its window (r * k + s, one byte per output element); the backward then reads
// (gradient, position) of the at most ceil(k/s)^2 windows over an input pixel instead of re-scanning k*k inputs for each.
struct PoolGeo { int K, S, P; };

template <typename T, int VEC, bool NHWC>
__device__ __forceinline__ void pool_window(const T* __restrict__ x, int64_t n, int cv, int C, int Hi, int Wi, int oy,
                                            int ox, PoolGeo g, float (&best)[VEC], int (&arg)[VEC]) {
    const int h0 = oy * g.S - g.P, w0 = ox * g.S - g.P;
    const int hs = h0 < 0 ? 0 : h0, ws = w0 < 0 ? 0 : w0;
    const int he = h0 + g.K > Hi ? Hi : h0 + g.K, we = w0 + g.K > Wi ? Wi : w0 + g.K;
#pragma unroll
    for (int k = 0; k < VEC; ++k) { best[k] = -INFINITY; arg[k] = (hs - h0) * g.K + (ws - w0); }
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
                if (v[k] > best[k] || v[k] != v[k]) { best[k] = v[k]; arg[k] = (h - h0) * g.K + (w - w0); }
        }
}

template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                            int C, int Hi, int Wi, int Ho, int Wo, PoolGeo g, int64_t total) {
    const int CV = C / VEC;
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLOCK) {
        int64_t t = v;
        int cv, ox, oy;
        int64_t n;
        if constexpr (NHWC) { cv = (int)(t % CV); t /= CV; ox = (int)(t % Wo); t /= Wo; oy = (int)(t % Ho); n = t / Ho; }
        else { ox = (int)(t % Wo); t /= Wo; oy = (int)(t % Ho); t /= Ho; cv = (int)(t % CV); n = t / CV; }
        float best[VEC];
        int arg[VEC];
        pool_window<T, VEC, NHWC>(x, n, cv, C, Hi, Wi, oy, ox, g, best, arg);
        if constexpr (NHWC) {
            const int64_t o = ((n * Ho + oy) * (int64_t)Wo + ox) * C + cv * VEC;
            if constexpr (VEC == 1) Elt<T>::st(y + o, best[0]);
            else Elt<T>::stv(y + o, reinterpret_cast<const float(&)[Elt<T>::VEC]>(best));
            if (idx) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) idx[o + k] = (uint8_t)arg[k];
            }
        } else {
            Elt<T>::st(y + v, best[0]);
            if (idx) idx[v] = (uint8_t)arg[0];
        }
    }
}

// gather: one thread = one input pixel x VEC channels; every window that contains it contributes its gradient if this
// pixel won it (position byte from the forward, or — without it — the window re-scanned)
template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void maxpool_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const uint8_t* __restrict__ idx, T* __restrict__ dx, int C, int Hi,
                                                            int Wi, int Ho, int Wo, PoolGeo g, int64_t total) {
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
        // windows oy with oy*S - P <= iy <= oy*S - P + K - 1
        const int ty = iy + g.P - g.K + 1, tx = ix + g.P - g.K + 1;
        const int oy0 = ty <= 0 ? 0 : (ty + g.S - 1) / g.S, oy1 = (iy + g.P) / g.S;
        const int ox0 = tx <= 0 ? 0 : (tx + g.S - 1) / g.S, ox1 = (ix + g.P) / g.S;
        for (int oy = oy0; oy <= oy1; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                if (ox >= Wo) continue;
                const int me = (iy - (oy * g.S - g.P)) * g.K + (ix - (ox * g.S - g.P));
                int arg[VEC];
                const int64_t o = NHWC ? ((n * Ho + oy) * (int64_t)Wo + ox) * C + cv * VEC
                                       : ((n * C + cv) * (int64_t)Ho + oy) * Wo + ox;
                if (idx) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) arg[k] = idx[o + k];
                } else {
                    float best[VEC];
                    pool_window<T, VEC, NHWC>(x, n, cv, C, Hi, Wi, oy, ox, g, best, arg);
                }
                float gr[VEC];
                if constexpr (VEC == 1) gr[0] = Elt<T>::ld(dy + o);
                else Elt<T>::ldv(dy + o, reinterpret_cast<float(&)[Elt<T>::VEC]>(gr));
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    if (arg[k] == me) acc[k] += gr[k];
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
constexpr int AP_WAVES = 16;     // 1024 threads: 16 pixel rows of a 64-channel column block in flight
template <typename T, typename TP>
__global__ __launch_bounds__(64 * AP_WAVES) void avgpool_nhwc_kernel(const T* __restrict__ x, TP* __restrict__ y, int C, int64_t HW) {
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int64_t p = r;
        for (; p + AP_WAVES < HW; p += 2 * AP_WAVES) {       // two independent loads per iteration
            s0 += Elt<T>::ld(x + ((int64_t)n * HW + p) * C + c);
            s1 += Elt<T>::ld(x + ((int64_t)n * HW + p + AP_WAVES) * C + c);
        }
        if (p < HW) s0 += Elt<T>::ld(x + ((int64_t)n * HW + p) * C + c);
    }
    __shared__ float red[AP_WAVES][64];
    red[r][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (r == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < AP_WAVES; ++i) t += red[i][threadIdx.x];
        Elt<TP>::st(y + (int64_t)n * C + c, t / (float)HW);
    }
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

// y[m][co] = b[co] + sum_ci x[m][ci] * w[co][ci]; weights in LDS.  FOUR lanes per pixel: lane q takes the 16-byte pieces
// q, q + 4, ... of the pixel's row (the quad reads 64 contiguous bytes per step), partial sums meet by two shuffles.
template <typename T>
__global__ __launch_bounds__(BLOCK) void pointwise_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ y,
                                                              int64_t M, int Ci, int Co) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [Co][Ci]
    for (int i = threadIdx.x; i < Co * Ci; i += BLOCK) wl[i] = w[i];
    __syncthreads();
    constexpr int V = Elt<T>::VEC;
    const int64_t m = (int64_t)blockIdx.x * (BLOCK / 4) + (threadIdx.x >> 2);
    const int q = threadIdx.x & 3;
    const bool live = m < M;
    float acc[PW_MAX_CO];
#pragma unroll
    for (int o = 0; o < PW_MAX_CO; ++o) acc[o] = (o < Co && b && q == 0) ? b[o] : 0.f;
    const T* row = x + (live ? m : 0) * Ci;
#pragma unroll 2
    for (int c = q * V; c < Ci; c += 4 * V) {
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
    for (int o = 0; o < PW_MAX_CO; ++o) {
        if (o < Co) {
            float v = acc[o];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            if (live && (o & 3) == q) y[m * Co + o] = v;
        }
    }
}

// dx[m][ci] = sum_co dy[m][co] * w[co][ci]; same quad mapping (no reduction: every lane owns its output pieces)
template <typename T>
__global__ __launch_bounds__(BLOCK) void pointwise_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             T* __restrict__ dx, int64_t M, int Ci, int Co) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    for (int i = threadIdx.x; i < Co * Ci; i += BLOCK) wl[i] = w[i];
    __syncthreads();
    constexpr int V = Elt<T>::VEC;
    const int64_t m = (int64_t)blockIdx.x * (BLOCK / 4) + (threadIdx.x >> 2);
    const int q = threadIdx.x & 3;
    if (m >= M) return;
    float g[PW_MAX_CO];
#pragma unroll
    for (int o = 0; o < PW_MAX_CO; ++o) g[o] = o < Co ? dy[m * Co + o] : 0.f;
    T* row = dx + m * Ci;
    for (int c = q * V; c < Ci; c += 4 * V) {
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

// dw partials: persistent workgroups of FOUR waves walk 64-pixel slices, a wave takes 16 of the slice's rows (four
// independent latency chains per workgroup instead of one: 103 -> ~30 us at 33 282 x 256 -> 21); a lane owns FOUR
// consecutive input channels (one 8- or 16-byte load per row, each broadcast LDS read of a dy value feeds four FMAs), the
// slice's dy tile sits in LDS; the four waves' sums are added in wave order through LDS (deterministic);
// slab[blk][Co][Ci] + bslab[blk][Co], folded in two stages (PW_FOLD groups).
constexpr int PW_SLICE = 64;
constexpr int PW_FOLD = 16;
constexpr int PW_DW_WAVES = 4;
constexpr int PW_DW_THREADS = 64 * PW_DW_WAVES;
constexpr int PW_RED_O = 8;                         // outputs per cross-wave reduction round
template <typename T>
__global__ __launch_bounds__(PW_DW_THREADS) void pointwise_dw_kernel(const float* __restrict__ dy, const T* __restrict__ x,
                                                                     float* __restrict__ slab, float* __restrict__ bslab,
                                                                     int64_t M, int Ci, int Co) {
    __shared__ float g[PW_SLICE][PW_MAX_CO];
    __shared__ float red[PW_DW_WAVES - 1][PW_RED_O][256];
    constexpr int RW = PW_SLICE / PW_DW_WAVES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t slices = (M + PW_SLICE - 1) / PW_SLICE;
    float bsum = 0.f;
    for (int c0 = 0; c0 < Ci; c0 += 4 * 64) {
        const int c = c0 + 4 * lane;
        float acc[PW_MAX_CO][4];
#pragma unroll
        for (int o = 0; o < PW_MAX_CO; ++o)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[o][k] = 0.f;
        for (int64_t sl = blockIdx.x; sl < slices; sl += gridDim.x) {
            const int64_t m0 = sl * PW_SLICE;
            const int rows = (int)((M - m0) < PW_SLICE ? (M - m0) : PW_SLICE);
            __syncthreads();
            for (int i = threadIdx.x; i < PW_SLICE * Co; i += PW_DW_THREADS) {
                const int r = i / Co, o = i - r * Co;
                g[r][o] = r < rows ? dy[(m0 + r) * Co + o] : 0.f;
            }
            __syncthreads();
            if (c0 == 0 && (int)threadIdx.x < Co)
                for (int r = 0; r < rows; ++r) bsum += g[r][threadIdx.x];
            if (c < Ci) {
#pragma unroll
                for (int r0 = wave * RW; r0 < wave * RW + RW; r0 += 4) {
                    float xv[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (r0 + j < rows) {
                            if constexpr (sizeof(T) == 2) {
                                const u16x4 t = *reinterpret_cast<const u16x4*>(x + (m0 + r0 + j) * Ci + c);
#pragma unroll
                                for (int k = 0; k < 4; ++k) xv[j][k] = bf2f(t[k]);
                            } else {
                                const f32x4 t = *reinterpret_cast<const f32x4*>(x + (m0 + r0 + j) * Ci + c);
                                xv[j][0] = t.x; xv[j][1] = t.y; xv[j][2] = t.z; xv[j][3] = t.w;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) xv[j][k] = 0.f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int o = 0; o < PW_MAX_CO; ++o) {
                            if (o < Co) {
                                const float gv = g[r0 + j][o];
#pragma unroll
                                for (int k = 0; k < 4; ++k) acc[o][k] = fmaf(gv, xv[j][k], acc[o][k]);
                            }
                        }
                }
            }
        }
        // waves 1..3 hand their sums to wave 0 through LDS, PW_RED_O outputs per round; wave 0 adds them in wave order
#pragma unroll
        for (int o0 = 0; o0 < PW_MAX_CO; o0 += PW_RED_O) {
            if (o0 < Co) {
                __syncthreads();
                if (wave > 0) {
#pragma unroll
                    for (int oo = 0; oo < PW_RED_O; ++oo)
#pragma unroll
                        for (int k = 0; k < 4; ++k) red[wave - 1][oo][lane * 4 + k] = acc[o0 + oo][k];
                }
                __syncthreads();
                if (wave == 0 && c < Ci) {
#pragma unroll
                    for (int oo = 0; oo < PW_RED_O; ++oo) {
                        if (o0 + oo < Co) {
                            f32x4 v = {acc[o0 + oo][0], acc[o0 + oo][1], acc[o0 + oo][2], acc[o0 + oo][3]};
#pragma unroll
                            for (int w = 0; w < PW_DW_WAVES - 1; ++w) {
                                v.x += red[w][oo][lane * 4 + 0]; v.y += red[w][oo][lane * 4 + 1];
                                v.z += red[w][oo][lane * 4 + 2]; v.w += red[w][oo][lane * 4 + 3];
                            }
                            *reinterpret_cast<f32x4*>(slab + ((int64_t)blockIdx.x * Co + (o0 + oo)) * Ci + c) = v;
                        }
                    }
                }
            }
        }
    }
    if ((int)threadIdx.x < Co) bslab[(int64_t)blockIdx.x * Co + threadIdx.x] = bsum;
}
// stage A: partial[f][i] = sum over slabs g = f, f + PW_FOLD, ... ; i runs over the Co*Ci weights, then the Co biases
__global__ __launch_bounds__(BLOCK) void pointwise_dw_fold_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                                  float* __restrict__ part, int G, int Ci, int Co) {
    const int i = blockIdx.x * BLOCK + threadIdx.x, f = blockIdx.y, nw = Co * Ci;
    if (i >= nw + Co) return;
    float s = 0.f;
    if (i < nw) for (int g = f; g < G; g += PW_FOLD) s += slab[(int64_t)g * nw + i];
    else for (int g = f; g < G; g += PW_FOLD) s += bslab[(int64_t)g * Co + (i - nw)];
    part[(int64_t)f * (nw + Co) + i] = s;
}
__global__ __launch_bounds__(BLOCK) void pointwise_dw_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                    float* __restrict__ db, int Ci, int Co, int accumulate) {
    const int i = blockIdx.x * BLOCK + threadIdx.x, nw = Co * Ci;
    if (i >= nw + Co) return;
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < PW_FOLD; ++f) s += part[(int64_t)f * (nw + Co) + i];
    if (i < nw) dw[i] = accumulate ? dw[i] + s : s;
    else if (db) db[i - nw] = accumulate ? db[i - nw] + s : s;
}

// ---- fp32 linear layer on a handful of rows -------------------------------------------------------------------------------
// The ASPP pooling branch (_deeplab.py:152-163) works on ONE vector per image: with 2 images per GPU its BatchNorm sees two
// samples per channel and normalises their DIFFERENCE, which bf16 storage of the pooled vector would round away (measured:
// backbone gradients off by 30-60 %).  The branch therefore stays in fp32 from the pooled vector to the broadcast:
//   y[n][co] = sum_ci x[n][ci] * w[co][ci]      (n <= LIN_MAX_N rows, fp32 master weights)
constexpr int LIN_MAX_N = 16;
// NN = the row count rounded up to {2, 4, 8, 16}: a compile-time row loop without branches (as a run-time `if (n < N)` inside
// the channel loop every load sat behind its own scalar branch: ~96 dependent latencies, 28 us for 2 MB of weights); rows
// beyond N re-read row N - 1 and are not stored.  Each lane's own sum keeps its order.
template <int NN>
__global__ __launch_bounds__(BLOCK) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 float* __restrict__ y, int N, int Ci, int Co) {
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (co >= Co) return;
    float acc[NN];
    const float* xr[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        acc[n] = 0.f;
        xr[n] = x + (int64_t)(n < N ? n : N - 1) * Ci;
    }
    const float* wr = w + (int64_t)co * Ci;
#pragma unroll 4
    for (int c = lane; c < Ci; c += 64) {
        const float wv = wr[c];
#pragma unroll
        for (int n = 0; n < NN; ++n) acc[n] = fmaf(xr[n][c], wv, acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const float s = wave_sum(acc[n]);
        if (n < N && lane == 0) y[(int64_t)n * Co + co] = s;
    }
}
// dx[n][ci] = sum_co dy[n][co] * w[co][ci]
// block = 64 input channels x 16 waves; wave w sums output channels w, w + 16, ... (4 loads in flight), then the waves fold
constexpr int LDX_WAVES = 16;
__global__ __launch_bounds__(64 * LDX_WAVES) void linear_small_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                         float* __restrict__ dx, int N, int Ci, int Co) {
    extern __shared__ float g[];      // [N][Co] then [LDX_WAVES][N][64] partials
    float* part = g + N * Co;
    for (int i = threadIdx.x; i < N * Co; i += 64 * LDX_WAVES) g[i] = dy[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float acc[LIN_MAX_N];
#pragma unroll
    for (int n = 0; n < LIN_MAX_N; ++n) acc[n] = 0.f;
    if (c < Ci) {
#pragma unroll 4
        for (int co = wv_; co < Co; co += LDX_WAVES) {
            const float wv = w[(int64_t)co * Ci + c];
#pragma unroll
            for (int n = 0; n < LIN_MAX_N; ++n)
                if (n < N) acc[n] = fmaf(g[n * Co + co], wv, acc[n]);
        }
    }
#pragma unroll
    for (int n = 0; n < LIN_MAX_N; ++n)
        if (n < N) part[(wv_ * N + n) * 64 + lane] = acc[n];
    __syncthreads();
    for (int i = threadIdx.x; i < N * 64; i += 64 * LDX_WAVES) {
        const int n = i >> 6, l = i & 63;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < LDX_WAVES; ++q) s += part[(q * N + n) * 64 + l];
        if (blockIdx.x * 64 + l < Ci) dx[(int64_t)n * Ci + blockIdx.x * 64 + l] = s;
    }
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

// ---- backward of y = [relu](x * alpha[c] + beta[c] [+ res]) with CONSTANT per-channel coefficients: a frozen BatchNorm
// (Detection/model.py:27-35,46-47: eval mode, no parameter gradients) or a convolution's bias (+ ReLU) (rpn/
// region_proposal_network.py:19-22): g = relu-masked dy (mask from the stored y), dx = g * alpha[c], d_res = g.
template <typename T, int VEC, bool NHWC>
__global__ __launch_bounds__(BLOCK) void affine_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const float* __restrict__ alpha, T* __restrict__ dx, T* __restrict__ dres,
                                                           int C, int64_t HW, int relu, int64_t total) {
    const int CV = C / VEC;
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < total; v += (int64_t)gridDim.x * BLOCK) {
        const int c0 = NHWC ? (int)(v % CV) * VEC : (int)((v / HW) % C);
        float g[VEC], o[VEC];
        if constexpr (VEC == 1) g[0] = Elt<T>::ld(dy + v);
        else Elt<T>::ldv(dy + v * VEC, reinterpret_cast<float(&)[Elt<T>::VEC]>(g));
        if (relu) {
            if constexpr (VEC == 1) o[0] = Elt<T>::ld(y + v);
            else Elt<T>::ldv(y + v * VEC, reinterpret_cast<float(&)[Elt<T>::VEC]>(o));
#pragma unroll
            for (int k = 0; k < VEC; ++k) g[k] = (o[k] > 0.f) ? g[k] : 0.f;
        }
        if (dres) {
            if constexpr (VEC == 1) Elt<T>::st(dres + v, g[0]);
            else Elt<T>::stv(dres + v * VEC, reinterpret_cast<const float(&)[Elt<T>::VEC]>(g));
        }
        if (dx) {
            if (alpha) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) g[k] *= alpha[c0 + k];
            }
            if constexpr (VEC == 1) Elt<T>::st(dx + v, g[0]);
            else Elt<T>::stv(dx + v * VEC, reinterpret_cast<const float(&)[Elt<T>::VEC]>(g));
        }
    }
}

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

// ld > 0: y (forward) / dy (backward) is a channel slice of a wider channels-last tensor — its pixels are ld elements apart
// (the decoder's concat, Segmentation/network/_deeplab.py:54-56: the resized ASPP output is written straight into channels
// 48..303 of the 304-channel tensor and its gradient is read from there).  NHWC only; ld = 0: dense.
static int upsample_fwd_impl(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                             int64_t ho, int64_t wo, int64_t ld, afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0) return AFAN_ESHAPE;
    if (ld && (layout != AFAN_NHWC || ld < c || ld > 0x7fffffffLL)) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x || !y) return AFAN_ENULL;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, es) || !aligned(y, es)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {x, y}) : vec_for<uint16_t>(c, {x, y})) : 1;
    if (ld && (ld * es) % 16 != 0) vec = 1;
    const int64_t total = n * ho * wo * c / vec;
    const float sh = (float)hi / (float)ho, sw = (float)wi / (float)wo;
    AFAN_PROF("upsample_bilinear_fwd_kernel", (double)es * n * c * (ho * wo + hi * wi), st);
    const int64_t rows = layout == AFAN_NHWC ? n * ho : n * c * ho, per_row = layout == AFAN_NHWC ? wo * (c / vec) : wo;
    if (rows > 0x7fffffffLL || (per_row + BLOCK - 1) / BLOCK > 65535) return AFAN_ESHAPE;
    (void)total;
    const dim3 ugrid((unsigned)rows, (unsigned)((per_row + BLOCK - 1) / BLOCK));
#define K_(T, V, L, ...) upsample_fwd_kernel<T, V, L><<<ugrid, BLOCK, 0, st>>>((const T*)x, (T*)y, (int)c, (int)hi, (int)wi, (int)ho, (int)wo, sh, sw, (int)(ld ? ld : c))
    AFAN_SEG_DISPATCH(K_, 0);
#undef K_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

static int upsample_bwd_impl(const void* dy, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                             int64_t ho, int64_t wo, int64_t ld, afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0) return AFAN_ESHAPE;
    if (ld && (layout != AFAN_NHWC || ld < c || ld > 0x7fffffffLL)) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!dy || !dx) return AFAN_ENULL;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(dy, es) || !aligned(dx, es)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {dy, dx}) : vec_for<uint16_t>(c, {dy, dx})) : 1;
    if (ld && (ld * es) % 16 != 0) vec = 1;
    const int64_t total = n * hi * wi * c / vec;
    const float sh = (float)hi / (float)ho, sw = (float)wi / (float)wo;
    const float ish = (float)ho / (float)hi, isw = (float)wo / (float)wi;
    AFAN_PROF("upsample_bilinear_bwd_kernel", (double)es * n * c * (ho * wo + hi * wi), st);
    const int64_t rows = layout == AFAN_NHWC ? n * hi : n * c * hi, per_row = layout == AFAN_NHWC ? wi * (c / vec) : wi;
    if (rows > 0x7fffffffLL || (per_row + BLOCK - 1) / BLOCK > 65535) return AFAN_ESHAPE;
    (void)total;
    const dim3 ugrid((unsigned)rows, (unsigned)((per_row + BLOCK - 1) / BLOCK));
#define K_(T, V, L, ...) upsample_bwd_kernel<T, V, L><<<ugrid, BLOCK, 0, st>>>((const T*)dy, (T*)dx, (int)c, (int)hi, (int)wi, (int)ho, (int)wo, sh, sw, ish, isw, (int)(ld ? ld : c))
    AFAN_SEG_DISPATCH(K_, 0);
#undef K_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_upsample_bilinear_fwd(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                               int64_t ho, int64_t wo, afan_stream_t stream) {
    return upsample_fwd_impl(x, y, dtype, layout, n, c, hi, wi, ho, wo, 0, stream);
}

int afan_upsample_bilinear_bwd(const void* dy, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                               int64_t ho, int64_t wo, afan_stream_t stream) {
    return upsample_bwd_impl(dy, dx, dtype, layout, n, c, hi, wi, ho, wo, 0, stream);
}

int afan_upsample_bilinear_fwd_slice(const void* x, void* y, int dtype, int64_t n, int64_t c, int64_t hi, int64_t wi, int64_t ho,
                                     int64_t wo, int64_t ld, afan_stream_t stream) {
    if (ld <= 0) return AFAN_ESHAPE;
    return upsample_fwd_impl(x, y, dtype, AFAN_NHWC, n, c, hi, wi, ho, wo, ld, stream);
}

int64_t afan_upsample_bilinear_bwd_workspace_floats(int64_t n, int64_t c, int64_t wi, int64_t ho) {
    return (n > 0 && c > 0 && wi > 0 && ho > 0) ? n * ho * wi * c : 0;
}

int afan_upsample_bilinear_bwd_slice(const void* dy, void* dx, int dtype, int64_t n, int64_t c, int64_t hi, int64_t wi, int64_t ho,
                                     int64_t wo, int64_t ld, float* ws, afan_stream_t stream) {
    if (ld <= 0) return AFAN_ESHAPE;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    const int vec = 16 / es;
    // two separable passes through ws (afan_upsample_bilinear_bwd_workspace_floats) when every access is a 16-byte vector;
    // otherwise (or without a workspace) the one-pass gather
    if (!ws || (dtype != AFAN_F32 && dtype != AFAN_BF16) || c % vec || (ld * es) % 16 || !dy || !dx || !aligned(dy, 16) ||
        !aligned(dx, 16) || !aligned(ws, 16) || n <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0 || ld < c || ld > 0x7fffffffLL)
        return upsample_bwd_impl(dy, dx, dtype, AFAN_NHWC, n, c, hi, wi, ho, wo, ld, stream);
    hipStream_t st = (hipStream_t)stream;
    const int64_t per_row = wi * (c / vec);
    if (n * ho > 0x7fffffffLL || (per_row + BLOCK - 1) / BLOCK > 65535) return AFAN_ESHAPE;
    const float sh = (float)hi / (float)ho, sw = (float)wi / (float)wo;
    const float ish = (float)ho / (float)hi, isw = (float)wo / (float)wi;
    AFAN_PROF("upsample_bilinear_bwd_kernel", (double)es * n * c * (ho * wo + hi * wi) + 8.0 * n * ho * wi * c, st);
    const dim3 g1((unsigned)(n * ho), (unsigned)((per_row + BLOCK - 1) / BLOCK)), g2((unsigned)(n * hi), (unsigned)((per_row + BLOCK - 1) / BLOCK));
    if (dtype == AFAN_F32) {
        upsample_bwd_rows_kernel<float, 4><<<g1, BLOCK, 0, st>>>((const float*)dy, ws, (int)c, (int)wi, (int)ho, (int)wo, sw, isw, (int)ld);
        upsample_bwd_cols_kernel<float, 4><<<g2, BLOCK, 0, st>>>(ws, (float*)dx, (int)c, (int)hi, (int)wi, (int)ho, sh, ish);
    } else {
        upsample_bwd_rows_kernel<uint16_t, 8><<<g1, BLOCK, 0, st>>>((const uint16_t*)dy, ws, (int)c, (int)wi, (int)ho, (int)wo, sw, isw, (int)ld);
        upsample_bwd_cols_kernel<uint16_t, 8><<<g2, BLOCK, 0, st>>>(ws, (uint16_t*)dx, (int)c, (int)hi, (int)wi, (int)ho, sh, ish);
    }
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
    ce2d_fold_count_kernel<<<1, BLOCK, 0, st>>>(workspace, G);
    AFAN_LAUNCH_CHECK();
    if (layout == AFAN_NHWC) ce2d_kernel<true><<<G, BLOCK, (size_t)BLOCK * c * 4, st>>>(logits, target, dlogits, (int)c, hw, P, ignore_index, grad_scale, workspace, G);
    else ce2d_kernel<false><<<G, BLOCK, 0, st>>>(logits, target, dlogits, (int)c, hw, P, ignore_index, grad_scale, workspace, G);
    AFAN_LAUNCH_CHECK();
    ce2d_finalize_kernel<<<1, BLOCK, 0, st>>>(workspace, G, loss);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// host copy of src_index's i0 (same fp32 expression): the widest source window an output tile reads
static int host_src_i0(float scale, int dst, int in_size) {
    float s = fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (s < 0.f) s = 0.f;
    int i0 = (int)s;
    return i0 > in_size - 1 ? in_size - 1 : i0;
}
static int up_window(int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    int worst = 0;
    for (int y0 = 0; y0 < out_size; y0 += UP_OT) {
        const int y1 = (y0 + UP_OT < out_size ? y0 + UP_OT : out_size) - 1;
        const int s0 = host_src_i0(scale, y0, in_size), s1 = host_src_i0(scale, y1, in_size) + 1;
        if (s1 - s0 + 1 > worst) worst = s1 - s0 + 1;
    }
    return worst;
}

int64_t afan_ce2d_upsampled_workspace_floats(int64_t n, int64_t c, int64_t h, int64_t w, int64_t ho, int64_t wo) {
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || ho <= 0 || wo <= 0) return 0;
    const int64_t tiles = n * ((ho + UP_OT - 1) / UP_OT) * ((wo + UP_OT - 1) / UP_OT);
    return 1 + (int64_t)ce2d_blocks(n * ho * wo) + tiles + tiles * UP_SW * UP_SW * c;
}

int afan_ce2d_upsampled(const float* logits, const int64_t* target, int64_t n, int64_t c, int64_t h, int64_t w, int64_t ho,
                        int64_t wo, int64_t ignore_index, float grad_scale, float* workspace, float* loss, float* dlogits,
                        afan_stream_t stream) {
    if (n <= 0 || h <= 0 || w <= 0 || ho < h || wo < w || c <= 0 || c > CE_MAX_C) return AFAN_ESHAPE;
    if (!logits || !target || !workspace || !loss) return AFAN_ENULL;
    if (!aligned(logits, 4) || !aligned(target, 8) || !aligned(workspace, 4)) return AFAN_EALIGN;
    if (up_window((int)h, (int)ho) > UP_SW || up_window((int)w, (int)wo) > UP_SW) return AFAN_ESHAPE;   // (down-scaling: not this kernel)
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = n * ho * wo;
    const int Gc = ce2d_blocks(P);
    const int tx = (int)((wo + UP_OT - 1) / UP_OT), ty = (int)((ho + UP_OT - 1) / UP_OT);
    const int G2 = (int)(n * tx * ty);
    float* loss_part = workspace + 1 + Gc;
    float* part = loss_part + G2;
    const size_t lds = ((size_t)BLOCK * c + (size_t)UP_SW * UP_OT * c) * 4;      // gt + max(source window, row-folded T)
    AFAN_PROF("ce2d_upsampled_kernel", (double)P * 8.0 + 4.0 * n * h * w * c * (dlogits ? 2 : 1), st);
    ce2d_count_kernel<<<Gc, BLOCK, 0, st>>>(target, P, ignore_index, workspace + 1);
    AFAN_LAUNCH_CHECK();
    ce2d_fold_count_kernel<<<1, BLOCK, 0, st>>>(workspace, Gc);
    AFAN_LAUNCH_CHECK();
    const float sh = (float)h / (float)ho, sw = (float)w / (float)wo, ish = (float)ho / (float)h, isw = (float)wo / (float)w;
    ce2d_up_kernel<<<G2, BLOCK, lds, st>>>(logits, target, part, (int)c, (int)h, (int)w, (int)ho, (int)wo, sh, sw, ignore_index,
                                           grad_scale, workspace, loss_part, tx, ty, dlogits ? 1 : 0);
    AFAN_LAUNCH_CHECK();
    if (dlogits) {
        const int64_t total = n * h * w * c;
        ce2d_up_gather_kernel<<<grid_for(total, BLOCK, 4096), BLOCK, 0, st>>>(part, dlogits, (int)c, (int)h, (int)w, (int)ho, (int)wo, sh,
                                                                             sw, ish, isw, tx, ty, total);
        AFAN_LAUNCH_CHECK();
    }
    ce2d_up_finalize_kernel<<<1, BLOCK, 0, st>>>(workspace, loss_part, G2, loss);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                       int k, int stride, int pad, afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hi <= 0 || wi <= 0 || k < 1 || k > 15 || stride < 1 || pad < 0 || 2 * pad > k) return AFAN_ESHAPE;
    if (hi + 2 * pad < k || wi + 2 * pad < k) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x || !y) return AFAN_ENULL;
    const int64_t ho = (hi + 2 * pad - k) / stride + 1, wo = (wi + 2 * pad - k) / stride + 1;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    hipStream_t st = (hipStream_t)stream;
    const int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {x, y}) : vec_for<uint16_t>(c, {x, y})) : 1;
    const int64_t total = n * ho * wo * c / vec;
    const PoolGeo g{k, stride, pad};
    AFAN_PROF("maxpool_fwd_kernel", (double)es * n * c * (ho * wo + hi * wi) + (idx ? (double)n * c * ho * wo : 0.0), st);
#define K_(T, V, L, ...) maxpool_fwd_kernel<T, V, L><<<grid_for(total, BLOCK, 4096), BLOCK, 0, st>>>((const T*)x, (T*)y, idx, (int)c, (int)hi, (int)wi, (int)ho, (int)wo, g, total)
    AFAN_SEG_DISPATCH(K_, 0);
#undef K_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_maxpool2d_bwd(const void* dy, const void* x, const uint8_t* idx, void* dx, int dtype, int layout, int64_t n, int64_t c,
                       int64_t hi, int64_t wi, int k, int stride, int pad, afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hi <= 0 || wi <= 0 || k < 1 || k > 15 || stride < 1 || pad < 0 || 2 * pad > k) return AFAN_ESHAPE;
    if (hi + 2 * pad < k || wi + 2 * pad < k) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!dy || !dx || (!x && !idx)) return AFAN_ENULL;
    const int64_t ho = (hi + 2 * pad - k) / stride + 1, wo = (wi + 2 * pad - k) / stride + 1;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    hipStream_t st = (hipStream_t)stream;
    const int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {dy, x, dx}) : vec_for<uint16_t>(c, {dy, x, dx})) : 1;
    const int64_t total = n * hi * wi * c / vec;
    const PoolGeo g{k, stride, pad};
    AFAN_PROF("maxpool_bwd_kernel", (double)es * n * c * (ho * wo + hi * wi) + (idx ? (double)n * c * ho * wo : (double)es * n * c * hi * wi), st);
#define K_(T, V, L, ...) maxpool_bwd_kernel<T, V, L><<<grid_for(total, BLOCK, 4096), BLOCK, 0, st>>>((const T*)dy, (const T*)x, idx, (T*)dx, (int)c, (int)hi, (int)wi, (int)ho, (int)wo, g, total)
    AFAN_SEG_DISPATCH(K_, 0);
#undef K_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_maxpool3x3s2_fwd(const void* x, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hi, int64_t wi,
                          afan_stream_t stream) {
    return afan_maxpool2d_fwd(x, y, nullptr, dtype, layout, n, c, hi, wi, 3, 2, 1, stream);
}

int afan_maxpool3x3s2_bwd(const void* dy, const void* x, void* dx, int dtype, int layout, int64_t n, int64_t c, int64_t hi,
                          int64_t wi, afan_stream_t stream) {
    return afan_maxpool2d_bwd(dy, x, nullptr, dx, dtype, layout, n, c, hi, wi, 3, 2, 1, stream);
}

int afan_affine_relu_bwd(const void* dy, const void* y, const float* alpha, void* dx, void* dres, int dtype, int layout,
                         int64_t n, int64_t c, int64_t hw, int relu, afan_stream_t stream) {
    int e = check_t(dtype, layout);
    if (e) return e;
    if (n < 0 || c <= 0 || hw <= 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!dy || (relu && !y) || (!dx && !dres)) return AFAN_ENULL;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    hipStream_t st = (hipStream_t)stream;
    const int vec = layout == AFAN_NHWC ? (dtype == AFAN_F32 ? vec_for<float>(c, {dy, y, dx, dres}) : vec_for<uint16_t>(c, {dy, y, dx, dres})) : 1;
    const int64_t total = n * hw * c / vec;
    AFAN_PROF("affine_bwd_kernel", (double)es * n * c * hw * (1 + (relu ? 1 : 0) + (dx ? 1 : 0) + (dres ? 1 : 0)), st);
#define K_(T, V, L, ...) affine_bwd_kernel<T, V, L><<<grid_for(total, BLOCK, 4096), BLOCK, 0, st>>>((const T*)dy, (const T*)y, alpha, (T*)dx, (T*)dres, (int)c, hw, relu, total)
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
        if (dtype == AFAN_F32) avgpool_nhwc_kernel<float, float><<<grid, 64 * AP_WAVES, 0, st>>>((const float*)x, (float*)y, (int)c, hw);
        else if (pf) avgpool_nhwc_kernel<uint16_t, float><<<grid, 64 * AP_WAVES, 0, st>>>((const uint16_t*)x, (float*)y, (int)c, hw);
        else avgpool_nhwc_kernel<uint16_t, uint16_t><<<grid, 64 * AP_WAVES, 0, st>>>((const uint16_t*)x, (uint16_t*)y, (int)c, hw);
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
    const unsigned g = (unsigned)((m + BLOCK / 4 - 1) / (BLOCK / 4));
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
    const unsigned g = (unsigned)((m + BLOCK / 4 - 1) / (BLOCK / 4));
    const size_t lds = (size_t)ci * co * 4;
    if (dx_dtype == AFAN_F32) pointwise_dx_kernel<float><<<g, BLOCK, lds, st>>>(dy, w, (float*)dx, m, (int)ci, (int)co);
    else pointwise_dx_kernel<uint16_t><<<g, BLOCK, lds, st>>>(dy, w, (uint16_t*)dx, m, (int)ci, (int)co);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

static int pw_blocks(int64_t m) {
    const int64_t slices = (m + PW_SLICE - 1) / PW_SLICE;
    return (int)(slices < 1024 ? slices : 1024);
}

int64_t afan_pointwise_workspace_floats(int64_t m, int64_t ci, int64_t co) {
    if (m <= 0 || ci <= 0 || co <= 0) return 0;
    return (int64_t)(pw_blocks(m) + PW_FOLD) * co * (ci + 1);
}

int afan_pointwise_bwd_dw(const float* dy, const void* x, int x_dtype, float* dw, float* db, int64_t m, int64_t ci,
                          int64_t co, float* workspace, int accumulate, afan_stream_t stream) {
    int e = pw_check(x_dtype, m, ci, co);
    if (e) return e;
    if (!dy || !x || !dw || !workspace) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const int G = pw_blocks(m);
    float* bslab = workspace + (int64_t)G * co * ci;
    float* part = bslab + (int64_t)G * co;
    const int es = x_dtype == AFAN_F32 ? 4 : 2;
    AFAN_PROF("pointwise_dw_kernel", (double)m * (es * ci + 4.0 * co) + 8.0 * G * co * ci, st);
    if (x_dtype == AFAN_F32) pointwise_dw_kernel<float><<<G, PW_DW_THREADS, 0, st>>>(dy, (const float*)x, workspace, bslab, m, (int)ci, (int)co);
    else pointwise_dw_kernel<uint16_t><<<G, PW_DW_THREADS, 0, st>>>(dy, (const uint16_t*)x, workspace, bslab, m, (int)ci, (int)co);
    AFAN_LAUNCH_CHECK();
    const unsigned gx = (unsigned)((co * ci + co + BLOCK - 1) / BLOCK);
    pointwise_dw_fold_kernel<<<dim3(gx, PW_FOLD), BLOCK, 0, st>>>(workspace, bslab, part, G, (int)ci, (int)co);
    AFAN_LAUNCH_CHECK();
    pointwise_dw_reduce_kernel<<<gx, BLOCK, 0, st>>>(part, dw, db, (int)ci, (int)co, accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_linear_small_max_rows(void) { return LIN_MAX_N; }

int afan_linear_small_fwd(const float* x, const float* w, float* y, int64_t n, int64_t ci, int64_t co, afan_stream_t stream) {
    if (n <= 0 || n > LIN_MAX_N || ci <= 0 || co <= 0) return AFAN_ESHAPE;
    if (!x || !w || !y) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("linear_small_fwd_kernel", 4.0 * (ci * co + n * (ci + co)), st);
    const unsigned lg = (unsigned)((co + 3) / 4);
    if (n <= 2) linear_small_fwd_kernel<2><<<lg, BLOCK, 0, st>>>(x, w, y, (int)n, (int)ci, (int)co);
    else if (n <= 4) linear_small_fwd_kernel<4><<<lg, BLOCK, 0, st>>>(x, w, y, (int)n, (int)ci, (int)co);
    else if (n <= 8) linear_small_fwd_kernel<8><<<lg, BLOCK, 0, st>>>(x, w, y, (int)n, (int)ci, (int)co);
    else linear_small_fwd_kernel<LIN_MAX_N><<<lg, BLOCK, 0, st>>>(x, w, y, (int)n, (int)ci, (int)co);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_linear_small_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, int64_t n, int64_t ci,
                          int64_t co, int accumulate, afan_stream_t stream) {
    if (n <= 0 || n > LIN_MAX_N || ci <= 0 || co <= 0 || (n * co + LDX_WAVES * n * 64) * 4 > 64 * 1024) return AFAN_ESHAPE;
    if (!dy || !w || !x) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("linear_small_bwd_kernel", 4.0 * (2 * ci * co + 2 * n * (ci + co)), st);
    if (dx) {
        linear_small_dx_kernel<<<(unsigned)((ci + 63) / 64), 64 * LDX_WAVES, (size_t)(n * co + LDX_WAVES * n * 64) * 4, st>>>(dy, w, dx, (int)n, (int)ci, (int)co);
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
