// mix_feature (the "N" of A-FAN) and SAT sample points for gfx950.
// Reference behaviour: Segmentation/attack_algo.py:108-118 (get_sample_points) and :121-130
// (mix_feature) == Detection/attack_algo.py:236-265.  Statistics run over the CHANNEL dimension per
// pixel, so in NCHW the reduction strides by H*W: lanes run along the pixel index (coalesced),
// each thread walks a subset of the channels, the clean column is parked in LDS so HBM sees the
// algorithmic 12 B/elt (read clean, read adv, write out) instead of eager PyTorch's ~48 B/elt.
#include "afan_common.h"

using namespace afan;

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_LDS_BYTES = 128 * 1024;

// Workgroup = PIX consecutive pixels of one sample x all C channels.  Thread (grp, px): px = tid % PIX,
// grp = tid / PIX walks channels grp, grp+G, ...  (G = BLOCK / PIX).
template <typename T, bool STAGE>
__global__ __launch_bounds__(BLOCK) void mix_feature_kernel(const T* __restrict__ clean,
                                                            const T* __restrict__ adv, T* __restrict__ out,
                                                            int C, int64_t HW, int PIX, int tiles_per_sample,
                                                            float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = BLOCK / PIX;
    float* stats = lds;                  // [2][G][PIX][3] partial moments, then [4][PIX] final stats
    float* tile = lds + 2 * G * PIX * 3; // [C][PIX] staged clean values (STAGE only)
    const int px = threadIdx.x % PIX, grp = threadIdx.x / PIX;
    const int64_t n = blockIdx.x / tiles_per_sample;
    const int64_t p0 = (int64_t)(blockIdx.x % tiles_per_sample) * PIX;
    const int64_t p = p0 + px;
    const bool live = p < HW;
    const int64_t base = n * (int64_t)C * HW + p;

    // pass 1: shifted sums over this thread's channels
    float sh_c = 0.f, s_c = 0.f, q_c = 0.f, sh_a = 0.f, s_a = 0.f, q_a = 0.f, cnt = 0.f;
    if (live) {
        bool first = true;
#pragma unroll 4
        for (int c = grp; c < C; c += G) {
            const float vc = Elt<T>::ld(clean + base + (int64_t)c * HW);
            const float va = Elt<T>::ld(adv + base + (int64_t)c * HW);
            if (STAGE) tile[c * PIX + px] = vc;
            if (first) { sh_c = vc; sh_a = va; first = false; }
            const float dc = vc - sh_c, da = va - sh_a;
            s_c += dc; q_c += dc * dc;
            s_a += da; q_a += da * da;
            cnt += 1.f;
        }
    }
    Moments mc{cnt, 0.f, 0.f}, ma{cnt, 0.f, 0.f};
    if (cnt > 0.f) {
        const float dmc = s_c / cnt, dma = s_a / cnt;
        mc.mean = sh_c + dmc; mc.m2 = fmaxf(q_c - s_c * dmc, 0.f);
        ma.mean = sh_a + dma; ma.m2 = fmaxf(q_a - s_a * dma, 0.f);
    }
    float* pc = stats + ((0 * G + grp) * PIX + px) * 3;
    float* pa = stats + ((1 * G + grp) * PIX + px) * 3;
    pc[0] = mc.n; pc[1] = mc.mean; pc[2] = mc.m2;
    pa[0] = ma.n; pa[1] = ma.mean; pa[2] = ma.m2;
    __syncthreads();
    float mean_c = 0.f, std_c = 1.f, mean_a = 0.f, std_a = 1.f;
    {
        // every thread folds its pixel's G partials (identical order => identical result in all groups)
        Moments tc{0.f, 0.f, 0.f}, ta{0.f, 0.f, 0.f};
        for (int g = 0; g < G; ++g) {
            const float* qc = stats + ((0 * G + g) * PIX + px) * 3;
            const float* qa = stats + ((1 * G + g) * PIX + px) * 3;
            tc = merge(tc, Moments{qc[0], qc[1], qc[2]});
            ta = merge(ta, Moments{qa[0], qa[1], qa[2]});
        }
        const float denom = (float)C - 1.0f;  // unbiased: torch.var default (C == 1 -> NaN, like the reference)
        mean_c = tc.mean; std_c = sqrtf(tc.m2 / denom + eps);
        mean_a = ta.mean; std_a = sqrtf(ta.m2 / denom + eps);
    }
    if (!live) return;
    // pass 2: (clean - mean_c) / std_c * std_a + mean_a, op by op as attack_algo.py:128-129
#pragma unroll 4
    for (int c = grp; c < C; c += G) {
        const float vc = STAGE ? tile[c * PIX + px] : Elt<T>::ld(clean + base + (int64_t)c * HW);
        float t = (vc - mean_c) / std_c;
        t = t * std_a;
        t = t + mean_a;
        Elt<T>::st(out + base + (int64_t)c * HW, t);
    }
}

// Channels-last variant: a pixel's C channels are contiguous, so ONE WAVE owns one pixel: lanes stride over the
// channels (coalesced 2/4-byte accesses; 16-byte vectors when C % (64 * VEC) == 0), the clean values of the lane stay in
// registers between the two passes (C <= 64 * KEEP) or are re-read, moments merge across lanes by butterfly (Chan).
template <typename T, int KEEP>
__global__ __launch_bounds__(BLOCK) void mix_feature_nhwc_kernel(const T* __restrict__ clean, const T* __restrict__ adv,
                                                                 T* __restrict__ out, int C, int64_t pixels, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (BLOCK / AFAN_WAVE) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (BLOCK / AFAN_WAVE);
    for (int64_t p = wave0; p < pixels; p += nwaves) {
        const T* pc = clean + p * C;
        const T* pa = adv + p * C;
        // keep[] is indexed by unrolled loop counters only.  (Round 1 indexed it with the run-time channel counter behind
        // an `if (k < KEEP)`: the compiler turned that into relative register addressing, and with C = 6000 — counter up
        // to 93 — the launch died with a memory-aperture violation.  Found by the fused kernel's test.)
        float keep[KEEP];
        float sh_c = 0.f, s_c = 0.f, q_c = 0.f, sh_a = 0.f, s_a = 0.f, q_a = 0.f, cnt = 0.f;
        bool first = true;
        auto take = [&](float vc, float va) {
            if (first) { sh_c = vc; sh_a = va; first = false; }
            const float dc = vc - sh_c, da = va - sh_a;
            s_c += dc; q_c += dc * dc;
            s_a += da; q_a += da * da;
            cnt += 1.f;
        };
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int c = lane + k * AFAN_WAVE;
            keep[k] = 0.f;
            if (c < C) {
                keep[k] = Elt<T>::ld(pc + c);
                take(keep[k], Elt<T>::ld(pa + c));
            }
        }
        for (int c = lane + KEEP * AFAN_WAVE; c < C; c += AFAN_WAVE) take(Elt<T>::ld(pc + c), Elt<T>::ld(pa + c));
        Moments mc{cnt, 0.f, 0.f}, ma{cnt, 0.f, 0.f};
        if (cnt > 0.f) {
            const float dmc = s_c / cnt, dma = s_a / cnt;
            mc.mean = sh_c + dmc; mc.m2 = fmaxf(q_c - s_c * dmc, 0.f);
            ma.mean = sh_a + dma; ma.m2 = fmaxf(q_a - s_a * dma, 0.f);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Moments oc{__shfl_xor(mc.n, o, 64), __shfl_xor(mc.mean, o, 64), __shfl_xor(mc.m2, o, 64)};
            Moments oa{__shfl_xor(ma.n, o, 64), __shfl_xor(ma.mean, o, 64), __shfl_xor(ma.m2, o, 64)};
            // merge in a lane-symmetric order so that every lane ends with the same bits
            const bool lo = (lane & o) == 0;
            mc = lo ? merge(mc, oc) : merge(oc, mc);
            ma = lo ? merge(ma, oa) : merge(oa, ma);
        }
        const float denom = (float)C - 1.0f;
        const float mean_c = mc.mean, std_c = sqrtf(mc.m2 / denom + eps);
        const float mean_a = ma.mean, std_a = sqrtf(ma.m2 / denom + eps);
        T* po = out + p * C;
        auto put = [&](int c, float vc) {
            float t = (vc - mean_c) / std_c;
            t = t * std_a;
            t = t + mean_a;
            Elt<T>::st(po + c, t);
        };
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int c = lane + k * AFAN_WAVE;
            if (c < C) put(c, keep[k]);
        }
        for (int c = lane + KEEP * AFAN_WAVE; c < C; c += AFAN_WAVE) put(c, Elt<T>::ld(pc + c));
    }
}

template <typename T>
int mix_nhwc_impl(const void* clean, const void* adv, void* out, int64_t pixels, int64_t c, float eps, hipStream_t st) {
    const int grid = grid_for(pixels * AFAN_WAVE, BLOCK, 8192);
    AFAN_PROF("mix_feature_nhwc_kernel", 3.0 * sizeof(T) * pixels * c, st);
    if (c <= 64 * 8) mix_feature_nhwc_kernel<T, 8><<<grid, BLOCK, 0, st>>>((const T*)clean, (const T*)adv, (T*)out, (int)c, pixels, eps);
    else mix_feature_nhwc_kernel<T, 20><<<grid, BLOCK, 0, st>>>((const T*)clean, (const T*)adv, (T*)out, (int)c, pixels, eps);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

struct LerpW {
    float w[8];
};

__global__ __launch_bounds__(BLOCK) void lerp_points_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ y,
                                                            float* __restrict__ out, int64_t n, LerpW lw,
                                                            int k_int, int vec) {
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    int64_t done = 0;
    if (vec) {
        const int64_t nvec = n >> 2;
        for (int64_t v = tid; v < nvec; v += nthreads) {
            const int64_t i = v << 2;
            float a[4], b[4];
            Elt<float>::ldv(x + i, a);
            Elt<float>::ldv(y + i, b);
            for (int k = 0; k < k_int; ++k) {
                const float w = lw.w[k];
                const bool small = fabsf(w) < 0.5f;  // ATen lerp: two forms around 0.5
                const float coeff = small ? w : w - 1.0f;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(coeff, b[e] - a[e], small ? a[e] : b[e]);
                Elt<float>::stv(out + (int64_t)k * n + i, o);
            }
        }
        done = nvec << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        const float a = x[i], b = y[i];
        for (int k = 0; k < k_int; ++k) {
            const float w = lw.w[k];
            const bool small = fabsf(w) < 0.5f;
            out[(int64_t)k * n + i] = fmaf(small ? w : w - 1.0f, b - a, small ? a : b);
        }
    }
}


// ---- fused SAT sample points + mix_feature (SURVEY 8(a) a11 "fuse with a10") ----------------------------------------
// Points j = 1 .. npts of get_sample_points(clean, adv, npts + 1) — the interior points lerp(clean, adv, w_j) and the end
// point adv itself — each optionally re-normalised by mix_feature(clean, point_j) (bit j-1 of `mask`), from ONE read of
// clean and adv: the Segmentation step's lerp + 2 x mix (main_aug_final.py:186-192) moves 9 tensors, this moves 4.
// Arithmetic and summation order are those of lerp_points_kernel / mix_feature*_kernel, so the results are bit-identical
// to the separate launches.  An end point without its mask bit is not written (the caller keeps adv).
constexpr int LM_MAX = 4;
__device__ __forceinline__ float lerp_one(float a, float b, float w) {
    const bool small = fabsf(w) < 0.5f;
    return fmaf(small ? w : w - 1.0f, b - a, small ? a : b);
}

template <int KEEP>
__global__ __launch_bounds__(BLOCK) void lerp_mix_nhwc_kernel(const float* __restrict__ clean, const float* __restrict__ adv,
                                                              float* __restrict__ out, int C, int64_t pixels, int64_t total,
                                                              LerpW lw, int npts, unsigned mask, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (BLOCK / AFAN_WAVE) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (BLOCK / AFAN_WAVE);
    for (int64_t p = wave0; p < pixels; p += nwaves) {
        const float* pc = clean + p * C;
        const float* pa = adv + p * C;
        float keep[KEEP];       // (indexed by unrolled loop counters only: a run-time index would put it in scratch memory)
        float sh_c = 0.f, s_c = 0.f, q_c = 0.f, cnt = 0.f;
        float sh[LM_MAX], sm[LM_MAX], sq[LM_MAX];
#pragma unroll
        for (int k = 0; k < LM_MAX; ++k) sh[k] = sm[k] = sq[k] = 0.f;
        bool first = true;
        auto take = [&](float vc, float vy) {
            if (first) sh_c = vc;
            const float dc = vc - sh_c;
            s_c += dc; q_c += dc * dc;
#pragma unroll
            for (int k = 0; k < LM_MAX; ++k)
                if (k < npts) {
                    const float v = k == npts - 1 ? vy : lerp_one(vc, vy, lw.w[k]);
                    if (first) sh[k] = v;
                    const float d = v - sh[k];
                    sm[k] += d; sq[k] += d * d;
                }
            first = false;
            cnt += 1.f;
        };
#pragma unroll
        for (int kk = 0; kk < KEEP; ++kk) {
            const int c = lane + kk * AFAN_WAVE;
            keep[kk] = 0.f;
            if (c < C) {
                const float vc = pc[c];
                keep[kk] = vc;
                take(vc, pa[c]);
            }
        }
        for (int c = lane + KEEP * AFAN_WAVE; c < C; c += AFAN_WAVE) take(pc[c], pa[c]);
        Moments mc{cnt, 0.f, 0.f}, mk[LM_MAX];
        if (cnt > 0.f) {
            const float dm = s_c / cnt;
            mc.mean = sh_c + dm; mc.m2 = fmaxf(q_c - s_c * dm, 0.f);
        }
#pragma unroll
        for (int k = 0; k < LM_MAX; ++k) {
            mk[k] = Moments{cnt, 0.f, 0.f};
            if (cnt > 0.f) {
                const float dm = sm[k] / cnt;
                mk[k].mean = sh[k] + dm; mk[k].m2 = fmaxf(sq[k] - sm[k] * dm, 0.f);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const bool lo = (lane & o) == 0;
            Moments oc{__shfl_xor(mc.n, o, 64), __shfl_xor(mc.mean, o, 64), __shfl_xor(mc.m2, o, 64)};
            mc = lo ? merge(mc, oc) : merge(oc, mc);
#pragma unroll
            for (int k = 0; k < LM_MAX; ++k) {
                Moments ok{__shfl_xor(mk[k].n, o, 64), __shfl_xor(mk[k].mean, o, 64), __shfl_xor(mk[k].m2, o, 64)};
                mk[k] = lo ? merge(mk[k], ok) : merge(ok, mk[k]);
            }
        }
        const float denom = (float)C - 1.0f;
        const float mean_c = mc.mean, std_c = sqrtf(mc.m2 / denom + eps);
        auto put = [&](int c, float vc, float vy) {
#pragma unroll
            for (int k = 0; k < LM_MAX; ++k)
                if (k < npts) {
                    float* po = out + (int64_t)k * total + p * C;
                    if ((mask >> k) & 1u) {
                        float t = (vc - mean_c) / std_c;
                        t = t * sqrtf(mk[k].m2 / denom + eps);
                        t = t + mk[k].mean;
                        po[c] = t;
                    } else if (k < npts - 1) {
                        po[c] = lerp_one(vc, vy, lw.w[k]);
                    }
                }
        };
#pragma unroll
        for (int kk = 0; kk < KEEP; ++kk) {
            const int c = lane + kk * AFAN_WAVE;
            if (c < C) put(c, keep[kk], pa[c]);
        }
        for (int c = lane + KEEP * AFAN_WAVE; c < C; c += AFAN_WAVE) put(c, pc[c], pa[c]);
    }
}

// NCHW: the workgroup / thread mapping of mix_feature_kernel (PIX pixels of one sample, channel groups), clean column staged.
template <bool STAGE>
__global__ __launch_bounds__(BLOCK) void lerp_mix_kernel(const float* __restrict__ clean, const float* __restrict__ adv,
                                                         float* __restrict__ out, int C, int64_t HW, int PIX, int tiles_per_sample,
                                                         int64_t total, LerpW lw, int npts, unsigned mask, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int G = BLOCK / PIX;
    float* stats = lds;                                    // [1 + LM_MAX][G][PIX][3]
    float* tile = lds + (1 + LM_MAX) * G * PIX * 3;        // [C][PIX]
    const int px = threadIdx.x % PIX, grp = threadIdx.x / PIX;
    const int64_t n = blockIdx.x / tiles_per_sample;
    const int64_t p = (int64_t)(blockIdx.x % tiles_per_sample) * PIX + px;
    const bool live = p < HW;
    const int64_t base = n * (int64_t)C * HW + p;
    float sh_c = 0.f, s_c = 0.f, q_c = 0.f, cnt = 0.f;
    float sh[LM_MAX], sm[LM_MAX], sq[LM_MAX];
#pragma unroll
    for (int k = 0; k < LM_MAX; ++k) sh[k] = sm[k] = sq[k] = 0.f;
    if (live) {
        bool first = true;
        for (int c = grp; c < C; c += G) {
            const float vc = clean[base + (int64_t)c * HW], vy = adv[base + (int64_t)c * HW];
            if (STAGE) tile[c * PIX + px] = vc;
            if (first) sh_c = vc;
            const float dc = vc - sh_c;
            s_c += dc; q_c += dc * dc;
#pragma unroll
            for (int k = 0; k < LM_MAX; ++k)
                if (k < npts) {
                    const float v = k == npts - 1 ? vy : lerp_one(vc, vy, lw.w[k]);
                    if (first) sh[k] = v;
                    const float d = v - sh[k];
                    sm[k] += d; sq[k] += d * d;
                }
            first = false;
            cnt += 1.f;
        }
    }
    {
        Moments m{cnt, 0.f, 0.f};
        if (cnt > 0.f) { const float dm = s_c / cnt; m.mean = sh_c + dm; m.m2 = fmaxf(q_c - s_c * dm, 0.f); }
        float* q = stats + ((0 * G + grp) * PIX + px) * 3;
        q[0] = m.n; q[1] = m.mean; q[2] = m.m2;
#pragma unroll
        for (int k = 0; k < LM_MAX; ++k) {
            Moments mk{cnt, 0.f, 0.f};
            if (cnt > 0.f) { const float dm = sm[k] / cnt; mk.mean = sh[k] + dm; mk.m2 = fmaxf(sq[k] - sm[k] * dm, 0.f); }
            float* qk = stats + (((1 + k) * G + grp) * PIX + px) * 3;
            qk[0] = mk.n; qk[1] = mk.mean; qk[2] = mk.m2;
        }
    }
    __syncthreads();
    const float denom = (float)C - 1.0f;
    float mean_c, std_c, mean_k[LM_MAX], std_k[LM_MAX];
    {
        Moments t{0.f, 0.f, 0.f};
        for (int g = 0; g < G; ++g) { const float* q = stats + ((0 * G + g) * PIX + px) * 3; t = merge(t, Moments{q[0], q[1], q[2]}); }
        mean_c = t.mean; std_c = sqrtf(t.m2 / denom + eps);
#pragma unroll
        for (int k = 0; k < LM_MAX; ++k) {
            Moments tk{0.f, 0.f, 0.f};
            for (int g = 0; g < G; ++g) { const float* q = stats + (((1 + k) * G + g) * PIX + px) * 3; tk = merge(tk, Moments{q[0], q[1], q[2]}); }
            mean_k[k] = tk.mean; std_k[k] = sqrtf(tk.m2 / denom + eps);
        }
    }
    if (!live) return;
    for (int c = grp; c < C; c += G) {
        const float vc = STAGE ? tile[c * PIX + px] : clean[base + (int64_t)c * HW];
        const float vy = adv[base + (int64_t)c * HW];
#pragma unroll
        for (int k = 0; k < LM_MAX; ++k)
            if (k < npts) {
                float* po = out + (int64_t)k * total + base + (int64_t)c * HW;
                if ((mask >> k) & 1u) {
                    float t = (vc - mean_c) / std_c;
                    t = t * std_k[k];
                    t = t + mean_k[k];
                    *po = t;
                } else if (k < npts - 1) {
                    *po = lerp_one(vc, vy, lw.w[k]);
                }
            }
    }
}

template <typename T>
int mix_impl(const void* clean, const void* adv, void* out, int64_t n, int64_t c, int64_t hw, float eps,
             hipStream_t st) {
    // largest power-of-two pixel tile (<= 64) whose staged clean column fits the LDS budget
    int pix = 64;
    while (pix >= 8 && (int64_t)c * pix * 4 + 2 * (BLOCK / pix) * pix * 3 * 4 > 64 * 1024) pix >>= 1;
    bool stage = pix >= 8;
    if (!stage) {
        pix = 64;
        while (pix >= 8 && (int64_t)c * pix * 4 + 2 * (BLOCK / pix) * pix * 3 * 4 > MAX_LDS_BYTES) pix >>= 1;
        stage = pix >= 8;
        if (!stage) pix = 64;
    }
    while (pix > 8 && pix / 2 >= hw) pix >>= 1;  // tiny planes: do not waste lanes
    const int tiles = (int)((hw + pix - 1) / pix);
    const int64_t blocks = n * tiles;
    if (blocks > 0x7fffffffLL) return AFAN_ESHAPE;
    const size_t stats_bytes = (size_t)2 * (BLOCK / pix) * pix * 3 * 4;
    const size_t lds = stats_bytes + (stage ? (size_t)c * pix * 4 : 0);
    AFAN_PROF("mix_feature_kernel", 3.0 * sizeof(T) * n * c * hw, st);
    if (stage) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)mix_feature_kernel<T, true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        mix_feature_kernel<T, true><<<(unsigned)blocks, BLOCK, lds, st>>>(
            (const T*)clean, (const T*)adv, (T*)out, (int)c, hw, pix, tiles, eps);
    } else {
        mix_feature_kernel<T, false><<<(unsigned)blocks, BLOCK, lds, st>>>(
            (const T*)clean, (const T*)adv, (T*)out, (int)c, hw, pix, tiles, eps);
    }
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // namespace

// ---- learnable mixing (Classification/main_learnable.py:226): out = clean + w*(adv - clean), three fp32 roundings like
// the eager expression; w is read from device memory (the i-th entry of the model's `w` parameter).  Backward: only
// d(loss)/dw = sum g*(adv - clean) is needed (clean is detached, adv's gradient is never used): block partials in
// fixed order, one wave folds them — deterministic.
template <typename TO>
__global__ __launch_bounds__(BLOCK) void mix_w_kernel(const float* __restrict__ clean, const float* __restrict__ adv,
                                                      const float* __restrict__ w, TO* __restrict__ out, int64_t n) {
    const float wv = *w;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const float c = clean[i];
        const float d = adv[i] - c;
        const float m = wv * d;
        Elt<TO>::st(out + i, c + m);
    }
}

template <typename TG>
__global__ __launch_bounds__(BLOCK) void mix_w_dot_kernel(const TG* __restrict__ g, const float* __restrict__ clean,
                                                          const float* __restrict__ adv, int64_t n,
                                                          float* __restrict__ partial) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK)
        s += Elt<TG>::ld(g + i) * (adv[i] - clean[i]);
    __shared__ float sh[BLOCK / AFAN_WAVE];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < BLOCK / AFAN_WAVE; ++k) t += sh[k];
        partial[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(AFAN_WAVE) void mix_w_dot_finalize_kernel(const float* __restrict__ partial, int G,
                                                                       float* __restrict__ dw, int accumulate) {
    float s = 0.f;
    for (int g = threadIdx.x; g < G; g += AFAN_WAVE) s += partial[g];
    s = wave_sum(s);
    if (threadIdx.x == 0) *dw = accumulate ? *dw + s : s;
}

constexpr int MIXW_MAX_BLOCKS = 1024;

extern "C" {

int afan_mix_feature_nhwc(const void* clean, const void* adv, void* out, int64_t n, int64_t c, int64_t hw, float eps,
                          int dtype, afan_stream_t stream) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n < 0 || c <= 0 || hw < 0 || c > 0x7fffffffLL) return AFAN_ESHAPE;
    if (n == 0 || hw == 0) return AFAN_OK;
    if (!clean || !adv || !out) return AFAN_ENULL;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(clean, a) || !aligned(adv, a) || !aligned(out, a)) return AFAN_EALIGN;
    if (dtype == AFAN_F32) return mix_nhwc_impl<float>(clean, adv, out, n * hw, c, eps, (hipStream_t)stream);
    return mix_nhwc_impl<uint16_t>(clean, adv, out, n * hw, c, eps, (hipStream_t)stream);
}

int afan_mix_feature(const void* clean, const void* adv, void* out, int64_t n, int64_t c, int64_t hw,
                     float eps, int dtype, afan_stream_t stream) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n < 0 || c <= 0 || hw < 0 || c > 0x7fffffffLL) return AFAN_ESHAPE;
    if (n == 0 || hw == 0) return AFAN_OK;
    if (!clean || !adv || !out) return AFAN_ENULL;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(clean, a) || !aligned(adv, a) || !aligned(out, a)) return AFAN_EALIGN;
    // hw == 1 (Detection's pooled ROI feature, [128, 2048, 1, 1]: attack_algo.py:254-265 on roi_feature_map): NCHW and NHWC are the
    // same bytes, and the pixel-tile kernel would run 1 live pixel column per workgroup on 32 of 256 threads (40.9 us for 3.1 MB);
    // one wave per row with coalesced channel accesses is the channels-last kernel
    if (hw == 1) {
        if (dtype == AFAN_F32) return mix_nhwc_impl<float>(clean, adv, out, n, c, eps, (hipStream_t)stream);
        return mix_nhwc_impl<uint16_t>(clean, adv, out, n, c, eps, (hipStream_t)stream);
    }
    if (dtype == AFAN_F32) return mix_impl<float>(clean, adv, out, n, c, hw, eps, (hipStream_t)stream);
    return mix_impl<uint16_t>(clean, adv, out, n, c, hw, eps, (hipStream_t)stream);
}

int afan_lerp_points(const float* x, const float* y, float* out, int64_t n, const float* weights,
                     int n_interior, afan_stream_t stream) {
    if (n < 0 || n_interior < 0 || n_interior > 8) return AFAN_ESHAPE;
    if (n == 0 || n_interior == 0) return AFAN_OK;
    if (!x || !y || !out || !weights) return AFAN_ENULL;
    if (!aligned(x, 4) || !aligned(y, 4) || !aligned(out, 4)) return AFAN_EALIGN;
    LerpW lw;
    for (int k = 0; k < 8; ++k) lw.w[k] = k < n_interior ? weights[k] : 0.f;
    const int vec = aligned(x, 16) && aligned(y, 16) && aligned(out, 16) && (n % 4 == 0);
    const int grid = grid_for(vec ? n / 4 : n, BLOCK);
    AFAN_PROF("lerp_points_kernel", 4.0 * n * (2 + n_interior), (hipStream_t)stream);
    lerp_points_kernel<<<grid, BLOCK, 0, (hipStream_t)stream>>>(x, y, out, n, lw, n_interior, vec);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_lerp_mix(const float* clean, const float* adv, float* out, int64_t n, int64_t c, int64_t hw, const float* weights,
                  int n_points, unsigned mix_mask, float eps, int layout, afan_stream_t stream) {
    const int npts = n_points - 1;                          // points 1 .. n_points-1 (the last one is adv itself)
    if (n < 0 || c <= 0 || hw < 0 || c > 0x7fffffffLL || npts < 1 || npts > LM_MAX || (mix_mask >> npts)) return AFAN_ESHAPE;
    if (layout != AFAN_NCHW && layout != AFAN_NHWC) return AFAN_ELAYOUT;
    if (n == 0 || hw == 0) return AFAN_OK;
    if (!clean || !adv || !out || (npts > 1 && !weights)) return AFAN_ENULL;
    if (!aligned(clean, 4) || !aligned(adv, 4) || !aligned(out, 4)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    LerpW lw;
    for (int k = 0; k < 8; ++k) lw.w[k] = k < npts - 1 ? weights[k] : 0.f;
    const int64_t total = n * c * hw;
    AFAN_PROF("lerp_mix_kernel", 4.0 * total * (2 + npts), st);
    if (layout == AFAN_NHWC || hw == 1) {       // (hw == 1: the same bytes either way; afan_mix_feature takes this form too — bit-equal)
        const int grid = grid_for(n * hw * AFAN_WAVE, BLOCK, 8192);
        if (c <= 64 * 8) lerp_mix_nhwc_kernel<8><<<grid, BLOCK, 0, st>>>(clean, adv, out, (int)c, n * hw, total, lw, npts, mix_mask, eps);
        else lerp_mix_nhwc_kernel<20><<<grid, BLOCK, 0, st>>>(clean, adv, out, (int)c, n * hw, total, lw, npts, mix_mask, eps);
    } else {
        // the pixel tile (hence the channel grouping and the summation order) is chosen exactly as afan_mix_feature does,
        // so that the fused result equals the separate launches bit for bit
        auto bytes = [&](int pix, bool stage) { return (size_t)(1 + LM_MAX) * (BLOCK / pix) * pix * 3 * 4 + (stage ? (size_t)c * pix * 4 : 0); };
        int pix = 64;
        while (pix >= 8 && (int64_t)c * pix * 4 + 2 * (BLOCK / pix) * pix * 3 * 4 > 64 * 1024) pix >>= 1;
        bool stage = pix >= 8;
        if (!stage) {
            pix = 64;
            while (pix >= 8 && (int64_t)c * pix * 4 + 2 * (BLOCK / pix) * pix * 3 * 4 > MAX_LDS_BYTES) pix >>= 1;
            stage = pix >= 8;
            if (!stage) pix = 64;
        }
        if (stage && bytes(pix, true) > (size_t)156 * 1024) stage = false;     // (same tile, clean re-read instead of staged)
        while (pix > 8 && pix / 2 >= hw) pix >>= 1;
        const int tiles = (int)((hw + pix - 1) / pix);
        const int64_t blocks = n * tiles;
        if (blocks > 0x7fffffffLL) return AFAN_ESHAPE;
        const size_t lds = bytes(pix, stage);
        if (stage) {
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)lerp_mix_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return (int)e;
            }
            lerp_mix_kernel<true><<<(unsigned)blocks, BLOCK, lds, st>>>(clean, adv, out, (int)c, hw, pix, tiles, total, lw, npts, mix_mask, eps);
        } else {
            lerp_mix_kernel<false><<<(unsigned)blocks, BLOCK, lds, st>>>(clean, adv, out, (int)c, hw, pix, tiles, total, lw, npts, mix_mask, eps);
        }
    }
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int64_t afan_mix_w_workspace_floats(void) { return MIXW_MAX_BLOCKS; }

int afan_mix_w(const float* clean, const float* adv, const float* w, void* out, int out_dtype, int64_t n,
               afan_stream_t stream) {
    if (out_dtype != AFAN_F32 && out_dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!clean || !adv || !w || !out) return AFAN_ENULL;
    if (!aligned(clean, 4) || !aligned(adv, 4) || !aligned(w, 4) || !aligned(out, out_dtype == AFAN_F32 ? 4 : 2)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(n, BLOCK);
    AFAN_PROF("mix_w_kernel", (8.0 + (out_dtype == AFAN_F32 ? 4 : 2)) * n, st);
    if (out_dtype == AFAN_F32) mix_w_kernel<float><<<grid, BLOCK, 0, st>>>(clean, adv, w, (float*)out, n);
    else mix_w_kernel<uint16_t><<<grid, BLOCK, 0, st>>>(clean, adv, w, (uint16_t*)out, n);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_mix_w_backward(const void* grad_out, int grad_dtype, const float* clean, const float* adv, int64_t n,
                        float* workspace, float* dw, int accumulate, afan_stream_t stream) {
    if (grad_dtype != AFAN_F32 && grad_dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n < 0) return AFAN_ESHAPE;
    if (!dw || !workspace) return AFAN_ENULL;
    if (n > 0 && (!grad_out || !clean || !adv)) return AFAN_ENULL;
    if (!aligned(clean, 4) || !aligned(adv, 4) || !aligned(dw, 4) || !aligned(workspace, 4) ||
        !aligned(grad_out, grad_dtype == AFAN_F32 ? 4 : 2))
        return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    int grid = n > 0 ? grid_for(n, BLOCK, MIXW_MAX_BLOCKS) : 1;
    {
        AFAN_PROF("mix_w_dot_kernel", (8.0 + (grad_dtype == AFAN_F32 ? 4 : 2)) * n, st);
        if (grad_dtype == AFAN_F32) mix_w_dot_kernel<float><<<grid, BLOCK, 0, st>>>((const float*)grad_out, clean, adv, n, workspace);
        else mix_w_dot_kernel<uint16_t><<<grid, BLOCK, 0, st>>>((const uint16_t*)grad_out, clean, adv, n, workspace);
    }
    AFAN_LAUNCH_CHECK();
    mix_w_dot_finalize_kernel<<<1, AFAN_WAVE, 0, st>>>(workspace, grid, dw, accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
