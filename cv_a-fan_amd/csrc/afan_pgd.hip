// PGD feature-ascent kernels for gfx950: sign-step (+ L-inf projection) fused with the per-sample
// perturbation norms and the bf16 shadow copy; host-noise random start.
// Reference behaviour restated (not translated) from Classification/attack_algo.py:9-19,35-36,41-56
// and Classification/main_perturb.py:188-192.  All of these are HBM-bound streaming kernels:
// 16 B per lane per access, grid-stride, >= 2048 workgroups when the tensor allows it.
#include "afan_common.h"

using namespace afan;

namespace {

constexpr int BLOCK = 256;
constexpr int NORM_CHUNK = 4096;  // elements of one sample handled by one workgroup in the norms kernels

__device__ __forceinline__ float sign_of(float g) {
    // torch.sign: 0 -> 0, NaN -> NaN
    return (g != g) ? g : (float)((g > 0.f) - (g < 0.f));
}

template <bool CLIP>
__device__ __forceinline__ float step_one(float xa, float g, float xc, float gamma, float eps) {
    float d = gamma * sign_of(g);  // exact: +-gamma, 0 or NaN
    float t = xa + d;              // the single rounding of attack_algo.py:53
    if (CLIP) {
        float lo = xc - eps;  // attack_algo.py:36 materialises centre-radius / centre+radius first
        float hi = xc + eps;
        if (t < lo) t = lo;  // attack_algo.py:14-15 (NaN compares false and passes through)
        if (t > hi) t = hi;  // attack_algo.py:16-17
    }
    return t;
}

template <typename G> struct GradVec4;
template <> struct GradVec4<float> {
    __device__ static __forceinline__ void ld(const float* p, float (&g)[4]) { Elt<float>::ldv(p, g); }
};
template <> struct GradVec4<uint16_t> {
    __device__ static __forceinline__ void ld(const uint16_t* p, float (&g)[4]) {
        u16x4 t = *reinterpret_cast<const u16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = bf2f(t[i]);
    }
};

// ---- plain step ---------------------------------------------------------------------------------
template <typename G, bool CLIP, bool SHADOW, bool VEC>
__global__ __launch_bounds__(BLOCK) void pgd_step_kernel(float* __restrict__ x_adv,
                                                         const G* __restrict__ grad,
                                                         const float* __restrict__ x_clean,
                                                         uint16_t* __restrict__ shadow, int64_t n,
                                                         float gamma, float eps) {
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    if (VEC) {
        const int64_t nvec = n >> 2;
        for (int64_t v = tid; v < nvec; v += nthreads) {
            const int64_t i = v << 2;
            float xa[4], g[4], xc[4] = {0.f, 0.f, 0.f, 0.f};
            Elt<float>::ldv(x_adv + i, xa);
            GradVec4<G>::ld(grad + i, g);
            if (CLIP) Elt<float>::ldv(x_clean + i, xc);
#pragma unroll
            for (int k = 0; k < 4; ++k) xa[k] = step_one<CLIP>(xa[k], g[k], xc[k], gamma, eps);
            Elt<float>::stv(x_adv + i, xa);
            if (SHADOW) {
                u16x4 s;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = f2bf(xa[k]);
                *reinterpret_cast<u16x4*>(shadow + i) = s;
            }
        }
        // ragged tail (n % 4 elements)
        const int64_t i = (nvec << 2) + tid;
        if (i < n) {
            float t = step_one<CLIP>(x_adv[i], Elt<G>::ld(grad + i), CLIP ? x_clean[i] : 0.f, gamma, eps);
            x_adv[i] = t;
            if (SHADOW) shadow[i] = f2bf(t);
        }
    } else {
        for (int64_t i = tid; i < n; i += nthreads) {
            float t = step_one<CLIP>(x_adv[i], Elt<G>::ld(grad + i), CLIP ? x_clean[i] : 0.f, gamma, eps);
            x_adv[i] = t;
            if (SHADOW) shadow[i] = f2bf(t);
        }
    }
}

// ---- block reduction of (sum of squares, max abs) into one partial per workgroup ----------------
__device__ __forceinline__ void block_reduce_store(float ss, float mx, float* __restrict__ partial,
                                                   int64_t slot) {
    __shared__ float s_ss[BLOCK / AFAN_WAVE];
    __shared__ float s_mx[BLOCK / AFAN_WAVE];
    ss = wave_sum(ss);
    // fmaxf drops NaNs: carry "saw a NaN" separately so the max stays NaN-propagating like torch.norm(p=inf)
    const float nanflag = wave_max((mx != mx) ? 1.f : 0.f);
    mx = wave_max((mx != mx) ? 0.f : mx);
    if (nanflag > 0.f) mx = __builtin_nanf("");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        s_ss[w] = ss;
        s_mx[w] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < BLOCK / AFAN_WAVE; ++k) {
            a += s_ss[k];
            b = (b != b || s_mx[k] != s_mx[k]) ? __builtin_nanf("") : fmaxf(b, s_mx[k]);
        }
        partial[2 * slot] = a;
        partial[2 * slot + 1] = b;
    }
}

// NaN-propagating |d| max (torch.norm(p=inf) returns NaN if any element is NaN)
__device__ __forceinline__ float absmax_nan(float m, float d) {
    float a = fabsf(d);
    return (a != a || m != m) ? __builtin_nanf("") : fmaxf(m, a);
}

// grid = (slices, batch): workgroup (s, b) owns elements [s*NORM_CHUNK, (s+1)*NORM_CHUNK) of sample b.
template <typename G, bool CLIP, bool SHADOW, bool VEC, bool STEP>
__global__ __launch_bounds__(BLOCK) void pgd_step_norms_kernel(
    float* __restrict__ x_adv, const G* __restrict__ grad, const float* __restrict__ x_clean,
    uint16_t* __restrict__ shadow, int64_t per_sample, float gamma, float eps,
    float* __restrict__ partial) {
    const int64_t b = blockIdx.y;
    const int64_t base = b * per_sample;
    const int64_t lo = (int64_t)blockIdx.x * NORM_CHUNK;
    const int64_t hi = (lo + NORM_CHUNK < per_sample) ? lo + NORM_CHUNK : per_sample;
    float ss = 0.f, mx = 0.f;
    if (VEC) {  // per_sample % 4 == 0 and 16-byte aligned bases
        for (int64_t j = lo + (int64_t)threadIdx.x * 4; j < hi; j += BLOCK * 4) {
            const int64_t i = base + j;
            float xa[4], xc[4], g[4];
            Elt<float>::ldv(x_adv + i, xa);
            Elt<float>::ldv(x_clean + i, xc);
            if (STEP) {
                GradVec4<G>::ld(grad + i, g);
#pragma unroll
                for (int k = 0; k < 4; ++k) xa[k] = step_one<CLIP>(xa[k], g[k], xc[k], gamma, eps);
                Elt<float>::stv(x_adv + i, xa);
                if (SHADOW) {
                    u16x4 s;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] = f2bf(xa[k]);
                    *reinterpret_cast<u16x4*>(shadow + i) = s;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float d = xa[k] - xc[k];  // main_perturb.py:189
                ss += d * d;
                mx = absmax_nan(mx, d);
            }
        }
    } else {
        for (int64_t j = lo + threadIdx.x; j < hi; j += BLOCK) {
            const int64_t i = base + j;
            float xa = x_adv[i], xc = x_clean[i];
            if (STEP) {
                xa = step_one<CLIP>(xa, Elt<G>::ld(grad + i), xc, gamma, eps);
                x_adv[i] = xa;
                if (SHADOW) shadow[i] = f2bf(xa);
            }
            float d = xa - xc;
            ss += d * d;
            mx = absmax_nan(mx, d);
        }
    }
    block_reduce_store(ss, mx, partial, b * gridDim.x + blockIdx.x);
}

// one wave per sample: fold the per-slice partials, L2 = sqrt(sum), Linf = max
__global__ __launch_bounds__(AFAN_WAVE) void norms_finalize_kernel(const float* __restrict__ partial,
                                                                   int slices, float* __restrict__ l2,
                                                                   float* __restrict__ linf) {
    const int64_t b = blockIdx.x;
    float ss = 0.f, mx = 0.f;
    for (int s = threadIdx.x; s < slices; s += AFAN_WAVE) {
        ss += partial[2 * (b * slices + s)];
        float m = partial[2 * (b * slices + s) + 1];
        mx = (m != m || mx != mx) ? __builtin_nanf("") : fmaxf(mx, m);
    }
    ss = wave_sum(ss);
    // NaN-aware wave max
    float isn = (mx != mx) ? 1.f : 0.f;
    isn = wave_max(isn);
    mx = wave_max((mx != mx) ? 0.f : mx);
    if (threadIdx.x == 0) {
        l2[b] = sqrtf(ss);
        linf[b] = (isn > 0.f) ? __builtin_nanf("") : mx;
    }
}

__global__ __launch_bounds__(BLOCK) void axpy_noise_kernel(float* __restrict__ x_adv,
                                                           const float* __restrict__ u, int64_t n,
                                                           float eps, uint16_t* __restrict__ shadow,
                                                           int vec) {
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    int64_t done = 0;
    if (vec) {
        const int64_t nvec = n >> 2;
        for (int64_t v = tid; v < nvec; v += nthreads) {
            const int64_t i = v << 2;
            float xa[4], uu[4];
            Elt<float>::ldv(x_adv + i, xa);
            Elt<float>::ldv(u + i, uu);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = 2.0f * uu[k];  // attack_algo.py:44, each op rounded on its own
                t = t - 1.0f;
                t = t * eps;
                xa[k] = xa[k] + t;
            }
            Elt<float>::stv(x_adv + i, xa);
            if (shadow) {
                u16x4 s;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = f2bf(xa[k]);
                *reinterpret_cast<u16x4*>(shadow + i) = s;
            }
        }
        done = nvec << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        float t = 2.0f * u[i];
        t = t - 1.0f;
        t = t * eps;
        t = x_adv[i] + t;
        x_adv[i] = t;
        if (shadow) shadow[i] = f2bf(t);
    }
}

// attack_algo.py:9-19 with arbitrary bound tensors
__global__ __launch_bounds__(BLOCK) void tensor_clamp_kernel(float* __restrict__ t, const float* __restrict__ lo,
                                                             const float* __restrict__ hi, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        float v = t[i];
        const float l = lo[i], h = hi[i];
        if (v < l) v = l;
        if (v > h) v = h;
        t[i] = v;
    }
}

__global__ __launch_bounds__(BLOCK) void cast_bf16_kernel(const float* __restrict__ src,
                                                          uint16_t* __restrict__ dst, int64_t n, int vec) {
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    int64_t done = 0;
    if (vec) {
        const int64_t nvec = n >> 3;
        for (int64_t v = tid; v < nvec; v += nthreads) {
            const int64_t i = v << 3;
            float a[4], b[4];
            Elt<float>::ldv(src + i, a);
            Elt<float>::ldv(src + i + 4, b);
            float o[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            Elt<uint16_t>::stv(dst + i, o);
        }
        done = nvec << 3;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) dst[i] = f2bf(src[i]);
}


// PGD's start (attack_algo.py:41 `x_adv = x.clone()` on the product's bf16 feature map): x in bf16 or fp32 -> the fp32
// reference copy x32 (the centre of the L-inf ball and of the norms; nullable when x is fp32 already), the fp32 iterate
// x_adv, and optionally its bf16 shadow — one pass instead of .float(), .clone() and a cast.
template <typename S>
__global__ __launch_bounds__(BLOCK) void pgd_init_kernel(const S* __restrict__ x, float* __restrict__ x32,
                                                         float* __restrict__ x_adv, uint16_t* __restrict__ shadow, int64_t n, int vec) {
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    int64_t done = 0;
    if (vec) {
        const int64_t nvec = n >> 3;
        for (int64_t v = tid; v < nvec; v += nthreads) {
            const int64_t i = v << 3;
            float o[8];
            if constexpr (sizeof(S) == 4) {
                float a[4], b[4];
                Elt<float>::ldv((const float*)x + i, a);
                Elt<float>::ldv((const float*)x + i + 4, b);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
            } else {
                Elt<uint16_t>::ldv((const uint16_t*)x + i, o);
            }
            const float lo[4] = {o[0], o[1], o[2], o[3]}, hi[4] = {o[4], o[5], o[6], o[7]};
            if (x32) { Elt<float>::stv(x32 + i, lo); Elt<float>::stv(x32 + i + 4, hi); }
            Elt<float>::stv(x_adv + i, lo);
            Elt<float>::stv(x_adv + i + 4, hi);
            if (shadow) Elt<uint16_t>::stv(shadow + i, o);
        }
        done = nvec << 3;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        const float v = Elt<S>::ld(x + i);
        if (x32) x32[i] = v;
        x_adv[i] = v;
        if (shadow) shadow[i] = f2bf(v);
    }
}

template <typename G, bool CLIP, bool SHADOW>
int launch_step(float* x_adv, const void* grad, const float* x_clean, uint16_t* shadow, int64_t n,
                float gamma, float eps, hipStream_t st) {
    const bool vec = aligned(x_adv, 16) && aligned(grad, 4 * sizeof(G)) &&
                     (!CLIP || aligned(x_clean, 16)) && (!SHADOW || aligned(shadow, 8));
    const int grid = grid_for(vec ? (n + 3) / 4 : n, BLOCK);
    AFAN_PROF("pgd_step_kernel", n * (8.0 + sizeof(G) + (CLIP ? 4 : 0) + (SHADOW ? 2 : 0)), st);
    if (vec)
        pgd_step_kernel<G, CLIP, SHADOW, true><<<grid, BLOCK, 0, st>>>(
            x_adv, (const G*)grad, x_clean, shadow, n, gamma, eps);
    else
        pgd_step_kernel<G, CLIP, SHADOW, false><<<grid, BLOCK, 0, st>>>(
            x_adv, (const G*)grad, x_clean, shadow, n, gamma, eps);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <typename G, bool CLIP, bool SHADOW, bool STEP>
int launch_step_norms(float* x_adv, const void* grad, const float* x_clean, uint16_t* shadow,
                      int64_t batch, int64_t per_sample, float gamma, float eps, float* partial,
                      float* l2, float* linf, hipStream_t st) {
    const bool vec = (per_sample % 4 == 0) && aligned(x_adv, 16) && aligned(x_clean, 16) &&
                     (!STEP || aligned(grad, 4 * sizeof(G))) && (!SHADOW || aligned(shadow, 8));
    const int slices = (int)((per_sample + NORM_CHUNK - 1) / NORM_CHUNK);
    dim3 grid(slices, (unsigned)batch);
    {
        const double elts = (double)batch * (double)per_sample;
        AFAN_PROF(STEP ? "pgd_step_norms_kernel" : "perturb_norms_kernel",
                  elts * (STEP ? (12.0 + sizeof(G) + (SHADOW ? 2 : 0)) : 8.0), st);
        if (vec)
            pgd_step_norms_kernel<G, CLIP, SHADOW, true, STEP><<<grid, BLOCK, 0, st>>>(
                x_adv, (const G*)grad, x_clean, shadow, per_sample, gamma, eps, partial);
        else
            pgd_step_norms_kernel<G, CLIP, SHADOW, false, STEP><<<grid, BLOCK, 0, st>>>(
                x_adv, (const G*)grad, x_clean, shadow, per_sample, gamma, eps, partial);
    }
    AFAN_LAUNCH_CHECK();
    AFAN_PROF("norms_finalize_kernel", 8.0 * batch * slices, st);
    norms_finalize_kernel<<<(unsigned)batch, AFAN_WAVE, 0, st>>>(partial, slices, l2, linf);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // namespace

extern "C" {

int afan_version(void) { return 100; }
const char* afan_arch(void) { return "gfx950"; }

int afan_pgd_step(float* x_adv, const void* grad, int grad_dtype, const float* x_clean,
                  uint16_t* shadow_bf16, int64_t n, float gamma, float eps, int clip,
                  afan_stream_t stream) {
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x_adv || !grad || (clip && !x_clean)) return AFAN_ENULL;
    if (grad_dtype != AFAN_F32 && grad_dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (!aligned(x_adv, 4) || !aligned(grad, grad_dtype == AFAN_F32 ? 4 : 2) ||
        (clip && !aligned(x_clean, 4)) || (shadow_bf16 && !aligned(shadow_bf16, 2)))
        return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
#define AFAN_DISPATCH(G)                                                                              \
    if (clip) {                                                                                       \
        if (shadow_bf16) return launch_step<G, true, true>(x_adv, grad, x_clean, shadow_bf16, n, gamma, eps, st); \
        return launch_step<G, true, false>(x_adv, grad, x_clean, shadow_bf16, n, gamma, eps, st);     \
    } else {                                                                                          \
        if (shadow_bf16) return launch_step<G, false, true>(x_adv, grad, x_clean, shadow_bf16, n, gamma, eps, st); \
        return launch_step<G, false, false>(x_adv, grad, x_clean, shadow_bf16, n, gamma, eps, st);    \
    }
    if (grad_dtype == AFAN_F32) { AFAN_DISPATCH(float) }
    AFAN_DISPATCH(uint16_t)
#undef AFAN_DISPATCH
}

int64_t afan_norms_workspace_floats(int64_t batch, int64_t per_sample) {
    if (batch <= 0 || per_sample <= 0) return 0;
    return 2 * batch * ((per_sample + NORM_CHUNK - 1) / NORM_CHUNK);
}

int afan_pgd_step_norms(float* x_adv, const void* grad, int grad_dtype, const float* x_clean,
                        uint16_t* shadow_bf16, int64_t batch, int64_t per_sample, float gamma,
                        float eps, int clip, float* partial, float* l2_out, float* linf_out,
                        afan_stream_t stream) {
    if (batch < 0 || per_sample <= 0 || batch > 65535) return AFAN_ESHAPE;
    if (batch == 0) return AFAN_OK;
    if (!x_adv || !grad || !x_clean || !partial || !l2_out || !linf_out) return AFAN_ENULL;
    if (grad_dtype != AFAN_F32 && grad_dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (!aligned(x_adv, 4) || !aligned(grad, grad_dtype == AFAN_F32 ? 4 : 2) || !aligned(x_clean, 4) ||
        !aligned(partial, 4) || (shadow_bf16 && !aligned(shadow_bf16, 2)))
        return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
#define AFAN_DISPATCH(G)                                                                             \
    if (clip) {                                                                                      \
        if (shadow_bf16)                                                                             \
            return launch_step_norms<G, true, true, true>(x_adv, grad, x_clean, shadow_bf16, batch,  \
                                                          per_sample, gamma, eps, partial, l2_out, linf_out, st); \
        return launch_step_norms<G, true, false, true>(x_adv, grad, x_clean, shadow_bf16, batch,     \
                                                       per_sample, gamma, eps, partial, l2_out, linf_out, st); \
    } else {                                                                                         \
        if (shadow_bf16)                                                                             \
            return launch_step_norms<G, false, true, true>(x_adv, grad, x_clean, shadow_bf16, batch, \
                                                           per_sample, gamma, eps, partial, l2_out, linf_out, st); \
        return launch_step_norms<G, false, false, true>(x_adv, grad, x_clean, shadow_bf16, batch,    \
                                                        per_sample, gamma, eps, partial, l2_out, linf_out, st); \
    }
    if (grad_dtype == AFAN_F32) { AFAN_DISPATCH(float) }
    AFAN_DISPATCH(uint16_t)
#undef AFAN_DISPATCH
}

int afan_perturb_norms(const float* x_adv, const float* x_clean, int64_t batch, int64_t per_sample,
                       float* partial, float* l2_out, float* linf_out, afan_stream_t stream) {
    if (batch < 0 || per_sample <= 0 || batch > 65535) return AFAN_ESHAPE;
    if (batch == 0) return AFAN_OK;
    if (!x_adv || !x_clean || !partial || !l2_out || !linf_out) return AFAN_ENULL;
    if (!aligned(x_adv, 4) || !aligned(x_clean, 4) || !aligned(partial, 4)) return AFAN_EALIGN;
    return launch_step_norms<float, false, false, false>(const_cast<float*>(x_adv), nullptr, x_clean,
                                                         nullptr, batch, per_sample, 0.f, 0.f, partial,
                                                         l2_out, linf_out, (hipStream_t)stream);
}

int afan_axpy_noise(float* x_adv, const float* u, int64_t n, float eps, uint16_t* shadow_bf16,
                    afan_stream_t stream) {
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x_adv || !u) return AFAN_ENULL;
    if (!aligned(x_adv, 4) || !aligned(u, 4) || (shadow_bf16 && !aligned(shadow_bf16, 2))) return AFAN_EALIGN;
    const int vec = aligned(x_adv, 16) && aligned(u, 16) && (!shadow_bf16 || aligned(shadow_bf16, 8));
    const int grid = grid_for(vec ? (n + 3) / 4 : n, BLOCK);
    AFAN_PROF("axpy_noise_kernel", n * (12.0 + (shadow_bf16 ? 2 : 0)), (hipStream_t)stream);
    axpy_noise_kernel<<<grid, BLOCK, 0, (hipStream_t)stream>>>(x_adv, u, n, eps, shadow_bf16, vec);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_tensor_clamp(float* t, const float* lo, const float* hi, int64_t n, afan_stream_t stream) {
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!t || !lo || !hi) return AFAN_ENULL;
    if (!aligned(t, 4) || !aligned(lo, 4) || !aligned(hi, 4)) return AFAN_EALIGN;
    AFAN_PROF("tensor_clamp_kernel", n * 16.0, (hipStream_t)stream);
    tensor_clamp_kernel<<<grid_for(n, BLOCK), BLOCK, 0, (hipStream_t)stream>>>(t, lo, hi, n);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_cast_bf16(const float* src, uint16_t* dst, int64_t n, afan_stream_t stream) {
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!src || !dst) return AFAN_ENULL;
    if (!aligned(src, 4) || !aligned(dst, 2)) return AFAN_EALIGN;
    const int vec = aligned(src, 16) && aligned(dst, 16);
    const int grid = grid_for(vec ? (n + 7) / 8 : n, BLOCK);
    AFAN_PROF("cast_bf16_kernel", n * 6.0, (hipStream_t)stream);
    cast_bf16_kernel<<<grid, BLOCK, 0, (hipStream_t)stream>>>(src, dst, n, vec);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_pgd_init(const void* x, int dtype, float* x32, float* x_adv, uint16_t* shadow, int64_t n, afan_stream_t stream) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!x || !x_adv || (dtype == AFAN_BF16 && !x32)) return AFAN_ENULL;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, a) || !aligned(x_adv, 4) || (x32 && !aligned(x32, 4)) || (shadow && !aligned(shadow, 2))) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int vec = aligned(x, 16) && aligned(x_adv, 16) && (!x32 || aligned(x32, 16)) && (!shadow || aligned(shadow, 16));
    const int grid = grid_for(vec ? (n + 7) / 8 : n, BLOCK);
    AFAN_PROF("pgd_init_kernel", n * (a + 4.0 + (x32 ? 4 : 0) + (shadow ? 2 : 0)), st);
    if (dtype == AFAN_F32) pgd_init_kernel<float><<<grid, BLOCK, 0, st>>>((const float*)x, x32, x_adv, shadow, n, vec);
    else pgd_init_kernel<uint16_t><<<grid, BLOCK, 0, st>>>((const uint16_t*)x, x32, x_adv, shadow, n, vec);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
