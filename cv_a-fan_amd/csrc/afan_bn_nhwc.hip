// Training-mode BatchNorm for NHWC (channels-last) activations on gfx950.
// The bf16 backbone keeps activations channels-last because the MFMA convolutions want K = (r, s, c)
// contiguous; the tensor is then one dense [M = N*H*W][C] matrix and BN statistics are column sums.
//
// Mapping: a thread owns ONE 16-byte channel vector (8 bf16 / 4 fp32 channels) for its whole life:
// flat vector index v = b*256 + t, stride G*256, and 256 % (C/VEC) == 0, so v % (C/VEC) is constant per
// thread.  Every wave-instruction therefore reads 1 KiB of consecutive memory.  Per-channel partial sums
// are shifted by the tensor's first row (kills the E[x^2]-mean^2 cancellation), reduced across the threads
// of a block that share a channel vector through LDS, and written as ONE partial per (block, channel):
// additive, so the finalize launch (one wave per channel) is a plain deterministic sum — no float atomics.
//   forward : stats -> finalize (mean, invstd, running stats) -> apply (+residual, +ReLU)
//   backward: reduce -> finalize (sum_g, sum_gx, dweight, dbias) -> dx (+d_residual)
// Where a convolution epilogue already took the sums (afan_conv.hip), the "accumulator" variants further down skip the
// slab and the finalize launch: f64 atomics into a few copies per channel, folded in the prologue of the apply pass.
// Channel counts that do not fit the mapping (C/VEC not a divisor of 256, e.g. C = 304) take the generic
// kernels: one thread per channel walking rows, lanes across channels (still coalesced, narrower accesses).
#include "afan_common.h"
#include <stdlib.h>

using namespace afan;

namespace afan_nhwc {

constexpr int BLOCK = 256;
constexpr int MAX_G = 512;  // partials per channel

template <typename T, int VEC> struct LdV {
    __device__ static __forceinline__ void ld(const T* p, float (&v)[VEC]) {
        if constexpr (VEC == 1) v[0] = Elt<T>::ld(p); else Elt<T>::ldv(p, v);
    }
    __device__ static __forceinline__ void st(T* p, const float (&v)[VEC]) {
        if constexpr (VEC == 1) Elt<T>::st(p, v[0]); else Elt<T>::stv(p, v);
    }
};

// per-channel coefficient vector of this thread's VEC channels, loaded as 16-byte accesses (a per-lane scalar
// gather here costs 10x the tensor's own traffic: 64 lanes x 32-byte stride = 16 cache lines per instruction)
template <int VEC>
__device__ __forceinline__ void ld_coef(const float* __restrict__ p, int c0, float (&v)[VEC]) {
    if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + c0 + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = p[c0 + k];
    }
}

__device__ __forceinline__ void affine_coeffs(float mu, float is, float w, float b, float& alpha, float& beta) {
    alpha = is * w;
    beta = fmaf(-mu, alpha, b);
}

// Accumulator type of the per-thread / per-block column sums: fp32 tensors (parity mode, where the reference's own
// arithmetic is the bar) sum in f64 — a channel mean taken as shift + sum(x - shift)/M and the sum of a BatchNorm
// gradient (which nearly cancels) otherwise carry an absolute error of eps * |shift - mean| that is COHERENT over the
// channel: a uniform offset of every output / gradient element (channel mean of a [4,64,65,65] map: 3e-6 -> 2e-8 relative
// error against float64, tools/diag_bn_layout.py).  bf16 tensors keep fp32 sums (their storage rounding is 2^16 times larger).
template <typename T> struct AccOf { typedef float type; };
template <> struct AccOf<float> { typedef double type; };

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Reduce per-thread values across the threads of the block that own the same channel vector (t % CV) and let
// the first CV threads write the block partial to ws[(q*C + c)*G + blockIdx.x], q = 0..NQ-1.
template <int VEC, int NQ, typename A>
__device__ __forceinline__ void block_fold_store(A (&acc)[NQ][VEC], int CV, int C, int G,
                                                 float* __restrict__ ws) {
    __shared__ A sh[NQ * VEC][BLOCK];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int k = 0; k < VEC; ++k) sh[q * VEC + k][threadIdx.x] = acc[q][k];
    __syncthreads();
    // all 256 threads take part (with 16 channels only 2 threads own a channel vector: letting them walk the 128 rows of
    // 16 columns alone cost 20 us per launch on ResNet-56s): thread t sums column (t % CV) of quantity/element (t / CV)
    if (CV < AFAN_WAVE) {
        // few channel vectors: a wave per quantity — 4 LDS reads per lane, then a butterfly over the lanes that share a
        // column (lane % CV); fixed order, so the partial is reproducible
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int qk = wave; qk < NQ * VEC; qk += BLOCK / AFAN_WAVE) {
            A s = (sh[qk][lane] + sh[qk][lane + 64]) + (sh[qk][lane + 128] + sh[qk][lane + 192]);
            for (int o = CV; o < AFAN_WAVE; o <<= 1) s += __shfl_xor(s, o, AFAN_WAVE);
            if (lane < CV) ws[((int64_t)(qk / VEC) * C + lane * VEC + qk % VEC) * G + blockIdx.x] = (float)s;
        }
        return;
    }
    const int R = BLOCK / CV;
    for (int item = threadIdx.x; item < NQ * VEC * CV; item += BLOCK) {
        const int cv = item % CV, qk = item / CV;
        A s = 0;
        for (int r = 0; r < R; ++r) s += sh[qk][cv + r * CV];
        const int q = qk / VEC, k = qk % VEC;
        ws[((int64_t)q * C + cv * VEC + k) * G + blockIdx.x] = (float)s;
    }
}

// Accumulator variant: the block's column sums are added into double-precision accumulators acc[slot][q][C] (zeroed
// by the caller) with native f64 atomics, slot = blockIdx.x % NS: atomics on ONE address serialise at the L2 (~100 ns
// each), so the adds of a launch are spread over NS copies and the consumer sums the copies.  The consumers
// (apply_acc / bwd_apply_acc kernels) derive their coefficients from the accumulators themselves, so neither a partial
// slab nor a finalize launch exists on this path.  fp32 block sums widened to f64 add exactly unless they differ by
// > 2^29 in magnitude, so the result is order-independent to ~1e-16 relative — far below the fp32 statistics.
inline int acc_slots(int64_t C) {   // NS * C <= 1024, 1 <= NS <= 16, power of two (host side; kernels get NS as a parameter)
    static const int64_t budget = [] { const char* v = getenv("AFAN_BN_SLOTS"); return v ? (int64_t)atoi(v) : (int64_t)1024; }();
    int ns = 16;
    while (ns > 1 && (int64_t)ns * C > budget) ns >>= 1;
    return ns;
}

int64_t acc_doubles(int64_t C);   // doubles per accumulator block (defined with the entry points below)

// How many identical train-mode forward passes the next BatchNorm forward launches of this host thread stand for (1 by
// default): the running statistics are updated that many times, in sequence, from the same batch moments —
// main_perturb.py:173 and :196 run the head twice on the same images with the same weights (afan_bn_set_running_updates).
static thread_local int g_running_updates = 1;
static inline int running_updates() { return g_running_updates; }

template <int VEC, int NQ, typename A>
__device__ __forceinline__ void block_fold_atomic(A (&acc)[NQ][VEC], int CV, int C, int NS,
                                                  double* __restrict__ out) {
    __shared__ A sh[NQ * VEC][BLOCK];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int k = 0; k < VEC; ++k) sh[q * VEC + k][threadIdx.x] = acc[q][k];
    __syncthreads();
    // all 256 threads take part: thread t sums column (t % CV) of quantity/element (t / CV) when it exists
    const int R = BLOCK / CV;
    double* dst = out + (int64_t)(blockIdx.x & (NS - 1)) * NQ * C;
    for (int item = threadIdx.x; item < NQ * VEC * CV; item += BLOCK) {
        const int cv = item % CV, qk = item / CV;
        A s = 0;
        for (int r = 0; r < R; ++r) s += sh[qk][cv + r * CV];
        const int q = qk / VEC, k = qk % VEC;
        unsafeAtomicAdd(dst + (int64_t)q * C + cv * VEC + k, (double)s);
    }
}

// ---- forward 1: shifted column sums -----------------------------------------------------------------
template <typename T, int VEC, bool ATOMIC = false>
__global__ __launch_bounds__(BLOCK) void stats_kernel(const T* __restrict__ x, int64_t nvec, int CV, int C,
                                                      float* __restrict__ ws, double* __restrict__ accd = nullptr,
                                                      int NS = 1) {
    const int cv = threadIdx.x % CV;
    float shift[VEC];
    LdV<T, VEC>::ld(x + (int64_t)cv * VEC, shift);  // first row of the tensor: same shift in every block
    typedef typename AccOf<T>::type A;
    A acc[2][VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[0][k] = acc[1][k] = 0;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
#pragma unroll 4
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += stride) {
        float e[VEC];
        LdV<T, VEC>::ld(x + v * VEC, e);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const A d = (A)e[k] - (A)shift[k];
            acc[0][k] += d;
            acc[1][k] += d * d;
        }
    }
    if constexpr (ATOMIC) block_fold_atomic<VEC, 2, A>(acc, CV, C, NS, accd);
    else block_fold_store<VEC, 2, A>(acc, CV, C, gridDim.x, ws);
}

// generic (channel counts the 16-byte mapping does not take, e.g. DeepLab's 48-channel projection): block per row slab;
// C <= BLOCK: the block's threads cover BLOCK / C rows at once — thread t reads element t of a run of consecutive rows
// (coalesced) — and the row lanes of a channel are summed through LDS; larger C: thread per channel.
template <typename A>
__device__ __forceinline__ void fold_row_lanes(A s1, A s2, int C, int R, int G, float* __restrict__ ws) {
    __shared__ A sh[2][BLOCK];
    sh[0][threadIdx.x] = s1;
    sh[1][threadIdx.x] = s2;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        A a = 0, b = 0;
        for (int k = 0; k < R; ++k) {
            a += sh[0][threadIdx.x + k * C];
            b += sh[1][threadIdx.x + k * C];
        }
        ws[((int64_t)0 * C + threadIdx.x) * G + blockIdx.x] = (float)a;
        ws[((int64_t)1 * C + threadIdx.x) * G + blockIdx.x] = (float)b;
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void stats_generic_kernel(const T* __restrict__ x, int64_t M, int C,
                                                              float* __restrict__ ws) {
    const int G = gridDim.x;
    const int64_t rows_per = (M + G - 1) / G;
    const int64_t r0 = blockIdx.x * rows_per, r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
    typedef typename AccOf<T>::type A;
    if (C <= BLOCK) {
        const int R = BLOCK / C, c = threadIdx.x % C, lane_r = threadIdx.x / C;
        A s1 = 0, s2 = 0;
        if (lane_r < R) {
            const float shift = Elt<T>::ld(x + c);
#pragma unroll 4
            for (int64_t r = r0 + lane_r; r < r1; r += R) {
                const A d = (A)Elt<T>::ld(x + r * C + c) - (A)shift;
                s1 += d;
                s2 += d * d;
            }
        }
        fold_row_lanes<A>(s1, s2, C, R, G, ws);
        return;
    }
    for (int c = threadIdx.x; c < C; c += BLOCK) {
        const float shift = Elt<T>::ld(x + c);
        A s1 = 0, s2 = 0;
        for (int64_t r = r0; r < r1; ++r) {
            const A d = (A)Elt<T>::ld(x + r * C + c) - (A)shift;
            s1 += d;
            s2 += d * d;
        }
        ws[((int64_t)0 * C + c) * G + blockIdx.x] = (float)s1;
        ws[((int64_t)1 * C + c) * G + blockIdx.x] = (float)s2;
    }
}

// ---- finalize: one wave per channel sums the G partials ----------------------------------------------
// MODE 0 (forward): stats[0..3][C] = mean, invstd, alpha = invstd*w, beta = b - mean*alpha (+ running stats).
// MODE 1 (backward): sum_g = sum g, sum_gx = invstd * sum g*(x-mean) -> dbias/dweight, and the dx coefficients
//   coef[0][c] = B = -alpha*invstd*sum_gx/M,  coef[1][c] = D = -alpha*sum_g/M     (dx = g*alpha + (x-mean)*B + D)
template <typename T, int MODE>
__global__ __launch_bounds__(BLOCK) void finalize_kernel(const float* __restrict__ ws, int G, int C, const T* x,
                                                         float inv_m, float m_count, float eps, float momentum,
                                                         const float* __restrict__ weight,
                                                         const float* __restrict__ bias, float* stats, float* coef,
                                                         float* rmean, float* rvar, int64_t* nbt, float* dweight,
                                                         float* dbias, int accumulate, const float* shift_ptr,
                                                         int updates) {
    const int c = blockIdx.x * (BLOCK / AFAN_WAVE) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double a = 0.0, b = 0.0;                // (f64 fold of the fp32 partials: see AccOf)
    for (int g = lane; g < G; g += AFAN_WAVE) {
        a += (double)ws[((int64_t)0 * C + c) * G + g];
        b += (double)ws[((int64_t)1 * C + c) * G + g];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (lane != 0) return;
    if (MODE == 0) {
        // the shift the partial sums were taken around: given explicitly (conv-epilogue partials), else row 0 of x
        const float shift = shift_ptr ? shift_ptr[c] : (x ? Elt<T>::ld(x + c) : 0.f);
        const double dmd = a / (double)m_count;        // (inv_m is a rounded fp32 reciprocal)
        const float mean = (float)((double)shift + dmd);
        const float m2 = (float)fmax(b - a * dmd, 0.0);
        const float is = 1.0f / sqrtf(m2 * inv_m + eps);
        float alpha, beta;
        affine_coeffs(mean, is, weight ? weight[c] : 1.f, bias ? bias[c] : 0.f, alpha, beta);
        stats[c] = mean;
        stats[C + c] = is;
        stats[2 * C + c] = alpha;
        stats[3 * C + c] = beta;
        if (rmean)
            for (int u = 0; u < updates; ++u) {    // updates > 1: this pass stands for that many identical train-mode passes
                rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
                rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (m2 / (m_count - 1.0f));
            }
        if (c == 0 && nbt) *nbt += updates;
    } else {
        const float is = stats[C + c], alpha = stats[2 * C + c];
        const float sum_g = (float)a, sum_gx = (float)b * is;
        coef[c] = -alpha * is * (sum_gx * inv_m);
        coef[C + c] = -alpha * (sum_g * inv_m);
        if (dbias) dbias[c] = accumulate ? dbias[c] + sum_g : sum_g;
        if (dweight) dweight[c] = accumulate ? dweight[c] + sum_gx : sum_gx;
    }
}

// eval mode: alpha/beta from given mean/invstd (running stats), one thread per channel
__global__ __launch_bounds__(BLOCK) void coef_kernel(int C, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd,
                                                     const float* __restrict__ weight,
                                                     const float* __restrict__ bias, float* __restrict__ stats) {
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= C) return;
    float alpha, beta;
    affine_coeffs(mean[c], invstd[c], weight ? weight[c] : 1.f, bias ? bias[c] : 0.f, alpha, beta);
    stats[c] = mean[c];
    stats[C + c] = invstd[c];
    stats[2 * C + c] = alpha;
    stats[3 * C + c] = beta;
}

// ---- forward 2: y = [relu](x*alpha + beta [+ res]) ------------------------------------------------------
template <typename T, int VEC, bool RES, bool RELU>
__global__ __launch_bounds__(BLOCK) void apply_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                      T* __restrict__ y, int64_t nvec, int CV, int C,
                                                      const float* __restrict__ stats) {
    const int c0 = (threadIdx.x % CV) * VEC;
    float alpha[VEC], beta[VEC];
    ld_coef<VEC>(stats + 2 * C, c0, alpha);
    ld_coef<VEC>(stats + 3 * C, c0, beta);
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
#pragma unroll 4
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += stride) {
        float e[VEC], r[VEC];
        LdV<T, VEC>::ld(x + v * VEC, e);
        if (RES) LdV<T, VEC>::ld(res + v * VEC, r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float t = fmaf(e[k], alpha[k], beta[k]);
            if (RES) t += r[k];
            if (RELU) t = (t > 0.f) ? t : ((t != t) ? t : 0.f);
            e[k] = t;
        }
        LdV<T, VEC>::st(y + v * VEC, e);
    }
}

template <typename T, bool RES, bool RELU>
__global__ __launch_bounds__(BLOCK) void apply_generic_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                              T* __restrict__ y, int64_t total, int C,
                                                              const float* __restrict__ stats) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int c = (int)(i % C);
        const float alpha = stats[2 * C + c], beta = stats[3 * C + c];
        float t = fmaf(Elt<T>::ld(x + i), alpha, beta);
        if (RES) t += Elt<T>::ld(res + i);
        if (RELU) t = (t > 0.f) ? t : ((t != t) ? t : 0.f);
        Elt<T>::st(y + i, t);
    }
}

// ---- backward 1: partial sums of g and g*xhat -------------------------------------------------------------
template <typename T, int VEC, bool RELU, bool HAVE_Y, bool ATOMIC = false>
__global__ __launch_bounds__(BLOCK) void bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, int64_t nvec, int CV, int C,
                                                           const float* __restrict__ stats, float* __restrict__ ws,
                                                           double* __restrict__ accd = nullptr, int NS = 1) {
    const int c0 = (threadIdx.x % CV) * VEC;
    float mu[VEC], alpha[VEC], beta[VEC];
    ld_coef<VEC>(stats, c0, mu);
    if (RELU && !HAVE_Y) {
        ld_coef<VEC>(stats + 2 * C, c0, alpha);
        ld_coef<VEC>(stats + 3 * C, c0, beta);
    }
    typedef typename AccOf<T>::type A;
    A acc[2][VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[0][k] = acc[1][k] = 0;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
#pragma unroll 2
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += stride) {
        float d[VEC], e[VEC], o[VEC];
        LdV<T, VEC>::ld(dy + v * VEC, d);
        LdV<T, VEC>::ld(x + v * VEC, e);
        if (RELU && HAVE_Y) LdV<T, VEC>::ld(y + v * VEC, o);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float g = d[k];
            if (RELU) {
                const float act = HAVE_Y ? o[k] : fmaf(e[k], alpha[k], beta[k]);
                g = (act > 0.f) ? g : 0.f;
            }
            acc[0][k] += (A)g;
            acc[1][k] += (A)g * ((A)e[k] - (A)mu[k]);   // invstd is applied once per channel in the finalize
        }
    }
    if constexpr (ATOMIC) block_fold_atomic<VEC, 2, A>(acc, CV, C, NS, accd);
    else block_fold_store<VEC, 2, A>(acc, CV, C, gridDim.x, ws);
}

template <typename T, bool RELU, bool HAVE_Y>
__global__ __launch_bounds__(BLOCK) void bwd_reduce_generic_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                   const T* __restrict__ y, int64_t M, int C,
                                                                   const float* __restrict__ stats,
                                                                   float* __restrict__ ws) {
    const int G = gridDim.x;
    const int64_t rows_per = (M + G - 1) / G;
    const int64_t r0 = blockIdx.x * rows_per, r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
    typedef typename AccOf<T>::type A;
    auto term = [&](int64_t i, float mu, float alpha, float beta, A& sg, A& sgx) {
        const float e = Elt<T>::ld(x + i);
        float g = Elt<T>::ld(dy + i);
        if (RELU) {
            const float act = HAVE_Y ? Elt<T>::ld(y + i) : fmaf(e, alpha, beta);
            g = (act > 0.f) ? g : 0.f;
        }
        sg += (A)g;
        sgx += (A)g * ((A)e - (A)mu);
    };
    if (C <= BLOCK) {       // BLOCK / C rows at once, as stats_generic_kernel
        const int R = BLOCK / C, c = threadIdx.x % C, lane_r = threadIdx.x / C;
        A sg = 0, sgx = 0;
        if (lane_r < R) {
            const float mu = stats[c], alpha = stats[2 * C + c], beta = stats[3 * C + c];
#pragma unroll 4
            for (int64_t r = r0 + lane_r; r < r1; r += R) term(r * C + c, mu, alpha, beta, sg, sgx);
        }
        fold_row_lanes<A>(sg, sgx, C, R, G, ws);
        return;
    }
    for (int c = threadIdx.x; c < C; c += BLOCK) {
        const float mu = stats[c], alpha = stats[2 * C + c], beta = stats[3 * C + c];
        A sg = 0, sgx = 0;
        for (int64_t r = r0; r < r1; ++r) term(r * C + c, mu, alpha, beta, sg, sgx);
        ws[((int64_t)0 * C + c) * G + blockIdx.x] = (float)sg;
        ws[((int64_t)1 * C + c) * G + blockIdx.x] = (float)sgx;
    }
}

// ---- backward 2: dx = g*alpha + (x-mean)*B + D (+ d_residual = g) ---------------------------------------------
template <typename T, int VEC, bool RELU, bool HAVE_Y, bool DRES>
__global__ __launch_bounds__(BLOCK) void bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const T* __restrict__ y, T* __restrict__ dx,
                                                          T* __restrict__ dres, int64_t nvec, int CV, int C,
                                                          const float* __restrict__ stats,
                                                          const float* __restrict__ coef) {
    const int c0 = (threadIdx.x % CV) * VEC;
    float mu[VEC], alpha[VEC], beta[VEC], B[VEC], D[VEC];
    ld_coef<VEC>(stats, c0, mu);
    ld_coef<VEC>(stats + 2 * C, c0, alpha);
    if (RELU && !HAVE_Y) ld_coef<VEC>(stats + 3 * C, c0, beta);
    ld_coef<VEC>(coef, c0, B);
    ld_coef<VEC>(coef + C, c0, D);
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
#pragma unroll 4
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += stride) {
        float d[VEC], e[VEC], o[VEC];
        LdV<T, VEC>::ld(dy + v * VEC, d);
        LdV<T, VEC>::ld(x + v * VEC, e);
        if (RELU && HAVE_Y) LdV<T, VEC>::ld(y + v * VEC, o);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float g = d[k];
            if (RELU) {
                const float act = HAVE_Y ? o[k] : fmaf(e[k], alpha[k], beta[k]);
                g = (act > 0.f) ? g : 0.f;
            }
            d[k] = g;
            e[k] = fmaf(g, alpha[k], fmaf(e[k] - mu[k], B[k], D[k]));
        }
        LdV<T, VEC>::st(dx + v * VEC, e);
        if (DRES) LdV<T, VEC>::st(dres + v * VEC, d);
    }
}

template <typename T, bool RELU, bool HAVE_Y, bool DRES>
__global__ __launch_bounds__(BLOCK) void bwd_apply_generic_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ dx,
    T* __restrict__ dres, int64_t total, int C, const float* __restrict__ stats, const float* __restrict__ coef) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int c = (int)(i % C);
        const float mu = stats[c], alpha = stats[2 * C + c], beta = stats[3 * C + c];
        const float e = Elt<T>::ld(x + i);
        float g = Elt<T>::ld(dy + i);
        if (RELU) {
            const float act = HAVE_Y ? Elt<T>::ld(y + i) : fmaf(e, alpha, beta);
            g = (act > 0.f) ? g : 0.f;
        }
        Elt<T>::st(dx + i, fmaf(g, alpha, fmaf(e - mu, coef[c], coef[C + c])));
        if (DRES) Elt<T>::st(dres + i, g);
    }
}

// sum of the NS accumulator copies of channel c; all loads are issued before the first add (NS <= 16)
__device__ __forceinline__ void fold_slots(const double* __restrict__ acc, int C, int NS, int c, double& a, double& b) {
    double av[16], bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        av[s] = (s < NS) ? acc[(int64_t)(2 * s) * C + c] : 0.0;
        bv[s] = (s < NS) ? acc[(int64_t)(2 * s + 1) * C + c] : 0.0;
    }
    a = b = 0.0;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        a += av[s];
        b += bv[s];
    }
}

// Many channels, few accumulator copies (NS * C <= 1024: C >= 512 has NS <= 2): a thread derives up to AHEAD_IT channels
// (c = tid + it * BLOCK).  One channel at a time each iteration waited for its own accumulator reads — memory-side data, ~1 us —
// before the next one's were asked for: 8 iterations at 2 048 channels, 10 of the kernel's 20 us (tools/probe/bn_apply_prologue.py:
// 20.6 us against 9.9 us with the coefficients given).  Here every iteration's reads leave first; the sums keep fold_slots' order
// ((0.0 + copy 0) + copy 1, the further zero terms change nothing).
constexpr int AHEAD_IT = 8;
__device__ __forceinline__ void fold_ahead(const double* __restrict__ acc, int C, int NS, double (&a)[AHEAD_IT], double (&b)[AHEAD_IT]) {
    double a0[AHEAD_IT], b0[AHEAD_IT], a1[AHEAD_IT], b1[AHEAD_IT];
#pragma unroll
    for (int it = 0; it < AHEAD_IT; ++it) {
        const int c = threadIdx.x + it * BLOCK;
        const bool ok = c < C, two = ok && NS == 2;
        a0[it] = ok ? acc[c] : 0.0;
        b0[it] = ok ? acc[(int64_t)C + c] : 0.0;
        a1[it] = two ? acc[(int64_t)2 * C + c] : 0.0;
        b1[it] = two ? acc[(int64_t)3 * C + c] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < AHEAD_IT; ++it) {
        a[it] = (0.0 + a0[it]) + a1[it];
        b[it] = (0.0 + b0[it]) + b1[it];
    }
}
__device__ __forceinline__ bool fold_ahead_ok(int C, int NS) { return C > BLOCK && C <= AHEAD_IT * BLOCK && NS <= 2; }

// ---- accumulator path: coefficients derived in the prologue of the streaming kernels themselves ----------------
// forward: acc[slot][0][c] = sum (x - shift), acc[slot][1][c] = sum (x - shift)^2 (f64); shift = given [C] floats, or
// row 0 of x.  Each block folds the NS copies and derives alpha/beta for all C channels once (one channel per thread,
// coalesced f64 loads), parks them in LDS, and its threads pick up their own VEC channels from there; block 0 also
// publishes stats[4][C] for the backward and updates the running statistics.  `shift` is never the live running_mean
// buffer (the producing convolution snapshots it behind the accumulators), so that in-place update races with nothing.
template <typename T, int VEC, bool RES, bool RELU, bool GROUPED>
__global__ __launch_bounds__(BLOCK) void apply_acc_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                          T* __restrict__ y, int64_t nvec, int CV, int C, int NS,
                                                          const double* __restrict__ acc,
                                                          const float* __restrict__ shift, double inv_m, float unbias,
                                                          float eps, float momentum, const float* __restrict__ weight,
                                                          const float* __restrict__ bias, float* __restrict__ stats,
                                                          float* rmean, float* rvar, int64_t* nbt, int shift_in_acc,
                                                          int64_t acc_stride, int updates) {
    // gridDim.y = image groups (half-batches with separate statistics): group g owns rows [g*nvec, (g+1)*nvec) vectors,
    // accumulator block acc + g*acc_stride and stats + g*4*C.  The running statistics see the groups as consecutive
    // forward passes: block (0, 0) alone applies all the updates, in group order.
    extern __shared__ __attribute__((aligned(16))) float coef[];   // [2][C]: alpha | beta
    // (GROUPED is a template flag so that the ordinary launch carries none of the multi-group code)
    const int grp = GROUPED ? blockIdx.y : 0, G = GROUPED ? gridDim.y : 1;
    x += (int64_t)grp * nvec * VEC;
    y += (int64_t)grp * nvec * VEC;
    if (RES) res += (int64_t)grp * nvec * VEC;
    const double* acc_g = acc + (int64_t)grp * acc_stride;
    float* stats_g = stats + (int64_t)grp * 4 * C;
    auto derive = [&](int c, double a, double b, float wv, float bv, float sh) {
        const double dm = a * inv_m;
        const double m2 = fmax(b - a * dm, 0.0);
        const float mean = (float)((double)sh + dm);      // (not sh + (float)dm: that rounds at |sh|, not at |mean|)
        const float varb = (float)(m2 * inv_m);
        const float is = 1.0f / sqrtf(varb + eps);
        float alpha, beta;
        affine_coeffs(mean, is, wv, bv, alpha, beta);
        coef[c] = alpha;
        coef[C + c] = beta;
        if (blockIdx.x == 0) {
            stats_g[c] = mean;
            stats_g[C + c] = is;
            stats_g[2 * C + c] = alpha;
            stats_g[3 * C + c] = beta;
            if (grp == 0) {
                if (rmean)
                    for (int u = 0; u < updates; ++u) {   // updates > 1: stands for that many identical passes (ungrouped)
                        rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
                        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (varb * unbias);
                    }
                if (c == 0 && nbt) *nbt += G * updates;
                // the later groups' running-stat updates, in order
                if constexpr (GROUPED) for (int gg = 1; gg < G; ++gg) {
                    const double* ag = acc + (int64_t)gg * acc_stride;
                    double a2, b2;
                    fold_slots(ag, C, NS, c, a2, b2);
                    const float sh2 = reinterpret_cast<const float*>(ag + (int64_t)2 * NS * C)[c];
                    const double dm2 = a2 * inv_m;
                    const float mean2 = (float)((double)sh2 + dm2);
                    const float var2 = (float)(fmax(b2 - a2 * dm2, 0.0) * inv_m);
                    if (rmean) {
                        rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean2;
                        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (var2 * unbias);
                    }
                }
            }
        }
    };
    auto shift_of = [&](int c) {
        return shift_in_acc ? reinterpret_cast<const float*>(acc_g + (int64_t)2 * NS * C)[c] : (shift ? shift[c] : Elt<T>::ld(x + c));
    };
    if (fold_ahead_ok(C, NS)) {
        double a_[AHEAD_IT], b_[AHEAD_IT];
        float w_[AHEAD_IT], bi_[AHEAD_IT], sh_[AHEAD_IT];
        fold_ahead(acc_g, C, NS, a_, b_);
#pragma unroll
        for (int it = 0; it < AHEAD_IT; ++it) {
            const int c = threadIdx.x + it * BLOCK;
            const bool ok = c < C;
            w_[it] = (ok && weight) ? weight[c] : 1.f;
            bi_[it] = (ok && bias) ? bias[c] : 0.f;
            sh_[it] = ok ? shift_of(c) : 0.f;
        }
#pragma unroll
        for (int it = 0; it < AHEAD_IT; ++it) {
            const int c = threadIdx.x + it * BLOCK;
            if (c < C) derive(c, a_[it], b_[it], w_[it], bi_[it], sh_[it]);
        }
    } else {
        for (int c = threadIdx.x; c < C; c += BLOCK) {
            const float wv = weight ? weight[c] : 1.f, bv = bias ? bias[c] : 0.f;
            double a, b;
            fold_slots(acc_g, C, NS, c, a, b);
            derive(c, a, b, wv, bv, shift_of(c));
        }
    }
    __syncthreads();
    const int c0 = (threadIdx.x % CV) * VEC;
    float alpha[VEC], beta[VEC];
    ld_coef<VEC>(coef, c0, alpha);
    ld_coef<VEC>(coef + C, c0, beta);
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
#pragma unroll 4
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += stride) {
        float e[VEC], r[VEC];
        LdV<T, VEC>::ld(x + v * VEC, e);
        if (RES) LdV<T, VEC>::ld(res + v * VEC, r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float t = fmaf(e[k], alpha[k], beta[k]);
            if (RES) t += r[k];
            if (RELU) t = (t > 0.f) ? t : ((t != t) ? t : 0.f);
            e[k] = t;
        }
        LdV<T, VEC>::st(y + v * VEC, e);
    }
}

// The end of a residual block with a projection shortcut (Classification/resnet_s.py:72-77, option B):
//   y = relu(bn_a(x_a) + bn_b(x_b))      x_a = the block's last convolution output, x_b = the 1x1 projection's output
// as ONE launch: both BatchNorms' coefficients are derived in the prologue from their own accumulator blocks (filled by the
// producing convolutions' epilogues), block 0 publishes both stats[4][C] blocks and applies both running-statistics
// updates; the projection's normalised tensor is never written (one launch, one tensor write and one read less than
// apply_acc(x_b) followed by apply_acc(x_a, res)).
struct DualBN {
    const double* acc;
    const float* weight;
    const float* bias;
    float* stats;
    float* rmean;
    float* rvar;
    int64_t* nbt;
    float eps, momentum;
};
template <typename T, int VEC>
__global__ __launch_bounds__(BLOCK) void apply_acc_dual_kernel(const T* __restrict__ xa, const T* __restrict__ xb,
                                                               T* __restrict__ y, int64_t nvec, int CV, int C, int NS,
                                                               DualBN A, DualBN B, double inv_m, float unbias, int updates) {
    extern __shared__ __attribute__((aligned(16))) float coef[];   // [4][C]: alpha_a | beta_a | alpha_b | beta_b
    for (int c = threadIdx.x; c < C; c += BLOCK) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const DualBN& P = q ? B : A;
            const float wv = P.weight ? P.weight[c] : 1.f, bv = P.bias ? P.bias[c] : 0.f;
            double a, b;
            fold_slots(P.acc, C, NS, c, a, b);
            const float sh = reinterpret_cast<const float*>(P.acc + (int64_t)2 * NS * C)[c];
            const double dm = a * inv_m;
            const double m2 = fmax(b - a * dm, 0.0);
            const float mean = (float)((double)sh + dm);
            const float varb = (float)(m2 * inv_m);
            const float is = 1.0f / sqrtf(varb + P.eps);
            float alpha, beta;
            affine_coeffs(mean, is, wv, bv, alpha, beta);
            coef[(2 * q) * C + c] = alpha;
            coef[(2 * q + 1) * C + c] = beta;
            if (blockIdx.x == 0) {
                P.stats[c] = mean;
                P.stats[C + c] = is;
                P.stats[2 * C + c] = alpha;
                P.stats[3 * C + c] = beta;
                if (P.rmean)
                    for (int u = 0; u < updates; ++u) {
                        P.rmean[c] = (1.0f - P.momentum) * P.rmean[c] + P.momentum * mean;
                        P.rvar[c] = (1.0f - P.momentum) * P.rvar[c] + P.momentum * (varb * unbias);
                    }
                if (c == 0 && P.nbt) *P.nbt += updates;
            }
        }
    }
    __syncthreads();
    const int c0 = (threadIdx.x % CV) * VEC;
    float aa[VEC], ba[VEC], ab[VEC], bb[VEC];
    ld_coef<VEC>(coef, c0, aa);
    ld_coef<VEC>(coef + C, c0, ba);
    ld_coef<VEC>(coef + 2 * C, c0, ab);
    ld_coef<VEC>(coef + 3 * C, c0, bb);
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
#pragma unroll 4
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += stride) {
        float e[VEC], r[VEC];
        LdV<T, VEC>::ld(xa + v * VEC, e);
        LdV<T, VEC>::ld(xb + v * VEC, r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float t = fmaf(e[k], aa[k], ba[k]) + fmaf(r[k], ab[k], bb[k]);
            e[k] = (t > 0.f) ? t : ((t != t) ? t : 0.f);
        }
        LdV<T, VEC>::st(y + v * VEC, e);
    }
}

// backward: acc[slot][0][c] = sum g, acc[slot][1][c] = sum g*(x - mean) (f64) -> B, D per channel, same block-level
// prologue; block 0 writes dweight / dbias.
template <typename T, int VEC, bool RELU, bool HAVE_Y, bool DRES, bool GROUPED>
__global__ __launch_bounds__(BLOCK) void bwd_apply_acc_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const T* __restrict__ y, T* __restrict__ dx,
                                                              T* __restrict__ dres, int64_t nvec, int CV, int C, int NS,
                                                              const float* __restrict__ stats,
                                                              const double* __restrict__ acc, double inv_m,
                                                              float* dweight, float* dbias, int accumulate,
                                                              int64_t acc_stride) {
    // gridDim.y = image groups, as in apply_acc_kernel; block (0, 0) alone adds every group's sums to dweight / dbias
    extern __shared__ __attribute__((aligned(16))) float coef[];   // [2][C]: B | D
    const int grp = GROUPED ? blockIdx.y : 0, G = GROUPED ? gridDim.y : 1;
    dy += (int64_t)grp * nvec * VEC;
    x += (int64_t)grp * nvec * VEC;
    dx += (int64_t)grp * nvec * VEC;
    if (RELU && HAVE_Y) y += (int64_t)grp * nvec * VEC;
    if (DRES) dres += (int64_t)grp * nvec * VEC;
    const double* acc_g = acc + (int64_t)grp * acc_stride;
    const float* stats_g = stats + (int64_t)grp * 4 * C;
    auto derive = [&](int c, double a, double b, float is, float alpha) {
        float sum_g = (float)a;
        float sum_gx = (float)b * is;
        coef[c] = -alpha * is * (float)((double)sum_gx * inv_m);
        coef[C + c] = -alpha * (float)((double)sum_g * inv_m);
        if (blockIdx.x == 0 && grp == 0) {
            if constexpr (GROUPED) for (int gg = 1; gg < G; ++gg) {        // the other groups' sums for dweight / dbias
                double a2, b2;
                fold_slots(acc + (int64_t)gg * acc_stride, C, NS, c, a2, b2);
                sum_g += (float)a2;
                sum_gx += (float)b2 * stats[(int64_t)gg * 4 * C + C + c];
            }
            if (dbias) dbias[c] = accumulate ? dbias[c] + sum_g : sum_g;
            if (dweight) dweight[c] = accumulate ? dweight[c] + sum_gx : sum_gx;
        }
    };
    if (fold_ahead_ok(C, NS)) {
        double a_[AHEAD_IT], b_[AHEAD_IT];
        float is_[AHEAD_IT], al_[AHEAD_IT];
        fold_ahead(acc_g, C, NS, a_, b_);
#pragma unroll
        for (int it = 0; it < AHEAD_IT; ++it) {
            const int c = threadIdx.x + it * BLOCK;
            is_[it] = c < C ? stats_g[C + c] : 0.f;
            al_[it] = c < C ? stats_g[2 * C + c] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < AHEAD_IT; ++it) {
            const int c = threadIdx.x + it * BLOCK;
            if (c < C) derive(c, a_[it], b_[it], is_[it], al_[it]);
        }
    } else {
        for (int c = threadIdx.x; c < C; c += BLOCK) {
            double a, b;
            fold_slots(acc_g, C, NS, c, a, b);
            derive(c, a, b, stats_g[C + c], stats_g[2 * C + c]);
        }
    }
    __syncthreads();
    stats += (int64_t)grp * 4 * C;
    const int c0 = (threadIdx.x % CV) * VEC;
    float mu[VEC], alpha[VEC], beta[VEC], B[VEC], D[VEC];
    ld_coef<VEC>(stats, c0, mu);
    ld_coef<VEC>(stats + 2 * C, c0, alpha);
    if (RELU && !HAVE_Y) ld_coef<VEC>(stats + 3 * C, c0, beta);
    ld_coef<VEC>(coef, c0, B);
    ld_coef<VEC>(coef + C, c0, D);
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
#pragma unroll 4
    for (int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += stride) {
        float d[VEC], e[VEC], o[VEC];
        LdV<T, VEC>::ld(dy + v * VEC, d);
        LdV<T, VEC>::ld(x + v * VEC, e);
        if (RELU && HAVE_Y) LdV<T, VEC>::ld(y + v * VEC, o);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float g = d[k];
            if (RELU) {
                const float act = HAVE_Y ? o[k] : fmaf(e[k], alpha[k], beta[k]);
                g = (act > 0.f) ? g : 0.f;
            }
            d[k] = g;
            e[k] = fmaf(g, alpha[k], fmaf(e[k] - mu[k], B[k], D[k]));
        }
        LdV<T, VEC>::st(dx + v * VEC, e);
        if (DRES) LdV<T, VEC>::st(dres + v * VEC, d);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
struct Plan {
    bool vec;      // vector mapping usable
    int CV;        // channel vectors per row
    int64_t nvec;  // total vectors
    int G;         // blocks == partials per channel
    double tensor_bytes;
};

template <typename T>
bool make_plan(int64_t M, int64_t C, std::initializer_list<const void*> ptrs, Plan& p) {
    constexpr int NV = Elt<T>::VEC;
    if (M <= 0 || C <= 0 || C > (1 << 20)) return false;
    bool ok = (C % NV == 0) && (C / NV <= BLOCK) && (BLOCK % (C / NV) == 0);
    for (const void* q : ptrs)
        if (q && !aligned(q, 16)) ok = false;
    p.vec = ok;
    p.CV = ok ? (int)(C / NV) : 0;
    p.nvec = ok ? M * (C / NV) : 0;
    int64_t work = ok ? (p.nvec + BLOCK - 1) / BLOCK : (M + 15) / 16;  // >= 1 vector/thread or >= 16 rows/block
    int64_t G = (work + 3) / 4;                                           // ~4 iterations per thread
    if (G > MAX_G) G = MAX_G;
    if (G < 1) G = 1;
    p.G = (int)G;
    p.tensor_bytes = (double)M * (double)C * sizeof(T);
    return true;
}

static inline int apply_grid(const Plan& p, int64_t total) {
    // ~4 vectors per thread: amortises the per-thread coefficient loads, still >= 2 blocks per CU on the big tensors
    static const int vpt = [] { const char* v = getenv("AFAN_BN_VPT"); return v ? atoi(v) : 4; }();
    static const int cap = [] { const char* v = getenv("AFAN_BN_MAXBLOCKS"); return v ? atoi(v) : 512; }();     // (512 vs 2048 blocks: +0.6 % of the step)
    return grid_for(p.vec ? (p.nvec + vpt - 1) / vpt : total, BLOCK, cap);
}

template <typename T>
int run_stats(const Plan& p, const T* x_, int64_t M, int64_t C, float eps, float momentum, const float* weight,
              const float* bias, float* ws, float* stats, float* rmean, float* rvar, int64_t* nbt, hipStream_t st) {
    {
        AFAN_PROF("bn_nhwc_stats_kernel", p.tensor_bytes, st);
        if (p.vec) stats_kernel<T, Elt<T>::VEC><<<p.G, BLOCK, 0, st>>>(x_, p.nvec, p.CV, (int)C, ws);
        else stats_generic_kernel<T><<<p.G, BLOCK, 0, st>>>(x_, M, (int)C, ws);
    }
    AFAN_LAUNCH_CHECK();
    AFAN_PROF("bn_nhwc_finalize_kernel", 8.0 * C * p.G, st);
    finalize_kernel<T, 0><<<(unsigned)((C + 3) / 4), BLOCK, 0, st>>>(ws, p.G, (int)C, x_, 1.0f / (float)M, (float)M, eps,
                                                                      momentum, weight, bias, stats, nullptr, rmean,
                                                                      rvar, nbt, nullptr, nullptr, 0, nullptr, running_updates());
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// stats: [4][C] = mean, invstd, alpha, beta.  train: computed here; eval: mean/invstd given, alpha/beta derived.
template <typename T>
int forward(const void* x, const void* res, void* y, int64_t M, int64_t C, float eps, float momentum,
            const float* weight, const float* bias, int relu, float* ws, float* stats, const float* mean_in,
            const float* invstd_in, float* rmean, float* rvar, int64_t* nbt, bool train, hipStream_t st,
            const float* partials = nullptr, int64_t partials_g = 0, const float* partials_shift = nullptr) {
    Plan p;
    if (!make_plan<T>(M, C, {x, res, y}, p)) return AFAN_ESHAPE;
    p.vec = p.vec && aligned(stats, 16);
    constexpr int NV = Elt<T>::VEC;
    const T* x_ = (const T*)x; const T* r_ = (const T*)res; T* y_ = (T*)y;
    if (train && partials) {
        // moments already summed per tile by the producing convolution's epilogue: only the fold is left
        AFAN_PROF("bn_nhwc_finalize_kernel", 8.0 * C * partials_g, st);
        finalize_kernel<T, 0><<<(unsigned)((C + 3) / 4), BLOCK, 0, st>>>(
            partials, (int)partials_g, (int)C, nullptr, 1.0f / (float)M, (float)M, eps, momentum, weight, bias, stats,
            nullptr, rmean, rvar, nbt, nullptr, nullptr, 0, partials_shift, running_updates());
        AFAN_LAUNCH_CHECK();
    } else if (train) {
        int e = run_stats<T>(p, x_, M, C, eps, momentum, weight, bias, ws, stats, rmean, rvar, nbt, st);
        if (e) return e;
    } else if (mean_in) {       // (mean_in == NULL: `stats` already holds mean | invstd | alpha | beta — afan_affine_apply)
        AFAN_PROF("bn_nhwc_coef_kernel", 24.0 * C, st);
        coef_kernel<<<(unsigned)((C + BLOCK - 1) / BLOCK), BLOCK, 0, st>>>((int)C, mean_in, invstd_in, weight, bias, stats);
        AFAN_LAUNCH_CHECK();
    }
    AFAN_PROF("bn_nhwc_apply_kernel", p.tensor_bytes * (res ? 3 : 2), st);
    const int grid = apply_grid(p, M * C);
#define AFAN_GO(RES, RELU)                                                                                         \
    do {                                                                                                           \
        if (p.vec) apply_kernel<T, NV, RES, RELU><<<grid, BLOCK, 0, st>>>(x_, r_, y_, p.nvec, p.CV, (int)C, stats); \
        else apply_generic_kernel<T, RES, RELU><<<grid, BLOCK, 0, st>>>(x_, r_, y_, M * C, (int)C, stats);          \
    } while (0)
    if (res) { if (relu) AFAN_GO(true, true); else AFAN_GO(true, false); }
    else { if (relu) AFAN_GO(false, true); else AFAN_GO(false, false); }
#undef AFAN_GO
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <typename T>
int stats_only(const void* x, int64_t M, int64_t C, float eps, float momentum, float* ws, float* stats,
               float* rmean, float* rvar, int64_t* nbt, hipStream_t st) {
    Plan p;
    if (!make_plan<T>(M, C, {x}, p)) return AFAN_ESHAPE;
    return run_stats<T>(p, (const T*)x, M, C, eps, momentum, nullptr, nullptr, ws, stats, rmean, rvar, nbt, st);
}

template <typename T>
int backward(const void* dy, const void* x, const void* y, void* dx, void* dres, int64_t M, int64_t C,
             const float* stats, int relu, float* ws, float* dweight, float* dbias, int accumulate, hipStream_t st,
             const float* partials = nullptr, int64_t partials_g = 0) {
    Plan p;
    if (!make_plan<T>(M, C, {dy, x, y, dx, dres}, p)) return AFAN_ESHAPE;
    p.vec = p.vec && aligned(stats, 16);
    constexpr int NV = Elt<T>::VEC;
    const T* dy_ = (const T*)dy; const T* x_ = (const T*)x; const T* y_ = (const T*)y;
    float* coef = ws + (int64_t)2 * C * MAX_G;  // [2][C] after the partials (16-byte aligned: C*MAX_G*8 bytes)
    const float* red = ws;
    int red_g = p.G;
    if (partials) {   // the producing dgrad's epilogue already took the sums
        red = partials;
        red_g = (int)partials_g;
    } else {
        AFAN_PROF("bn_nhwc_bwd_reduce_kernel", p.tensor_bytes * ((relu && y) ? 3 : 2), st);
#define AFAN_RED(RELU, HY)                                                                                        \
    do {                                                                                                          \
        if (p.vec) bwd_reduce_kernel<T, NV, RELU, HY><<<p.G, BLOCK, 0, st>>>(dy_, x_, y_, p.nvec, p.CV, (int)C, stats, ws); \
        else bwd_reduce_generic_kernel<T, RELU, HY><<<p.G, BLOCK, 0, st>>>(dy_, x_, y_, M, (int)C, stats, ws);      \
    } while (0)
        if (!relu) AFAN_RED(false, false);
        else if (y) AFAN_RED(true, true);
        else AFAN_RED(true, false);
#undef AFAN_RED
    }
    AFAN_LAUNCH_CHECK();
    {
        AFAN_PROF("bn_nhwc_finalize_kernel", 8.0 * C * red_g, st);
        finalize_kernel<T, 1><<<(unsigned)((C + 3) / 4), BLOCK, 0, st>>>(
            red, red_g, (int)C, nullptr, 1.0f / (float)M, (float)M, 0.f, 0.f, nullptr, nullptr, const_cast<float*>(stats),
            coef, nullptr, nullptr, nullptr, dweight, dbias, accumulate, nullptr, 1);
    }
    AFAN_LAUNCH_CHECK();
    const int grid = apply_grid(p, M * C);
    AFAN_PROF("bn_nhwc_bwd_apply_kernel", p.tensor_bytes * (3 + ((relu && y) ? 1 : 0) + (dres ? 1 : 0)), st);
#define AFAN_APP(RELU, HY, DR)                                                                                    \
    do {                                                                                                          \
        if (p.vec) bwd_apply_kernel<T, NV, RELU, HY, DR><<<grid, BLOCK, 0, st>>>(dy_, x_, y_, (T*)dx, (T*)dres, p.nvec, p.CV, (int)C, stats, coef); \
        else bwd_apply_generic_kernel<T, RELU, HY, DR><<<grid, BLOCK, 0, st>>>(dy_, x_, y_, (T*)dx, (T*)dres, M * C, (int)C, stats, coef); \
    } while (0)
    if (!relu) { if (dres) AFAN_APP(false, false, true); else AFAN_APP(false, false, false); }
    else if (y) { if (dres) AFAN_APP(true, true, true); else AFAN_APP(true, true, false); }
    else { if (dres) AFAN_APP(true, false, true); else AFAN_APP(true, false, false); }
#undef AFAN_APP
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// ---- accumulator path, host side.  acc: double[acc_doubles(C)] — NS = acc_slots(C) copies of [2][C] accumulators
// (zeroed by the caller before the producer ran), then C floats: the shift the producer used (written by the producing
// convolution's epilogue).
template <typename T>
int forward_acc(const void* x, const void* res, void* y, int64_t M, int64_t C, float eps, float momentum,
                const float* weight, const float* bias, int relu, double* acc, int acc_ready, float* stats,
                float* rmean, float* rvar, int64_t* nbt, hipStream_t st, int groups = 1) {
    // M: rows per group; groups > 1 needs accumulators that a convolution epilogue already filled
    if (groups < 1 || (groups > 1 && !acc_ready)) return AFAN_ESHAPE;
    Plan p;
    if (!make_plan<T>(M, C, {x, res, y}, p)) return AFAN_ESHAPE;
    if (!p.vec || !aligned(stats, 16) || !aligned(acc, 16)) return AFAN_ESHAPE;
    constexpr int NV = Elt<T>::VEC;
    const T* x_ = (const T*)x; const T* r_ = (const T*)res; T* y_ = (T*)y;
    if (!acc_ready) {
        AFAN_PROF("bn_nhwc_stats_kernel", p.tensor_bytes, st);
        stats_kernel<T, NV, true><<<p.G, BLOCK, 0, st>>>(x_, p.nvec, p.CV, (int)C, nullptr, acc, acc_slots(C));
        AFAN_LAUNCH_CHECK();
    }
    const int NS = acc_slots(C);
    const float* shift = acc_ready ? reinterpret_cast<const float*>(acc + (int64_t)2 * NS * C) : nullptr;
    const size_t lds = (size_t)2 * C * sizeof(float);
    const double inv_m = 1.0 / (double)M;
    const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f;
    AFAN_PROF("bn_nhwc_apply_kernel", p.tensor_bytes * (res ? 3 : 2) * groups, st);
    const dim3 grid((unsigned)apply_grid(p, M * C), (unsigned)groups);
    const int64_t acc_stride = (acc_doubles(C) + 1) & ~(int64_t)1;
#define AFAN_GO1(RES, RELU, GR)                                                                                   \
    apply_acc_kernel<T, NV, RES, RELU, GR><<<grid, BLOCK, lds, st>>>(x_, r_, y_, p.nvec, p.CV, (int)C, NS, acc, shift,   \
                                                                   inv_m, unbias, eps, momentum, weight, bias, stats,  \
                                                                   rmean, rvar, nbt, acc_ready, acc_stride,        \
                                                                   groups > 1 ? 1 : running_updates())
#define AFAN_GO(RES, RELU) do { if (groups > 1) AFAN_GO1(RES, RELU, true); else AFAN_GO1(RES, RELU, false); } while (0)
    if (res) { if (relu) AFAN_GO(true, true); else AFAN_GO(true, false); }
    else { if (relu) AFAN_GO(false, true); else AFAN_GO(false, false); }
#undef AFAN_GO
#undef AFAN_GO1
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <typename T>
int forward_acc_dual(const void* xa, const void* xb, void* y, int64_t M, int64_t C, const DualBN& A, const DualBN& B,
                     hipStream_t st) {
    Plan p;
    if (!make_plan<T>(M, C, {xa, xb, y}, p)) return AFAN_ESHAPE;
    if (!p.vec || !aligned(A.stats, 16) || !aligned(B.stats, 16) || !aligned(A.acc, 16) || !aligned(B.acc, 16)) return AFAN_ESHAPE;
    constexpr int NV = Elt<T>::VEC;
    const int NS = acc_slots(C);
    const size_t lds = (size_t)4 * C * sizeof(float);
    const double inv_m = 1.0 / (double)M;
    const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f;
    AFAN_PROF("bn_nhwc_apply_kernel", p.tensor_bytes * 3, st);
    apply_acc_dual_kernel<T, NV><<<apply_grid(p, M * C), BLOCK, lds, st>>>((const T*)xa, (const T*)xb, (T*)y, p.nvec, p.CV, (int)C,
                                                                           NS, A, B, inv_m, unbias, running_updates());
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <typename T>
int backward_acc(const void* dy, const void* x, const void* y, void* dx, void* dres, int64_t M, int64_t C,
                 const float* stats, int relu, double* acc, int acc_ready, float* dweight, float* dbias,
                 int accumulate, hipStream_t st, int groups = 1) {
    if (groups < 1 || (groups > 1 && !acc_ready)) return AFAN_ESHAPE;
    Plan p;
    if (!make_plan<T>(M, C, {dy, x, y, dx, dres}, p)) return AFAN_ESHAPE;
    if (!p.vec || !aligned(stats, 16) || !aligned(acc, 16)) return AFAN_ESHAPE;
    constexpr int NV = Elt<T>::VEC;
    const T* dy_ = (const T*)dy; const T* x_ = (const T*)x; const T* y_ = (const T*)y;
    if (!acc_ready) {
        AFAN_PROF("bn_nhwc_bwd_reduce_kernel", p.tensor_bytes * ((relu && y) ? 3 : 2), st);
#define AFAN_RED(RELU, HY)                                                                                        \
    bwd_reduce_kernel<T, NV, RELU, HY, true><<<p.G, BLOCK, 0, st>>>(dy_, x_, y_, p.nvec, p.CV, (int)C, stats, nullptr, acc, acc_slots(C))
        if (!relu) AFAN_RED(false, false);
        else if (y) AFAN_RED(true, true);
        else AFAN_RED(true, false);
#undef AFAN_RED
        AFAN_LAUNCH_CHECK();
    }
    const dim3 grid((unsigned)apply_grid(p, M * C), (unsigned)groups);
    const int64_t acc_stride = (acc_doubles(C) + 1) & ~(int64_t)1;
    const double inv_m = 1.0 / (double)M;
    const int NS = acc_slots(C);
    const size_t lds = (size_t)2 * C * sizeof(float);
    AFAN_PROF("bn_nhwc_bwd_apply_kernel", p.tensor_bytes * (3 + ((relu && y) ? 1 : 0) + (dres ? 1 : 0)) * groups, st);
#define AFAN_APP1(RELU, HY, DR, GR)                                                                               \
    bwd_apply_acc_kernel<T, NV, RELU, HY, DR, GR><<<grid, BLOCK, lds, st>>>(dy_, x_, y_, (T*)dx, (T*)dres, p.nvec, p.CV, \
                                                                            (int)C, NS, stats, acc, inv_m, dweight, \
                                                                            dbias, accumulate, acc_stride)
#define AFAN_APP(RELU, HY, DR) do { if (groups > 1) AFAN_APP1(RELU, HY, DR, true); else AFAN_APP1(RELU, HY, DR, false); } while (0)
    if (!relu) { if (dres) AFAN_APP(false, false, true); else AFAN_APP(false, false, false); }
    else if (y) { if (dres) AFAN_APP(true, true, true); else AFAN_APP(true, true, false); }
    else { if (dres) AFAN_APP(true, false, true); else AFAN_APP(true, false, false); }
#undef AFAN_APP1
#undef AFAN_APP
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int fwd_acc(int dtype, const void* x, const void* res, void* y, int64_t M, int64_t C, float eps, float mom,
            const float* w, const float* b, int relu, double* acc, int acc_ready, float* stats, float* rm, float* rv,
            int64_t* nbt, hipStream_t st, int groups) {
    return dtype == AFAN_F32 ? forward_acc<float>(x, res, y, M, C, eps, mom, w, b, relu, acc, acc_ready, stats, rm, rv, nbt, st, groups)
                             : forward_acc<uint16_t>(x, res, y, M, C, eps, mom, w, b, relu, acc, acc_ready, stats, rm, rv, nbt, st, groups);
}
int fwd_acc_dual(int dtype, const void* xa, const void* xb, void* y, int64_t M, int64_t C, float eps_a, float mom_a,
                 const float* w_a, const float* b_a, double* acc_a, float* stats_a, float* rm_a, float* rv_a, int64_t* nbt_a,
                 float eps_b, float mom_b, const float* w_b, const float* b_b, double* acc_b, float* stats_b, float* rm_b,
                 float* rv_b, int64_t* nbt_b, hipStream_t st) {
    const DualBN A{acc_a, w_a, b_a, stats_a, rm_a, rv_a, nbt_a, eps_a, mom_a}, B{acc_b, w_b, b_b, stats_b, rm_b, rv_b, nbt_b, eps_b, mom_b};
    return dtype == AFAN_F32 ? forward_acc_dual<float>(xa, xb, y, M, C, A, B, st) : forward_acc_dual<uint16_t>(xa, xb, y, M, C, A, B, st);
}
int bwd_acc(int dtype, const void* dy, const void* x, const void* y, void* dx, void* dres, int64_t M, int64_t C,
            const float* stats_in, int relu, double* acc, int acc_ready, float* dw, float* db, int accumulate,
            hipStream_t st, int groups) {
    return dtype == AFAN_F32 ? backward_acc<float>(dy, x, y, dx, dres, M, C, stats_in, relu, acc, acc_ready, dw, db, accumulate, st, groups)
                             : backward_acc<uint16_t>(dy, x, y, dx, dres, M, C, stats_in, relu, acc, acc_ready, dw, db, accumulate, st, groups);
}
int64_t acc_doubles(int64_t C) { return C > 0 ? (int64_t)2 * acc_slots(C) * C + (C + 1) / 2 : 0; }
int acc_slot_count(int64_t C) { return acc_slots(C); }
// can the accumulator path take this channel count (vector mapping: C/VEC divides the block)
int acc_supported(int dtype, int64_t C) {
    const int nv = dtype == AFAN_F32 ? Elt<float>::VEC : Elt<uint16_t>::VEC;
    return C > 0 && C % nv == 0 && C / nv <= BLOCK && BLOCK % (C / nv) == 0;
}

// entry points used by afan_bn.hip's extern "C" dispatch (dtype: 0 = f32, 1 = bf16)
int fwd(int dtype, const void* x, const void* res, void* y, int64_t M, int64_t C, float eps, float mom, const float* w,
        const float* b, int relu, float* ws, float* stats, const float* mean_in, const float* invstd_in, float* rm,
        float* rv, int64_t* nbt, bool train, hipStream_t st, const float* partials, int64_t partials_g,
        const float* partials_shift) {
    return dtype == AFAN_F32
               ? forward<float>(x, res, y, M, C, eps, mom, w, b, relu, ws, stats, mean_in, invstd_in, rm, rv, nbt, train, st, partials, partials_g, partials_shift)
               : forward<uint16_t>(x, res, y, M, C, eps, mom, w, b, relu, ws, stats, mean_in, invstd_in, rm, rv, nbt, train, st, partials, partials_g, partials_shift);
}
int stats(int dtype, const void* x, int64_t M, int64_t C, float eps, float mom, float* ws, float* stats_out, float* rm,
          float* rv, int64_t* nbt, hipStream_t st) {
    return dtype == AFAN_F32 ? stats_only<float>(x, M, C, eps, mom, ws, stats_out, rm, rv, nbt, st)
                             : stats_only<uint16_t>(x, M, C, eps, mom, ws, stats_out, rm, rv, nbt, st);
}
int bwd(int dtype, const void* dy, const void* x, const void* y, void* dx, void* dres, int64_t M, int64_t C,
        const float* stats_in, int relu, float* ws, float* dw, float* db, int acc, hipStream_t st,
        const float* partials, int64_t partials_g) {
    return dtype == AFAN_F32
               ? backward<float>(dy, x, y, dx, dres, M, C, stats_in, relu, ws, dw, db, acc, st, partials, partials_g)
               : backward<uint16_t>(dy, x, y, dx, dres, M, C, stats_in, relu, ws, dw, db, acc, st, partials, partials_g);
}
int coefs(int64_t C, const float* mean, const float* invstd, const float* w, const float* b, float* out, hipStream_t st) {
    AFAN_PROF("bn_nhwc_coef_kernel", 24.0 * C, st);
    coef_kernel<<<(unsigned)((C + BLOCK - 1) / BLOCK), BLOCK, 0, st>>>((int)C, mean, invstd, w, b, out);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}
int set_running_updates(int n) {
    const int old = g_running_updates;
    g_running_updates = n < 0 ? 1 : n;     // 0: normalise with the batch moments, leave the running statistics alone (a deferred update)
    return old;
}
// partials [2][C][MAX_G] + coef [2][C] + eval-mode stats [4][C]
int64_t workspace_floats(int64_t c) { return c > 0 ? 2 * c * MAX_G + 2 * c + 4 * c : 0; }

}  // namespace afan_nhwc
