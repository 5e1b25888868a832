// Classifier head of the slice protocol: global average pool -> flatten -> linear (resnet_s.py:108-110, run at the end of
// every tail pass: K times inside PGD and for both final passes).  In eager PyTorch this is a mean, a cast, an addmm and,
// backward, two GEMMs, a column sum, an expand/divide and a cast — ten launches of a few microseconds on tensors of a
// few hundred KB.  Here: one forward launch, two backward launches, fp32 arithmetic throughout.
//   forward : pooled[b][c] = mean_hw x[b][hw][c]   (channels-last: lanes along c, coalesced)
//             logits[b][k] = sum_c pooled[b][c] * W[k][c] + bias[k]
//   backward: dx[b][hw][c] = (sum_k dlogits[b][k] * W[k][c]) / HW
//             dW[k][c] (+)= sum_b dlogits[b][k] * pooled[b][c] ;  db[k] (+)= sum_b dlogits[b][k]
#include "afan_common.h"

using namespace afan;

namespace {

constexpr int BLOCK = 256;
constexpr int MAXK = 16;      // classes handled with register accumulators; larger heads take the vendor GEMM (caller)

template <typename T>
__global__ __launch_bounds__(BLOCK) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ pooled,
                                                         float* __restrict__ logits, int C, int HW, int K) {
    const int b = blockIdx.x;
    const T* xb = x + (int64_t)b * HW * C;
    float part[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) part[k] = 0.f;
    const float inv = 1.0f / (float)HW;
    for (int c = threadIdx.x; c < C; c += BLOCK) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += Elt<T>::ld(xb + (int64_t)p * C + c);
        const float m = s * inv;
        pooled[(int64_t)b * C + c] = m;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) part[k] = fmaf(m, W[(int64_t)k * C + c], part[k]);
    }
    __shared__ float red[BLOCK / AFAN_WAVE][MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const float v = wave_sum(part[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float s = bias ? bias[threadIdx.x] : 0.f;
        for (int w = 0; w < BLOCK / AFAN_WAVE; ++w) s += red[w][threadIdx.x];
        logits[(int64_t)b * K + threadIdx.x] = s;
    }
}

// bf16 channels-last with C % 8 == 0 and 256 % (C / 8) == 0 (512 or 64 channels here): 16-byte loads, lanes along
// 8-channel pieces, 256 / (C/8) pixel groups side by side; groups and channels are summed in fixed order.
// (The scalar kernel above took 19.8 us at 512 channels x 16 pixels: 32 dependent 2-byte loads per thread.)
__global__ __launch_bounds__(BLOCK) void head_fwd_vec_kernel(const uint16_t* __restrict__ x, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ pooled,
                                                             float* __restrict__ logits, int C, int HW, int K) {
    __shared__ float grp[BLOCK * 8];                    // [G][C]
    __shared__ float red[BLOCK / AFAN_WAVE][MAXK];
    const int b = blockIdx.x, P = C >> 3, G = BLOCK / P;
    const int pc = threadIdx.x % P, pg = threadIdx.x / P;
    const uint16_t* xb = x + (int64_t)b * HW * C + pc * 8;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    for (int p = pg; p < HW; p += G) {
        const u16x8 v = *reinterpret_cast<const u16x8*>(xb + (int64_t)p * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += bf2f(v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) grp[pg * C + pc * 8 + j] = s[j];
    __syncthreads();
    float part[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) part[k] = 0.f;
    const float inv = 1.0f / (float)HW;
    for (int c = threadIdx.x; c < C; c += BLOCK) {
        float t = 0.f;
        for (int g = 0; g < G; ++g) t += grp[g * C + c];
        const float m = t * inv;
        pooled[(int64_t)b * C + c] = m;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) part[k] = fmaf(m, W[(int64_t)k * C + c], part[k]);
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const float v = wave_sum(part[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float t = bias ? bias[threadIdx.x] : 0.f;
        for (int w = 0; w < BLOCK / AFAN_WAVE; ++w) t += red[w][threadIdx.x];
        logits[(int64_t)b * K + threadIdx.x] = t;
    }
}

// dx: block per image, thread per channel
template <typename T>
__global__ __launch_bounds__(BLOCK) void head_bwd_dx_kernel(const float* __restrict__ dlogits, const float* __restrict__ W,
                                                            T* __restrict__ dx, int C, int HW, int K) {
    const int b = blockIdx.x;
    __shared__ float dl[MAXK];
    if (threadIdx.x < K) dl[threadIdx.x] = dlogits[(int64_t)b * K + threadIdx.x];
    __syncthreads();
    const float inv = 1.0f / (float)HW;
    T* xb = dx + (int64_t)b * HW * C;
    for (int c = threadIdx.x; c < C; c += BLOCK) {
        float g = 0.f;
        for (int k = 0; k < K; ++k) g = fmaf(dl[k], W[(int64_t)k * C + c], g);
        g *= inv;
        for (int p = 0; p < HW; ++p) Elt<T>::st(xb + (int64_t)p * C + c, g);
    }
}

// dW / db: 64 (k, c) columns x 16 batch slices per block; a slice walks its images in order, the 16 slice sums are added
// in fixed order (deterministic).  db = the K extra columns whose second factor is 1.  (A thread per column walking the
// whole batch took 148 us at batch 512: a 512-long dependent chain per thread on 20 blocks.)
constexpr int DW_SLICES = 16;
__global__ __launch_bounds__(64 * DW_SLICES) void head_bwd_dw_kernel(const float* __restrict__ dlogits,
                                                                    const float* __restrict__ pooled, float* __restrict__ dW,
                                                                    float* __restrict__ db, int B, int C, int K, int accumulate) {
    __shared__ float red[DW_SLICES][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    const int64_t KC = (int64_t)K * C, total = KC + (db ? K : 0);
    float s = 0.f;
    if (i < KC) {
        const int k = (int)(i / C), c = (int)(i % C);
        for (int b = q; b < B; b += DW_SLICES) s = fmaf(dlogits[(int64_t)b * K + k], pooled[(int64_t)b * C + c], s);
    } else if (i < total) {
        const int k = (int)(i - KC);
        for (int b = q; b < B; b += DW_SLICES) s += dlogits[(int64_t)b * K + k];
    }
    red[q][e] = s;
    __syncthreads();
    if (q == 0 && i < total) {
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < DW_SLICES; ++j) sum += red[j][e];
        float* dst = i < KC ? dW + i : db + (i - KC);
        *dst = accumulate ? *dst + sum : sum;
    }
}

// Mean cross-entropy of the classifier's logits (nn.CrossEntropyLoss defaults: main_perturb.py:71, attack_algo.py:51) and
// its gradient in ONE launch: per row  lse = max + log sum exp(l - max),  loss_r = lse - l[target],
// dlogits[r][k] = (exp(l[k] - lse) - [k == target]) / N;  loss = (sum_r loss_r) / N, rows summed in fixed order.
// One block (N * K is a few thousand values here); in torch the same is log_softmax, nll_loss, a ones_like fill, two
// backward kernels and a zero fill — at the end of every tail pass, K + 2 times per iteration.
constexpr int CE_BLOCK = 256;
__global__ __launch_bounds__(CE_BLOCK) void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                          float* __restrict__ loss, float* __restrict__ dlogits, int N, int K) {
    __shared__ float red[CE_BLOCK / AFAN_WAVE];
    const float invn = 1.0f / (float)N;
    float local = 0.f;
    for (int r = threadIdx.x; r < N; r += CE_BLOCK) {
        const float* l = logits + (int64_t)r * K;
        float m = l[0];
        for (int k = 1; k < K; ++k) m = fmaxf(m, l[k]);
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += expf(l[k] - m);
        const float lse = m + logf(s);
        const int64_t t64 = target[r];
        const bool bad = t64 < 0 || t64 >= K;       // torch asserts here: poison the loss instead of reading out of range
        const int t = bad ? 0 : (int)t64;
        local += bad ? NAN : lse - l[t];
        float* d = dlogits + (int64_t)r * K;
        for (int k = 0; k < K; ++k) d[k] = bad ? NAN : (expf(l[k] - lse) - (k == t ? 1.f : 0.f)) * invn;
    }
    const float w = wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < CE_BLOCK / AFAN_WAVE; ++i) s += red[i];
        loss[0] = s * invn;
    }
}

// Many classes (ImageNet-shape heads): one WAVE per row, lanes stride over the classes; the 16 waves of the single
// workgroup take rows w, w + 16, ...; row losses are summed per wave in row order and the waves in index order
// (deterministic).  64 x 1000 logits: 313 us with the thread-per-row kernel above, ~10 us here.
constexpr int CEW_WAVES = 16;
__global__ __launch_bounds__(64 * CEW_WAVES) void ce_fwd_wide_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                                     float* __restrict__ loss, float* __restrict__ dlogits, int N, int K) {
    __shared__ float red[CEW_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float invn = 1.0f / (float)N;
    float local = 0.f;
    for (int r = w; r < N; r += CEW_WAVES) {
        const float* l = logits + (int64_t)r * K;
        float m = -INFINITY;
        for (int k = lane; k < K; k += 64) m = fmaxf(m, l[k]);
        m = wave_max(m);
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += expf(l[k] - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        const int64_t t64 = target[r];
        const bool bad = t64 < 0 || t64 >= K;
        const int t = bad ? 0 : (int)t64;
        if (lane == 0) local += bad ? NAN : lse - l[t];
        float* d = dlogits + (int64_t)r * K;
        for (int k = lane; k < K; k += 64) d[k] = bad ? NAN : (expf(l[k] - lse) - (k == t ? 1.f : 0.f)) * invn;
    }
    if (lane == 0) red[w] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < CEW_WAVES; ++i) s += red[i];
        loss[0] = s * invn;
    }
}

}  // namespace

extern "C" {

int afan_head_max_classes(void) { return MAXK; }

int afan_head_forward(const void* x, int dtype, int64_t n, int64_t c, int64_t hw, const float* weight, const float* bias,
                      int64_t k, float* pooled, float* logits, afan_stream_t stream) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n <= 0 || c <= 0 || hw <= 0 || k <= 0 || k > MAXK) return AFAN_ESHAPE;
    if (!x || !weight || !pooled || !logits) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("head_fwd_kernel", (double)n * hw * c * (dtype == AFAN_F32 ? 4 : 2), st);
    if (dtype == AFAN_F32)
        head_fwd_kernel<float><<<(unsigned)n, BLOCK, 0, st>>>((const float*)x, weight, bias, pooled, logits, (int)c, (int)hw, (int)k);
    else if (c % 8 == 0 && c / 8 <= BLOCK && BLOCK % (c / 8) == 0 && aligned(x, 16))
        head_fwd_vec_kernel<<<(unsigned)n, BLOCK, 0, st>>>((const uint16_t*)x, weight, bias, pooled, logits, (int)c, (int)hw, (int)k);
    else
        head_fwd_kernel<uint16_t><<<(unsigned)n, BLOCK, 0, st>>>((const uint16_t*)x, weight, bias, pooled, logits, (int)c, (int)hw, (int)k);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_head_backward(const float* dlogits, const float* weight, const float* pooled, int64_t n, int64_t c, int64_t hw,
                       int64_t k, void* dx, int dx_dtype, float* dweight, float* dbias, int accumulate,
                       afan_stream_t stream) {
    if (dx && dx_dtype != AFAN_F32 && dx_dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n <= 0 || c <= 0 || hw <= 0 || k <= 0 || k > MAXK) return AFAN_ESHAPE;
    if (!dlogits || !weight || !pooled) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        AFAN_PROF("head_bwd_dx_kernel", (double)n * hw * c * (dx_dtype == AFAN_F32 ? 4 : 2), st);
        if (dx_dtype == AFAN_F32)
            head_bwd_dx_kernel<float><<<(unsigned)n, BLOCK, 0, st>>>(dlogits, weight, (float*)dx, (int)c, (int)hw, (int)k);
        else
            head_bwd_dx_kernel<uint16_t><<<(unsigned)n, BLOCK, 0, st>>>(dlogits, weight, (uint16_t*)dx, (int)c, (int)hw, (int)k);
        AFAN_LAUNCH_CHECK();
    }
    if (dweight) {
        AFAN_PROF("head_bwd_dw_kernel", 4.0 * n * (c + k), st);
        const unsigned grid = (unsigned)((k * c + (dbias ? k : 0) + 63) / 64);
        head_bwd_dw_kernel<<<grid, 64 * DW_SLICES, 0, st>>>(dlogits, pooled, dweight, dbias, (int)n, (int)c, (int)k, accumulate);
        AFAN_LAUNCH_CHECK();
    }
    return AFAN_OK;
}

int afan_cross_entropy(const float* logits, const int64_t* target, int64_t n, int64_t k, float* loss, float* dlogits,
                       afan_stream_t stream) {
    if (n <= 0 || k <= 0 || n * k > (1 << 16)) return AFAN_ESHAPE;        // one block; larger heads stay with the caller
    if (!logits || !target || !loss || !dlogits) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("ce_fwd_kernel", 8.0 * n * k, st);
    if (k >= 64) ce_fwd_wide_kernel<<<1, 64 * CEW_WAVES, 0, st>>>(logits, target, loss, dlogits, (int)n, (int)k);
    else ce_fwd_kernel<<<1, CE_BLOCK, 0, st>>>(logits, target, loss, dlogits, (int)n, (int)k);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
