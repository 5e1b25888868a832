// fp32-ACCURATE convolutions on gfx950: forward, input gradient, weight gradient on v_mfma_f32_32x32x2_f32
// (f32 operands, f32 accumulate: bit-for-bit a k-ordered fmaf chain — the arithmetic the reference's fp32 training
// runs in, Classification/main_perturb.py:173-201, attack_algo.py:50-52).  This is the library's GENERAL convolution:
//   * any channel counts (the 3-channel image stems included), k x k with any padding / stride 1 or 2 / dilation,
//     optional bias (Segmentation/network/_deeplab.py:45);
//   * channels-last (NHWC activations, KRSC weights — the product layout) or the reference's NCHW / KCRS, by element
//     strides: the kernels only see strides, the extern "C" entries take the library's layout codes;
//   * fp32 storage (parity mode: north_star's "within 1e-4 fp32" is held on THESE kernels) or bf16 storage with fp32
//     arithmetic (the shapes the tuned bf16 kernels of afan_conv.hip decline).
// It replaces what rounds 1-2 sent to the vendor library through aten: the package now contains no vendor convolution.
//
// GEMM views (same tap-class scheme as afan_conv.hip):
//   forward / dgrad   Y[m, co] = sum_{t, c} X[pixel(m) + (dh_t, dw_t), c] * W[co, tap_t, c]      K = T * C (flattened)
//     dgrad at stride 2 = four output-parity classes with their own tap lists (no products with inserted zeros);
//     the weight operand is addressed [row][tap][c] by three element strides, so the input gradient reads the
//     untransposed KRSC / KCRS weights directly (rows = Ci, reduction = Co).
//   wgrad             dW[co, (t, ci)] = sum_p dY[p, co] * X[pixel(p) + tap t, ci]                K = pixels, split in
//     slices whose fp32 partial tiles are summed in slice order (deterministic) into the gradient tensor.
// Tiles: 128 x {128, 64, 32} outputs per workgroup of 4 waves, BK = 16; operands go global -> registers -> LDS (double
// buffered, one barrier per K-step).  Staging per operand: 16-byte vectors along the contiguous reduction / channel
// index where the layout gives one (template flags), otherwise scalar gathers with the lanes laid along whichever index
// is contiguous in memory (run-time flag).  LDS rows are [row][16 + 4 pad] floats: ds_write_b128 / ds_read_b128 are
// conflict-free with the 80-byte row stride; a lane (i, h) fetches k = 4h..4h+3 and 8+4h..8+4h+3 of its row with two
// ds_read_b128 and uses component j for MFMA j — A and B use the same (permuted) k order, and a sum does not care.
// Peak of the instruction: 157 TFLOP/s (1/16 of the bf16 MFMA rate): this path exists for accuracy and generality.
#include "afan_common.h"

using namespace afan;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAXT = 49;         // taps of a 7x7
constexpr int BK = 16;
constexpr int LDA = BK + 4;      // floats per LDS row, forward / dgrad ([row][k])
constexpr int THREADS = 256;
constexpr int BM = 128;          // row tile (pixels) of forward / dgrad

struct GClass {
    int Hg, Wg;                  // grid of output positions of this class
    int out_h0, out_w0;          // output coordinate = g * out_s + out_0
    int T;                       // taps
    short dh[MAXT], dw[MAXT];    // input pixel = g * in_s + (dh, dw)
    short tap[MAXT];             // index of the tap inside the weight tensor (r * k + s)
    short pad_;
};

struct GP {
    const void* x;
    const void* w;
    void* y;
    const float* bias;           // optional [Co]
    int N, Hi, Wi, Ci;           // gathered tensor [N, Hi, Wi, Ci] (Ci = reduction channels)
    int Ho, Wo, Co;              // output tensor
    int64_t x_sN, x_sH, x_sW, x_sC;      // element strides
    int64_t y_sN, y_sH, y_sW, y_sC;
    int64_t w_sRow, w_sTap, w_sC;        // weight element (row, tap, c)
    int in_s, out_s, n_classes;
    int a_rows, b_rows;          // scalar staging: lanes along the rows (1) or along the reduction index (0)
    GClass cls[4];
};

template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<uint16_t>(const uint16_t* p, float (&v)[4]) {
    const u16x4 t = *reinterpret_cast<const u16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf2f(t[i]);
}

// ---------------------------------------------------------------------------------------------- forward / dgrad
// WGM waves along the rows (pixels), 4 / WGM along the output channels.
template <typename T, int BN, int WGM, bool AVEC, bool BVEC>
__global__ __launch_bounds__(THREADS) void conv_f32_kernel(const GP p) {
    constexpr int WGN = 4 / WGM;
    constexpr int TM = BM / WGM, TN = BN / WGN;
    constexpr int MI = TM / 32, NI = TN / 32;
    static_assert(MI >= 1 && NI >= 1, "wave tile must hold a 32x32 MFMA tile");
    constexpr int A_V = BM / 64;                  // vector mode: float4 pieces per thread (row = tid/4 + 64 i)
    constexpr int B_V = (BN + 63) / 64;
    constexpr int A_S = BM / 16;                  // scalar mode: elements per thread
    constexpr int B_S = BN / 16;

    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDA];
    __shared__ int64_t row_base[BM];
    __shared__ int64_t out_off[BM];
    __shared__ int row_h0[BM], row_w0[BM];
    __shared__ short s_dh[MAXT], s_dw[MAXT], s_tap[MAXT];

    const GClass& cc = p.cls[blockIdx.z];
    const uint32_t Wg = (uint32_t)cc.Wg, Hg = (uint32_t)cc.Hg;
    const uint32_t M = (uint32_t)p.N * Hg * Wg;
    const uint32_t m0 = blockIdx.x * BM;
    if (m0 >= M) return;
    const int n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lh = lane >> 5;
    const int Tn = cc.T, Ci = p.Ci, Hi = p.Hi, Wi = p.Wi;
    const int Ktot = Tn * Ci;
    const int nsteps = (Ktot + BK - 1) / BK;
    const T* __restrict__ X = static_cast<const T*>(p.x);
    const T* __restrict__ Wt = static_cast<const T*>(p.w);

    if (tid < BM) {
        const uint32_t m = m0 + tid;
        if (m < M) {
            const uint32_t t1 = m / Wg, wg = m - t1 * Wg;
            const uint32_t n = t1 / Hg, hg = t1 - n * Hg;
            const int h0 = (int)hg * p.in_s, w0 = (int)wg * p.in_s;
            row_h0[tid] = h0; row_w0[tid] = w0;
            row_base[tid] = (int64_t)n * p.x_sN + (int64_t)h0 * p.x_sH + (int64_t)w0 * p.x_sW;
            out_off[tid] = (int64_t)n * p.y_sN + (int64_t)((int)hg * p.out_s + cc.out_h0) * p.y_sH +
                           (int64_t)((int)wg * p.out_s + cc.out_w0) * p.y_sW;
        } else {
            row_h0[tid] = -(1 << 28); row_w0[tid] = -(1 << 28);     // every tap lands outside the image
            row_base[tid] = 0; out_off[tid] = -1;
        }
    }
    if (tid < Tn) { s_dh[tid] = cc.dh[tid]; s_dw[tid] = cc.dw[tid]; s_tap[tid] = cc.tap[tid]; }
    __syncthreads();

    // one element (or 4 consecutive reduction channels) of the gathered operand: row r of the tile, reduction index k
    auto a_addr = [&](int r, int k, bool& ok) -> int64_t {
        int t = (int)((uint32_t)k / (uint32_t)Ci);
        const int c = k - t * Ci;
        t = t < Tn ? t : 0;                          // (k >= Ktot: masked below)
        const int dh = s_dh[t], dw = s_dw[t];
        const int hi = row_h0[r] + dh, wi = row_w0[r] + dw;
        ok = (k < Ktot) && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
        return row_base[r] + (int64_t)dh * p.x_sH + (int64_t)dw * p.x_sW + (int64_t)c * p.x_sC;
    };
    auto b_addr = [&](int r, int k, bool& ok) -> int64_t {
        int t = (int)((uint32_t)k / (uint32_t)Ci);
        const int c = k - t * Ci;
        t = t < Tn ? t : 0;
        ok = (k < Ktot) && (n0 + r) < p.Co;
        return (int64_t)(n0 + r) * p.w_sRow + (int64_t)s_tap[t] * p.w_sTap + (int64_t)c * p.w_sC;
    };

    float ra[AVEC ? A_V * 4 : A_S];
    float rb[BVEC ? B_V * 4 : B_S];

    auto fetch = [&](int s) {
        const int k0 = s * BK;
        if constexpr (AVEC) {
#pragma unroll
            for (int i = 0; i < A_V; ++i) {
                const int r = (tid >> 2) + 64 * i, k = k0 + (tid & 3) * 4;
                bool ok;
                const int64_t a = a_addr(r, k, ok);
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (ok) ld4<T>(X + a, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[i * 4 + j] = v[j];
            }
        } else {
#pragma unroll
            for (int e = 0; e < A_S; ++e) {
                const int r = p.a_rows ? (tid % BM) : (tid >> 4) + 16 * e;
                const int kk = p.a_rows ? (tid / BM) + (THREADS / BM) * e : (tid & 15);
                bool ok;
                const int64_t a = a_addr(r, k0 + kk, ok);
                ra[e] = ok ? Elt<T>::ld(X + a) : 0.f;
            }
        }
        if constexpr (BVEC) {
#pragma unroll
            for (int i = 0; i < B_V; ++i) {
                const int r = (tid >> 2) + 64 * i, k = k0 + (tid & 3) * 4;
                bool ok;
                const int64_t a = b_addr(r, k, ok);
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (ok && r < BN) ld4<T>(Wt + a, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) rb[i * 4 + j] = v[j];
            }
        } else {
#pragma unroll
            for (int e = 0; e < B_S; ++e) {
                const int r = p.b_rows ? (tid % BN) : (tid >> 4) + 16 * e;
                const int kk = p.b_rows ? (tid / BN) + (THREADS / BN) * e : (tid & 15);
                bool ok;
                const int64_t a = b_addr(r, k0 + kk, ok);
                rb[e] = ok ? Elt<T>::ld(Wt + a) : 0.f;
            }
        }
    };
    auto stash = [&](int buf) {
        float* As = lds[buf];
        float* Bs = lds[buf] + BM * LDA;
        if constexpr (AVEC) {
#pragma unroll
            for (int i = 0; i < A_V; ++i) {
                const int r = (tid >> 2) + 64 * i;
                *reinterpret_cast<f32x4*>(As + r * LDA + (tid & 3) * 4) = f32x4{ra[i * 4], ra[i * 4 + 1], ra[i * 4 + 2], ra[i * 4 + 3]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < A_S; ++e) {
                const int r = p.a_rows ? (tid % BM) : (tid >> 4) + 16 * e;
                const int kk = p.a_rows ? (tid / BM) + (THREADS / BM) * e : (tid & 15);
                As[r * LDA + kk] = ra[e];
            }
        }
        if constexpr (BVEC) {
#pragma unroll
            for (int i = 0; i < B_V; ++i) {
                const int r = (tid >> 2) + 64 * i;
                if (r < BN)
                    *reinterpret_cast<f32x4*>(Bs + r * LDA + (tid & 3) * 4) = f32x4{rb[i * 4], rb[i * 4 + 1], rb[i * 4 + 2], rb[i * 4 + 3]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < B_S; ++e) {
                const int r = p.b_rows ? (tid % BN) : (tid >> 4) + 16 * e;
                const int kk = p.b_rows ? (tid / BN) + (THREADS / BN) * e : (tid & 15);
                Bs[r * LDA + kk] = rb[e];
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nsteps > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const bool more = s + 1 < nsteps;
        if (more) fetch(s + 1);
        const float* As = lds[s & 1] + (wm * TM + li) * LDA + lh * 4;
        const float* Bs = lds[s & 1] + BM * LDA + (wn * TN + li) * LDA + lh * 4;
        f32x4 fa[MI][2], fb[NI][2];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) fa[i][q] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDA + q * 8);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) fb[j][q] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDA + q * 8);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q][c], fb[j][q][c], acc[i][j], 0, 0, 0);
        if (more) stash((s + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane holds column li of each 32x32 tile, rows (e & 3) + 8 (e >> 2) + 4 lh
    T* __restrict__ Y = static_cast<T*>(p.y);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * TN + j * 32 + li;
        if (col >= p.Co) continue;
        const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int64_t o = out_off[r];
                if (o >= 0) Elt<T>::st(Y + o + (int64_t)col * p.y_sC, acc[i][j][e] + b);
            }
    }
}

// ------------------------------------------------------------------------------------------------------- wgrad
struct WP {
    const void* x;
    const void* dy;
    float* slab;                 // [slices][Co][T * Ci]
    int N, Hi, Wi, Ci, Ho, Wo, Co;
    int k, stride, pad, dil;
    int64_t x_sN, x_sH, x_sW, x_sC;
    int64_t d_sN, d_sH, d_sW, d_sC;
    int P;                       // pixels N * Ho * Wo (the reduction)
    int steps_per_slice;         // K-steps of BK pixels per slice
    int a_k, b_k;                // scalar staging: lanes along the pixel index (1: NCHW) or along the channels (0)
};

// WM x WN: BMw x BNw tile of (Co) x (T * Ci); LDS operand images are [k][rows + 32 pad] (pixel-major, like memory in NHWC)
template <typename T, int BMw, int BNw, int WGM, bool AVEC, bool BVEC>
__global__ __launch_bounds__(THREADS) void wgrad_f32_kernel(const WP p) {
    constexpr int WGN = 4 / WGM;
    constexpr int TM = BMw / WGM, TN = BNw / WGN;
    constexpr int MI = TM / 32, NI = TN / 32;
    static_assert(MI >= 1 && NI >= 1, "wave tile must hold a 32x32 MFMA tile");
    constexpr int LA = BMw + 32, LB = BNw + 32;
    constexpr int A_PER = BMw / 4, A_KPT = THREADS / A_PER, A_V = (BK + A_KPT - 1) / A_KPT;   // vector mode: float4 (m4, k)
    constexpr int B_PER = BNw / 4, B_KPT = THREADS / B_PER, B_V = (BK + B_KPT - 1) / B_KPT;
    constexpr int A_S = BMw / 16, B_S = BNw / 16;

    __shared__ __attribute__((aligned(16))) float lds[2][BK * (LA + LB)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * BMw, n0 = blockIdx.x * BNw;
    const int Ci = p.Ci, Ncols = p.k * p.k * Ci;
    const int step0 = blockIdx.z * p.steps_per_slice;
    int nsteps = (p.P + BK - 1) / BK - step0;
    if (nsteps > p.steps_per_slice) nsteps = p.steps_per_slice;
    const T* __restrict__ X = static_cast<const T*>(p.x);
    const T* __restrict__ D = static_cast<const T*>(p.dy);
    const uint32_t Wo = (uint32_t)p.Wo, Ho = (uint32_t)p.Ho;

    // pixel q of the reduction -> dy offset and the input window's origin
    auto pix = [&](int q, int64_t& doff, int64_t& xoff, int& h0, int& w0) -> bool {
        if (q >= p.P) return false;
        const uint32_t t1 = (uint32_t)q / Wo, wo = (uint32_t)q - t1 * Wo;
        const uint32_t n = t1 / Ho, ho = t1 - n * Ho;
        doff = (int64_t)n * p.d_sN + (int64_t)ho * p.d_sH + (int64_t)wo * p.d_sW;
        h0 = (int)ho * p.stride - p.pad; w0 = (int)wo * p.stride - p.pad;
        xoff = (int64_t)n * p.x_sN;
        return true;
    };
    // column c of the tile -> (tap, ci)
    auto b_elem = [&](int col, int64_t xoff, int h0, int w0, bool okp, bool& ok) -> int64_t {
        const int n = n0 + col;
        const int t = (int)((uint32_t)n / (uint32_t)Ci), ci = n - t * Ci;
        const int r = t / p.k, s = t - r * p.k;
        const int hi = h0 + r * p.dil, wi = w0 + s * p.dil;
        ok = okp && n < Ncols && hi >= 0 && hi < p.Hi && wi >= 0 && wi < p.Wi;
        return xoff + (int64_t)hi * p.x_sH + (int64_t)wi * p.x_sW + (int64_t)ci * p.x_sC;
    };

    float ra[AVEC ? A_V * 4 : A_S];
    float rb[BVEC ? B_V * 4 : B_S];

    auto fetch = [&](int s) {
        const int q0 = (step0 + s) * BK;
        if constexpr (AVEC) {
#pragma unroll
            for (int i = 0; i < A_V; ++i) {
                const int kk = tid / A_PER + A_KPT * i, m = (tid % A_PER) * 4;
                int64_t doff = 0, xoff; int h0, w0;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (kk < BK && pix(q0 + kk, doff, xoff, h0, w0) && m0 + m < p.Co) ld4<T>(D + doff + (m0 + m), v);
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[i * 4 + j] = v[j];
            }
        } else {
#pragma unroll
            for (int e = 0; e < A_S; ++e) {
                const int kk = p.a_k ? (tid & 15) : tid / BMw + (THREADS / BMw) * e;
                const int m = p.a_k ? (tid >> 4) + 16 * e : tid % BMw;
                int64_t doff = 0, xoff; int h0, w0;
                const bool ok = pix(q0 + kk, doff, xoff, h0, w0) && m0 + m < p.Co;
                ra[e] = ok ? Elt<T>::ld(D + doff + (int64_t)(m0 + m) * p.d_sC) : 0.f;
            }
        }
        if constexpr (BVEC) {
#pragma unroll
            for (int i = 0; i < B_V; ++i) {
                const int kk = tid / B_PER + B_KPT * i, c = (tid % B_PER) * 4;
                int64_t doff, xoff = 0; int h0 = 0, w0 = 0;
                const bool okp = kk < BK && pix(q0 + kk, doff, xoff, h0, w0);
                bool ok;
                const int64_t a = b_elem(c, xoff, h0, w0, okp, ok);
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (ok) ld4<T>(X + a, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) rb[i * 4 + j] = v[j];
            }
        } else {
#pragma unroll
            for (int e = 0; e < B_S; ++e) {
                const int kk = p.b_k ? (tid & 15) : tid / BNw + (THREADS / BNw) * e;
                const int c = p.b_k ? (tid >> 4) + 16 * e : tid % BNw;
                int64_t doff, xoff = 0; int h0 = 0, w0 = 0;
                const bool okp = pix(q0 + kk, doff, xoff, h0, w0);
                bool ok;
                const int64_t a = b_elem(c, xoff, h0, w0, okp, ok);
                rb[e] = ok ? Elt<T>::ld(X + a) : 0.f;
            }
        }
    };
    auto stash = [&](int buf) {
        float* As = lds[buf];
        float* Bs = lds[buf] + BK * LA;
        if constexpr (AVEC) {
#pragma unroll
            for (int i = 0; i < A_V; ++i) {
                const int kk = tid / A_PER + A_KPT * i, m = (tid % A_PER) * 4;
                if (kk < BK) *reinterpret_cast<f32x4*>(As + kk * LA + m) = f32x4{ra[i * 4], ra[i * 4 + 1], ra[i * 4 + 2], ra[i * 4 + 3]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < A_S; ++e) {
                const int kk = p.a_k ? (tid & 15) : tid / BMw + (THREADS / BMw) * e;
                const int m = p.a_k ? (tid >> 4) + 16 * e : tid % BMw;
                As[kk * LA + m] = ra[e];
            }
        }
        if constexpr (BVEC) {
#pragma unroll
            for (int i = 0; i < B_V; ++i) {
                const int kk = tid / B_PER + B_KPT * i, c = (tid % B_PER) * 4;
                if (kk < BK) *reinterpret_cast<f32x4*>(Bs + kk * LB + c) = f32x4{rb[i * 4], rb[i * 4 + 1], rb[i * 4 + 2], rb[i * 4 + 3]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < B_S; ++e) {
                const int kk = p.b_k ? (tid & 15) : tid / BNw + (THREADS / BNw) * e;
                const int c = p.b_k ? (tid >> 4) + 16 * e : tid % BNw;
                Bs[kk * LB + c] = rb[e];
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nsteps > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const bool more = s + 1 < nsteps;
        if (more) fetch(s + 1);
        const float* As = lds[s & 1] + lh * LA + wm * TM + li;
        const float* Bs = lds[s & 1] + BK * LA + lh * LB + wn * TN + li;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = As[kk * LA + i * 32];
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = Bs[kk * LB + j * 32];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash((s + 1) & 1);
        __syncthreads();
    }

    float* __restrict__ S = p.slab + (int64_t)blockIdx.z * p.Co * Ncols;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * TN + j * 32 + li;
        if (col >= Ncols) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = m0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (r < p.Co) S[(int64_t)r * Ncols + col] = acc[i][j][e];
            }
    }
}

// grad[co][tap][ci] (element strides g_sRow / g_sTap / g_sC) (+)= sum over slices, in slice order
__global__ __launch_bounds__(256) void wgrad_f32_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad,
                                                               int64_t co, int64_t ncols, int64_t ci, int slices,
                                                               int64_t g_sRow, int64_t g_sTap, int64_t g_sC, int accumulate) {
    const int64_t total = co * ncols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        int z = 0;
        for (; z + 8 <= slices; z += 8) {                 // eight requests in flight, added in slice order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(z + u) * total + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < slices; ++z) s += slab[(int64_t)z * total + i];
        const int64_t r = i / ncols, n = i - r * ncols;
        const int64_t t = n / ci, c = n - t * ci;
        float* g = grad + r * g_sRow + t * g_sTap + c * g_sC;
        *g = accumulate ? *g + s : s;
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct Strides { int64_t sN, sH, sW, sC; };
Strides act_strides(int layout, int64_t c, int64_t h, int64_t w) {
    if (layout == AFAN_NHWC) return Strides{h * w * c, w * c, c, 1};
    return Strides{c * h * w, w, 1, h * w};
}

bool vec_ok(const void* base, int esize, int64_t c, const Strides& s) {
    const size_t al = esize * 4;       // 16 B (fp32) or 8 B (bf16) pieces of 4 channels
    return s.sC == 1 && c % 4 == 0 && aligned(base, al) && s.sN % 4 == 0 && s.sH % 4 == 0 && s.sW % 4 == 0;
}

template <typename T, bool AV, bool BV>
int launch_tile(const GP& p, int64_t M, hipStream_t st) {
    const int co = p.Co;
    if (co <= 32) {
        dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((co + 31) / 32), (unsigned)p.n_classes);
        conv_f32_kernel<T, 32, 4, AV, BV><<<grid, THREADS, 0, st>>>(p);
    } else if (co <= 64) {
        dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((co + 63) / 64), (unsigned)p.n_classes);
        conv_f32_kernel<T, 64, 2, AV, BV><<<grid, THREADS, 0, st>>>(p);
    } else {
        dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((co + 127) / 128), (unsigned)p.n_classes);
        conv_f32_kernel<T, 128, 2, AV, BV><<<grid, THREADS, 0, st>>>(p);
    }
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <typename T>
int launch_modes(const GP& p, int64_t M, bool av, bool bv, hipStream_t st) {
    if (av && bv) return launch_tile<T, true, true>(p, M, st);
    if (av) return launch_tile<T, true, false>(p, M, st);
    if (bv) return launch_tile<T, false, true>(p, M, st);
    return launch_tile<T, false, false>(p, M, st);
}

int launch_conv(const GP& p, int dtype, bool av, bool bv, hipStream_t st) {
    int64_t M = 0;
    for (int c = 0; c < p.n_classes; ++c) {
        const int64_t v = (int64_t)p.N * p.cls[c].Hg * p.cls[c].Wg;
        if (v > M) M = v;
    }
    if (M <= 0) return AFAN_OK;
    if (dtype == AFAN_F32) return launch_modes<float>(p, M, av, bv, st);
    return launch_modes<uint16_t>(p, M, av, bv, st);
}

int check_general(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride, int pad, int dilation,
                  int dtype, int layout, int w_layout) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if ((layout != AFAN_NCHW && layout != AFAN_NHWC) || (w_layout != AFAN_NCHW && w_layout != AFAN_NHWC)) return AFAN_ELAYOUT;
    if (n < 0 || hi <= 0 || wi <= 0 || ci <= 0 || co <= 0) return AFAN_ESHAPE;
    if (k < 1 || k > 7 || !(stride == 1 || stride == 2) || pad < 0 || pad > 127 || dilation < 1 || dilation > 64) return AFAN_ESHAPE;
    if ((hi + 2 * pad - dilation * (k - 1) - 1) < 0 || (wi + 2 * pad - dilation * (k - 1) - 1) < 0) return AFAN_ESHAPE;
    if (n * hi * wi > 0x7fffffffLL || (int64_t)k * k * (ci > co ? ci : co) > 0x7fffffffLL) return AFAN_ESHAPE;   // 32-bit row / reduction indices
    return AFAN_OK;
}

inline int out_dim(int64_t i, int k, int stride, int pad, int dil) { return (int)((i + 2 * pad - dil * (k - 1) - 1) / stride + 1); }

}  // namespace

extern "C" {

int afan_conv_fwd(const void* x, const void* w, const float* bias, void* y, int dtype, int layout, int w_layout, int64_t n,
                  int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride, int pad, int dilation, afan_stream_t stream) {
    int e = check_general(n, hi, wi, ci, co, k, stride, pad, dilation, dtype, layout, w_layout);
    if (e) return e;
    if (n == 0) return AFAN_OK;
    if (!x || !w || !y) return AFAN_ENULL;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, es) || !aligned(w, es) || !aligned(y, es) || (bias && !aligned(bias, 4))) return AFAN_EALIGN;
    GP p{};
    p.x = x; p.w = w; p.y = y; p.bias = bias;
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci;
    p.Ho = out_dim(hi, k, stride, pad, dilation); p.Wo = out_dim(wi, k, stride, pad, dilation); p.Co = (int)co;
    const Strides xs = act_strides(layout, ci, hi, wi), ys = act_strides(layout, co, p.Ho, p.Wo);
    p.x_sN = xs.sN; p.x_sH = xs.sH; p.x_sW = xs.sW; p.x_sC = xs.sC;
    p.y_sN = ys.sN; p.y_sH = ys.sH; p.y_sW = ys.sW; p.y_sC = ys.sC;
    const int64_t kk = (int64_t)k * k;
    if (w_layout == AFAN_NHWC) { p.w_sRow = kk * ci; p.w_sTap = ci; p.w_sC = 1; }      // KRSC
    else { p.w_sRow = ci * kk; p.w_sTap = 1; p.w_sC = kk; }                              // KCRS
    p.in_s = stride; p.out_s = 1; p.n_classes = 1;
    p.a_rows = layout == AFAN_NCHW ? 1 : 0;       // NCHW: pixels along w are contiguous
    p.b_rows = 0;
    GClass& c0 = p.cls[0];
    c0.Hg = p.Ho; c0.Wg = p.Wo; c0.out_h0 = 0; c0.out_w0 = 0; c0.T = k * k;
    for (int r = 0; r < k; ++r)
        for (int s = 0; s < k; ++s) {
            const int t = r * k + s;
            c0.dh[t] = (short)(r * dilation - pad); c0.dw[t] = (short)(s * dilation - pad); c0.tap[t] = (short)t;
        }
    const bool av = vec_ok(x, es, ci, xs);
    const bool bv = p.w_sC == 1 && ci % 4 == 0 && aligned(w, es * 4);
    hipStream_t st = (hipStream_t)stream;
    const double M = (double)n * p.Ho * p.Wo;
    AFAN_PROF_FLOPS("conv_f32_fwd_kernel", (double)es * (M * co + (double)n * hi * wi * ci + (double)co * kk * ci),
                    2.0 * M * co * kk * ci, st);
    return launch_conv(p, dtype, av, bv, st);
}

/* dx[N,Hi,Wi,Ci] for y = conv(x, w): reads the UNTRANSPOSED weight w[Co,Ci,k,k] (KRSC or KCRS memory). */
int afan_conv_dgrad(const void* dy, const void* w, void* dx, int dtype, int layout, int w_layout, int64_t n, int64_t hi,
                    int64_t wi, int64_t ci, int64_t co, int k, int stride, int pad, int dilation, afan_stream_t stream) {
    int e = check_general(n, hi, wi, ci, co, k, stride, pad, dilation, dtype, layout, w_layout);
    if (e) return e;
    if (n == 0) return AFAN_OK;
    if (!dy || !w || !dx) return AFAN_ENULL;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(dy, es) || !aligned(w, es) || !aligned(dx, es)) return AFAN_EALIGN;
    const int ho = out_dim(hi, k, stride, pad, dilation), wo = out_dim(wi, k, stride, pad, dilation);
    GP p{};
    p.x = dy; p.w = w; p.y = dx;
    p.N = (int)n; p.Hi = ho; p.Wi = wo; p.Ci = (int)co;         // gathered tensor = dy, reduction over co
    p.Ho = (int)hi; p.Wo = (int)wi; p.Co = (int)ci;             // output = dx, rows of the weight operand = ci
    const Strides xs = act_strides(layout, co, ho, wo), ys = act_strides(layout, ci, hi, wi);
    p.x_sN = xs.sN; p.x_sH = xs.sH; p.x_sW = xs.sW; p.x_sC = xs.sC;
    p.y_sN = ys.sN; p.y_sH = ys.sH; p.y_sW = ys.sW; p.y_sC = ys.sC;
    const int64_t kk = (int64_t)k * k;
    if (w_layout == AFAN_NHWC) { p.w_sRow = 1; p.w_sTap = ci; p.w_sC = kk * ci; }       // element (co, t, ci) of KRSC
    else { p.w_sRow = kk; p.w_sTap = 1; p.w_sC = ci * kk; }                               // ... of KCRS
    p.in_s = 1;
    p.a_rows = layout == AFAN_NCHW ? 1 : 0;
    p.b_rows = 1;                                  // KRSC: consecutive ci are consecutive addresses
    if (stride == 1) {
        // dx[h, w] = sum_{r, s} dy[h + pad - r d, w + pad - s d] * w[., r, s, .]
        p.out_s = 1; p.n_classes = 1;
        GClass& c0 = p.cls[0];
        c0.Hg = (int)hi; c0.Wg = (int)wi; c0.out_h0 = 0; c0.out_w0 = 0; c0.T = k * k;
        for (int r = 0; r < k; ++r)
            for (int s = 0; s < k; ++s) {
                const int t = r * k + s;
                c0.dh[t] = (short)(pad - r * dilation); c0.dw[t] = (short)(pad - s * dilation); c0.tap[t] = (short)t;
            }
    } else {
        // output pixel (2h'+ph, 2w'+pw) receives tap (r, s) iff (ph + pad - r d) and (pw + pad - s d) are even
        p.out_s = 2;
        int nc = 0;
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                GClass& c = p.cls[nc];
                c.Hg = (int)((hi - ph + 1) / 2); c.Wg = (int)((wi - pw + 1) / 2);
                if (c.Hg <= 0 || c.Wg <= 0) continue;
                c.out_h0 = ph; c.out_w0 = pw;
                int T = 0;
                for (int r = 0; r < k; ++r)
                    for (int s = 0; s < k; ++s) {
                        const int a = ph + pad - r * dilation, b = pw + pad - s * dilation;
                        if ((a & 1) || (b & 1)) continue;
                        c.dh[T] = (short)(a >> 1); c.dw[T] = (short)(b >> 1); c.tap[T] = (short)(r * k + s);   // (>> 1: exact, a is even)
                        ++T;
                    }
                c.T = T;               // 0 taps (1x1 at stride 2): the class's gradient is zero, written as such
                ++nc;
            }
        p.n_classes = nc;
    }
    const bool av = vec_ok(dy, es, co, xs);
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF_FLOPS("conv_f32_dgrad_kernel", (double)es * ((double)n * ho * wo * co + (double)n * hi * wi * ci + (double)co * kk * ci),
                    2.0 * (double)n * ho * wo * co * kk * ci, st);
    return launch_conv(p, dtype, av, false, st);
}

static int wgrad_plan(int64_t P, int64_t co, int64_t ncols, int* bm, int* slices, int* steps_per_slice) {
    *bm = co <= 32 ? 32 : (co <= 64 ? 64 : 128);
    const int64_t tiles = ((co + *bm - 1) / *bm) * ((ncols + 127) / 128);
    const int64_t steps = (P + BK - 1) / BK;
    int64_t s = (1024 + tiles - 1) / tiles;          // ~4 workgroups per CU over the launch
    if (s > steps / 8) s = steps / 8;                // at least 8 K-steps (128 pixels) per slice
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    const int64_t sps = (steps + s - 1) / s;
    *steps_per_slice = (int)sps;
    *slices = (int)((steps + sps - 1) / sps);
    return 0;
}

int64_t afan_conv_wgrad_f32_workspace_floats(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride,
                                             int pad, int dilation) {
    if (check_general(n, hi, wi, ci, co, k, stride, pad, dilation, AFAN_F32, AFAN_NHWC, AFAN_NHWC) || n == 0) return 0;
    const int64_t P = n * out_dim(hi, k, stride, pad, dilation) * out_dim(wi, k, stride, pad, dilation);
    int bm, slices, sps;
    wgrad_plan(P, co, (int64_t)k * k * ci, &bm, &slices, &sps);
    return (int64_t)slices * co * k * k * ci;
}

/* grad[Co,Ci,k,k] fp32 (KRSC or KCRS memory by w_layout) (+)= d(loss)/dw from x and dy (`dtype` storage, `layout`). */
int afan_conv_wgrad(const void* x, const void* dy, float* grad, int dtype, int layout, int w_layout, int64_t n, int64_t hi,
                    int64_t wi, int64_t ci, int64_t co, int k, int stride, int pad, int dilation, float* workspace,
                    int accumulate, afan_stream_t stream) {
    int e = check_general(n, hi, wi, ci, co, k, stride, pad, dilation, dtype, layout, w_layout);
    if (e) return e;
    if (!grad) return AFAN_ENULL;
    if (!aligned(grad, 4)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int64_t kk = (int64_t)k * k, ncols = kk * ci;
    int64_t g_sRow, g_sTap, g_sC;
    if (w_layout == AFAN_NHWC) { g_sRow = ncols; g_sTap = ci; g_sC = 1; }
    else { g_sRow = ncols; g_sTap = 1; g_sC = kk; }
    if (n == 0) {
        if (!accumulate) {
            hipError_t he = hipMemsetAsync(grad, 0, (size_t)co * ncols * 4, st);
            if (he != hipSuccess) return (int)he;
        }
        return AFAN_OK;
    }
    if (!x || !dy || !workspace) return AFAN_ENULL;
    const int es = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, es) || !aligned(dy, es) || !aligned(workspace, 16)) return AFAN_EALIGN;
    WP p{};
    p.x = x; p.dy = dy; p.slab = workspace;
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci; p.Co = (int)co;
    p.Ho = out_dim(hi, k, stride, pad, dilation); p.Wo = out_dim(wi, k, stride, pad, dilation);
    p.k = k; p.stride = stride; p.pad = pad; p.dil = dilation;
    const Strides xs = act_strides(layout, ci, hi, wi), ds = act_strides(layout, co, p.Ho, p.Wo);
    p.x_sN = xs.sN; p.x_sH = xs.sH; p.x_sW = xs.sW; p.x_sC = xs.sC;
    p.d_sN = ds.sN; p.d_sH = ds.sH; p.d_sW = ds.sW; p.d_sC = ds.sC;
    p.P = (int)(n * p.Ho * p.Wo);
    int bm, slices;
    wgrad_plan(p.P, co, ncols, &bm, &slices, &p.steps_per_slice);
    p.a_k = layout == AFAN_NCHW ? 1 : 0;
    p.b_k = layout == AFAN_NCHW ? 1 : 0;
    const bool av = vec_ok(dy, es, co, ds);
    const bool bv = vec_ok(x, es, ci, xs);
    const double Pd = (double)p.P;
    AFAN_PROF_FLOPS("wgrad_f32_kernel", (double)es * (Pd * co + (double)n * hi * wi * ci) + 4.0 * (slices + 1) * co * ncols,
                    2.0 * Pd * co * ncols, st);
    dim3 grid((unsigned)((ncols + 127) / 128), (unsigned)((co + bm - 1) / bm), (unsigned)slices);
#define AFAN_WG_GO(TT, BMV, WGMV)                                                                     \
    do {                                                                                              \
        if (av && bv) wgrad_f32_kernel<TT, BMV, 128, WGMV, true, true><<<grid, THREADS, 0, st>>>(p);        \
        else if (av) wgrad_f32_kernel<TT, BMV, 128, WGMV, true, false><<<grid, THREADS, 0, st>>>(p);        \
        else if (bv) wgrad_f32_kernel<TT, BMV, 128, WGMV, false, true><<<grid, THREADS, 0, st>>>(p);        \
        else wgrad_f32_kernel<TT, BMV, 128, WGMV, false, false><<<grid, THREADS, 0, st>>>(p);               \
    } while (0)
    if (dtype == AFAN_F32) {
        if (bm == 32) AFAN_WG_GO(float, 32, 1);
        else if (bm == 64) AFAN_WG_GO(float, 64, 2);
        else AFAN_WG_GO(float, 128, 2);
    } else {
        if (bm == 32) AFAN_WG_GO(uint16_t, 32, 1);
        else if (bm == 64) AFAN_WG_GO(uint16_t, 64, 2);
        else AFAN_WG_GO(uint16_t, 128, 2);
    }
#undef AFAN_WG_GO
    AFAN_LAUNCH_CHECK();
    const int64_t total = co * ncols;
    wgrad_f32_reduce_kernel<<<grid_for(total, 256), 256, 0, st>>>(workspace, grad, co, ncols, ci, slices, g_sRow, g_sTap, g_sC,
                                                                 accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
