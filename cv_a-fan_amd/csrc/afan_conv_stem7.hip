// ImageNet-style image stem: 7x7 / stride 2 / pad 3 convolution, 3 -> 64 channels, bf16 channels-last — the first layer
// of the DeepLabv3+ / ResNet-50/101 backbones (Segmentation/network/backbone/resnet.py:143-144 `conv1`) and of the
// build-defined ImageNet-shape ResNet-50.  Forward and weight gradient; no input gradient (images carry none on the
// A-FAN feature path).  ~0.6 % of the 513x513 step's FLOPs and K = 147 with 6-byte pixels (nothing wider than 2 bytes is
// aligned when the image width is odd), so this is a plain fp32 FMA kernel out of LDS, not an MFMA one: a workgroup
// stages the 7 input rows of a 64-pixel output segment once (133 pixels each) and all 147 x 64 weights, then every lane
// owns one output channel (weights: conflict-free LDS reads; pixels: broadcasts).
#include "afan_common.h"

using namespace afan;

namespace {

constexpr int BLOCK = 256;
constexpr int CO = 64, KK = 7, K = KK * KK * 3;   // 147
constexpr int SEG = 64;                           // output pixels per tile (one row segment)
constexpr int PATCH = (2 * SEG + 5) * 3;          // 399 staged elements per input row
constexpr int ROWP = 400;
constexpr int PPT = SEG / 4;                      // forward: pixels per thread (4 pixel groups x 64 channels)
constexpr int KPT = (K + 3) / 4;                  // wgrad: reduction columns per thread (4 groups): 37

__device__ __forceinline__ void stage_patch(const uint16_t* __restrict__ x, float* xl, int n, int oy, int ox0, int Hi, int Wi) {
    const int ix0 = 2 * ox0 - 3;
    for (int i = threadIdx.x; i < KK * PATCH; i += BLOCK) {
        const int r = i / PATCH, e = i - r * PATCH;
        const int iy = 2 * oy - 3 + r, col = ix0 + e / 3;
        float v = 0.f;
        if (iy >= 0 && iy < Hi && col >= 0 && col < Wi) v = bf2f(x[(((int64_t)n * Hi + iy) * Wi + ix0) * 3 + e]);
        xl[r * ROWP + e] = v;
    }
}

__global__ __launch_bounds__(BLOCK) void stem7_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                          uint16_t* __restrict__ y, int Hi, int Wi, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;               // [K][CO]
    float* xl = lds + K * CO;      // [7][ROWP]
    const int n = blockIdx.z, oy = blockIdx.y, ox0 = blockIdx.x * SEG;
    for (int i = threadIdx.x; i < K * CO; i += BLOCK) {
        const int ch = i / K, k = i - ch * K;
        wl[k * CO + ch] = bf2f(w[i]);
    }
    stage_patch(x, xl, n, oy, ox0, Hi, Wi);
    __syncthreads();
    const int ch = threadIdx.x & 63, pg = threadIdx.x >> 6;
    float acc[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) acc[p] = 0.f;
    for (int r = 0; r < KK; ++r) {
#pragma unroll 3
        for (int j = 0; j < 21; ++j) {
            const float wv = wl[(r * 21 + j) * CO + ch];
            const float* xp = xl + r * ROWP + pg * PPT * 6 + j;
#pragma unroll
            for (int p = 0; p < PPT; ++p) acc[p] = fmaf(wv, xp[p * 6], acc[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int ox = ox0 + pg * PPT + p;
        if (ox < Wo) y[(((int64_t)n * Ho + oy) * Wo + ox) * CO + ch] = f2bf(acc[p]);
    }
}

// persistent workgroups; slab[blk][k][ch]
__global__ __launch_bounds__(BLOCK) void stem7_wgrad_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                            float* __restrict__ slab, int N, int Hi, int Wi, int Ho, int Wo,
                                                            int segs, int64_t tiles) {
    __shared__ float dyl[SEG * CO];
    __shared__ float xl[KK * ROWP];
    const int ch = threadIdx.x & 63, kg = threadIdx.x >> 6;
    float acc[KPT];
    int koff[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        acc[j] = 0.f;
        const int k = kg * KPT + j;
        koff[j] = k < K ? (k / 21) * ROWP + (k % 21) : -1;
    }
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int seg = (int)(t % segs);
        const int64_t t2 = t / segs;
        const int oy = (int)(t2 % Ho), n = (int)(t2 / Ho);
        const int ox0 = seg * SEG;
        __syncthreads();
        for (int i = threadIdx.x; i < SEG * CO; i += BLOCK) {
            const int p = i >> 6, c = i & 63;
            dyl[i] = (ox0 + p < Wo) ? bf2f(dy[(((int64_t)n * Ho + oy) * Wo + ox0 + p) * CO + c]) : 0.f;
        }
        stage_patch(x, xl, n, oy, ox0, Hi, Wi);
        __syncthreads();
        for (int p = 0; p < SEG; ++p) {
            const float d = dyl[p * CO + ch];
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if (koff[j] >= 0) acc[j] = fmaf(d, xl[koff[j] + p * 6], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int k = kg * KPT + j;
        if (k < K) slab[((int64_t)blockIdx.x * K + k) * CO + ch] = acc[j];
    }
}

__global__ __launch_bounds__(BLOCK) void stem7_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad,
                                                                   int G, int accumulate) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;      // index into [k][ch]
    if (i >= K * CO) return;
    float s = 0.f;
    int g = 0;
    for (; g + 8 <= G; g += 8) {                          // eight requests in flight, added in slice order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(g + u) * K * CO + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; g < G; ++g) s += slab[(int64_t)g * K * CO + i];
    const int k = i >> 6, ch = i & 63;
    float* dst = grad + ch * K + k;                      // KRSC: [ch][r][s][c]
    *dst = accumulate ? *dst + s : s;
}

// ---- im2col of the stem: cols[m][k], m = (n, oy, ox), k = (r, s, c) in the weights' KRSC order, K = 147 padded to KP = 152 ----
// With it the stem's forward and weight gradient are plain 1x1 problems for the MFMA kernels (reduction 152 channels,
// ragged last chunk masked): 0.04 GB written once per pass and read twice against ~0.7 ms of LDS-bound FMAs before.
constexpr int KP = 152;
__global__ __launch_bounds__(BLOCK) void stem7_im2col_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ cols,
                                                             int Hi, int Wi, int Ho, int Wo) {
    __shared__ uint16_t xl[KK * ROWP];
    const int n = blockIdx.z, oy = blockIdx.y, ox0 = blockIdx.x * SEG;
    const int ix0 = 2 * ox0 - 3;
    for (int i = threadIdx.x; i < KK * PATCH; i += BLOCK) {
        const int r = i / PATCH, e = i - r * PATCH;
        const int iy = 2 * oy - 3 + r, col = ix0 + e / 3;
        uint16_t v = 0;
        if (iy >= 0 && iy < Hi && col >= 0 && col < Wi) v = x[(((int64_t)n * Hi + iy) * Wi + ix0) * 3 + e];
        xl[r * ROWP + e] = v;
    }
    __syncthreads();
    constexpr int PIECES = KP / 8;     // 19 16-byte pieces per output row
    for (int i = threadIdx.x; i < SEG * PIECES; i += BLOCK) {
        const int px = i / PIECES, pc = i - px * PIECES;
        if (ox0 + px >= Wo) continue;
        u16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = pc * 8 + e;
            const int r = k / 21, j = k - r * 21;
            v[e] = k < K ? xl[r * ROWP + px * 6 + j] : (uint16_t)0;
        }
        *reinterpret_cast<u16x8*>(cols + ((((int64_t)n * Ho + oy) * Wo + ox0 + px) * KP + pc * 8)) = v;
    }
}

// The adjoint of the im2col: dx[n, h, w, c] = sum over the taps (r, s) whose window covers input pixel (h, w) — (h + 3 - r) and
// (w + 3 - s) even, output pixel ((h + 3 - r) / 2, (w + 3 - s) / 2) inside the map — of dcols[n, ho, wo, (r*7 + s)*3 + c].
// With dcols = dy x W (a 1x1 input-gradient problem on the MFMA kernel, 64 -> 152 columns) this is the stem's input gradient:
// the image-level perturbation of Detection (train_aug_sat_muti_advt.py:82-95) needs it in five of an iteration's eight
// forwards, and the general kernel spends 527 us on a 3-channel output (3 of 32 MFMA columns used).  One thread per input pixel,
// 12-16 taps x 3 channels; fp32 sums, taps in increasing (r, s).
__global__ __launch_bounds__(BLOCK) void stem7_col2im_kernel(const uint16_t* __restrict__ dcols, uint16_t* __restrict__ dx,
                                                             int Hi, int Wi, int Ho, int Wo, int64_t pixels) {
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= pixels) return;
    const int w = (int)(p % Wi), h = (int)((p / Wi) % Hi);
    const int64_t n = p / Wi / Hi;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int r = (h + 3) & 1; r < KK; r += 2) {            // (h + 3 - r) even
        const int ho = (h + 3 - r) >> 1;
        if (ho < 0 || ho >= Ho) continue;
        for (int s = (w + 3) & 1; s < KK; s += 2) {
            const int wo = (w + 3 - s) >> 1;
            if (wo < 0 || wo >= Wo) continue;
            const uint16_t* src = dcols + (((int64_t)n * Ho + ho) * Wo + wo) * KP + (r * KK + s) * 3;
            a0 += bf2f(src[0]);
            a1 += bf2f(src[1]);
            a2 += bf2f(src[2]);
        }
    }
    uint16_t* dst = dx + p * 3;
    dst[0] = f2bf(a0);
    dst[1] = f2bf(a1);
    dst[2] = f2bf(a2);
}

int wgrad_blocks(int64_t tiles) { return (int)(tiles < 512 ? tiles : 512); }

}  // namespace

extern "C" {

int afan_conv_stem7_supported(int64_t ci, int64_t co, int k, int stride) { return (ci == 3 && co == CO && k == KK && stride == 2) ? 1 : 0; }

// y[N,Ho,Wo,64] = conv7x7/2 pad 3 (x[N,Hi,Wi,3], w[64,7,7,3]); all bf16 channels-last.  Ho = (Hi - 1) / 2 + 1.
int afan_conv_stem7_fwd_nhwc_bf16(const void* x, const void* w, void* y, int64_t n, int64_t hi, int64_t wi, afan_stream_t stream) {
    if (n <= 0 || hi <= 0 || wi <= 0 || n > 65535) return AFAN_ESHAPE;
    if (!x || !w || !y) return AFAN_ENULL;
    if (!aligned(x, 2) || !aligned(w, 2) || !aligned(y, 2)) return AFAN_EALIGN;
    const int64_t ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
    if (ho > 65535) return AFAN_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(K * CO + KK * ROWP) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)stem7_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const double M = (double)n * ho * wo;
    AFAN_PROF_FLOPS("conv_stem7_fwd_kernel", 2.0 * (M * CO + (double)n * hi * wi * 3), 2.0 * M * CO * K, st);
    dim3 grid((unsigned)((wo + SEG - 1) / SEG), (unsigned)ho, (unsigned)n);
    stem7_fwd_kernel<<<grid, BLOCK, lds, st>>>((const uint16_t*)x, (const uint16_t*)w, (uint16_t*)y, (int)hi, (int)wi, (int)ho, (int)wo);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// cols[N*Ho*Wo][152] (bf16) = im2col of x[N,Hi,Wi,3] for the 7x7 / 2 / pad 3 stem: column k = (r*7 + s)*3 + c for k < 147
// (the weights' KRSC order), zero for k >= 147.  The stem's forward and weight gradient then run as 1x1 problems on
// afan_conv_fwd_nhwc_bf16 / afan_conv_wgrad_nhwc_bf16 with ci = 152.
int afan_conv_stem7_im2col_k(void) { return KP; }

int afan_conv_stem7_im2col(const void* x, void* cols, int64_t n, int64_t hi, int64_t wi, afan_stream_t stream) {
    if (n <= 0 || hi <= 0 || wi <= 0 || n > 65535) return AFAN_ESHAPE;
    if (!x || !cols) return AFAN_ENULL;
    if (!aligned(x, 2) || !aligned(cols, 16)) return AFAN_EALIGN;
    const int64_t ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
    if (ho > 65535) return AFAN_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("stem7_im2col_kernel", 2.0 * ((double)n * hi * wi * 3 + (double)n * ho * wo * KP), st);
    dim3 grid((unsigned)((wo + SEG - 1) / SEG), (unsigned)ho, (unsigned)n);
    stem7_im2col_kernel<<<grid, BLOCK, 0, st>>>((const uint16_t*)x, (uint16_t*)cols, (int)hi, (int)wi, (int)ho, (int)wo);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// dx[N,Hi,Wi,3] (bf16) = col2im of dcols[N*Ho*Wo][152] (bf16): the adjoint of afan_conv_stem7_im2col (columns >= 147 ignored).
int afan_conv_stem7_col2im(const void* dcols, void* dx, int64_t n, int64_t hi, int64_t wi, afan_stream_t stream) {
    if (n <= 0 || hi <= 0 || wi <= 0 || n * hi * wi > 0x7fffffffLL * BLOCK) return AFAN_ESHAPE;
    if (!dcols || !dx) return AFAN_ENULL;
    if (!aligned(dcols, 2) || !aligned(dx, 2)) return AFAN_EALIGN;
    const int64_t ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1, pixels = n * hi * wi;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("stem7_col2im_kernel", 2.0 * ((double)pixels * 3 + (double)n * ho * wo * K), st);
    stem7_col2im_kernel<<<(unsigned)((pixels + BLOCK - 1) / BLOCK), BLOCK, 0, st>>>((const uint16_t*)dcols, (uint16_t*)dx, (int)hi, (int)wi,
                                                                                   (int)ho, (int)wo, pixels);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int64_t afan_conv_stem7_wgrad_workspace_floats(int64_t n, int64_t hi, int64_t wi) {
    if (n <= 0 || hi <= 0 || wi <= 0) return 0;
    const int64_t ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
    return (int64_t)wgrad_blocks(n * ho * ((wo + SEG - 1) / SEG)) * K * CO;
}

// grad[64,7,7,3] (fp32, KRSC) (+)= sum over output pixels of dy[N,Ho,Wo,64] * window(x[N,Hi,Wi,3])
int afan_conv_stem7_wgrad_nhwc_bf16(const void* x, const void* dy, float* grad, int64_t n, int64_t hi, int64_t wi,
                                    float* workspace, int accumulate, afan_stream_t stream) {
    if (n <= 0 || hi <= 0 || wi <= 0) return AFAN_ESHAPE;
    if (!x || !dy || !grad || !workspace) return AFAN_ENULL;
    if (!aligned(x, 2) || !aligned(dy, 2) || !aligned(grad, 4) || !aligned(workspace, 4)) return AFAN_EALIGN;
    const int64_t ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
    const int segs = (int)((wo + SEG - 1) / SEG);
    const int64_t tiles = n * ho * segs;
    const int G = wgrad_blocks(tiles);
    hipStream_t st = (hipStream_t)stream;
    const double M = (double)n * ho * wo;
    AFAN_PROF_FLOPS("conv_stem7_wgrad_kernel", 2.0 * (M * CO + (double)n * hi * wi * 3) + 4.0 * G * K * CO, 2.0 * M * CO * K, st);
    stem7_wgrad_kernel<<<G, BLOCK, 0, st>>>((const uint16_t*)x, (const uint16_t*)dy, workspace, (int)n, (int)hi, (int)wi, (int)ho,
                                            (int)wo, segs, tiles);
    AFAN_LAUNCH_CHECK();
    stem7_wgrad_reduce_kernel<<<(K * CO + BLOCK - 1) / BLOCK, BLOCK, 0, st>>>(workspace, grad, G, accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
