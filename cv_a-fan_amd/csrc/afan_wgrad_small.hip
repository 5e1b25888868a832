// Weight gradient of the 3x3 convolutions with a 16/32-channel side — every layer of the first two stages of the
// reference's own CIFAR ResNets (resnet_s.py:88-106: ResNet-20s / ResNet-56s) and their stride-2 transitions.  With this
// kernel (and afan_conv_stem.hip) those networks' training step holds no vendor kernel either.
//
//   dW[co][r][q][ci] = sum over output pixels  dy[pix][co] * x[pix * s + (r - 1, q - 1)][ci]
//
// 0.3-0.6 GFLOP and 4-8 MB per layer: traffic and latency decide, not the MFMA rate.  A wave owns tiles of 32 consecutive
// output pixels (32 / Wo rows of one image).  Per tile it stages the dy tile (32 x Co, contiguous) and the input window
// the nine taps reach ((R-1) s + 3 rows x (Wo-1) s + 3 columns x Ci, zero outside the image) in wave-private LDS with
// 16-byte pieces, then for every tap runs 32x32x16 MFMAs with the PIXEL index as the reduction: A = dy (rows = output
// channels), B = the shifted window (columns = input channels), both read transposed out of LDS.  The nine 32 x 32
// accumulators stay in registers over all of the wave's tiles; waves are summed through LDS, each workgroup writes one
// fp32 slab [Co][3][3][Ci], and a second launch sums the slabs in fixed order (reproducible, no float atomics).
// blockIdx.y = group of 32 output channels (two for the 32 -> 64 transition).
#include "afan_wgrad_small.h"
#include <stdlib.h>

using namespace afan;

namespace afan_wgrad_small {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr uint32_t OOB = 0x80000000u;
constexpr int MAX_WIN = 9 * 17 * (32 + 8);       // largest window: 32 -> 64, stride 2, 16x16 input (elements)
constexpr int MAX_DY = 32 * (32 + 8);            // dy tile of one 32-channel group

struct P {
    const uint16_t* x; const uint16_t* dy; float* slab;
    const uint16_t* x2; const uint16_t* dy2;   // optional second operand pair (another pass over the same layer), N2 images
    int N2;
    int N, Hi, Wi, Ci, Ho, Wo, Co, s;
    int R;                 // output rows per tile (32 / Wo, or 1 when Wo >= 32)
    int wr, wc;            // window rows / columns
};

__global__ __launch_bounds__(THREADS) void wgrad_small_kernel(const P p) {
    __shared__ __attribute__((aligned(16))) uint16_t win_s[WAVES][MAX_WIN];
    __shared__ __attribute__((aligned(16))) uint16_t dy_s[WAVES][MAX_DY];
    static_assert(sizeof(float) * WAVES * 32 * 33 <= sizeof(uint16_t) * WAVES * MAX_WIN, "part aliases win_s");
    float (*part)[32][33] = reinterpret_cast<float (*)[32][33]>(&win_s[0][0]);   // used after the tile loop only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int Ci = p.Ci, Co = p.Co, s = p.s, Wo = p.Wo, Hi = p.Hi, Wi = p.Wi;
    const int cg = blockIdx.y;                             // output channels cg*32 .. +32
    const int cow = Co - cg * 32 < 32 ? Co - cg * 32 : 32; // channels of this group (16 or 32)
    const int LDW = Ci + 8, LDD = 32 + 8;
    const __amdgpu_buffer_rsrc_t xr1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x), 0, (int)((int64_t)p.N * Hi * Wi * Ci * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x2), 0, p.x2 ? (int)((int64_t)p.N2 * Hi * Wi * Ci * 2) : 0, 0x00020000);
    uint16_t* win = win_s[wave];
    uint16_t* dyt = dy_s[wave];

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int cpp = Ci >> 3;                               // 16-byte pieces per input pixel
    const int win_pieces = p.wr * p.wc * cpp;
    const int dpp = cow >> 3;                              // pieces per dy pixel (this group's channels)
    const int tiles_per_img = (p.Ho * Wo) >> 5;
    const int tiles1 = p.N * tiles_per_img, tiles = tiles1 + p.N2 * tiles_per_img;
    for (int tg = blockIdx.x * WAVES + wave; tg < tiles; tg += gridDim.x * WAVES) {
        const bool seg2 = tg >= tiles1;                                // wave-uniform: a tile lies in one operand pair
        const int t = seg2 ? tg - tiles1 : tg;
        const __amdgpu_buffer_rsrc_t xr = seg2 ? xr2 : xr1;
        const uint16_t* dyb = seg2 ? p.dy2 : p.dy;
        const int n = t / tiles_per_img, ti = t - n * tiles_per_img;
        const int ho0 = Wo >= 32 ? ti / (Wo >> 5) : ti * p.R;          // first output row of the tile
        const int wo0 = Wo >= 32 ? (ti - ho0 * (Wo >> 5)) << 5 : 0;    // first output column
        const int hi0 = ho0 * s - 1, wi0 = wo0 * s - 1;                // input coordinate of window element (0, 0)
        for (int q = lane; q < win_pieces; q += 64) {
            const int px = q / cpp, c8 = q - px * cpp;
            const int wy = px / p.wc, wx = px - wy * p.wc;
            const int hi = hi0 + wy, wi = wi0 + wx;
            const bool ok = hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
            const uint32_t off = ok ? (uint32_t)((((n * Hi + hi) * Wi + wi) * Ci + c8 * 8) * 2) : OOB;
            *reinterpret_cast<u32x4*>(win + px * LDW + c8 * 8) =
                __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
        }
        const uint16_t* dsrc = dyb + ((int64_t)t * 32) * Co + cg * 32;
        for (int q = lane; q < 32 * dpp; q += 64) {
            const int px = q / dpp, c8 = q - px * dpp;
            *reinterpret_cast<u16x8*>(dyt + px * LDD + c8 * 8) = *reinterpret_cast<const u16x8*>(dsrc + (int64_t)px * Co + c8 * 8);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int p0 = 16 * ks + 8 * half;             // this lane's 8 reduction pixels p0 .. p0+7 of the tile
            u32x4 a;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t lo = col < cow ? dyt[(p0 + 2 * q) * LDD + col] : 0u;
                const uint32_t hi = col < cow ? dyt[(p0 + 2 * q + 1) * LDD + col] : 0u;
                a[q] = lo | (hi << 16);
            }
            const bf16x8 fa = __builtin_bit_cast(bf16x8, a);
            // window element of pixel p0 + j under tap (0, 0): row (prow * s), column (pcol * s); 8 pixels = one row segment
            int base[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pp = p0 + j;
                const int prow = Wo >= 32 ? 0 : pp / Wo, pcol = Wo >= 32 ? pp : pp - prow * Wo;
                base[j] = ((prow * s) * p.wc + pcol * s) * LDW + col;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int tap = (r * p.wc + q) * LDW;
                    u32x4 b;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t lo = col < Ci ? win[base[2 * k] + tap] : 0u;
                        const uint32_t hi = col < Ci ? win[base[2 * k + 1] + tap] : 0u;
                        b[k] = lo | (hi << 16);
                    }
                    acc[r * 3 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, b), acc[r * 3 + q], 0, 0, 0);
                }
        }
        __builtin_amdgcn_wave_barrier();
    }

    __syncthreads();                                       // every wave is done with its window: `part` may take the space
    // acc[t][4g + e] = dW[co = cg*32 + 8g + 4 half + e][tap t][ci = col]; waves summed in fixed order, tap by tap
    float* slab = p.slab + (int64_t)blockIdx.x * Co * 9 * Ci;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[wave][8 * g + 4 * half + e][col] = acc[t][4 * g + e];
        __syncthreads();
        for (int i = tid; i < cow * Ci; i += THREADS) {
            const int co = i / Ci, ci = i - co * Ci;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += part[w][co][ci];
            slab[((cg * 32 + co) * 9 + t) * Ci + ci] = v;
        }
        __syncthreads();
    }
}

// grad[i] (+)= sum over the S slabs, fixed order: 64 elements x 16 slab groups per block
__global__ __launch_bounds__(1024) void wgrad_small_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad,
                                                                  int total, int S, int accumulate) {
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    float v = 0.f;
    if (i < total)
        for (int s = q; s < S; s += 16) v += slab[(int64_t)s * total + i];
    red[q][e] = v;
    __syncthreads();
    if (q == 0 && i < total) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += red[k][e];
        grad[i] = accumulate ? grad[i] + sum : sum;
    }
}

int slabs(int64_t tiles, int64_t co) {
    static const int64_t target = [] { const char* v = getenv("AFAN_WGRAD_SMALL_WGS"); return v ? (int64_t)atoi(v) : (int64_t)256; }();
    int64_t g = target / ((co + 31) / 32);                 // ~one workgroup per CU over both channel groups
    const int64_t need = (tiles + WAVES - 1) / WAVES;
    if (g > need) g = need;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

bool eligible(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride) {
    static const bool on = [] { const char* v = getenv("AFAN_WGRAD_SMALL"); return !v || atoi(v) != 0; }();
    if (!on || k != 3 || !(stride == 1 || stride == 2) || n <= 0 || hi <= 0 || wi <= 0) return false;
    if (!(ci == 16 || ci == 32) || !(co == 16 || co == 32 || co == 64)) return false;   // (64 x 64 and up: afan_wgrad.hip)
    if (stride == 2 && ((hi | wi) & 1)) return false;
    const int64_t ho = (hi - 1) / stride + 1, wo = (wi - 1) / stride + 1;
    if (wo >= 32 ? wo % 32 != 0 : (32 % wo != 0 || ho % (32 / wo) != 0)) return false;
    const int64_t R = wo >= 32 ? 1 : 32 / wo, wr = (R - 1) * stride + 3, wc = ((wo >= 32 ? 32 : wo) - 1) * stride + 3;
    if (wr * wc * (ci + 8) > MAX_WIN) return false;
    return n * hi * wi * ci * 2 <= 0x7fffffffLL && n * ho * wo * co * 2 <= 0x7fffffffLL;
}

int64_t workspace_floats(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int stride) {
    const int64_t ho = (hi - 1) / stride + 1, wo = (wi - 1) / stride + 1;
    return (int64_t)slabs(n * ho * wo / 32, co) * co * 9 * ci;
}

int launch(const void* x, const void* dy, float* grad, int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int stride,
           float* ws, int accumulate, hipStream_t st, const void* x2, const void* dy2, int64_t n2) {
    P p{};
    p.x = (const uint16_t*)x; p.dy = (const uint16_t*)dy; p.slab = ws;
    p.x2 = (const uint16_t*)x2; p.dy2 = (const uint16_t*)dy2; p.N2 = (int)n2;
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci; p.Co = (int)co; p.s = stride;
    p.Ho = (int)((hi - 1) / stride + 1); p.Wo = (int)((wi - 1) / stride + 1);
    p.R = p.Wo >= 32 ? 1 : 32 / p.Wo;
    p.wr = (p.R - 1) * stride + 3;
    p.wc = ((p.Wo >= 32 ? 32 : p.Wo) - 1) * stride + 3;
    const int S = slabs((int64_t)(n + n2) * p.Ho * p.Wo / 32, co);
    dim3 grid((unsigned)S, (unsigned)((co + 31) / 32));
    wgrad_small_kernel<<<grid, THREADS, 0, st>>>(p);
    AFAN_LAUNCH_CHECK();
    const int total = (int)(co * 9 * ci);
    wgrad_small_reduce_kernel<<<(total + 63) / 64, 1024, 0, st>>>(ws, grad, total, S, accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // namespace afan_wgrad_small
