// Convolutions with 16 / 32 channels on one side: the layers of the reference's own CIFAR ResNets (resnet_s.py:88-106:
// 16-32-64 channels; ResNet-20s / ResNet-56s, the networks run_perturb.sh trains).  Same GEMM view and tap lists as
// afan_conv.hip (forward, stride-1 dgrad, stride-2 dgrad as four parity classes), different regime: 0.3-0.6 GFLOP per
// layer, a reduction of only 16-64 channels per tap and 16-64 output channels.  Staging tiles through LDS buys nothing
// here; instead
//   * the weights of a 32-channel output block live in registers for the life of a persistent workgroup
//     (T taps x Ci/16 MFMA operand fragments, at most 36 x 4 VGPRs);
//   * a wave owns tiles of 32 consecutive output pixels; the MFMA B operand of tap t is read STRAIGHT from global memory:
//     lane l holds the 8 channels [8*(l/32), +8) (+16 per k-step) of pixel l%32 — one 16-byte buffer load per lane, padding
//     by out-of-range offsets; neighbouring lanes read neighbouring pixels, the two lane halves the two halves of a
//     32-byte channel row;
//   * no LDS, no barrier in the loop: occupancy (4 waves per SIMD) hides the loads;
//   * outputs leave as 8-byte pieces (4 channels) per lane; the epilogue fusions of afan_conv.hip (addend, BatchNorm
//     moments, BatchNorm-backward sums, image groups) are kept: per-lane partial sums across all of a wave's tiles, one
//     butterfly + one LDS fold per workgroup at the end, then the f64 accumulators.
#include "afan_conv_params.h"
#include <stdlib.h>

using namespace afan;

namespace afan_conv {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int THREADS = 256;
constexpr int NB = 32;       // output channels per workgroup column

template <int KK>            // MFMA k-steps per tap = reduction channels / 16
__global__ __launch_bounds__(THREADS) void conv_small_kernel(const ConvP pp) {
    const ConvClass& cc = pp.cls[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int co0 = blockIdx.z * NB;
    const int T = cc.T, Ci = pp.Ci, Hi = pp.Hi, Wi = pp.Wi, Co = pp.Co;
    const uint32_t Wg = (uint32_t)cc.Wg, Hg = (uint32_t)cc.Hg, HWg = Hg * Wg;
    const uint32_t M = (uint32_t)pp.N * HWg;
    constexpr uint32_t OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(pp.x), 0, (int)((int64_t)pp.N * Hi * Wi * Ci * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(pp.w), 0, (int)((int64_t)Co * pp.w_row_stride * 2), 0x00020000);
    const int out_bytes = (int)((int64_t)pp.N * pp.Ho * pp.Wo * Co * 2);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pp.addend), 0, pp.addend ? out_bytes : 0, 0x00020000);
    const bool want_stats = pp.acc != nullptr;
    const bool bn_bwd = want_stats && pp.bnx != nullptr;
    const __amdgpu_buffer_rsrc_t bxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pp.bnx), 0, bn_bwd ? out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t byr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pp.bny), 0, (bn_bwd && pp.bny) ? out_bytes : 0, 0x00020000);

    // weights -> registers: fragment (t, kk) = output channel co0 + col, reduction channels kk*16 + half*8 .. +8
    bf16x8 wreg[MAX_TAPS][KK];
    {
        const uint32_t row = (uint32_t)(co0 + col);
        const uint32_t base = row < (uint32_t)Co ? (row * (uint32_t)pp.w_row_stride + (uint32_t)half * 8u) * 2u : OOB;
#pragma unroll
        for (int t = 0; t < MAX_TAPS; ++t)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const uint32_t off = (t < T && base != OOB) ? base + (uint32_t)(cc.wofs[t] + kk * 16) * 2u : OOB;
                wreg[t][kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)off, 0, 0));
            }
    }

    // tiles of this wave; with two image groups the first half of the workgroups walks the first half-batch
    const uint32_t tiles = (M + 31) / 32;
    uint32_t wid = blockIdx.x * (THREADS / 64) + wave, W = gridDim.x * (THREADS / 64);
    uint32_t t_begin = 0, t_end = tiles;
    int grp = 0;
    if (pp.groups == 2) {
        const uint32_t half_tiles = ((uint32_t)(pp.N / 2) * HWg) / 32;       // exact (host check)
        W >>= 1;
        if (wid >= W) { grp = 1; wid -= W; }
        t_begin = grp * half_tiles;
        t_end = t_begin + half_tiles;
    }
    const float* bn_stats = pp.bn_stats ? pp.bn_stats + (int64_t)grp * 4 * Co : nullptr;

    // per-lane partial sums of the lane's 16 channels (8g + 4*half + e) over all its pixels
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;

    for (uint32_t tile = t_begin + wid; tile < t_end; tile += W) {
        const uint32_t m = tile * 32 + col;
        const bool valid = m < M;
        const uint32_t n = m / HWg, rem = m - n * HWg, hg = rem / Wg, wg = rem - hg * Wg;
        const int hi0 = (int)hg * pp.in_s, wi0 = (int)wg * pp.in_s;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < MAX_TAPS; ++t) {
            if (t < T) {
                const int hi = hi0 + cc.dh[t], wi = wi0 + cc.dw[t];
                const bool ok = valid && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
                const uint32_t off = ok ? ((((n * Hi + hi) * Wi + wi) * Ci) + half * 8) * 2u : OOB;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const bf16x8 fx = __builtin_bit_cast(
                        bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(ok ? off + kk * 32u : OOB), 0, 0));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[t][kk], fx, acc, 0, 0, 0);
                }
            }
        }
        // ---- epilogue: lane = pixel `col`; channels co0 + 8g + 4*half + e -------------------------------------------
        const uint32_t opix = (n * pp.Ho + hg * pp.out_s + cc.out_h0) * pp.Wo + wg * pp.out_s + cc.out_w0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = co0 + 8 * g + 4 * half;
            const bool st_ok = valid && ch < Co;
            const uint32_t bo = st_ok ? (opix * (uint32_t)Co + (uint32_t)ch) * 2u : OOB;
            u16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = f2bf(acc[4 * g + e]);
            if (pp.addend) {
                const u16x4 a = __builtin_bit_cast(u16x4, __builtin_amdgcn_raw_buffer_load_b64(ar, (int)bo, 0, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(a[e]));
            }
            if (st_ok) *reinterpret_cast<u16x4*>(pp.y + (int64_t)opix * Co + ch) = v;
            if (bn_bwd) {
                const u16x4 xv = __builtin_bit_cast(u16x4, __builtin_amdgcn_raw_buffer_load_b64(bxr, (int)bo, 0, 0));
                u16x4 yv = xv;
                if (pp.bny) yv = __builtin_bit_cast(u16x4, __builtin_amdgcn_raw_buffer_load_b64(byr, (int)bo, 0, 0));
                if (st_ok) {
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(bn_stats + ch);
                    const f32x4 al = *reinterpret_cast<const f32x4*>(bn_stats + 2 * Co + ch);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(bn_stats + 3 * Co + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xf = bf2f(xv[e]);
                        float gval = bf2f(v[e]);
                        if (pp.bny) gval = (bf2f(yv[e]) > 0.f) ? gval : 0.f;
                        else if (pp.bn_relu) gval = (fmaf(xf, al[e], be[e]) > 0.f) ? gval : 0.f;
                        s1[4 * g + e] += gval;
                        s2[4 * g + e] += gval * (xf - mu[e]);
                    }
                }
            } else if (want_stats && st_ok) {
                const f32x4 sh = pp.shift ? *reinterpret_cast<const f32x4*>(pp.shift + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = bf2f(v[e]) - sh[e];
                    s1[4 * g + e] += f;
                    s2[4 * g + e] += f * f;
                }
            }
        }
    }

    if (want_stats) {
        // sums over the 32 pixels-lanes of each half (same channel set), then over the 4 waves, then one f64 atomic
        // per channel and workgroup
        __shared__ float red[THREADS / 64][2][NB];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s1[r] += __shfl_xor(s1[r], o, 64);
                s2[r] += __shfl_xor(s2[r], o, 64);
            }
        if (col == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 8 * (r >> 2) + 4 * half + (r & 3);
                red[wave][0][c] = s1[r];
                red[wave][1][c] = s2[r];
            }
        }
        __syncthreads();
        if (tid < NB && co0 + tid < Co) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) {
                a += red[w][0][tid];
                b += red[w][1][tid];
            }
            const int c = co0 + tid;
            double* blk = pp.acc + (int64_t)grp * pp.acc_stride;
            double* dst = blk + (int64_t)((blockIdx.x + blockIdx.y) & (pp.acc_ns - 1)) * 2 * Co;
            unsafeAtomicAdd(dst + c, (double)a);
            unsafeAtomicAdd(dst + Co + c, (double)b);
            const bool first = blockIdx.y == 0 && blockIdx.x == (pp.groups == 2 && grp ? gridDim.x / 2 : 0);
            if (!bn_bwd && first) reinterpret_cast<float*>(blk + (int64_t)2 * pp.acc_ns * Co)[c] = pp.shift ? pp.shift[c] : 0.f;
        }
    }
}

}  // namespace

bool small_eligible(const ConvP& p) {
    static const bool on = [] { const char* v = getenv("AFAN_CONV_SMALL"); return !v || atoi(v) != 0; }();
    if (!on || p.stats) return false;                       // (the partial-slab form is not implemented here)
    if (!(p.Ci == 16 || p.Ci == 32 || p.Ci == 64) || p.Co % 16 != 0 || p.Co > 64) return false;
    return p.Ci < 64 || p.Co < 64;
}

int small_launch(const ConvP& p, hipStream_t st) {
    int64_t mmax = 0;
    for (int c = 0; c < p.n_classes; ++c) {
        const int64_t m = (int64_t)p.N * p.cls[c].Hg * p.cls[c].Wg;
        if (m > mmax) mmax = m;
    }
    const int64_t tiles = (mmax + 31) / 32;
    int64_t gx = (tiles + 3) / 4;                              // one tile per wave ...
    if (gx > 1024) gx = 1024;                                  // ... up to 4 resident workgroups per CU, then persistent
    if (gx < 1) gx = 1;
    if (p.groups == 2) gx = (gx + 1) & ~(int64_t)1;            // a workgroup never mixes the two image groups
    dim3 grid((unsigned)gx, (unsigned)p.n_classes, (unsigned)((p.Co + NB - 1) / NB));
    if (p.Ci == 16) conv_small_kernel<1><<<grid, THREADS, 0, st>>>(p);
    else if (p.Ci == 32) conv_small_kernel<2><<<grid, THREADS, 0, st>>>(p);
    else conv_small_kernel<4><<<grid, THREADS, 0, st>>>(p);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // namespace afan_conv
