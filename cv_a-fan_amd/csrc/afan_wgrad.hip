// Weight-gradient convolution for the bf16 channels-last backbone on gfx950 (MFMA 32x32x16 bf16, fp32 accumulate).
// Part of the joint backward of the A-FAN step (Classification/main_perturb.py:200 `loss.backward()`): every tail
// convolution gets two weight-gradient passes per iteration (adv branch + clean branch), every head convolution one.
//
//   dW[co][r][s][ci] += sum over output pixels p of  dy[p][co] * x[pixel(p) + (r - pad, s - pad)][ci]
//
// GEMM view per tap: rows = co, cols = ci, reduction = pixels.  Both operands are stored pixel-major (channels-last), i.e.
// the reduction index is the SLOW index in memory — the transpose an MFMA operand needs is done by the LDS read itself:
// tiles are staged row-major ([pixel][channel], plain 16-byte coalesced copies, hardware zero-fill for the padding)
// and fragments are fetched with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered
// channel-per-lane).  Row stride = channels + 32 elements puts the 4 pixel rows of a read on disjoint bank quarters.
//
// Work split: one workgroup = one (co tile, ci tile, tap, pixel slice); it writes an fp32 partial tile to a slab
// [slice][tap][co][ci]; a second launch sums the slices IN ORDER (deterministic, no float atomics) and ADDS the result
// into the fp32 gradient arena (KRSC), so autograd's AccumulateGrad, the bf16->fp32 cast and the split-K workspace
// zero/cast passes of the vendor path all disappear.
//
// Measured in round 4 and NOT kept (profiles/r04_wgrad_experiments.txt):
//  * in-launch reduction (ticket per (tile, tap); the last slice to arrive folds the slices in slice order — bit-identical results):
//    with __threadfence() the kernel ran 3.6x slower (215 vs 60 us: the agent-scope release writes back and invalidates the L2
//    every other workgroup streams its operands through); in the write-through form of cdna_hip_programming.md (16-byte sc1 slab
//    stores, sc1 loads, agent-scope ticket) the last arriver reads S x 64 KB alone at the cross-XCD rate (~65 GB/s) — S = 14-56
//    here, a 14-55 us serial tail per tile against the 8-17 us of a reduce launch that uses the whole chip: 147 vs 52 + 10 us per
//    layer inside the step.  The guide's rule (combine in-launch only when S x slab bytes per tile is a few tens of KB) says the same.
//  * operand tiles three K-steps ahead in registers with unconditional loads (vmcnt(23..16) in the steady state instead of a
//    drain per step): 8.93 vs 8.92 ms per step, every layer within 2 % — the K loop is not waiting for its loads.
#include "afan_common.h"
#include "afan_conv_stem.h"
#include "afan_wgrad_small.h"
#include <stdlib.h>

using namespace afan;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;
constexpr int BKP = 64;   // pixels per step

struct FastDiv {  // q = x / d for x < 2^31 (Granlund-Montgomery): (umulhi(x, m) + x) >> l
    uint32_t m, l;
};
static FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while ((1u << l) < d) ++l;
    f.l = l;
    f.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t x, FastDiv f) { return (__umulhi(x, f.m) + x) >> f.l; }

struct WgradP {
    const uint16_t* x;   // [N, Hi, Wi, Ci]
    const uint16_t* dy;  // [N, Ho, Wo, Co]
    float* slab;         // [S][taps][Co][Ci]
    int N, Hi, Wi, Ci, Ho, Wo, Co;
    int k, stride, pad, dil;     // pad = dil * (k / 2)
    int S;               // pixel slices
    int steps_per_slice; // BKP-pixel steps per slice (last slice may be ragged; rows beyond P read as zero)
    uint32_t P;          // output pixels in total (both segments)
    FastDiv dWo, dHo;
    // optional second segment: pixels [P1, P) come from (x2, dy2) — the same layer's operands of another pass (the clean
    // and the adversarial tail pass of one iteration: ONE launch and one slab reduction for both).  P1 % BKP == 0, so
    // a K-step never straddles the segments; without a second segment P1 == P.
    const uint16_t* x2;
    const uint16_t* dy2;
    uint32_t P1;
    int N2;
    // incremental pixel decomposition (INC kernels): Wo divides 64, so a staged row keeps its output column over the K
    // loop and advances by 64 pixels = dn images + dho rows per step; N1 = images of the first segment
    int dn, dho, N1;
    int xcd_remap;       // 1: XCD-aware workgroup order (see the kernel)
    // S == 1 (few pixels: the 33 x 33 stages of DeepLab at 2 images per GPU): every (tile, tap) has ONE writer, so the tile
    // goes straight into the gradient [co][tap][ci] (added to it when `accumulate`) — no slab, no reduce launch
    float* direct;
    int accumulate;
};

template <int BM, int BN, bool INC>   // BM = co tile, BN = ci tile; INC: see WgradP::dn
__device__ __forceinline__ void wgrad_body(const WgradP& p, const uint32_t lin, const uint32_t tiles_, const uint32_t taps_,
                                           const uint32_t total) {
    constexpr int TM = BM / 2, TN = BN / 2, MI = TM / 32, NI = TN / 32;
    constexpr int LDA = BM + 32, LDB = BN + 32;            // elements per LDS row
    constexpr int PA = BM / 8, PB = BN / 8;                // 16-byte pieces per row
    constexpr int RA = THREADS / PA, RB = THREADS / PB;    // rows covered per pass
    constexpr int NA = BKP / RA, NB = BKP / RB;            // pieces per thread per step
    __shared__ __attribute__((aligned(16))) uint16_t As[BKP * LDA];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[BKP * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (p.Ci + BN - 1) / BN;      // ragged channel counts (multiples of 8): pieces beyond Co / Ci read as zero
    // XCD-aware order: workgroups go to the 8 XCDs round-robin in dispatch order, and each XCD has its own L2.  The taps
    // of one (tile, pixel slice) read the same dy tile and the same x pixels (shifted by a column or a row): with the tap
    // as the grid's y dimension they ran far apart in time and on different XCDs, and every tap fetched its operands from
    // HBM again (PMC: 267 MB per launch against 33 MB of operands).  Here dispatch slot `lin` is mapped to a logical
    // index that is contiguous per XCD, with the tap fastest: the nine taps of a group run back to back on ONE XCD.
    int tap, slice, tile;
    {
        uint32_t logical = lin;
        if (p.xcd_remap) {
            const uint32_t per = total >> 3, main_ = per << 3;           // the last total % 8 slots keep their index
            if (lin < main_) logical = (lin & 7u) * per + (lin >> 3);
        }
        tap = (int)(logical % taps_);
        const uint32_t rest = logical / taps_;
        tile = (int)(rest % tiles_);
        slice = (int)(rest / tiles_);
    }
    const int co0 = (tile / tiles_n) * BM, ci0 = (tile % tiles_n) * BN;
    const int dh = (tap / p.k) * p.dil - p.pad, dw = (tap % p.k) * p.dil - p.pad;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x), 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.Ci * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.dy), 0, (int)((int64_t)p.P1 * p.Co * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x2), 0, p.x2 ? (int)((int64_t)p.N2 * p.Hi * p.Wi * p.Ci * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.dy2), 0, p.dy2 ? (int)((int64_t)(p.P - p.P1) * p.Co * 2) : 0, 0x00020000);
    constexpr uint32_t OOB = 0x80000000u;

    const int pa = tid % PA, ra0 = tid / PA;   // dy tile: piece / first row
    const int pb = tid % PB, rb0 = tid / PB;   // x tile
    const bool a_ok = co0 + pa * 8 < p.Co, b_ok = ci0 + pb * 8 < p.Ci;
    const uint32_t p_begin = (uint32_t)slice * p.steps_per_slice * BKP;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 va[NA], vb[NB];
    // INC: per staged x row — image and output row of its pixel at the slice's first step (over the concatenated batch of
    // both segments: P1 is a whole number of images), validity and byte offset of its (constant) input column
    int rn[NB], rho[NB];
    bool wi_ok[NB];
    uint32_t col_off[NB];
    if constexpr (INC) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint32_t pix = p_begin + rb0 + i * RB;
            const uint32_t t1 = fdiv(pix, p.dWo), wo = pix - t1 * p.Wo;
            const uint32_t n = fdiv(t1, p.dHo);
            rn[i] = (int)n;
            rho[i] = (int)(t1 - n * p.Ho);
            const int wi = (int)wo * p.stride + dw;
            wi_ok[i] = wi >= 0 && wi < p.Wi;
            col_off[i] = (uint32_t)((wi * p.Ci + ci0 + pb * 8) * 2);
        }
    }
    auto gload = [&](int ks) {
        const uint32_t pbase = p_begin + ks * BKP;
        const bool seg2 = pbase >= p.P1;                       // wave-uniform: the whole K-step lies in one segment
        const uint32_t shift = seg2 ? p.P1 : 0u, plim = seg2 ? p.P : p.P1;
        const __amdgpu_buffer_rsrc_t dsel = seg2 ? dr2 : dr, xsel = seg2 ? xr2 : xr;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const uint32_t pix = pbase + ra0 + i * RA;
            const uint32_t off = (pix < plim && a_ok) ? ((pix - shift) * p.Co + co0 + pa * 8) * 2u : OOB;
            va[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(dsel, (int)off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint32_t pix = pbase + rb0 + i * RB;
            if constexpr (INC) {
                // (n, ho) of this row are carried from step to step (two fast divisions and their multiplies per row and
                // step otherwise: ~120 VALU instructions per K-step against 16 MFMA); the column part is constant
                const int hi = rho[i] * p.stride + dh;
                const int nl = rn[i] - (seg2 ? p.N1 : 0);
                const bool ok = pix < plim && wi_ok[i] && hi >= 0 && hi < p.Hi && b_ok;
                const uint32_t off = ok ? ((uint32_t)((nl * p.Hi + hi) * p.Wi) * (uint32_t)p.Ci) * 2u + col_off[i] : OOB;
                vb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xsel, (int)off, 0, 0));
                rn[i] += p.dn;
                rho[i] += p.dho;
                if (rho[i] >= p.Ho) { rho[i] -= p.Ho; rn[i] += 1; }
            } else {
                const uint32_t pl = pix - shift;
                const uint32_t t1 = fdiv(pl, p.dWo), wo = pl - t1 * p.Wo;
                const uint32_t n = fdiv(t1, p.dHo), ho = t1 - n * p.Ho;
                const int hi = (int)ho * p.stride + dh, wi = (int)wo * p.stride + dw;
                const bool ok = pix < plim && hi >= 0 && hi < p.Hi && wi >= 0 && wi < p.Wi && b_ok;
                const uint32_t off = ok ? (((n * p.Hi + hi) * p.Wi + wi) * p.Ci + ci0 + pb * 8) * 2u : OOB;
                vb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xsel, (int)off, 0, 0));
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(As + (ra0 + i * RA) * LDA + pa * 8) = va[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(Bs + (rb0 + i * RB) * LDB + pb * 8) = vb[i];
    };
    // transposed fragment: lane l gets channel c_base + (l & 31), pixels kbase + 8*(l >> 5) + 0..7
    const int fr = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int fc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto frag = [&](const uint16_t* T, int ld, int kbase, int cbase) -> bf16x8 {
        typedef s16x4 __attribute__((address_space(3))) * lptr;
        const uint16_t* a0 = T + (kbase + fr) * ld + cbase + fc;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * ld));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    // ragged last slice: steps that start beyond P contribute nothing
    int KS = p.steps_per_slice;
    {
        const int64_t remaining = (int64_t)p.P - (int64_t)p_begin;
        const int64_t need = remaining <= 0 ? 0 : (remaining + BKP - 1) / BKP;
        if (need < KS) KS = (int)need;
    }
    if (KS > 0) gload(0);
    for (int ks = 0; ks < KS; ++ks) {
        lstore();
        __syncthreads();
        if (ks + 1 < KS) gload(ks + 1);   // in flight under the MFMAs
#pragma unroll
        for (int kk = 0; kk < BKP / 16; ++kk) {
            bf16x8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = frag(As, LDA, kk * 16, wr * TM + i * 32);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = frag(Bs, LDB, kk * 16, wc * TN + j * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // partial tile -> slab[slice][tap][co][ci] (ci contiguous: 32 lanes write 128 consecutive bytes)
    const int taps = p.k * p.k;
    float* out = p.slab + (((int64_t)slice * taps + tap) * p.Co) * p.Ci;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wr * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ci = ci0 + wc * TN + j * 32 + (lane & 31);
                if (co < p.Co && ci < p.Ci) {
                    if (p.direct) {
                        float* g = p.direct + ((int64_t)co * taps + tap) * p.Ci + ci;
                        *g = p.accumulate ? *g + acc[i][j][r] : acc[i][j][r];
                    } else {
                        out[(int64_t)co * p.Ci + ci] = acc[i][j][r];
                    }
                }
            }
}

template <int BM, int BN, bool INC>
__global__ __launch_bounds__(THREADS) void wgrad_kernel(const WgradP p) {
    wgrad_body<BM, BN, INC>(p, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y,
                            gridDim.x * gridDim.y * gridDim.z);
}

// 0 + p[0] + p[stride] + ... + p[(S - 1) * stride], added in THAT order, eight requests in flight: written as `for k: s += p[k * stride]`
// the loop was one memory round trip per slice (a 64-slice reduction of an 8 K-element gradient: 32 us on 8 workgroups,
// profiles/r06_r18_trace_by_grid.txt)
__device__ __forceinline__ f32x4 sum_slices(const float* __restrict__ p, int64_t stride, int S) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= S; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(k + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < S; ++k) s += *reinterpret_cast<const f32x4*>(p + (int64_t)k * stride);
    return s;
}

// Up to MAXQ independent weight-gradient problems (different layers, the same tile configuration) in ONE launch: the three
// convolutions of a bottleneck at 33 x 33 pixels and 2 images are 35-step reductions — 13-25 us launches that do not fill the
// chip and cost mostly their own ramp.  Workgroup `blockIdx.x` belongs to problem q with first[q] <= blockIdx.x < first[q+1]
// and keeps that problem's own (tile, tap, slice) order, XCD remap included.
constexpr int MAXQ = 4;
struct WgradPN {
    int n;
    uint32_t first[MAXQ + 1];      // multiples of 8: the XCD remap of wgrad_body reads the dispatch slot's XCD as lin % 8
    uint32_t count[MAXQ];          // workgroups of problem q (the slots up to first[q + 1] beyond them stay idle)
    uint32_t tiles[MAXQ], taps[MAXQ];
    WgradP p[MAXQ];
};
template <int BM, int BN, bool INC>
__global__ __launch_bounds__(THREADS) void wgrad_multi_kernel(const WgradPN pn) {
    int q = 0;
    while (q + 1 < pn.n && blockIdx.x >= pn.first[q + 1]) ++q;
    const uint32_t lin = blockIdx.x - pn.first[q];
    if (lin >= pn.count[q]) return;
    wgrad_body<BM, BN, INC>(pn.p[q], lin, pn.tiles[q], pn.taps[q], pn.count[q]);
}

struct RedPN {
    const float* slab[MAXQ];
    float* grad[MAXQ];
    int S[MAXQ], taps[MAXQ], Co[MAXQ], Ci[MAXQ];
};
// the slab reductions of those problems in one launch (blockIdx.y = problem; same arithmetic and order as wgrad_reduce_kernel)
__global__ __launch_bounds__(THREADS) void wgrad_reduce_multi_kernel(const RedPN r, int accumulate) {
    const int q = blockIdx.y;
    const float* __restrict__ slab = r.slab[q];
    float* __restrict__ grad = r.grad[q];
    const int S = r.S[q], taps = r.taps[q], Co = r.Co[q], Ci = r.Ci[q];
    const int64_t per = (int64_t)taps * Co * Ci;
    const int64_t nvec = per / 4;
    for (int64_t v = (int64_t)blockIdx.x * THREADS + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * THREADS) {
        const int64_t e = v * 4;
        const int ci = (int)(e % Ci);
        const int64_t t2 = e / Ci;
        const int co = (int)(t2 % Co);
        const int tap = (int)(t2 / Co);
        f32x4 s = sum_slices(slab + e, per, S);
        float* g = grad + ((int64_t)co * taps + tap) * Ci + ci;
        if (accumulate) s += *reinterpret_cast<const f32x4*>(g);
        *reinterpret_cast<f32x4*>(g) = s;
    }
}

// grad[co][tap][ci] (+)= sum_s slab[s][tap][co][ci], slices added in index order
__global__ __launch_bounds__(THREADS) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad,
                                                               int S, int taps, int Co, int Ci, int accumulate) {
    const int64_t per = (int64_t)taps * Co * Ci;
    const int64_t nvec = per / 4;
    for (int64_t v = (int64_t)blockIdx.x * THREADS + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * THREADS) {
        const int64_t e = v * 4;                       // index into [tap][co][ci]
        const int ci = (int)(e % Ci);
        const int64_t t2 = e / Ci;
        const int co = (int)(t2 % Co);
        const int tap = (int)(t2 / Co);
        f32x4 s = sum_slices(slab + e, per, S);
        float* g = grad + ((int64_t)co * taps + tap) * Ci + ci;
        if (accumulate) s += *reinterpret_cast<const f32x4*>(g);
        *reinterpret_cast<f32x4*>(g) = s;
    }
}

struct Plan {
    int bm, bn, S, steps;
};

Plan make_plan(int64_t P, int co, int ci, int taps) {
    Plan pl;
    pl.bm = (co % 128 == 0) ? 128 : 64;
    pl.bn = (ci % 128 == 0) ? 128 : 64;
    const int64_t tiles = (int64_t)((co + pl.bm - 1) / pl.bm) * ((ci + pl.bn - 1) / pl.bn) * taps;
    const int64_t total_steps = (P + BKP - 1) / BKP;
    static const int target = [] { const char* v = getenv("AFAN_WGRAD_WGS"); return v ? atoi(v) : 512; }();
    int64_t S = target / tiles;                    // ~2 workgroups per CU (measured: 512 beats 320 by 2 % of the step)
    // >= 16 steps per workgroup — except for the very short reductions (DeepLab's 33 x 33 stages at 2 images per GPU: 35
    // steps in all): there the launch is latency-bound and more, shorter slices win (DeepLab 27.1 -> 26.1 ms per iteration
    // at 4-8 steps; ResNet-50, whose shortest reduction is 49 steps, loses 0.6 % with the same rule, hence the bound)
    static const int min_steps_env = [] { const char* v = getenv("AFAN_WGRAD_MINSTEPS"); return v ? atoi(v) : 0; }();
    const int min_steps = min_steps_env > 0 ? min_steps_env : (total_steps <= 40 ? 5 : 16);
    const int64_t max_s = total_steps / min_steps > 0 ? total_steps / min_steps : 1;
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    // two short slices plus a reduce launch lose to one slice written straight into the gradient (see WgradP::direct)
    static const int direct_on = [] { const char* v = getenv("AFAN_WGRAD_DIRECT"); return v ? atoi(v) : 0; }();
    if (direct_on && S == 2 && total_steps <= 64) S = 1;
    pl.steps = (int)((total_steps + S - 1) / S);
    pl.S = (int)((total_steps + pl.steps - 1) / pl.steps);
    return pl;
}

int build_problem(const void* x, const void* dy, int64_t n, const void* x2, const void* dy2, int64_t n2, float* grad, int64_t hi,
                  int64_t wi, int64_t ci, int64_t co, int k, int stride, int dilation, float* workspace, int accumulate,
                  WgradP& p, Plan& pl);

template <int BM, int BN>
int launch(const WgradP& p, int taps, hipStream_t st) {
    dim3 grid((unsigned)(((p.Co + BM - 1) / BM) * ((p.Ci + BN - 1) / BN)), (unsigned)taps, (unsigned)p.S);
    if (p.dn >= 0) wgrad_kernel<BM, BN, true><<<grid, THREADS, 0, st>>>(p);
    else wgrad_kernel<BM, BN, false><<<grid, THREADS, 0, st>>>(p);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // namespace

extern "C" {

// floats of workspace the call below needs for its partial slabs
int64_t afan_conv_wgrad_workspace_floats(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride) {
    if (ci == 3) return afan_stem::eligible(n, hi, wi, ci, co, k, stride) ? afan_stem::wgrad_workspace_floats(n, hi, wi, co) : 0;
    if (afan_wgrad_small::eligible(n, hi, wi, ci, co, k, stride)) return afan_wgrad_small::workspace_floats(n, hi, wi, ci, co, stride);
    if (n <= 0 || hi <= 0 || wi <= 0 || ci % 8 || co % 8 || ci < 40 || co < 40 || !(k == 1 || k == 3) || !(stride == 1 || stride == 2))
        return 0;
    const int pad = k / 2;
    const int64_t P = n * ((hi + 2 * pad - k) / stride + 1) * ((wi + 2 * pad - k) / stride + 1);
    const Plan pl = make_plan(P, (int)co, (int)ci, k * k);
    return (int64_t)pl.S * k * k * co * ci;
}

// grad[Co,k,k,Ci] (fp32, KRSC) (+)= wgrad(x[N,Hi,Wi,Ci], dy[N,Ho,Wo,Co]); bf16 channels-last operands.
int afan_conv_wgrad_nhwc_bf16(const void* x, const void* dy, float* grad, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                              int64_t co, int k, int stride, int dilation, float* workspace, int accumulate,
                              afan_stream_t stream) {
    return afan_conv_wgrad2_nhwc_bf16(x, dy, n, nullptr, nullptr, 0, grad, hi, wi, ci, co, k, stride, dilation, workspace,
                                      accumulate, stream);
}

// The same over TWO operand pairs of one layer in one launch: grad (+)= wgrad(x, dy) + wgrad(x2, dy2) (n2 = 0: one pair).
// Workspace: afan_conv_wgrad_workspace_floats(n + n2, ...).  The first pair's output pixel count must be a multiple of 64.
int afan_conv_wgrad2_nhwc_bf16(const void* x, const void* dy, int64_t n, const void* x2, const void* dy2, int64_t n2,
                               float* grad, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride, int dilation,
                               float* workspace, int accumulate, afan_stream_t stream) {
    if (n <= 0 || hi <= 0 || wi <= 0 || ci <= 0 || co <= 0 || n2 < 0) return AFAN_ESHAPE;
    if (dilation < 1 || (dilation > 1 && (k != 3 || stride != 1 || ci % 8 || co % 8 || ci < 40 || co < 40))) return AFAN_ESHAPE;
    if (n2 > 0 && ci == 3) return AFAN_ESHAPE;                  // (the stem runs in one pass per iteration)
    if (n2 > 0 && (!x2 || !dy2)) return AFAN_ENULL;
    if (ci == 3) {                                              // the image stem has its own kernel
        if (!afan_stem::eligible(n, hi, wi, ci, co, k, stride)) return AFAN_ESHAPE;
        if (!x || !dy || !grad || !workspace) return AFAN_ENULL;
        if (!aligned(dy, 16) || !aligned(grad, 4) || !aligned(workspace, 16)) return AFAN_EALIGN;
        hipStream_t st = (hipStream_t)stream;
        const double M = (double)n * hi * wi;
        AFAN_PROF_FLOPS("conv_stem_wgrad_kernel", 2.0 * (M * co + M * 3), 2.0 * M * co * 27, st);
        return afan_stem::wgrad_launch(x, dy, grad, n, hi, wi, co, workspace, accumulate, st);
    }
    if (dilation == 1 && afan_wgrad_small::eligible(n, hi, wi, ci, co, k, stride)) {     // 16/32-channel layers
        if (!x || !dy || !grad || !workspace) return AFAN_ENULL;
        if (!aligned(x, 16) || !aligned(dy, 16) || !aligned(grad, 4) || !aligned(workspace, 16)) return AFAN_EALIGN;
        hipStream_t st = (hipStream_t)stream;
        const double P = (double)n * ((hi - 1) / stride + 1) * ((wi - 1) / stride + 1);
        AFAN_PROF_FLOPS("conv_wgrad_small_kernel", 2.0 * (P * co + (double)n * hi * wi * ci), 2.0 * P * co * ci * 9, st);
        if (n2 > 0 && (!aligned(x2, 16) || !aligned(dy2, 16))) return AFAN_EALIGN;
        return afan_wgrad_small::launch(x, dy, grad, n, hi, wi, ci, co, stride, workspace, accumulate, st, x2, dy2, n2);
    }
    WgradP p{};
    Plan pl;
    int e = build_problem(x, dy, n, x2, dy2, n2, grad, hi, wi, ci, co, k, stride, dilation, workspace, accumulate, p, pl);
    if (e) return e;
    const int taps = k * k;
    const int64_t P = (n + n2) * (int64_t)p.Ho * p.Wo;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    {
        AFAN_PROF_FLOPS("conv_wgrad_kernel", 2.0 * (double)(P * co + n * hi * wi * ci) + 4.0 * pl.S * taps * co * ci,
                        2.0 * (double)P * co * ci * taps, st);
        if (pl.bm == 128) rc = pl.bn == 128 ? launch<128, 128>(p, taps, st) : launch<128, 64>(p, taps, st);
        else rc = pl.bn == 128 ? launch<64, 128>(p, taps, st) : launch<64, 64>(p, taps, st);
    }
    if (rc || p.direct) return rc;
    const int64_t per = (int64_t)taps * co * ci;
    AFAN_PROF("conv_wgrad_reduce_kernel", 4.0 * per * (pl.S + 1 + (accumulate ? 1 : 0)), st);
    wgrad_reduce_kernel<<<grid_for(per / 4, THREADS, 1024), THREADS, 0, st>>>(workspace, grad, pl.S, taps, (int)co, (int)ci,
                                                                              accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"

namespace {
// the tiled kernel's problem description (operands checked, plan chosen): shared by the single and the multi launch
int build_problem(const void* x, const void* dy, int64_t n, const void* x2, const void* dy2, int64_t n2, float* grad, int64_t hi,
                  int64_t wi, int64_t ci, int64_t co, int k, int stride, int dilation, float* workspace, int accumulate,
                  WgradP& p, Plan& pl) {
    if (ci % 8 || co % 8 || ci < 40 || co < 40 || !(k == 1 || k == 3) || !(stride == 1 || stride == 2)) return AFAN_ESHAPE;
    if (!x || !dy || !grad || !workspace) return AFAN_ENULL;
    if (!aligned(x, 16) || !aligned(dy, 16) || !aligned(grad, 16) || !aligned(workspace, 16)) return AFAN_EALIGN;
    const int pad = dilation * (k / 2);
    const int64_t ho = (hi + 2 * (k / 2) - k) / stride + 1, wo = (wi + 2 * (k / 2) - k) / stride + 1;
    const int64_t P1 = n * ho * wo, P = (n + n2) * ho * wo;
    if (P * co * 2 > 0x7fffffffLL || (n + n2) * hi * wi * ci * 2 > 0x7fffffffLL) return AFAN_ESHAPE;
    if (n2 > 0 && (P1 % BKP != 0 || !aligned(x2, 16) || !aligned(dy2, 16))) return n2 > 0 && P1 % BKP != 0 ? AFAN_ESHAPE : AFAN_EALIGN;
    const int taps = k * k;
    pl = make_plan(P, (int)co, (int)ci, taps);
    p.x = (const uint16_t*)x; p.dy = (const uint16_t*)dy; p.slab = workspace;
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci; p.Ho = (int)ho; p.Wo = (int)wo; p.Co = (int)co;
    p.k = k; p.stride = stride; p.pad = pad; p.dil = dilation; p.S = pl.S; p.steps_per_slice = pl.steps; p.P = (uint32_t)P;
    p.x2 = (const uint16_t*)x2; p.dy2 = (const uint16_t*)dy2; p.P1 = (uint32_t)P1; p.N2 = (int)n2;
    p.N1 = (int)n;
    {
        static const bool remap_on = [] { const char* v = getenv("AFAN_WGRAD_XCD"); return !v || atoi(v) != 0; }();
        p.xcd_remap = remap_on ? 1 : 0;
    }
    p.dn = -1; p.dho = 0;
    {   // 64 pixels = dn images + dho output rows exactly (the column does not move): Wo | 64 and dho <= Ho
        static const bool inc_on = [] { const char* v = getenv("AFAN_WGRAD_INC"); return !v || atoi(v) != 0; }();
        if (inc_on && BKP % wo == 0) {
            const int64_t hw = ho * wo;
            p.dn = (int)(BKP / hw);
            p.dho = (int)((BKP % hw) / wo);
        }
    }
    p.dWo = make_fastdiv((uint32_t)wo); p.dHo = make_fastdiv((uint32_t)ho);
    {
        static const int direct_on = [] { const char* v = getenv("AFAN_WGRAD_DIRECT"); return v ? atoi(v) : 0; }();
        p.direct = (direct_on && pl.S == 1) ? grad : nullptr;
        p.accumulate = accumulate;
    }
    return AFAN_OK;
}
}  // namespace

extern "C" {

// what the tiled kernel would do with this problem: 0 = not its problem (stem / small-channel / unsupported), else
// bm | bn << 8 | (incremental pixel walk ? 1 << 16 : 0) | 1 << 17 — problems with equal codes can share a multi launch
int afan_conv_wgrad_plan(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride) {
    if (ci == 3 || afan_wgrad_small::eligible(n, hi, wi, ci, co, k, stride)) return 0;
    if (n <= 0 || hi <= 0 || wi <= 0 || ci % 8 || co % 8 || ci < 40 || co < 40 || !(k == 1 || k == 3) || !(stride == 1 || stride == 2))
        return 0;
    const int64_t ho = (hi + 2 * (k / 2) - k) / stride + 1, wo = (wi + 2 * (k / 2) - k) / stride + 1;
    const Plan pl = make_plan(n * ho * wo, (int)co, (int)ci, k * k);
    static const bool inc_on = [] { const char* v = getenv("AFAN_WGRAD_INC"); return !v || atoi(v) != 0; }();
    const int inc = (inc_on && BKP % wo == 0) ? 1 : 0;
    return pl.bm | (pl.bn << 8) | (inc << 16) | (1 << 17);
}

// nb (2..4) weight gradients of DIFFERENT layers in one launch + one reduction launch: grad[b] (+)= wgrad(x[b], dy[b]) with the
// per-problem shapes given as arrays; all problems must have the same afan_conv_wgrad_plan code.  workspace: the sum of the
// problems' afan_conv_wgrad_workspace_floats.  Results are bit-identical to nb separate afan_conv_wgrad_nhwc_bf16 calls.
int afan_conv_wgrad_multi_nhwc_bf16(int nb, const void* const* x, const void* const* dy, float* const* grad, const int64_t* n,
                                    const int64_t* hi, const int64_t* wi, const int64_t* ci, const int64_t* co, const int* k,
                                    const int* stride, const int* dilation, float* workspace, int accumulate,
                                    afan_stream_t stream) {
    if (nb < 1 || nb > MAXQ) return AFAN_ESHAPE;
    if (!x || !dy || !grad || !n || !hi || !wi || !ci || !co || !k || !stride || !dilation || !workspace) return AFAN_ENULL;
    WgradPN pn{};
    RedPN rp{};
    pn.n = nb;
    int code0 = 0;
    int64_t woff = 0, max_vec = 0;
    double bytes = 0, flops = 0;
    for (int b = 0; b < nb; ++b) {
        const int code = afan_conv_wgrad_plan(n[b], hi[b], wi[b], ci[b], co[b], k[b], stride[b]);
        if (!code || (b && code != code0)) return AFAN_ESHAPE;
        code0 = code;
        if (dilation[b] < 1 || (dilation[b] > 1 && (k[b] != 3 || stride[b] != 1))) return AFAN_ESHAPE;
        Plan pl;
        int e = build_problem(x[b], dy[b], n[b], nullptr, nullptr, 0, grad[b], hi[b], wi[b], ci[b], co[b], k[b], stride[b],
                              dilation[b], workspace + woff, accumulate, pn.p[b], pl);
        if (e) return e;
        pn.p[b].direct = nullptr;                                    // (a one-slice problem takes the slab + reduce here too)
        const int taps = k[b] * k[b];
        const int bm = code & 0xff, bn = (code >> 8) & 0xff;
        pn.tiles[b] = (uint32_t)(((co[b] + bm - 1) / bm) * ((ci[b] + bn - 1) / bn));
        pn.taps[b] = (uint32_t)taps;
        pn.count[b] = pn.tiles[b] * taps * (uint32_t)pl.S;
        pn.first[b + 1] = (pn.first[b] + pn.count[b] + 7u) & ~7u;
        rp.slab[b] = workspace + woff; rp.grad[b] = grad[b]; rp.S[b] = pl.S; rp.taps[b] = taps; rp.Co[b] = (int)co[b]; rp.Ci[b] = (int)ci[b];
        const int64_t per = (int64_t)taps * co[b] * ci[b];
        woff += (int64_t)pl.S * per;
        if (per / 4 > max_vec) max_vec = per / 4;
        const double P = (double)n[b] * pn.p[b].Ho * pn.p[b].Wo;
        bytes += 2.0 * (P * co[b] + (double)n[b] * hi[b] * wi[b] * ci[b]) + 4.0 * pl.S * per;
        flops += 2.0 * P * co[b] * ci[b] * taps;
    }
    hipStream_t st = (hipStream_t)stream;
    const int bm = code0 & 0xff, bn = (code0 >> 8) & 0xff;
    const bool inc = (code0 >> 16) & 1;
    const unsigned G = pn.first[nb];
    {
    AFAN_PROF_FLOPS("conv_wgrad_kernel", bytes, flops, st);
#define AFAN_WM(BM_, BN_)                                                                                   \
    do {                                                                                                    \
        if (inc) wgrad_multi_kernel<BM_, BN_, true><<<G, THREADS, 0, st>>>(pn);                             \
        else wgrad_multi_kernel<BM_, BN_, false><<<G, THREADS, 0, st>>>(pn);                                \
    } while (0)
    if (bm == 128) { if (bn == 128) AFAN_WM(128, 128); else AFAN_WM(128, 64); }
    else { if (bn == 128) AFAN_WM(64, 128); else AFAN_WM(64, 64); }
#undef AFAN_WM
    AFAN_LAUNCH_CHECK();
    }
    AFAN_PROF("conv_wgrad_reduce_kernel", 4.0 * (double)woff, st);
    wgrad_reduce_multi_kernel<<<dim3((unsigned)grid_for(max_vec, THREADS, 1024), (unsigned)nb), THREADS, 0, st>>>(rp, accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
