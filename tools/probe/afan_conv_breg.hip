// Implicit-GEMM convolution with the WEIGHT operand streamed global -> VGPRs from a fragment-major packed copy (no LDS for
// it) and only the activation operand staged in LDS by LDS-DMA.  Why (NOTES.md section 8, "Where the convolution stands"): the tiled
// kernel's K loop is bound by memory latency x bytes in flight, and the bytes in flight are capped by LDS capacity
// (2 workgroups x 1 tile, or 1 x 3 in the deep variant).  Here the weights ride in registers NB K-steps ahead (the VGPR file
// is 3x the LDS), which both adds in-flight bytes and halves the LDS traffic (writes: A only; reads: 8 instead of 12
// ds_read_b128 per wave and K-step).
//
// Packed weights: Wp[n_tile = row / 32][ks = tap * chunks + q][kk (4)][lane (64)][8]  bf16, element
//   (row = n_tile * 32 + (lane & 31),  k = q * 64 + kk * 16 + (lane >> 5) * 8 + e)  of tap `tap` —
// exactly the MFMA 32x32x16 operand of lane `lane`, so one K-step of a wave's 32 weight rows is 4 KiB contiguous and every
// buffer_load_dwordx4 is one fully coalesced 1-KiB request.  (afan_pack_weights builds it once per SGD step.)
//
// The B loads are inline asm (hipcc would wait vmcnt(0) for an ordinary VGPR load while LDS-DMA is in flight, draining the
// pipeline every K-step: cdna_hip_programming.md 5.7 item 1); their completion is counted by hand with the LDS-DMA ops on the
// same in-order vmcnt: per K-step and wave A_ROWS DMA ops then 4 B loads, issued in the order A(s) B(s) from the prologue on.
#include "afan_common.h"
#include "afan_conv_params.h"

using namespace afan;
using namespace afan_conv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int BN = 128;
constexpr int WM = 2, WN = 4;
constexpr int THREADS = 64 * WM * WN;
constexpr int RPP = THREADS / 8;

constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }

#define AFAN_BLOAD(dst, voff, rs, soff, imm)                                                                     \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:" #imm : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory")

template <int BM, int NSA, int NB>
__global__ __launch_bounds__(THREADS) void conv_igemm_breg_kernel(const ConvP pp, const uint16_t* __restrict__ wp) {
    static_assert(NSA == NB + 1, "A(s) and B(s) retire together only with one more LDS stage than register slots");
    constexpr int A_ROWS = BM / RPP;
    constexpr int TM = BM / WM, MI = TM / 32;
    constexpr int STAGE = BM * BK;                      // elements per A stage
    constexpr int LPT = A_ROWS + 4;                     // vm ops per thread and K-step
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    __shared__ int out_off[BM];

    const ConvClass& cc = pp.cls[0];
    const uint32_t Wg = (uint32_t)cc.Wg, Hg = (uint32_t)cc.Hg;
    const uint32_t M = (uint32_t)pp.N * Hg * Wg;
    const uint32_t m0 = blockIdx.y * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int n0 = blockIdx.x * BN;
    const int T = cc.T, Ci = pp.Ci, Hi = pp.Hi, Wi = pp.Wi;
    constexpr uint32_t OOB = 0x80000000u;

    const int piece = (tid & 7) ^ ((tid >> 4) & 7);
    const int row0 = tid >> 3;
    uint32_t a_off[A_ROWS], a_valid[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const uint32_t m = m0 + row0 + RPP * i;
        a_off[i] = 0;
        a_valid[i] = 0;
        if (m < M) {
            const uint32_t t1 = m / Wg, wg = m - t1 * Wg;
            const uint32_t n = t1 / Hg, hg = t1 - n * Hg;
            const int hi0 = (int)hg * pp.in_s, wi0 = (int)wg * pp.in_s;
            a_off[i] = (((n * Hi + hi0) * Wi + wi0) * Ci + piece * 8) * 2u;
            uint32_t v = 0;
#pragma unroll
            for (int t = 0; t < MAX_TAPS; ++t) {
                const int hi = hi0 + cc.dh[t], wi = wi0 + cc.dw[t];
                if (t < T && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi) v |= 1u << t;
            }
            a_valid[i] = v;
        }
    }
    for (int r = tid; r < BM; r += THREADS) {
        const uint32_t m = m0 + r;
        int off = -1;
        if (m < M) {
            const uint32_t t1 = m / Wg, wg = m - t1 * Wg;
            const uint32_t n = t1 / Hg, hg = t1 - n * Hg;
            off = (int)(((n * pp.Ho + hg * pp.out_s + cc.out_h0) * pp.Wo + wg * pp.out_s + cc.out_w0) * pp.Co);
        }
        out_off[r] = off;
    }

    f32x16 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int chunks = Ci / BK;
    const int KS = T * chunks;
    const uint32_t A_BIAS = (uint32_t)(((pp.max_pad * Wi + pp.max_pad) * Ci) * 2);
    const __amdgpu_buffer_rsrc_t xdma = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(const_cast<uint16_t*>(pp.x)) - A_BIAS, 0, (int)((int64_t)pp.N * Hi * Wi * Ci * 2 + A_BIAS), 0x00020000);

    // ---- A: LDS-DMA, K-steps issued strictly in order (tap state carried) ----
    int dma_t = 0, dma_q = 0;
    int dma_a = ((cc.dh[0] * Wi + cc.dw[0]) * Ci) * 2 + (int)A_BIAS;
    auto gdmaA = [&](int buf) {
        const int t = dma_t;
        const int a_tap = dma_a;
        uint16_t* A = lds + buf * STAGE + wave * 512;
        typedef __attribute__((address_space(3))) void* lptr;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const uint32_t voff = ((a_valid[i] >> t) & 1u) ? a_off[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xdma, (lptr)(A + i * (RPP * 64)), 16, (int)voff, a_tap, 0, 0);
        }
        if (++dma_q == chunks) {
            dma_q = 0;
            ++dma_t;
            if (dma_t < T) dma_a = ((cc.dh[dma_t] * Wi + cc.dw[dma_t]) * Ci) * 2 + (int)A_BIAS;
        } else {
            dma_a += BK * 2;
        }
    };

    // ---- B: packed weights straight into registers ----
    const int n_tile = blockIdx.x * (BN / 32) + wc;
    const int64_t wp_bytes = (int64_t)((pp.Co + 31) / 32) * KS * 4096;
    u32x4 wrs;                                                       // raw buffer descriptor in SGPRs
    {
        const uint64_t base = (uint64_t)reinterpret_cast<uintptr_t>(wp);
        wrs[0] = __builtin_amdgcn_readfirstlane((uint32_t)base);
        wrs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffffu);
        wrs[2] = __builtin_amdgcn_readfirstlane((uint32_t)wp_bytes);
        wrs[3] = __builtin_amdgcn_readfirstlane(0x00020000u);
    }
    const uint32_t b_voff = (uint32_t)lane * 16u;
    uint32_t b_soff = __builtin_amdgcn_readfirstlane((uint32_t)((int64_t)n_tile * KS * 4096));
    asm volatile("s_nop 4");      // SGPRs written by v_readfirstlane -> buffer_load descriptor / soffset
    u32x4 breg[NB][4];
    auto loadB = [&](int slot) {
        AFAN_BLOAD(breg[slot][0], b_voff, wrs, b_soff, 0);
        AFAN_BLOAD(breg[slot][1], b_voff, wrs, b_soff, 1024);
        AFAN_BLOAD(breg[slot][2], b_voff, wrs, b_soff, 2048);
        AFAN_BLOAD(breg[slot][3], b_voff, wrs, b_soff, 3072);
        b_soff += 4096;
    };

    auto compute = [&](int buf, int slot) {
        const uint16_t* A = lds + buf * STAGE;
        const int frow = lane & 31;
        const int sw = (frow >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 fx[MI];
            const int koff = ((kk * 2 + (lane >> 5)) ^ sw) * 8;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fx[i] = *reinterpret_cast<const bf16x8*>(A + (wr * TM + i * 32 + frow) * BK + koff);
            const bf16x8 fw = __builtin_bit_cast(bf16x8, breg[slot][kk]);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx[i], acc[i], 0, 0, 0);
        }
    };

    // prologue: A(s) B(s) for s = 0 .. NB-1 (A has one more stage than B has slots: A(NB) is issued at step 0)
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        if (s < KS) { gdmaA(s); loadB(s); }
    }
    int buf = 0;
    // main loop, unrolled by NB so that the register slots are static
    for (int ks0 = 0; ks0 < KS; ks0 += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int ks = ks0 + j;
            if (ks < KS) {
                const int rem = KS - 1 - ks;                    // K-steps after this one; min(rem, NB - 1) of them are in flight
#define AFAN_BWAIT(N_) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(breg[j][0]), "+v"(breg[j][1]), "+v"(breg[j][2]), "+v"(breg[j][3]) : "i"(N_))
                if (rem >= NB - 1) AFAN_BWAIT(LPT * (NB - 1));
                else if (NB >= 4 && rem == 2) AFAN_BWAIT(LPT * 2);
                else if (NB >= 3 && rem == 1) AFAN_BWAIT(LPT);
                else AFAN_BWAIT(0);
#undef AFAN_BWAIT
                __builtin_amdgcn_s_barrier();                    // A tile ks is in LDS for everyone; the stage of tile ks-1 is free
                if (ks + NB < KS) gdmaA(buf == 0 ? NSA - 1 : buf - 1);
                compute(buf, j);
                if (ks + NB < KS) loadB(j);                      // refill the slot just consumed
                buf = buf + 1 == NSA ? 0 : buf + 1;
            }
        }
    }
    __syncthreads();

    // ---- epilogue: accumulators -> bf16 tile in LDS -> 16-byte channels-last stores (no fusions in this variant) ----
    constexpr int LDC = BN + 8;
    uint16_t* C = lds;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int pix = wr * TM + i * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = wc * 32 + 8 * g + 4 * (lane >> 5);
            u16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = f2bf(acc[i][4 * g + e]);
            *reinterpret_cast<u16x4*>(C + pix * LDC + ch) = v;
        }
    }
    __syncthreads();
    constexpr int PIECES = BN / 8, ROWS_PER_PASS = THREADS / PIECES, EPI_ROWS = BM / ROWS_PER_PASS;
    const int pc = tid % PIECES, pr = tid / PIECES;
    const bool ch_ok = n0 + pc * 8 < pp.Co;
#pragma unroll
    for (int q = 0; q < EPI_ROWS; ++q) {
        const int r = pr + q * ROWS_PER_PASS;
        const int off = out_off[r];
        if (off >= 0 && ch_ok) *reinterpret_cast<u16x8*>(pp.y + (int64_t)off + n0 + pc * 8) = *reinterpret_cast<const u16x8*>(C + r * LDC + pc * 8);
    }
}

// packed[n_tile][ks][kk][lane][8] from w[rows][taps][red] (row stride = taps * red elements); rows / red padded with zeros
__global__ __launch_bounds__(256) void pack_weights_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ out, int rows, int taps,
                                                           int red, int chunks, int64_t total_vec) {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total_vec; v += (int64_t)gridDim.x * 256) {
        const int lane = (int)(v & 63);
        const int kk = (int)((v >> 6) & 3);
        const int64_t rest = v >> 8;                                     // n_tile * KS + ks
        const int KS = taps * chunks;
        const int ks = (int)(rest % KS), nt = (int)(rest / KS);
        const int tap = ks / chunks, q = ks - tap * chunks;
        const int row = nt * 32 + (lane & 31), k = q * 64 + kk * 16 + (lane >> 5) * 8;
        u16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < rows && k < red) val = *reinterpret_cast<const u16x8*>(w + ((int64_t)row * taps + tap) * red + k);
        *reinterpret_cast<u16x8*>(out + v * 8) = val;
    }
}

}  // namespace

extern "C" {

// elements (bf16) of the packed copy of a [rows][taps][red] weight tensor
int64_t afan_pack_weights_elems(int64_t rows, int64_t taps, int64_t red) {
    if (rows <= 0 || taps <= 0 || red <= 0) return 0;
    return ((rows + 31) / 32) * taps * ((red + 63) / 64) * 2048;
}

int afan_pack_weights(const void* w, void* packed, int64_t rows, int64_t taps, int64_t red, afan_stream_t stream) {
    if (rows <= 0 || taps <= 0 || red <= 0 || red % 8) return AFAN_ESHAPE;
    if (!w || !packed) return AFAN_ENULL;
    if (!aligned(w, 16) || !aligned(packed, 16)) return AFAN_EALIGN;
    const int chunks = (int)((red + 63) / 64);
    const int64_t total_vec = afan_pack_weights_elems(rows, taps, red) / 8;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("pack_weights_kernel", 4.0 * total_vec * 8, st);
    pack_weights_kernel<<<grid_for(total_vec, 256, 4096), 256, 0, st>>>((const uint16_t*)w, (uint16_t*)packed, (int)rows, (int)taps, (int)red,
                                                                        chunks, total_vec);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// EXPERIMENTAL entry (tools/conv_breg_bench.py): stride-1 forward, k in {1,3}, Ci % 64 == 0, Co % 128 == 0, weights packed.
// variant: 0 = 128-row tiles / 4 LDS stages / 3 register slots, 1 = 64-row tiles, 2 = 128 rows / 3 stages / 2 slots
int afan_conv_fwd_breg_exp(const void* x, const void* wp, void* y, int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k,
                           int variant, afan_stream_t stream) {
    if (n <= 0 || hi <= 0 || wi <= 0 || ci % 64 || co % 128 || !(k == 1 || k == 3)) return AFAN_ESHAPE;
    if (!x || !wp || !y) return AFAN_ENULL;
    const int pad = k / 2;
    ConvP p{};
    p.x = (const uint16_t*)x; p.w = nullptr; p.y = (uint16_t*)y;
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci; p.Ho = (int)hi; p.Wo = (int)wi; p.Co = (int)co;
    p.in_s = 1; p.out_s = 1; p.n_classes = 1; p.max_pad = 1;
    ConvClass& c0 = p.cls[0];
    c0.Hg = p.Ho; c0.Wg = p.Wo; c0.out_h0 = 0; c0.out_w0 = 0; c0.T = k * k;
    for (int r = 0; r < k; ++r)
        for (int s = 0; s < k; ++s) {
            const int t = r * k + s;
            c0.dh[t] = r - pad; c0.dw[t] = s - pad; c0.wofs[t] = 0;
        }
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = n * hi * wi;
#define BREG_LAUNCH(BM_, NSA_, NB_)                                                                                             \
    do {                                                                                                                          \
        constexpr size_t stage_bytes = (size_t)NSA_ * BM_ * BK * 2, epi = (size_t)BM_ * (BN + 8) * 2;                              \
        constexpr size_t lds = stage_bytes > epi ? stage_bytes : epi;                                                             \
        static bool done = false;                                                                                                 \
        if (!done && lds > 64 * 1024) {                                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_breg_kernel<BM_, NSA_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                                   \
            done = true;                                                                                                          \
        }                                                                                                                         \
        dim3 grid((unsigned)(co / BN), (unsigned)((M + BM_ - 1) / BM_), 1);                                                       \
        conv_igemm_breg_kernel<BM_, NSA_, NB_><<<grid, THREADS, lds, st>>>(p, (const uint16_t*)wp);                                \
    } while (0)
    if (variant == 1) BREG_LAUNCH(64, 4, 3);
    else if (variant == 2) BREG_LAUNCH(128, 3, 2);
    else if (variant == 3) BREG_LAUNCH(128, 5, 4);
    else BREG_LAUNCH(128, 4, 3);
#undef BREG_LAUNCH
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
