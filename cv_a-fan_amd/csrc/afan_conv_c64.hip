// 3x3, stride 1, 64 -> 64 channels on small images: ResNet-18's layer1 inside the PGD tail (4 convolutions, each run
// forward and as an input gradient K+2 times per A-FAN iteration: Classification/attack_algo.py:50-52 through
// resnet_s.py:52-54).  In the generic implicit-GEMM kernel (afan_conv.hip) these layers are bound by L2 -> LDS
// traffic, not by the matrix cores: N = 64 output channels is a narrow GEMM, every one of the 9 taps re-fetches the
// activation tile and every workgroup re-fetches the weights.  This kernel removes both re-fetches:
//   * weights live in REGISTERS for the life of a persistent workgroup: a wave owns 32 output channels and all
//     9 taps x 64 input channels of them = 36 MFMA operand fragments = 144 VGPRs; two workgroups (the two channel
//     halves) share a CU, two waves per SIMD, 256 registers each;
//   * the activation tile is fetched ONCE with its halo: 128 output pixels = 128/W full image rows, plus one row /
//     column of padding each side, lands in LDS by LDS-DMA (zero page for the padding) and all 9 taps read their
//     shifted fragments from it; the next tile's halo is in flight while this one is multiplied;
//   * the two workgroups of a CU drift out of step, so one's epilogue (bf16 staging, 16-byte stores, BatchNorm sums),
//     barriers and halo requests sit under the other's MFMAs — with a single 4-wave workgroup per CU (first version:
//     all 64 channels per wave, 288 weight registers) those phases were serial and took as long as the MFMAs.
// Workgroup pair b walks tiles b, b + G, ...; per tile and wave: 36 ds_read_b128 + 36 MFMA 32x32x16.
// The input gradient of such a layer is the same convolution with mirrored taps and CRSK weights (flip = 1).
// Epilogue fusions are those of afan_conv.hip (residual-gradient addend, BatchNorm moments, BatchNorm-backward sums);
// the per-channel sums stay in registers across the workgroup's tiles and reach the f64 accumulators once.
#include "afan_conv_c64.h"
#include <stdio.h>
#include <stdlib.h>

using namespace afan;

namespace afan_c64 {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CH = 64;                 // channels in and out
constexpr int COH = 32;                // output channels per workgroup (two workgroups share a tile's pixels)
constexpr int TP = 128;                // output pixels per tile
constexpr int THREADS = 256;           // 4 waves, one per SIMD; two workgroups per CU
constexpr int HALO_ALLOC = 208;        // halo pixels per buffer (>= 204 used), whole 1 KiB wave pieces (8 pixels each)
constexpr int ROUNDS = (HALO_ALLOC + 31) / 32;   // DMA rounds of the workgroup (the last one: two waves only)
constexpr int RTAPS = 7;               // taps whose weights stay in registers; the other two are read from LDS every tile
constexpr int LDC = COH + 8;           // epilogue staging row (elements)
constexpr int NBUF = 2;                // halo buffers: tile i is multiplied while tile i+1 lands
constexpr size_t LDS_BYTES = (size_t)NBUF * HALO_ALLOC * CH * 2 + (size_t)TP * LDC * 2;

// Diagnostic build only (make EXTRA=-DAFAN_C64_STAMPS): per-phase s_memtime totals of wave 0 of every workgroup, printed
// by launch().  No stamp exists in the product build.
#ifdef AFAN_C64_STAMPS
#define C64_STAMP(slot)                                                          \
    do {                                                                         \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();                      \
        stamp_acc[slot] += now_ - stamp_last;                                    \
        stamp_last = now_;                                                       \
    } while (0)
#else
#define C64_STAMP(slot) do { } while (0)
#endif

__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_c64_kernel(const Params p, int tiles, int logW, uint64_t* stamps, int xcd_pair) {
#ifdef AFAN_C64_STAMPS
    uint64_t stamp_acc[6] = {0, 0, 0, 0, 0, 0};
    uint64_t stamp_last = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* const halo0 = lds;
    uint16_t* const Cst = lds + NBUF * HALO_ALLOC * CH;
    __shared__ float red[THREADS / 64][2][COH];
    __shared__ __attribute__((aligned(16))) uint16_t w8[COH][(9 - RTAPS) * CH + 8];   // weights of the LDS-resident taps (padded rows)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    // XCD-aware pairing: the two workgroups of a tile (channel halves) read the same halo; dispatch slots go to the XCDs
    // round-robin, so slots are mapped to logical ids that are contiguous per XCD — a pair shares one XCD's L2
    uint32_t bid = blockIdx.x;
    if (xcd_pair) {
        const uint32_t per = gridDim.x >> 3;
        if (bid < (per << 3)) bid = (bid & 7u) * per + (bid >> 3);
    }
    const int coh = bid & 1;                             // which 32 output channels
    const int W = p.W, H = p.H, W2 = W + 2, TR = TP >> logW;
    const int HP = (TR + 2) * W2;                       // halo pixels actually used
    const int tiles_per_img = H / TR;

    // ---- static part of the halo fetch: which pixel / 16-byte piece this thread brings in each round ---------------
    // LDS position pos = round * 256 + tid  ->  halo pixel pos / 8, slot pos % 8; the slot holds logical piece
    // slot ^ ((pixel >> 1) & 7): the XOR swizzle that keeps the ds_read_b128 fragments below conflict-free on
    // unpadded 128-byte rows (the DMA writes 1 KiB of consecutive LDS per wave instruction, so padding is not an option)
    // (registers are the scarce resource here: the image-row offset of the piece rides in the low byte of `rel`, whose
    // own low 4 bits are zero, and a piece that is padding in every tile gets row offset 127)
    // ... so scarce that the table itself lives in LDS (7 dwords per thread), read back with ds_read: a register copy
    // gets spilled to scratch, and a scratch reload between two DMA requests waits for the first one (same counter)
    __shared__ int rel_tab[ROUNDS][THREADS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int pos = i * THREADS + tid, hp = pos >> 3, slot = pos & 7;
        const int piece = slot ^ ((hp >> 1) & 7);
        const int hr = hp / W2, wc = hp - hr * W2;
        const bool ok = hp < HP && wc >= 1 && wc <= W;
        const int bytes = (((hr - 1) * W + (wc - 1)) * CH + piece * 8) * 2;   // from the tile's first pixel; multiple of 16
        rel_tab[i][tid] = (bytes << 4) | (ok ? hr : 127);                     // hr = image row offset + 1, in [0, 33]
    }
    // buffer-addressed LDS-DMA: one 32-bit offset per piece (no 64-bit pointers held across the loop), and padding
    // pieces take an out-of-range offset that the hardware turns into zeros
    typedef __attribute__((address_space(3))) void* lptr;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x), 0, (int)((int64_t)p.N * H * W * CH * 2), 0x00020000);
    auto fetch = [&](int tile, int buf) {
        const int n = tile / tiles_per_img, h0 = (tile - n * tiles_per_img) * TR;
        const int base = ((n * H + h0) * W) * (CH * 2);                    // bytes; < 2^31 (eligible())
        uint16_t* dst = halo0 + buf * (HALO_ALLOC * CH) + wave * 512;     // wave-uniform; hardware adds lane * 16 B
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            if ((i * THREADS + wave * 64) >= HALO_ALLOC * 8) break;       // wave-uniform: beyond the buffer
            const int rel = rel_tab[i][tid];
            const bool v = (unsigned)(h0 + (rel & 255) - 1) < (unsigned)H;
            const int off = v ? base + (rel >> 8 << 4) : (int)0x80000000;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr)(dst + i * (THREADS * 8)), 16, off, 0, 0, 0);
        }
    };

    // Two workgroups per CU (one per channel half) run out of step, so one's stores, barriers and halo fetches sit
    // under the other's MFMAs; inside a workgroup the schedule is the plain one: fetch next, multiply, stage, store.
    int tile = bid >> 1;
    const int G = gridDim.x >> 1;
    if (tile < tiles) fetch(tile, 0);

    // ---- weights -> registers, through LDS: fragment (t, kk) = rows lane%32 (output channel), k = kk*16 + half*8 .. +8.
    // Read straight from global memory a fragment is 64 lanes x 16 B on 64 different cache lines and all four waves
    // fetch the same fragments (measured: a third of the kernel).  Instead this half's 36 KiB are copied once with
    // linear 16-byte loads into LDS — over halo buffer 1 and the staging tile, neither of which is live yet — with
    // rows padded to 73 slots of 16 B, so the 16 lanes of a ds_read_b128 group land on 16 different slots.
    bf16x8 wreg[RTAPS][4];
    {
        constexpr int WROW = RTAPS * CH + 8;             // padded row of the register-resident taps (elements): 65 slots
        uint16_t* wl = halo0 + HALO_ALLOC * CH;
        static_assert((size_t)COH * WROW * 2 <= (size_t)(NBUF - 1) * HALO_ALLOC * CH * 2 + (size_t)TP * LDC * 2, "alias");
        constexpr int PIECES = COH * 9 * CH / 8;         // 2304 16-byte pieces, 9 per thread
        const uint16_t* wsrc = p.w + (int64_t)coh * COH * 9 * CH;
        u16x8 tmp[PIECES / THREADS];
#pragma unroll
        for (int i = 0; i < PIECES / THREADS; ++i)
            tmp[i] = *reinterpret_cast<const u16x8*>(wsrc + (int64_t)(i * THREADS + tid) * 8);
#pragma unroll
        for (int i = 0; i < PIECES / THREADS; ++i) {
            const int q = i * THREADS + tid, row = q / 72, col = q - row * 72;
            uint16_t* d = col < RTAPS * 8 ? wl + row * WROW + col * 8 : &w8[row][(col - RTAPS * 8) * 8];
            *reinterpret_cast<u16x8*>(d) = tmp[i];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RTAPS; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                wreg[t][kk] = *reinterpret_cast<const bf16x8*>(wl + (lane & 31) * WROW + t * CH + kk * 16 + half * 8);
        __syncthreads();                                 // every wave holds its fragments: the region is free again
    }

    // this lane's output pixel inside the tile and its centre position in the halo
    const int pix = wave * 32 + (lane & 31);
    const int hpc = ((pix >> logW) + 1) * W2 + (pix & (W - 1)) + 1;
    const int sgn = p.flip ? -1 : 1;

    // epilogue roles: 4 pieces of 8 channels per output row, 64 rows per pass
    const int pc = tid & 3, pr = tid >> 2;
    const bool want_stats = p.acc != nullptr;
    const bool bn_bwd = want_stats && p.bnx != nullptr;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    const int c0 = coh * COH + pc * 8;   // this thread's 8 channels in the epilogue
    C64_STAMP(0);

    for (int it = 0; tile < tiles; ++it, tile += G) {
        const int buf = it & 1;
        if (tile + G < tiles) fetch(tile + G, buf ^ 1);   // lands under this tile's MFMAs
        C64_STAMP(3);
        const uint16_t* halo = halo0 + buf * (HALO_ALLOC * CH);
        f32x16 acc0;   // one chain: the other wave of this SIMD issues into the gaps a dependent MFMA leaves
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
        // (no explicit read-ahead: the registers it needs are not there, and the other wave on this SIMD covers the
        // LDS latency of this one)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hp = hpc + sgn * ((t / 3 - 1) * W2 + (t % 3 - 1));
            const uint16_t* row = halo + hp * CH;
            const int sw = (hp >> 1) & 7;
            bf16x8 fx[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fx[kk] = *reinterpret_cast<const bf16x8*>(row + ((kk * 2 + half) ^ sw) * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 fw = t < RTAPS ? wreg[t < RTAPS ? t : 0][kk]
                                            : *reinterpret_cast<const bf16x8*>(&w8[lane & 31][(t - RTAPS) * CH + kk * 16 + half * 8]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx[kk], acc0, 0, 0, 0);
            }
        }
        C64_STAMP(1);
        // fp32 accumulators -> bf16 tile [pixel][channel]: a lane holds 4 consecutive channels of its pixel per quad
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u16x4 v0;
#pragma unroll
            for (int e = 0; e < 4; ++e) v0[e] = f2bf(acc0[4 * g + e]);
            *reinterpret_cast<u16x4*>(Cst + pix * LDC + 8 * g + 4 * half) = v0;
        }
        __syncthreads();   // staging tile visible; the next halo has landed
        C64_STAMP(2);
        // the tile's 128 output pixels are 128/W full image rows: one contiguous span of y
        const int n = tile / tiles_per_img, h0 = (tile - n * tiles_per_img) * TR;
        const int64_t out0 = ((int64_t)(n * H + h0) * W) * CH + coh * COH + pc * 8;
        // per-channel coefficients of this thread's 8 channels: fetched per tile as 16-byte loads (cache hits) so that
        // they do not occupy registers across the MFMA phase
        float sh[8], al[8], be[8];
        if (want_stats) {
            const float* src = bn_bwd ? p.bn_stats : p.shift;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a = src ? *reinterpret_cast<const f32x4*>(src + c0 + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
                sh[4 * h] = a.x; sh[4 * h + 1] = a.y; sh[4 * h + 2] = a.z; sh[4 * h + 3] = a.w;
                if (bn_bwd && p.bn_relu) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bn_stats + 2 * CH + c0 + 4 * h);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.bn_stats + 3 * CH + c0 + 4 * h);
                    al[4 * h] = b.x; al[4 * h + 1] = b.y; al[4 * h + 2] = b.z; al[4 * h + 3] = b.w;
                    be[4 * h] = c.x; be[4 * h + 1] = c.y; be[4 * h + 2] = c.z; be[4 * h + 3] = c.w;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < TP / 64; ++q) {
            const int r = pr + 64 * q;
            u16x8 v = *reinterpret_cast<const u16x8*>(Cst + r * LDC + pc * 8);
            const int64_t go = out0 + (int64_t)r * CH;
            if (p.addend) {
                const u16x8 a = *reinterpret_cast<const u16x8*>(p.addend + go);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) + bf2f(a[j]));
            }
            *reinterpret_cast<u16x8*>(p.y + go) = v;
            if (bn_bwd) {
                const u16x8 xv = *reinterpret_cast<const u16x8*>(p.bnx + go);
                u16x8 yv = xv;
                if (p.bny) yv = *reinterpret_cast<const u16x8*>(p.bny + go);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xf = bf2f(xv[j]);
                    float g = bf2f(v[j]);
                    if (p.bny) g = (bf2f(yv[j]) > 0.f) ? g : 0.f;
                    else if (p.bn_relu) g = (fmaf(xf, al[j], be[j]) > 0.f) ? g : 0.f;
                    s1[j] += g;
                    s2[j] += g * (xf - sh[j]);
                }
            } else if (want_stats) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f(v[j]) - sh[j];
                    s1[j] += f;
                    s2[j] += f * f;
                }
            }
        }
        __syncthreads();   // staging tile and this halo buffer are free again
        C64_STAMP(4);
    }

    if (want_stats) {
        // lanes l, l+4, l+8, ... of a wave hold the same 8 channels: butterfly over those, then over the 4 waves
#pragma unroll
        for (int o = 4; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
        if (lane < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[wave][0][lane * 8 + j] = s1[j];
                red[wave][1][lane * 8 + j] = s2[j];
            }
        }
        __syncthreads();
        if (tid < COH) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) {
                a += red[w][0][tid];
                b += red[w][1][tid];
            }
            const int c = coh * COH + tid;
            double* dst = p.acc + (int64_t)((bid >> 1) & (p.acc_ns - 1)) * 2 * CH;
            unsafeAtomicAdd(dst + c, (double)a);
            unsafeAtomicAdd(dst + CH + c, (double)b);
            if (!bn_bwd && (bid >> 1) == 0)
                reinterpret_cast<float*>(p.acc + (int64_t)2 * p.acc_ns * CH)[c] = p.shift ? p.shift[c] : 0.f;
        }
    }
#ifdef AFAN_C64_STAMPS
    C64_STAMP(5);
    if (stamps && tid == 0)
        for (int i = 0; i < 6; ++i) stamps[blockIdx.x * 6 + i] = stamp_acc[i];
#endif
}

}  // namespace

bool eligible(int64_t n, int64_t h, int64_t w, int64_t ci, int64_t co, int k, int stride) {
    static const bool on = [] { const char* v = getenv("AFAN_CONV_C64"); return !v || atoi(v) != 0; }();
    if (!on || ci != CH || co != CH || k != 3 || stride != 1) return false;
    if (w < 4 || w > 32 || (w & (w - 1)) != 0) return false;
    const int64_t tr = TP / w;
    return h % tr == 0 && n * h * w * CH * 2 <= 0x7fffffffLL;
}

int launch(const Params& p, hipStream_t st) {
    int logW = 0;
    while ((1 << logW) < p.W) ++logW;
    const int tiles = (int)((int64_t)p.N * p.H * p.W / TP);
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return AFAN_ESHAPE;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)LDS_BYTES);
        if (e != hipSuccess) { cus = 0; return (int)e; }
    }
    const int grid = 2 * (tiles < cus ? tiles : cus);   // a pair of workgroups (channel halves) per tile column
#ifdef AFAN_C64_STAMPS
    static uint64_t* stamps = nullptr;
    if (!stamps && hipMalloc(&stamps, 6 * 1024 * sizeof(uint64_t)) != hipSuccess) return AFAN_ESHAPE;
    conv3x3_c64_kernel<<<grid, THREADS, LDS_BYTES, st>>>(p, tiles, logW, stamps, 1);
    {
        static int calls = 0;
        if (++calls % 50 == 0) {
            uint64_t h[6 * 1024];
            hipStreamSynchronize(st);
            hipMemcpy(h, stamps, sizeof(uint64_t) * 6 * grid, hipMemcpyDeviceToHost);
            double tot[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < grid; ++b)
                for (int i = 0; i < 6; ++i) tot[i] += (double)h[b * 6 + i] / grid;
            fprintf(stderr, "[c64 stamps] tiles/wg %.1f  prologue %.0f  mfma %.0f  stage+barrier %.0f  fetch %.0f  epilogue %.0f  tail %.0f cycles\n",
                    (double)tiles / (grid / 2), tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]);
        }
    }
#else
    static const int xcd_pair = [] { const char* v = getenv("AFAN_C64_XCD"); return v ? atoi(v) : 1; }();
    conv3x3_c64_kernel<<<grid, THREADS, LDS_BYTES, st>>>(p, tiles, logW, nullptr, xcd_pair);
#endif
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // namespace afan_c64
