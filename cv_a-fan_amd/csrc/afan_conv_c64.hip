// 3x3, stride 1, 64 -> 64 channels on small images: ResNet-18's layer1 inside the PGD tail (4 convolutions, each run
// forward and as an input gradient K+2 times per A-FAN iteration: Classification/attack_algo.py:50-52 through
// resnet_s.py:52-54).  In the generic implicit-GEMM kernel (afan_conv.hip) these layers are bound by L2 -> LDS
// traffic, not by the matrix cores: N = 64 output channels is a narrow GEMM, every one of the 9 taps re-fetches the
// activation tile and every workgroup re-fetches the weights.  This kernel removes both re-fetches:
//   * weights live in REGISTERS for the life of a persistent workgroup: a wave owns all 64 output channels and all
//     9 taps x 64 input channels of them = 72 MFMA operand fragments = 288 VGPRs (one wave per SIMD, 512-entry file);
//   * the activation tile is fetched ONCE with its halo: 128 output pixels = 128/W full image rows, plus one row /
//     column of padding each side, lands in LDS by LDS-DMA (zero page for the padding) and all 9 taps read their
//     shifted fragments from it; the next tile's halo is in flight while this one is multiplied;
//   * an activation fragment feeds two MFMAs (both channel halves), so LDS reads run at half the MFMA operand rate.
// One workgroup (4 waves) per CU walks tiles b, b + G, ...; per tile and wave: 36 ds_read_b128 + 72 MFMA 32x32x16.
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
constexpr int TP = 128;                // output pixels per tile
constexpr int THREADS = 256;           // 4 waves, one per SIMD
constexpr int HALO_ALLOC = 224;        // halo pixels per buffer, rounded up to whole DMA rounds (32 pixels each)
constexpr int ROUNDS = HALO_ALLOC / 32;
constexpr int LDC = CH + 8;            // epilogue staging row (elements)
constexpr int NBUF = 3;                // halo buffers: tile i is multiplied while i+1 has landed and i+2 is in flight
constexpr size_t LDS_BYTES = (size_t)NBUF * HALO_ALLOC * CH * 2 + (size_t)2 * TP * LDC * 2;

__device__ const uint4 c64_zero_page[4] = {};

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

__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv3x3_c64_kernel(const Params p, int tiles, int logW, uint64_t* stamps) {
#ifdef AFAN_C64_STAMPS
    uint64_t stamp_acc[6] = {0, 0, 0, 0, 0, 0};
    uint64_t stamp_last = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* const halo0 = lds;
    uint16_t* const Cst0 = lds + NBUF * HALO_ALLOC * CH;   // two staging tiles (epilogue of tile i overlaps tile i+1)
    __shared__ float red[THREADS / 64][2][CH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int W = p.W, H = p.H, W2 = W + 2, TR = TP >> logW;
    const int HP = (TR + 2) * W2;                       // halo pixels actually used
    const int tiles_per_img = H / TR;

    // ---- static part of the halo fetch: which pixel / 16-byte piece this thread brings in each round ---------------
    // LDS position pos = round * 256 + tid  ->  halo pixel pos / 8, slot pos % 8; the slot holds logical piece
    // slot ^ ((pixel >> 1) & 7): the XOR swizzle that keeps the ds_read_b128 fragments below conflict-free on
    // unpadded 128-byte rows (the DMA writes 1 KiB of consecutive LDS per wave instruction, so padding is not an option)
    int rel[ROUNDS], hrow[ROUNDS];
    bool ok[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int pos = i * THREADS + tid, hp = pos >> 3, slot = pos & 7;
        const int piece = slot ^ ((hp >> 1) & 7);
        const int hr = hp / W2, wc = hp - hr * W2;
        hrow[i] = hr - 1;
        ok[i] = hp < HP && wc >= 1 && wc <= W;
        rel[i] = (((hr - 1) * W + (wc - 1)) * CH + piece * 8) * 2;     // bytes from the tile's first pixel
    }
    typedef __attribute__((address_space(1))) const void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    auto fetch = [&](int tile, int buf) {
        const int n = tile / tiles_per_img, h0 = (tile - n * tiles_per_img) * TR;
        const char* base = reinterpret_cast<const char*>(p.x) + ((int64_t)(n * H + h0) * W) * (CH * 2);
        uint16_t* dst = halo0 + buf * (HALO_ALLOC * CH) + wave * 512;     // wave-uniform; hardware adds lane * 16 B
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const bool v = ok[i] && (unsigned)(h0 + hrow[i]) < (unsigned)H;
            const char* src = v ? base + rel[i] : reinterpret_cast<const char*>(c64_zero_page);
            __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(dst + i * (THREADS * 8)), 16, 0, 0);
        }
    };

    // Schedule: ONE barrier per tile.  After the barrier of tile i (all waves are done multiplying it), the halo of
    // tile i+2 is requested into the buffer tile i-1 used, then tile i's outputs are stored; the barrier of tile i+1 then
    // only waits for memory operations that were issued a whole multiply phase earlier.
    int tile = blockIdx.x;
    const int G = gridDim.x;
    if (tile < tiles) fetch(tile, 0);

    // ---- weights -> registers, through LDS: fragment (t, kk, j) = rows j*32 + lane%32 (output channel), k = kk*16 +
    // half*8 .. +8.  Read straight from global memory a fragment is 64 lanes x 16 B on 64 different cache lines and all
    // four waves fetch the same 72 fragments (measured: 25 k cycles, a third of the kernel).  Instead the 72 KiB tensor
    // is copied once with linear 16-byte loads into LDS — over the halo buffers 1, 2 and the staging tiles, none of
    // which is live yet — with rows padded to 73 slots of 16 B, so the 16 lanes of a ds_read_b128 group land on 16
    // different slots, and every wave picks its fragments from there.
    bf16x8 wreg[9][4][2];
    {
        constexpr int WROW = 9 * CH + 8;                 // padded weight row in LDS (elements): 1168 B
        uint16_t* wl = halo0 + HALO_ALLOC * CH;          // 64 * 1168 B = 73 KiB <= 2 halo buffers + 2 staging tiles
        static_assert((size_t)CH * WROW * 2 <= (size_t)(NBUF - 1) * HALO_ALLOC * CH * 2 + (size_t)2 * TP * LDC * 2, "alias");
        constexpr int PIECES = CH * 9 * CH / 8;          // 4608 16-byte pieces, 18 per thread
        u16x8 tmp[PIECES / THREADS];
#pragma unroll
        for (int i = 0; i < PIECES / THREADS; ++i)
            tmp[i] = *reinterpret_cast<const u16x8*>(p.w + (int64_t)(i * THREADS + tid) * 8);
#pragma unroll
        for (int i = 0; i < PIECES / THREADS; ++i) {
            const int q = i * THREADS + tid, row = q / 72, col = q - row * 72;
            *reinterpret_cast<u16x8*>(wl + row * WROW + col * 8) = tmp[i];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    wreg[t][kk][j] = *reinterpret_cast<const bf16x8*>(wl + (j * 32 + (lane & 31)) * WROW + t * CH +
                                                                      kk * 16 + half * 8);
        __syncthreads();                                 // every wave holds its fragments: the region is free again
    }
    if (tile + G < tiles) fetch(tile + G, 1);

    // this lane's output pixel inside the tile and its centre position in the halo
    const int pix = wave * 32 + (lane & 31);
    const int hpc = ((pix >> logW) + 1) * W2 + (pix & (W - 1)) + 1;
    const int sgn = p.flip ? -1 : 1;

    // epilogue roles
    const int pc = tid & 7, pr = tid >> 3;
    const bool want_stats = p.acc != nullptr;
    const bool bn_bwd = want_stats && p.bnx != nullptr;
    float s1[8], s2[8], sh[8], al[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = s2[j] = 0.f;
        const int c = pc * 8 + j;
        sh[j] = bn_bwd ? p.bn_stats[c] : ((want_stats && p.shift) ? p.shift[c] : 0.f);
        al[j] = bn_bwd ? p.bn_stats[2 * CH + c] : 0.f;
        be[j] = bn_bwd ? p.bn_stats[3 * CH + c] : 0.f;
    }
    __syncthreads();   // first halo landed (vmcnt drained before the barrier)
    C64_STAMP(0);

    int buf = 0;
    for (int it = 0; tile < tiles; ++it, tile += G) {
        const uint16_t* halo = halo0 + buf * (HALO_ALLOC * CH);
        uint16_t* Cst = Cst0 + (it & 1) * (TP * LDC);
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        // tap t+1's four activation fragments are requested before tap t's eight MFMAs (~256 cycles) are issued
        bf16x8 fx[2][4];
        auto load_tap = [&](int t, bf16x8 (&f)[4]) {
            const int hp = hpc + sgn * ((t / 3 - 1) * W2 + (t % 3 - 1));
            const uint16_t* row = halo + hp * CH;
            const int sw = (hp >> 1) & 7;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) f[kk] = *reinterpret_cast<const bf16x8*>(row + ((kk * 2 + half) ^ sw) * 8);
        };
        load_tap(0, fx[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) load_tap(t + 1, fx[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);           // keep the reads ahead of this tap's MFMAs (the scheduler sinks them)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[t][kk][0], fx[t & 1][kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[t][kk][1], fx[t & 1][kk], acc1, 0, 0, 0);
            }
        }
        C64_STAMP(1);
        // fp32 accumulators -> bf16 tile [pixel][channel]: a lane holds 4 consecutive channels of its pixel per quad
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u16x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = f2bf(acc0[4 * g + e]);
                v1[e] = f2bf(acc1[4 * g + e]);
            }
            *reinterpret_cast<u16x4*>(Cst + pix * LDC + 8 * g + 4 * half) = v0;
            *reinterpret_cast<u16x4*>(Cst + pix * LDC + 32 + 8 * g + 4 * half) = v1;
        }
        __syncthreads();
        C64_STAMP(2);
        {
            const int nb = buf == 0 ? NBUF - 1 : buf - 1;          // the buffer tile it-1 was multiplied from
            if (tile + 2 * G < tiles) fetch(tile + 2 * G, nb);
            buf = buf + 1 == NBUF ? 0 : buf + 1;
        }
        C64_STAMP(3);
        // the tile's 128 output pixels are 128/W full image rows: one contiguous 16 KiB span of y
        const int n = tile / tiles_per_img, h0 = (tile - n * tiles_per_img) * TR;
        const int64_t out0 = ((int64_t)(n * H + h0) * W) * CH + pc * 8;
#pragma unroll
        for (int q = 0; q < TP / 32; ++q) {
            const int r = pr + 32 * q;
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
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xf = bf2f(xv[j]);
                    float g = bf2f(v[j]);
                    if (p.bn_relu) g = (fmaf(xf, al[j], be[j]) > 0.f) ? g : 0.f;
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
        C64_STAMP(4);
    }

    if (want_stats) {
        // lanes l, l+8, l+16, ... of a wave hold the same 8 channels: butterfly over those, then over the 4 waves
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
        if (lane < 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[wave][0][lane * 8 + j] = s1[j];
                red[wave][1][lane * 8 + j] = s2[j];
            }
        }
        __syncthreads();
        if (tid < CH) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) {
                a += red[w][0][tid];
                b += red[w][1][tid];
            }
            double* dst = p.acc + (int64_t)(blockIdx.x & (p.acc_ns - 1)) * 2 * CH;
            unsafeAtomicAdd(dst + tid, (double)a);
            unsafeAtomicAdd(dst + CH + tid, (double)b);
            if (!bn_bwd && blockIdx.x == 0)
                reinterpret_cast<float*>(p.acc + (int64_t)2 * p.acc_ns * CH)[tid] = p.shift ? p.shift[tid] : 0.f;
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
    const int grid = tiles < cus ? tiles : cus;
#ifdef AFAN_C64_STAMPS
    static uint64_t* stamps = nullptr;
    if (!stamps && hipMalloc(&stamps, 6 * 1024 * sizeof(uint64_t)) != hipSuccess) return AFAN_ESHAPE;
    conv3x3_c64_kernel<<<grid, THREADS, LDS_BYTES, st>>>(p, tiles, logW, stamps);
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
                    (double)tiles / grid, tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]);
        }
    }
#else
    conv3x3_c64_kernel<<<grid, THREADS, LDS_BYTES, st>>>(p, tiles, logW, nullptr);
#endif
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // namespace afan_c64
