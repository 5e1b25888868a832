// The image stem: 3x3 / stride 1 convolution from the 3 image channels (resnet_s.py:88 `conv1 = nn.Conv2d(3, 16|64, 3, 1, 1)`),
// forward and weight gradient (the images carry no gradient).  With these two the training step holds no vendor kernel:
// the captured hipGraph is self-contained and `--seed` (cudnn.deterministic) costs nothing.
//
// 0.9 GFLOP against 33 MB of output (batch 256, 64 channels): HBM-bound on the output, everything else is bookkeeping.
//   * a wave owns tiles of 32 consecutive pixels of one image row (W % 32 == 0); the three input rows it needs
//     (34 pixels x 3 channels each) are staged in wave-private LDS with 2-byte loads (pixels are 6 bytes: nothing wider
//     is aligned), zero outside the image;
//   * the 27-long reduction is laid out in 32 slots: row r (dh = r - 1) -> slots 10r .. 10r+8 = (dw, channel) pairs,
//     slot 10r+9 and slots 30, 31 are zero.  Two 32x32x16 MFMAs per 32 output channels; the weight fragments (same slot
//     order) live in registers for the life of the persistent workgroup;
//   * forward: the 32 pixel x Co tile is contiguous in channels-last memory (up to 4 KiB): transposed through LDS and
//     stored as 16-byte pieces, whole 128-byte lines per pixel.  Optional BatchNorm moments of the stored values
//     (sum (y - shift), sum (y - shift)^2 per channel) into the f64 accumulator block, as the other forward kernels;
//   * weight gradient: dW[co][slot] = sum over pixels dy[pix][co] * window[pix][slot] — the same two operands with the
//     pixel index as the reduction: both come out of LDS transposed (dy tile staged with 16-byte loads).  Each workgroup
//     keeps its partial in MFMA accumulators over all its tiles, writes one slab; a second tiny kernel sums the slabs in
//     fixed order (run-to-run reproducible, no float atomics).
#include "afan_conv_stem.h"
#include <stdlib.h>

using namespace afan;

namespace afan_stem {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 256;          // 4 waves
constexpr int WAVES = THREADS / 64;
constexpr int ROWE = 104;             // staged input row: 34 pixels x 3 channels = 102 elements (+2)
constexpr uint32_t OOB = 0x80000000u;

// slot s of the 32-slot reduction -> element offset inside a pixel's 3x3x3 window (r * 9 + dw * 3 + c), or -1 (zero slot)
__device__ __forceinline__ int slot_elem(int s) {
    const int d = s >> 1, r = d / 5, m = d - 5 * r, e = 2 * m + (s & 1);
    return (d < 15 && e < 9) ? r * 9 + e : -1;
}

// stage rows h-1, h, h+1, pixels w0-1 .. w0+32 of image n into `in` (zero outside the image)
__device__ __forceinline__ void stage_rows(const __amdgpu_buffer_rsrc_t xr, uint16_t (*in)[ROWE], int n, int h, int w0, int H,
                                           int W, int lane) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int hh = h + r - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = lane + 64 * j;
            const int px = i / 3, c = i - 3 * px, ww = w0 - 1 + px;
            const bool ok = i < 102 && hh >= 0 && hh < H && ww >= 0 && ww < W;
            const uint32_t off = ok ? (uint32_t)((((n * H + hh) * W + ww) * 3 + c) * 2) : OOB;
            const uint16_t v = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(xr, (int)off, 0, 0);
            if (i < ROWE) in[r][i] = v;
        }
    }
}

// the lane's two MFMA operand fragments of pixel `col`'s window (k-steps 0 and 1; half selects the upper 8 slots)
__device__ __forceinline__ void window_frags(const uint16_t (*in)[ROWE], int col, int half, bf16x8& f0, bf16x8& f1) {
    uint32_t D[16];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        uint32_t e[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = in[r][3 * col + k];
        D[5 * r + 0] = e[0] | (e[1] << 16);
        D[5 * r + 1] = e[2] | (e[3] << 16);
        D[5 * r + 2] = e[4] | (e[5] << 16);
        D[5 * r + 3] = e[6] | (e[7] << 16);
        D[5 * r + 4] = e[8];
    }
    D[15] = 0;
    u32x4 a, b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a[q] = half ? D[4 + q] : D[q];
        b[q] = half ? D[12 + q] : D[8 + q];
    }
    f0 = __builtin_bit_cast(bf16x8, a);
    f1 = __builtin_bit_cast(bf16x8, b);
}

struct FwdP {
    const uint16_t* x; const uint16_t* w; uint16_t* y;
    int N, H, W, Co;
    double* acc; int acc_ns; const float* shift;
};

template <int CG>   // groups of 32 output channels
__global__ __launch_bounds__(THREADS) void stem_fwd_kernel(const FwdP p) {
    constexpr int LDO = CG * 32 + 8;
    __shared__ __attribute__((aligned(16))) uint16_t in_s[WAVES][3][ROWE];
    __shared__ __attribute__((aligned(16))) uint16_t out_s[WAVES][32][LDO];
    __shared__ float red[WAVES][2][CG * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int H = p.H, W = p.W, Co = p.Co;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x), 0, (int)((int64_t)p.N * H * W * 3 * 2), 0x00020000);

    // weight fragments: row = output channel cg*32 + col; dword q of k-step ks = slots 16 ks + 8 half + 2q, +1
    bf16x8 wf[CG][2];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        const int co = cg * 32 + col;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = 16 * ks + 8 * half + 2 * q;
                const int e0 = slot_elem(s), e1 = slot_elem(s + 1);
                const uint32_t lo = (co < Co && e0 >= 0) ? p.w[co * 27 + e0] : 0u;
                const uint32_t hi = (co < Co && e1 >= 0) ? p.w[co * 27 + e1] : 0u;
                v[q] = lo | (hi << 16);
            }
            wf[cg][ks] = __builtin_bit_cast(bf16x8, v);
        }
    }

    const int ppr = Co >> 3;                       // 16-byte pieces per pixel row of the output (2, 4 or 8)
    const int pieces = 32 * ppr;
    const int chunk = lane & (ppr - 1);            // this lane's 8 channels, the same for all its pieces
    const bool want_stats = p.acc != nullptr;
    float s1[8], s2[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = s2[j] = 0.f;
        sh[j] = (want_stats && p.shift) ? p.shift[chunk * 8 + j] : 0.f;
    }

    const int tpr = W >> 5;
    const int tiles = p.N * H * tpr;
    for (int t = blockIdx.x * WAVES + wave; t < tiles; t += gridDim.x * WAVES) {
        const int rowidx = t / tpr, w0 = (t - rowidx * tpr) << 5;
        const int n = rowidx / H, h = rowidx - n * H;
        stage_rows(xr, in_s[wave], n, h, w0, H, W, lane);
        __builtin_amdgcn_wave_barrier();
        bf16x8 f0, f1;
        window_frags(in_s[wave], col, half, f0, f1);
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cg][0], f0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cg][1], f1, acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f2bf(acc[4 * g + e]);
                *reinterpret_cast<u16x4*>(&out_s[wave][col][cg * 32 + 8 * g + 4 * half]) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        uint16_t* ytile = p.y + ((int64_t)rowidx * W + w0) * Co;
        for (int q = lane; q < pieces; q += 64) {
            const int px = q / ppr;
            const u16x8 v = *reinterpret_cast<const u16x8*>(&out_s[wave][px][chunk * 8]);
            *reinterpret_cast<u16x8*>(ytile + px * Co + chunk * 8) = v;
            if (want_stats) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f(v[j]) - sh[j];
                    s1[j] += f;
                    s2[j] += f * f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (want_stats) {
        // lanes with the same chunk hold the same 8 channels: butterfly over them, then over the waves
        for (int o = ppr; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
        if (lane < ppr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[wave][0][lane * 8 + j] = s1[j];
                red[wave][1][lane * 8 + j] = s2[j];
            }
        }
        __syncthreads();
        if (tid < Co) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                a += red[w][0][tid];
                b += red[w][1][tid];
            }
            double* dst = p.acc + (int64_t)(blockIdx.x & (p.acc_ns - 1)) * 2 * Co;
            unsafeAtomicAdd(dst + tid, (double)a);
            unsafeAtomicAdd(dst + Co + tid, (double)b);
            if (blockIdx.x == 0) reinterpret_cast<float*>(p.acc + (int64_t)2 * p.acc_ns * Co)[tid] = p.shift ? p.shift[tid] : 0.f;
        }
    }
}

struct WgradP {
    const uint16_t* x; const uint16_t* dy; float* slab;
    int N, H, W, Co;
};

template <int CG>
__global__ __launch_bounds__(THREADS) void stem_wgrad_kernel(const WgradP p) {
    constexpr int LDY = CG * 32 + 8;
    __shared__ __attribute__((aligned(16))) uint16_t in_s[WAVES][3][ROWE];
    __shared__ __attribute__((aligned(16))) uint16_t dy_s[WAVES][32][LDY];
    __shared__ float part[WAVES][CG * 32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int H = p.H, W = p.W, Co = p.Co;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x), 0, (int)((int64_t)p.N * H * W * 3 * 2), 0x00020000);
    // this lane's slot (MFMA column) of the window operand: row r, element e inside the staged row — or a zero slot
    const int se = slot_elem(col);
    const int sr = se >= 0 ? se / 9 : 0, sofs = se >= 0 ? se - 9 * sr : 0;

    f32x16 acc[CG];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cg][r] = 0.f;

    const int ppr = Co >> 3, pieces = 32 * ppr, chunk = lane & (ppr - 1);
    const int tpr = W >> 5;
    const int tiles = p.N * H * tpr;
    for (int t = blockIdx.x * WAVES + wave; t < tiles; t += gridDim.x * WAVES) {
        const int rowidx = t / tpr, w0 = (t - rowidx * tpr) << 5;
        const int n = rowidx / H, h = rowidx - n * H;
        stage_rows(xr, in_s[wave], n, h, w0, H, W, lane);
        const uint16_t* dyt = p.dy + ((int64_t)rowidx * W + w0) * Co;
        for (int q = lane; q < pieces; q += 64) {
            const int px = q / ppr;
            *reinterpret_cast<u16x8*>(&dy_s[wave][px][chunk * 8]) = *reinterpret_cast<const u16x8*>(dyt + px * Co + chunk * 8);
        }
        if (CG * 32 > Co)                          // channel rows beyond Co: zero (16-channel stem)
            for (int q = lane; q < 32 * (CG * 32 - Co); q += 64) dy_s[wave][q / (CG * 32 - Co)][Co + q % (CG * 32 - Co)] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int p0 = 16 * ks + 8 * half;     // this lane's 8 reduction pixels
            u32x4 b;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t lo = se >= 0 ? in_s[wave][sr][3 * (p0 + 2 * q) + sofs] : 0u;
                const uint32_t hi = se >= 0 ? in_s[wave][sr][3 * (p0 + 2 * q + 1) + sofs] : 0u;
                b[q] = lo | (hi << 16);
            }
            const bf16x8 fb = __builtin_bit_cast(bf16x8, b);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                u32x4 a;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t lo = dy_s[wave][p0 + 2 * q][cg * 32 + col];
                    const uint32_t hi = dy_s[wave][p0 + 2 * q + 1][cg * 32 + col];
                    a[q] = lo | (hi << 16);
                }
                acc[cg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), fb, acc[cg], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // acc[cg][4g + e] = dW[channel cg*32 + 8g + 4 half + e][slot col]: waves summed in fixed order, one slab per workgroup
#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[wave][cg * 32 + 8 * g + 4 * half + e][col] = acc[cg][4 * g + e];
    __syncthreads();
    float* slab = p.slab + (int64_t)blockIdx.x * Co * 27;
    for (int i = tid; i < Co * 32; i += THREADS) {
        const int co = i >> 5, s = i & 31, el = slot_elem(s);
        if (el >= 0) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += part[w][co][s];
            slab[co * 27 + el] = v;
        }
    }
}

// grad[i] (+)= sum over the S slabs, fixed order: 64 elements x 16 slab groups per block (16 independent loads each)
__global__ __launch_bounds__(1024) void stem_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad,
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

int wgrad_slabs(int64_t tiles) {
    int64_t g = (tiles + WAVES - 1) / WAVES;
    if (g > 256) g = 256;                                  // one workgroup per CU, persistent over its tiles
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

bool eligible(int64_t n, int64_t h, int64_t w, int64_t ci, int64_t co, int k, int stride) {
    static const bool on = [] { const char* v = getenv("AFAN_CONV_STEM"); return !v || atoi(v) != 0; }();
    if (!on || ci != 3 || k != 3 || stride != 1) return false;
    if (!(co == 16 || co == 32 || co == 64) || w % 32 != 0 || n <= 0 || h <= 0) return false;
    return n * h * w * co * 2 <= 0x7fffffffLL;
}

int fwd_launch(const void* x, const void* w, void* y, int64_t n, int64_t h, int64_t wd, int64_t co, double* acc, int acc_ns,
               const float* shift, hipStream_t st) {
    FwdP p{(const uint16_t*)x, (const uint16_t*)w, (uint16_t*)y, (int)n, (int)h, (int)wd, (int)co, acc, acc_ns, shift};
    const int64_t tiles = n * h * (wd / 32);
    int64_t g = (tiles + WAVES - 1) / WAVES;
    static const int64_t cap = [] { const char* v = getenv("AFAN_STEM_WGS"); return v ? (int64_t)atoi(v) : (int64_t)1024; }();
    if (g > cap) g = cap;                                  // 4 workgroups per CU, then persistent (2048 measured 24.6 us against 19.0: the weight-fragment setup per workgroup)
    if (co > 32) stem_fwd_kernel<2><<<(unsigned)g, THREADS, 0, st>>>(p);
    else stem_fwd_kernel<1><<<(unsigned)g, THREADS, 0, st>>>(p);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int64_t wgrad_workspace_floats(int64_t n, int64_t h, int64_t w, int64_t co) {
    return (int64_t)wgrad_slabs(n * h * (w / 32)) * co * 27;
}

int wgrad_launch(const void* x, const void* dy, float* grad, int64_t n, int64_t h, int64_t wd, int64_t co, float* ws,
                 int accumulate, hipStream_t st) {
    WgradP p{(const uint16_t*)x, (const uint16_t*)dy, ws, (int)n, (int)h, (int)wd, (int)co};
    const int S = wgrad_slabs(n * h * (wd / 32));
    if (co > 32) stem_wgrad_kernel<2><<<S, THREADS, 0, st>>>(p);
    else stem_wgrad_kernel<1><<<S, THREADS, 0, st>>>(p);
    AFAN_LAUNCH_CHECK();
    const int total = (int)co * 27;
    stem_wgrad_reduce_kernel<<<(total + 63) / 64, 1024, 0, st>>>(ws, grad, total, S, accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // namespace afan_stem
