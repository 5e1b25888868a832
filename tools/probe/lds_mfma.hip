// Consumer-side rate of the convolution's K loop: 4 waves (one per SIMD), each a 64x64 tile of a 128x128x64 LDS stage
// (16 ds_read_b128 + 16 MFMA 32x32x16 per K-step), no global traffic.  Variants of the read/MFMA order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int BM = 128, BN = 128, BK = 64, STAGE = (BM + BN) * BK;
__device__ int g_random = 0;     // 1: operands are random normal-ish bf16 (MFMA power, hence clock, depends on the data)
template <int V, int NWAVE, bool BAR>
__global__ __launch_bounds__(64 * NWAVE) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * STAGE; i += blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // random sign, exponent 2^-4 .. 2^0, random mantissa
        lds[i] = g_random ? (uint16_t)(((h & 1) << 15) | ((123 + ((h >> 1) % 5)) << 7) | ((h >> 8) & 127)) : (uint16_t)(0x3c00 + (i & 63));
    }
    __syncthreads();
    constexpr int WN = (NWAVE == 4 || V == 3) ? 2 : 4, TM = 64, TN = BN / WN, MI = 2, NI = TN / 32;
    const int wr = wave / WN, wc = wave % WN;
    const int frow = lane & 31, sw = (frow >> 1) & 7;
    f32x16 acc[NI][MI];
    for (int j = 0; j < NI; ++j) for (int i = 0; i < MI; ++i) for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    auto ldA = [&](int buf, int kk, int i) { const int koff = ((kk * 2 + (lane >> 5)) ^ sw) * 8;
        return *reinterpret_cast<const bf16x8*>(lds + buf * STAGE + (wr * TM + i * 32 + frow) * BK + koff); };
    auto ldB = [&](int buf, int kk, int j) { const int koff = ((kk * 2 + (lane >> 5)) ^ sw) * 8;
        return *reinterpret_cast<const bf16x8*>(lds + buf * STAGE + BM * BK + (wc * TN + j * 32 + frow) * BK + koff); };
    if (V == 3 && wave >= 4) {          // partner waves: barrier only (the producer waves of the convolution, without their DMA)
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
        out[blockIdx.x * blockDim.x + tid] = 0.f;
        return;
    }
    if (V == 0 || V == 3) {             // all 4 k16-slices requested, then the MFMAs
        for (int it = 0; it < iters; ++it) {
            const int buf = it & 1;
            if (BAR) __builtin_amdgcn_s_barrier();
            bf16x8 fx[4][MI], fw[4][NI];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { for (int i = 0; i < MI; ++i) fx[kk][i] = ldA(buf, kk, i); for (int j = 0; j < NI; ++j) fw[kk][j] = ldB(buf, kk, j); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < NI; ++j) for (int i = 0; i < MI; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk][j], fx[kk][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (V == 1) {                // half-step software pipeline: slices {0,1} of the NEXT K-step are requested before the
                                        // MFMAs of slices {2,3}; the barrier sits between the two
        bf16x8 ax[2][MI], aw[2][NI], bx[2][MI], bw[2][NI];
        if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) { for (int i = 0; i < MI; ++i) ax[h][i] = ldA(0, h, i); for (int j = 0; j < NI; ++j) aw[h][j] = ldB(0, h, j); }
        for (int it = 0; it < iters; ++it) {
            const int buf = it & 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) { for (int i = 0; i < MI; ++i) bx[h][i] = ldA(buf, 2 + h, i); for (int j = 0; j < NI; ++j) bw[h][j] = ldB(buf, 2 + h, j); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) for (int j = 0; j < NI; ++j) for (int i = 0; i < MI; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[h][j], ax[h][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (BAR) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }    // lgkmcnt(0): this tile's reads are done
#pragma unroll
            for (int h = 0; h < 2; ++h) { for (int i = 0; i < MI; ++i) ax[h][i] = ldA(buf ^ 1, h, i); for (int j = 0; j < NI; ++j) aw[h][j] = ldB(buf ^ 1, h, j); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) for (int j = 0; j < NI; ++j) for (int i = 0; i < MI; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[h][j], bx[h][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {                            // the compiler's order (no sched barriers)
        for (int it = 0; it < iters; ++it) {
            const int buf = it & 1;
            if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 fx[MI], fw[NI];
                for (int i = 0; i < MI; ++i) fx[i] = ldA(buf, kk, i);
                for (int j = 0; j < NI; ++j) fw[j] = ldB(buf, kk, j);
                for (int j = 0; j < NI; ++j) for (int i = 0; i < MI; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fx[i], acc[j][i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < NI; ++j) for (int i = 0; i < MI; ++i) for (int r = 0; r < 16; ++r) s += acc[j][i][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}
template <int V, int NWAVE, bool BAR>
void run() {
    float* o; (void)hipMalloc(&o, 256 * 1024 * 4);
    const int iters = 2000;
    (void)hipFuncSetAttribute((const void*)k<V, NWAVE, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 2);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, NWAVE, BAR>), dim3(256), dim3(64 * NWAVE), 2 * STAGE * 2, 0, o, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, NWAVE, BAR>), dim3(256), dim3(64 * NWAVE), 2 * STAGE * 2, 0, o, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * BM * BN * BK * iters * 256;
    printf("variant %d, %d waves, barrier %d: %.0f ns per K-step (MFMA floor ~%d), %.0f TFLOP/s\n", V, NWAVE, (int)BAR, ms * 1e6 / iters, 16 * 18, flop / ms / 1e9);
    (void)hipFree(o);
}
int main(int argc, char** argv) {
    int r = argc > 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_random), &r, sizeof(int));
    printf("operands: %s\n", r ? "random" : "smooth");
    run<2, 4, false>(); run<0, 4, false>(); run<1, 4, false>();
    run<2, 4, true>(); run<0, 4, true>(); run<1, 4, true>();
    run<2, 8, false>(); run<0, 8, false>(); run<1, 8, false>();
    run<2, 8, true>(); run<0, 8, true>(); run<1, 8, true>();
    run<3, 8, true>();
    return 0;
}
