// Bare MFMA issue rate of ONE wave (and of W waves per SIMD) as a function of the number of independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC, bool SMALL>
__global__ void k(float* out, long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    f32x16 acc[NACC];
    f32x4 acs[NACC];
    for (int j = 0; j < NACC; ++j) { for (int r = 0; r < 16; ++r) acc[j][r] = 0.f; for (int r = 0; r < 4; ++r) acs[j][r] = 0.f; }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (SMALL) acs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acs[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) { for (int r = 0; r < 16; ++r) s += acc[j][r]; for (int r = 0; r < 4; ++r) s += acs[j][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, bool SMALL>
void run(int threads) {
    float* o; long long* c; hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&c, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(256), dim3(threads), 0, 0, o, c, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(256), dim3(threads), 0, 0, o, c, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 16;                       // MFMAs per wave
    const double flop = mf * (SMALL ? 16384.0 : 32768.0) * (threads / 64) * 256;
    printf("%s nacc %d waves/SIMD %d: %.1f ns per MFMA per wave, %.1f ns per MFMA per SIMD (counter %.1f ticks/MFMA), %.0f TFLOP/s\n", SMALL ? "16x16x32" : "32x32x16", NACC,
           threads / 256, ms * 1e6 / mf, ms * 1e6 / mf / (threads / 256), (double)h / mf, flop / ms / 1e9);
    hipFree(o); hipFree(c);
}
int main() {
    for (int t : {256, 512, 1024}) { run<1, false>(t); run<2, false>(t); run<4, false>(t); run<8, false>(t); run<16, false>(t); }
    for (int t : {256, 512, 1024}) { run<1, true>(t); run<2, true>(t); run<4, true>(t); run<8, true>(t); run<16, true>(t); }
    return 0;
}
