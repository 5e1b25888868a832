// Round 6: what the in-launch BatchNorm's "sums -> grid barrier -> totals" chain pays per part, and how the totals' read-back depends
// on the number of accumulator copies (NS) and on the load width.  256 workgroups x 768 threads x 145 KB LDS (one per CU); per round:
// every workgroup adds C x 2 f64 partial sums into copy (id & (NS - 1)) [drained], meets the others at the SHIPPED barrier protocol
// (sharded arrival counters + one monotonic release counter), reads the totals of its C channels back past L2 (sc1) and checks them.
//   layout 0: acc[NS][2][C] (the library's), totals by 8-byte loads;  layout 1: acc[NS][C][2] (s1, s2 adjacent), totals by 16-byte loads.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/grid_barrier3.hip -o /tmp/gb3 && /tmp/gb3
#include <hip/hip_runtime.h>
#include <cstdio>

struct Bar { unsigned shard_cnt[8][16]; unsigned global_cnt[16]; unsigned err[16]; };

__device__ __forceinline__ unsigned episode(Bar* b) { return (__hip_atomic_load(&b->global_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / 8u + 1u) * 8u; }
__device__ __forceinline__ void arrive(Bar* b, unsigned id, unsigned nwg) {
    const unsigned sh = id & 7u, per = nwg / 8u + (sh < (nwg & 7u) ? 1u : 0u);
    const unsigned a = __hip_atomic_fetch_add(&b->shard_cnt[sh][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == per) {
        __hip_atomic_store(&b->shard_cnt[sh][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&b->global_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void wait(Bar* b, unsigned target) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(&b->global_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000LL) { __hip_atomic_store(&b->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int LAYOUT>   // MODE 0: atomics only; 1: + barrier; 2: + totals
__global__ __launch_bounds__(768) void k(Bar* bar, double* acc, unsigned* bad, int rounds, int NS, int C) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(acc, 0, NS * 2 * C * 8, 0x00020000);
    for (int rd = 1; rd <= rounds; ++rd) {
        unsigned target = 0;
        if (tid == 0 && MODE >= 1) target = episode(bar);
        if (tid < C) {
            const size_t s = (size_t)(blockIdx.x & (NS - 1));
            if (LAYOUT == 0) { unsafeAtomicAdd(acc + s * 2 * C + tid, 1.0); unsafeAtomicAdd(acc + s * 2 * C + C + tid, 2.0); }
            else { unsafeAtomicAdd(acc + (s * C + tid) * 2, 1.0); unsafeAtomicAdd(acc + (s * C + tid) * 2 + 1, 2.0); }
        }
        if (MODE >= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) { arrive(bar, blockIdx.x, gridDim.x); wait(bar, target); }
            __syncthreads();
        }
        if (MODE >= 2) {
            if (tid < C) {
                double a = 0, b = 0;
                if (LAYOUT == 0) {
                    double av[8], bv[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        av[s] = s < NS ? __builtin_bit_cast(double, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(r, ((2 * s) * C + tid) * 8, 0, 16)) : 0.0;
                        bv[s] = s < NS ? __builtin_bit_cast(double, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(r, ((2 * s + 1) * C + tid) * 8, 0, 16)) : 0.0;
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) { a += av[s]; b += bv[s]; }
                } else {
                    u32x4 v[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) v[s] = s < NS ? (u32x4)__builtin_amdgcn_raw_buffer_load_b128(r, (s * C + tid) * 16, 0, 16) : u32x4{0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        a += __builtin_bit_cast(double, u32x2{v[s][0], v[s][1]});
                        b += __builtin_bit_cast(double, u32x2{v[s][2], v[s][3]});
                    }
                }
                lds[tid] = (float)(a + b);
                if (a < (double)rd * gridDim.x || b < 2.0 * rd * gridDim.x) atomicAdd(bad, 1u);
            }
            __syncthreads();
        }
    }
    if (tid == 0) lds[0] += 1.f;
}

template <int MODE, int LAYOUT> float run(Bar* bar, double* acc, unsigned* bad, int blocks, int NS, int C, unsigned* hb, unsigned* he) {
    const int rounds = 200, lds = 145 * 1024;
    (void)hipFuncSetAttribute((const void*)k<MODE, LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(bar, 0, sizeof(Bar)); (void)hipMemset(acc, 0, sizeof(double) * 8 * 2 * 512); (void)hipMemset(bad, 0, 8);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, LAYOUT>), dim3(blocks), dim3(768), lds, 0, bar, acc, bad, rounds, NS, C);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(hb, bad, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(he, (char*)bar + offsetof(Bar, err), 4, hipMemcpyDeviceToHost);
        if (ms / rounds < best) best = ms / rounds;
    }
    return best * 1e3f;
}

int main() {
    Bar* bar; double* acc; unsigned* bad;
    (void)hipMalloc(&bar, sizeof(Bar)); (void)hipMalloc(&acc, sizeof(double) * 8 * 2 * 512); (void)hipMalloc(&bad, 8);
    printf("256 workgroups, us per round: atomics | + barrier | + totals (8-byte loads, acc[NS][2][C]) | + totals (16-byte loads, acc[NS][C][2])   early reads / spin limits\n");
    for (int C : {128, 64}) for (int NS : {8, 4, 2, 1}) {
        unsigned hb = 0, he = 0, hb2 = 0, he2 = 0;
        const float a = run<0, 0>(bar, acc, bad, 256, NS, C, &hb, &he), b = run<1, 0>(bar, acc, bad, 256, NS, C, &hb, &he);
        const float c = run<2, 0>(bar, acc, bad, 256, NS, C, &hb, &he), d = run<2, 1>(bar, acc, bad, 256, NS, C, &hb2, &he2);
        const float a1 = run<0, 1>(bar, acc, bad, 256, NS, C, &hb2, &he2);
        printf("C %3d NS %d: %5.2f (adjacent layout %5.2f) | %5.2f | %5.2f | %5.2f    %u %u / %u %u\n", C, NS, a, a1, b, c, d, hb, hb2, he, he2);
    }
    return 0;
}
