// What a convolution epilogue would pay for "BatchNorm sums -> grid barrier -> totals" inside ONE launch (round 5):
// 256 workgroups x 768 threads with 145 KB of dynamic LDS (one per CU, like the halo-form convolutions); per round every
// workgroup adds 2 x 128 f64 partial sums into NS accumulator copies with native atomics, drains them (s_waitcnt vmcnt(0)),
// meets the others at a sense-reversing barrier sharded over 8 counters (RELAXED agent-scope atomics only: what the barrier orders
// travels through memory-side atomics itself), then reads the totals back with sc1 loads (the per-XCD L2s are not coherent).
// Checks the totals every round (wrong = a workgroup read before every add had landed).  Bounded spin.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/grid_barrier2.hip -o /tmp/gb2 && /tmp/gb2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Bar {            // 64-byte separated words
    unsigned shard_cnt[8][16];
    unsigned global_cnt[16];
    unsigned flag[8][16];
    unsigned err[16];
};

__device__ __forceinline__ void grid_barrier(Bar* b, unsigned nwg) {
    // caller: all of the workgroup's memory operations drained, __syncthreads() done; one thread calls
    const unsigned id = blockIdx.x, sh = id & 7u;
    const unsigned per = nwg / 8u + (sh < (nwg & 7u) ? 1u : 0u);
    const unsigned shards = nwg < 8u ? nwg : 8u;
    const unsigned gen = __hip_atomic_load(&b->flag[sh][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned a = __hip_atomic_fetch_add(&b->shard_cnt[sh][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == per) {
        (void)__hip_atomic_exchange(&b->shard_cnt[sh][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // returns: performed
        const unsigned g = __hip_atomic_fetch_add(&b->global_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g + 1 == shards) {
            (void)__hip_atomic_exchange(&b->global_cnt[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned q = 0; q < shards; ++q)
                __hip_atomic_store(&b->flag[q][0], gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    while (__hip_atomic_load(&b->flag[sh][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000LL) {   // 0.2 s
            __hip_atomic_store(&b->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

template <int MODE>   // 0: atomics only; 1: + barrier; 2: + barrier + totals read back and checked
__global__ __launch_bounds__(768) void k(Bar* bar, double* acc, unsigned* bad, int rounds, int NS, int C) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        if (tid < C) {
            double* dst = acc + (size_t)(blockIdx.x & (NS - 1)) * 2 * C;
            unsafeAtomicAdd(dst + tid, 1.0);
            unsafeAtomicAdd(dst + C + tid, 2.0);
        }
        if (MODE >= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) grid_barrier(bar, gridDim.x);
            __syncthreads();
        }
        if (MODE >= 2) {
            if (tid < C) {
                double a = 0, b = 0;
                for (int s = 0; s < NS; ++s) {
                    a += __hip_atomic_load(acc + (size_t)s * 2 * C + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    b += __hip_atomic_load(acc + (size_t)s * 2 * C + C + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                lds[tid] = (float)a;
                // too small = read before every add of this round had landed (a real failure); too large = a faster workgroup's
                // adds of the NEXT round (this probe re-uses the totals without a second barrier; a launch has one episode)
                if (a < (double)r * gridDim.x || b < 2.0 * r * gridDim.x) atomicAdd(bad, 1u);
                else if (a != (double)r * gridDim.x) atomicAdd(bad + 1, 1u);
            }
            __syncthreads();
            // (a second barrier would be needed before the NEXT round's adds if the totals were reset; they are not: monotonic)
        }
    }
    if (tid == 0) lds[0] += 1.f;
}

int main() {
    Bar* bar; double* acc; unsigned* bad;
    const int NS = 8, C = 128, rounds = 200;
    (void)hipMalloc(&bar, sizeof(Bar)); (void)hipMalloc(&acc, sizeof(double) * NS * 2 * C); (void)hipMalloc(&bad, 8);
    const int lds = 145 * 1024;
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int blocks : {256, 128, 255}) for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f; unsigned hb = 0, he = 0, hl = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(bar, 0, sizeof(Bar)); (void)hipMemset(acc, 0, sizeof(double) * NS * 2 * C); (void)hipMemset(bad, 0, 8);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(768), lds, 0, bar, acc, bad, rounds, NS, C);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(768), lds, 0, bar, acc, bad, rounds, NS, C);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(768), lds, 0, bar, acc, bad, rounds, NS, C);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&hl, bad + 1, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&he, (char*)bar + offsetof(Bar, err), 4, hipMemcpyDeviceToHost);
            if (ms / rounds < best) best = ms / rounds;
            if (hb || he) break;
        }
        printf("blocks %3d mode %d (%s): %.2f us per round  totals read EARLY %u (late, next round's adds: %u)  spin limit %u\n", blocks, mode,
               mode == 0 ? "atomics only" : mode == 1 ? "+ barrier" : "+ barrier + sc1 totals", best * 1e3, hb, hl, he);
    }
    return 0;
}
