// Cost of a grid-wide barrier between resident workgroups (the construct a fused convolution + BatchNorm launch needs).
// One device-scope arrival counter per barrier (monotonic: target = round * blocks).  Default build: RELAXED atomics (enough
// when what the barrier orders travels through memory-side atomics itself); -DORD_ARRIVE=__ATOMIC_RELEASE
// -DORD_POLL=__ATOMIC_ACQUIRE adds the L2 write-back / invalidate a release-acquire pair costs at agent scope;
// the spin is BOUNDED (gives up and flags) so that a placement where not all workgroups are resident cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef ORD_ARRIVE
#define ORD_ARRIVE __ATOMIC_RELAXED
#define ORD_POLL __ATOMIC_RELAXED
#endif
// Sharded form: workgroup b belongs to shard b % 8 (the dispatcher deals workgroups round-robin over the 8 XCDs); arrivals
// count per shard, the last arriver of a shard bumps the global counter, the last shard to arrive publishes the round in
// all 8 per-shard flags; a workgroup polls only its shard's flag (32-64 pollers per address instead of 256-512).
__global__ void ks(unsigned* mem, unsigned* failed, float* sink, int rounds, int work) {
    unsigned* shard_cnt = mem;            // [8] (64-byte apart)
    unsigned* global_cnt = mem + 8 * 16;
    unsigned* flag = mem + 9 * 16;        // [8] (64-byte apart)
    const int sh = blockIdx.x & 7, per = gridDim.x >> 3;
    float acc = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        for (int i = 0; i < work; ++i) acc = acc * 1.0001f + 0.5f;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned a = __hip_atomic_fetch_add(shard_cnt + sh * 16, 1u, ORD_ARRIVE, __HIP_MEMORY_SCOPE_AGENT);
            if (a + 1 == (unsigned)r * per) {
                const unsigned g = __hip_atomic_fetch_add(global_cnt, 1u, ORD_ARRIVE, __HIP_MEMORY_SCOPE_AGENT);
                if (g + 1 == (unsigned)r * 8)
                    for (int q = 0; q < 8; ++q) __hip_atomic_store(flag + q * 16, (unsigned)r, ORD_ARRIVE, __HIP_MEMORY_SCOPE_AGENT);
            }
            long spins = 0;
            while (__hip_atomic_load(flag + sh * 16, ORD_POLL, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {
                if (++spins > 20000000L) { *failed = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k(unsigned* counter, unsigned* failed, float* sink, int rounds, int work) {
    float acc = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        for (int i = 0; i < work; ++i) acc = acc * 1.0001f + 0.5f;          // a little per-round work
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, ORD_ARRIVE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)r * gridDim.x;
            long spins = 0;
            while (__hip_atomic_load(counter, ORD_POLL, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 20000000L) { *failed = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    unsigned *c, *f; float* s;
    (void)hipMalloc(&c, 4096); (void)hipMalloc(&f, 4); (void)hipMalloc(&s, 1024 * 512 * 4);
    for (int sharded : {0, 1}) for (int blocks : {256, 512}) for (int threads : {256, 512}) for (int work : {0, 2000}) {
        float best = 1e9f; unsigned hf = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(c, 0, 4096); (void)hipMemset(f, 0, 4);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            const int rounds = 200;
            (void)hipEventRecord(e0);
            if (sharded) hipLaunchKernelGGL(ks, dim3(blocks), dim3(threads), 0, 0, c, f, s, rounds, work);
            else hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, c, f, s, rounds, work);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(&hf, f, 4, hipMemcpyDeviceToHost);
            if (ms / rounds < best) best = ms / rounds;
        }
        printf("%s blocks %d x %d threads, work %d: %.2f us per round%s\n", sharded ? "sharded" : "single ", blocks, threads, work, best * 1e3, hf ? "  (SPIN LIMIT HIT: not all resident?)" : "");
    }
    return 0;
}
