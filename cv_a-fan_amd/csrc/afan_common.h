// Shared device helpers for the A-FAN gfx950 kernels.  wave = 64 lanes; 16-byte accesses per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/afan_hip.h"

#define AFAN_WAVE 64

namespace afan {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

// bf16 <-> f32.  The plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 (RNE, NaN stays NaN).
__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t u) {
    return __builtin_bit_cast(float, ((uint32_t)u) << 16);
}


template <typename T> struct Elt;
template <> struct Elt<float> {
    static constexpr int VEC = 4;  // elements per 16-byte access
    typedef f32x4 vec_t;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ void ldv(const float* p, float (&v)[4]) {
        f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void stv(float* p, const float (&v)[4]) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = t;
    }
};
template <> struct Elt<uint16_t> {  // bf16 storage
    static constexpr int VEC = 8;
    typedef u16x8 vec_t;
    __device__ static __forceinline__ float ld(const uint16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(uint16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ void ldv(const uint16_t* p, float (&v)[8]) {
        u16x8 t = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bf2f(t[i]);
    }
    __device__ static __forceinline__ void stv(uint16_t* p, const float (&v)[8]) {
        u16x8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = f2bf(v[i]);
        *reinterpret_cast<u16x8*>(p) = t;
    }
};

// wave-level butterfly reductions over 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Chan et al. merge of two (count, mean, M2) moment triples.
struct Moments {
    float n, mean, m2;
};
__device__ __forceinline__ Moments merge(Moments a, Moments b) {
    float n = a.n + b.n;
    if (n == 0.f) return Moments{0.f, 0.f, 0.f};
    float d = b.mean - a.mean;
    float f = b.n / n;
    Moments r;
    r.n = n;
    r.mean = a.mean + d * f;
    r.m2 = a.m2 + b.m2 + d * d * a.n * f;
    return r;
}
__device__ __forceinline__ Moments wave_merge(Moments m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Moments t;
        t.n = __shfl_xor(m.n, o, 64);
        t.mean = __shfl_xor(m.mean, o, 64);
        t.m2 = __shfl_xor(m.m2, o, 64);
        m = merge(m, t);
    }
    return m;
}

static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace afan

// ---- optional per-launch event timing (afan_prof.hip) ----
namespace afan { namespace prof {
extern int g_enabled;
extern thread_local unsigned g_launches;     // kernel launches issued by this thread (AFAN_LAUNCH_CHECK counts them)
void begin(const char* name, double bytes, double flops, hipStream_t st, size_t* slot);
void end(size_t slot, hipStream_t st);
void cancel(size_t slot);
struct Scope {
    size_t slot; hipStream_t st; bool on; unsigned launches0;
    Scope(const char* name, double bytes, hipStream_t s, double flops = 0.0) : slot((size_t)-1), st(s), on(g_enabled != 0), launches0(g_launches) {
        if (on) begin(name, bytes, flops, st, &slot);
    }
    // a scope inside which NOTHING was launched (an entry point that declined the problem: AFAN_ESHAPE, "not this launch") is not a
    // launch: its FLOPs and bytes must not be booked against ~zero time
    ~Scope() { if (on) { if (g_launches != launches0) end(slot, st); else cancel(slot); } }
};
} }
// time the launches issued in the rest of the enclosing block as kernel `name` moving `bytes` algorithmic bytes
#define AFAN_PROF(name, bytes, st) afan::prof::Scope afan_prof_scope__(name, (double)(bytes), st)
// same, for MFMA-bound kernels: also records the launch's algorithmic FLOPs
#define AFAN_PROF_FLOPS(name, bytes, flops, st) afan::prof::Scope afan_prof_scope__(name, (double)(bytes), st, (double)(flops))

#define AFAN_LAUNCH_CHECK()                     \
    do {                                        \
        ++afan::prof::g_launches;               \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)
