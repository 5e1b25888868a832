// SGD with momentum over one flat parameter arena (gfx950).
// Reference behaviour: torch.optim.SGD as configured at Classification/main_perturb.py:72-74
// (momentum 0.9, weight_decay 5e-4, dampening 0, nesterov off) with the warm-up learning rate of
// main_perturb.py:288-293.  One launch updates every tensor of the model: parameters, gradients and
// momentum live in three flat fp32 buffers (also the single all-reduce payload for data parallel).
#include "afan_common.h"

using namespace afan;

namespace {
constexpr int BLOCK = 256;

template <bool FIRST, bool SHADOW>
__global__ __launch_bounds__(BLOCK) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, uint16_t* __restrict__ shadow,
                                                    int64_t n, const float* __restrict__ lr_dev, float mom,
                                                    float wd, float gscale, int vec, const unsigned* __restrict__ guard) {
    // guard (nullable): the grid barrier's error word (afan_conv.hip GridBar::err) — non-zero means a convolution + BatchNorm
    // launch of this step went on with partial batch totals: the step's gradients are invalid and the update must not happen
    if (guard && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const float lr = *lr_dev;
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    auto upd = [&](float& pv, float gv, float& mv) {
        float gg = gv * gscale;
        gg = gg + wd * pv;                    // grad.add(param, alpha=weight_decay)
        mv = FIRST ? gg : (mv * mom + gg);    // buf = clone(grad) on the first step, else buf*mom + grad
        pv = pv - lr * mv;                    // param.add_(buf, alpha=-lr)
    };
    int64_t done = 0;
    if (vec) {
        const int64_t nvec = n >> 2;
        for (int64_t v = tid; v < nvec; v += nthreads) {
            const int64_t i = v << 2;
            float pv[4], gv[4], mv[4] = {0.f, 0.f, 0.f, 0.f};
            Elt<float>::ldv(p + i, pv);
            Elt<float>::ldv(g + i, gv);
            if (!FIRST) Elt<float>::ldv(m + i, mv);
#pragma unroll
            for (int k = 0; k < 4; ++k) upd(pv[k], gv[k], mv[k]);
            Elt<float>::stv(p + i, pv);
            Elt<float>::stv(m + i, mv);
            if (SHADOW) {
                u16x4 s;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = f2bf(pv[k]);
                *reinterpret_cast<u16x4*>(shadow + i) = s;
            }
        }
        done = nvec << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        float pv = p[i], mv = FIRST ? 0.f : m[i];
        upd(pv, g[i], mv);
        p[i] = pv;
        m[i] = mv;
        if (SHADOW) shadow[i] = f2bf(pv);
    }
}
}  // namespace

static int sgd_impl(float* param, const float* grad, float* momentum_buf, uint16_t* shadow_bf16,
                    int64_t n, const float* lr_dev, float momentum, float weight_decay,
                    float grad_scale, int first_step, const unsigned* guard, afan_stream_t stream) {
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!param || !grad || !momentum_buf || !lr_dev) return AFAN_ENULL;
    if (!aligned(param, 4) || !aligned(grad, 4) || !aligned(momentum_buf, 4) ||
        (shadow_bf16 && !aligned(shadow_bf16, 2)))
        return AFAN_EALIGN;
    const int vec = aligned(param, 16) && aligned(grad, 16) && aligned(momentum_buf, 16) &&
                    (!shadow_bf16 || aligned(shadow_bf16, 8));
    const int grid = grid_for(vec ? (n + 3) / 4 : n, BLOCK);
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("sgd_kernel", n * (20.0 + (shadow_bf16 ? 2 : 0)), st);
#define AFAN_GO(F, S) \
    sgd_kernel<F, S><<<grid, BLOCK, 0, st>>>(param, grad, momentum_buf, shadow_bf16, n, lr_dev, momentum, weight_decay, grad_scale, vec, guard)
    if (first_step) { if (shadow_bf16) AFAN_GO(true, true); else AFAN_GO(true, false); }
    else { if (shadow_bf16) AFAN_GO(false, true); else AFAN_GO(false, false); }
#undef AFAN_GO
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

extern "C" int afan_sgd_step(float* param, const float* grad, float* momentum_buf, uint16_t* shadow_bf16,
                             int64_t n, const float* lr_dev, float momentum, float weight_decay,
                             float grad_scale, int first_step, afan_stream_t stream) {
    return sgd_impl(param, grad, momentum_buf, shadow_bf16, n, lr_dev, momentum, weight_decay, grad_scale, first_step, nullptr, stream);
}

// The same update, skipped ON THE DEVICE when *guard != 0 (guard = the grid barrier's error word): a step whose in-launch BatchNorm
// went on with partial totals never reaches the weights, the momentum or the bf16 shadow — without a host read inside the step.
// The word is sticky until the host clears it, so every later step is skipped too until the host has noticed (train_step.GridGuard).
extern "C" int afan_sgd_step_guarded(float* param, const float* grad, float* momentum_buf, uint16_t* shadow_bf16,
                                     int64_t n, const float* lr_dev, float momentum, float weight_decay,
                                     float grad_scale, int first_step, const void* guard, afan_stream_t stream) {
    if (!guard) return AFAN_ENULL;
    if (!aligned(guard, 4)) return AFAN_EALIGN;
    return sgd_impl(param, grad, momentum_buf, shadow_bf16, n, lr_dev, momentum, weight_decay, grad_scale, first_step,
                    (const unsigned*)guard, stream);
}

namespace {
// dst = src (16-byte pieces) unless *guard != 0; one thread also counts the copies that happened (how many steps started clean)
__global__ __launch_bounds__(BLOCK) void guarded_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int64_t n16,
                                                              const unsigned* __restrict__ guard, unsigned* __restrict__ counter) {
    if (__hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const int64_t nthreads = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += nthreads) dst[i] = src[i];
    if (counter && blockIdx.x == 0 && threadIdx.x == 0) *counter = *counter + 1u;
}

// `workgroups` workgroups that each hold `lds_bytes` of LDS and spin for `microseconds` (s_memrealtime: 100 MHz): stands for another
// stream's long-lived kernels (an RCCL channel kernel, a weight gradient) in tests and probes
__global__ __launch_bounds__(64) void occupy_kernel(long long ticks) {
    extern __shared__ unsigned occ_lds[];
    occ_lds[threadIdx.x] = threadIdx.x;
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (occ_lds[threadIdx.x] == 0xffffffffu) occ_lds[0] = 1;     // (keeps the allocation alive)
}
}  // namespace

// The pre-step snapshot of a trainer's BatchNorm buffers: dst = src (bytes a multiple of 16, both 16-byte aligned) — skipped when
// *guard != 0, so that after a barrier gave up the snapshot keeps the state at the START of the step that failed; *counter
// (nullable, device) counts the copies that happened.
extern "C" int afan_guarded_copy(void* dst, const void* src, int64_t bytes, const void* guard, void* counter, afan_stream_t stream) {
    if (bytes < 0 || (bytes & 15)) return AFAN_ESHAPE;
    if (bytes == 0) return AFAN_OK;
    if (!dst || !src || !guard) return AFAN_ENULL;
    if (!aligned(dst, 16) || !aligned(src, 16) || !aligned(guard, 4) || (counter && !aligned(counter, 4))) return AFAN_EALIGN;
    const int64_t n16 = bytes / 16;
    guarded_copy_kernel<<<grid_for(n16, BLOCK), BLOCK, 0, (hipStream_t)stream>>>((uint4*)dst, (const uint4*)src, n16, (const unsigned*)guard,
                                                                                (unsigned*)counter);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// Test / probe aid (no reference counterpart): park `workgroups` single-wave workgroups with `lds_bytes` of LDS each on `stream` for
// `microseconds` — what an RCCL channel kernel or a side-stream weight gradient looks like to the launches beside it
// (tests/test_grid_guard_gpu.py forces a grid barrier to give up with it; tools/probe/cu_sharing.py measures the slowdown of the
// 256-workgroup launches beside 8 / 16 / 32 of them).
extern "C" int afan_occupy_cus(int workgroups, int lds_bytes, int microseconds, afan_stream_t stream) {
    if (workgroups < 0 || lds_bytes < 256 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 2000000) return AFAN_ESHAPE;
    if (workgroups == 0) return AFAN_OK;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    occupy_kernel<<<workgroups, 64, (size_t)lds_bytes, (hipStream_t)stream>>>((long long)microseconds * 100LL);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}
