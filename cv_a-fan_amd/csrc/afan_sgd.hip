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
                                                    float wd, float gscale, int vec) {
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

extern "C" int afan_sgd_step(float* param, const float* grad, float* momentum_buf, uint16_t* shadow_bf16,
                             int64_t n, const float* lr_dev, float momentum, float weight_decay,
                             float grad_scale, int first_step, afan_stream_t stream) {
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
    sgd_kernel<F, S><<<grid, BLOCK, 0, st>>>(param, grad, momentum_buf, shadow_bf16, n, lr_dev, momentum, weight_decay, grad_scale, vec)
    if (first_step) { if (shadow_bf16) AFAN_GO(true, true); else AFAN_GO(true, false); }
    else { if (shadow_bf16) AFAN_GO(false, true); else AFAN_GO(false, false); }
#undef AFAN_GO
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}
