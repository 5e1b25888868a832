// Training-mode BatchNorm for NCHW activations on gfx950 — the "feature-norm stats" kernels.
// The A-FAN step runs every tail BN K+2 times and every head BN twice per iteration
// (reference: Classification/main_perturb.py:173,195-196 + attack_algo.py:50 through nn.BatchNorm2d,
// resnet_s.py:52-55,88), so BN is the bandwidth-bound part of the step.
//
// Mapping: grid = (S slices, C channels).  Workgroup (s, c) streams every S-th 256-vector chunk of
// channel c's N*HW elements with 16-byte accesses (lanes run along W, so NCHW planes are read
// coalesced), keeps shifted sums in registers, folds them with wave64 shuffles, then across the 4
// waves through LDS, and writes ONE partial per workgroup.  The second launch uses the SAME
// (s, c) -> data mapping: its prologue folds the <= 64 partials of its channel (L2-resident) with
// one wave, so no separate finalize launch and no float atomics (bitwise reproducible).
//   forward : stats partials -> [fold] normalise + affine (+ residual) (+ ReLU), running stats by s==0
//   backward: (sum g, sum g*xhat) partials -> [fold] dx (+ d_residual), dweight/dbias by s==0
#include "afan_common.h"

using namespace afan;

namespace afan_nhwc {  // afan_bn_nhwc.hip
int fwd(int dtype, const void* x, const void* res, void* y, int64_t M, int64_t C, float eps, float mom, const float* w,
        const float* b, int relu, float* ws, float* stats, const float* mean_in, const float* invstd_in, float* rm,
        float* rv, int64_t* nbt, bool train, hipStream_t st, const float* partials, int64_t partials_g,
        const float* partials_shift);
int stats(int dtype, const void* x, int64_t M, int64_t C, float eps, float mom, float* ws, float* stats_out, float* rm,
          float* rv, int64_t* nbt, hipStream_t st);
int bwd(int dtype, const void* dy, const void* x, const void* y, void* dx, void* dres, int64_t M, int64_t C,
        const float* stats_in, int relu, float* ws, float* dw, float* db, int acc, hipStream_t st,
        const float* partials, int64_t partials_g);
int64_t workspace_floats(int64_t c);
int fwd_acc(int dtype, const void* x, const void* res, void* y, int64_t M, int64_t C, float eps, float mom,
            const float* w, const float* b, int relu, double* acc, int acc_ready, float* stats, float* rm, float* rv,
            int64_t* nbt, hipStream_t st, int groups);
int bwd_acc(int dtype, const void* dy, const void* x, const void* y, void* dx, void* dres, int64_t M, int64_t C,
            const float* stats_in, int relu, double* acc, int acc_ready, float* dw, float* db, int accumulate,
            hipStream_t st, int groups);
int acc_supported(int dtype, int64_t C);
int64_t acc_doubles(int64_t C);
int set_running_updates(int n);
int fwd_acc_dual(int dtype, const void* xa, const void* xb, void* y, int64_t M, int64_t C, float eps_a, float mom_a,
                 const float* w_a, const float* b_a, double* acc_a, float* stats_a, float* rm_a, float* rv_a, int64_t* nbt_a,
                 float eps_b, float mom_b, const float* w_b, const float* b_b, double* acc_b, float* stats_b, float* rm_b,
                 float* rv_b, int64_t* nbt_b, hipStream_t st);
int coefs(int64_t C, const float* mean, const float* invstd, const float* w, const float* b, float* out, hipStream_t st);
}

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_SLICES = 64;
constexpr int WS_STRIDE = 4;  // floats per partial

struct Geo {
    uint32_t hwv;      // vectors per (n,c) plane
    uint32_t V;        // vectors per channel = N*hwv
    int64_t plane;     // HW elements
    int64_t nstride;   // C*HW elements
    int shift;         // log2(hwv) if power of two else -1
};

__device__ __forceinline__ int64_t vec_offset(const Geo& g, uint32_t v, int c, int vec) {
    uint32_t n, j;
    if (g.shift >= 0) {
        n = v >> g.shift;
        j = v & (g.hwv - 1);
    } else {
        n = v / g.hwv;
        j = v - n * g.hwv;
    }
    return (int64_t)n * g.nstride + (int64_t)c * g.plane + (int64_t)j * vec;
}

template <typename T, int VEC> struct Ld {
    __device__ static __forceinline__ void ld(const T* p, float (&v)[VEC]) {
        if constexpr (VEC == 1) v[0] = Elt<T>::ld(p);
        else Elt<T>::ldv(p, v);
    }
    __device__ static __forceinline__ void st(T* p, const float (&v)[VEC]) {
        if constexpr (VEC == 1) Elt<T>::st(p, v[0]);
        else Elt<T>::stv(p, v);
    }
};

// alpha/beta of y = x*alpha + beta; ONE definition so forward and the backward mask recompute agree bitwise
__device__ __forceinline__ void affine_coeffs(float mu, float is, const float* weight, const float* bias,
                                              int c, float& alpha, float& beta) {
    alpha = is * (weight ? weight[c] : 1.f);
    beta = fmaf(-mu, alpha, bias ? bias[c] : 0.f);
}

// ---- fold helpers (one wave folds the channel's partials, result broadcast through LDS) ---------
__device__ __forceinline__ Moments fold_moments(const float* __restrict__ ws, int c, int S) {
    __shared__ float sh[3];
    if (threadIdx.x < AFAN_WAVE) {
        Moments m{0.f, 0.f, 0.f};
        if ((int)threadIdx.x < S) {
            const float* p = ws + ((int64_t)c * MAX_SLICES + threadIdx.x) * WS_STRIDE;
            m.n = p[0]; m.mean = p[1]; m.m2 = p[2];
        }
        m = wave_merge(m);
        if (threadIdx.x == 0) { sh[0] = m.n; sh[1] = m.mean; sh[2] = m.m2; }
    }
    __syncthreads();
    return Moments{sh[0], sh[1], sh[2]};
}

__device__ __forceinline__ void fold_sums(const float* __restrict__ ws, int c, int S, float& a, float& b) {
    __shared__ float sh[2];
    if (threadIdx.x < AFAN_WAVE) {
        float x = 0.f, y = 0.f;
        if ((int)threadIdx.x < S) {
            const float* p = ws + ((int64_t)c * MAX_SLICES + threadIdx.x) * WS_STRIDE;
            x = p[0]; y = p[1];
        }
        x = wave_sum(x);
        y = wave_sum(y);
        if (threadIdx.x == 0) { sh[0] = x; sh[1] = y; }
    }
    __syncthreads();
    a = sh[0];
    b = sh[1];
}

// ---- forward launch 1: per-(channel, slice) moments ----------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(BLOCK) void bn_stats_kernel(const T* __restrict__ x, Geo g,
                                                         float* __restrict__ ws) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    float shift = 0.f, sum = 0.f, ssq = 0.f, cnt = 0.f;
    bool first = true;
#pragma unroll 2
    for (uint32_t v = s * BLOCK + threadIdx.x; v < g.V; v += S * BLOCK) {
        float e[VEC];
        Ld<T, VEC>::ld(x + vec_offset(g, v, c, VEC), e);
        if (first) { shift = e[0]; first = false; }  // shifted sums: no cancellation when |mean| >> std
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float d = e[k] - shift;
            sum += d;
            ssq += d * d;
        }
        cnt += (float)VEC;
    }
    Moments m{cnt, 0.f, 0.f};
    if (cnt > 0.f) {
        float dm = sum / cnt;
        m.mean = shift + dm;
        m.m2 = fmaxf(ssq - sum * dm, 0.f);
    }
    m = wave_merge(m);
    __shared__ float sh[BLOCK / AFAN_WAVE][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[w][0] = m.n; sh[w][1] = m.mean; sh[w][2] = m.m2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        Moments t{sh[0][0], sh[0][1], sh[0][2]};
#pragma unroll
        for (int k = 1; k < BLOCK / AFAN_WAVE; ++k) t = merge(t, Moments{sh[k][0], sh[k][1], sh[k][2]});
        float* p = ws + ((int64_t)c * MAX_SLICES + s) * WS_STRIDE;
        p[0] = t.n; p[1] = t.mean; p[2] = t.m2;
    }
}

// ---- stand-alone finalize (afan_bn_stats) --------------------------------------------------------
__device__ __forceinline__ void write_stats(const Moments& m, int c, float eps, float momentum,
                                            float* mean, float* invstd, float* rmean, float* rvar) {
    const float var_b = m.m2 / m.n;
    mean[c] = m.mean;
    invstd[c] = 1.0f / sqrtf(var_b + eps);
    if (rmean) {
        const float var_u = m.m2 / (m.n - 1.0f);  // torch uses the unbiased estimate for running_var
        rmean[c] = (1.0f - momentum) * rmean[c] + momentum * m.mean;
        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * var_u;
    }
}

__global__ __launch_bounds__(AFAN_WAVE) void bn_finalize_kernel(const float* __restrict__ ws, int S,
                                                                float eps, float momentum,
                                                                float* mean, float* invstd, float* rmean,
                                                                float* rvar, int64_t* nbt) {
    const int c = blockIdx.x;
    Moments m = fold_moments(ws, c, S);
    if (threadIdx.x == 0) {
        write_stats(m, c, eps, momentum, mean, invstd, rmean, rvar);
        if (c == 0 && nbt) *nbt += 1;
    }
}

// ---- forward launch 2: fold + apply ---------------------------------------------------------------
template <typename T, int VEC, bool RES, bool RELU, bool TRAIN>
__global__ __launch_bounds__(BLOCK) void bn_apply_kernel(const T* __restrict__ x,
                                                         const T* __restrict__ res, T* __restrict__ y,
                                                         Geo g, const float* __restrict__ ws, float eps,
                                                         float momentum, const float* __restrict__ weight,
                                                         const float* __restrict__ bias, float* mean,
                                                         float* invstd, float* rmean, float* rvar,
                                                         int64_t* nbt) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    float mu, is;
    if (TRAIN) {
        Moments m = fold_moments(ws, c, S);
        mu = m.mean;
        is = 1.0f / sqrtf(m.m2 / m.n + eps);
        if (s == 0 && threadIdx.x == 0) {
            write_stats(m, c, eps, momentum, mean, invstd, rmean, rvar);
            if (c == 0 && nbt) *nbt += 1;
        }
    } else {
        mu = mean[c];
        is = invstd[c];
    }
    float alpha, beta;
    affine_coeffs(mu, is, weight, bias, c, alpha, beta);
#pragma unroll 2
    for (uint32_t v = s * BLOCK + threadIdx.x; v < g.V; v += S * BLOCK) {
        const int64_t off = vec_offset(g, v, c, VEC);
        float e[VEC], r[VEC];
        Ld<T, VEC>::ld(x + off, e);
        if (RES) Ld<T, VEC>::ld(res + off, r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float t = fmaf(e[k], alpha, beta);
            if (RES) t += r[k];
            if (RELU) t = (t > 0.f) ? t : ((t != t) ? t : 0.f);  // NaN propagates like torch.relu
            e[k] = t;
        }
        Ld<T, VEC>::st(y + off, e);
    }
}

// ---- backward launch 1: per-(channel, slice) sums of g and g*xhat ---------------------------------
// mask source: y (when given) else recomputed fmaf(x,alpha,beta) exactly as the forward did.
template <typename T, int VEC, bool RELU, bool HAVE_Y>
__global__ __launch_bounds__(BLOCK) void bn_bwd_reduce_kernel(const T* __restrict__ dy,
                                                              const T* __restrict__ x,
                                                              const T* __restrict__ y, Geo g,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ weight,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ ws) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const float mu = mean[c], is = invstd[c];
    float alpha, beta;
    affine_coeffs(mu, is, weight, bias, c, alpha, beta);
    float sg = 0.f, sgx = 0.f;
#pragma unroll 2
    for (uint32_t v = s * BLOCK + threadIdx.x; v < g.V; v += S * BLOCK) {
        const int64_t off = vec_offset(g, v, c, VEC);
        float d[VEC], e[VEC], o[VEC];
        Ld<T, VEC>::ld(dy + off, d);
        Ld<T, VEC>::ld(x + off, e);
        if (RELU && HAVE_Y) Ld<T, VEC>::ld(y + off, o);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float gk = d[k];
            if (RELU) {
                const float act = HAVE_Y ? o[k] : fmaf(e[k], alpha, beta);
                gk = (act > 0.f) ? gk : 0.f;
            }
            sg += gk;
            sgx += gk * ((e[k] - mu) * is);
        }
    }
    sg = wave_sum(sg);
    sgx = wave_sum(sgx);
    __shared__ float sh[BLOCK / AFAN_WAVE][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[w][0] = sg; sh[w][1] = sgx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < BLOCK / AFAN_WAVE; ++k) { a += sh[k][0]; b += sh[k][1]; }
        float* p = ws + ((int64_t)c * MAX_SLICES + s) * WS_STRIDE;
        p[0] = a; p[1] = b;
    }
}

// ---- backward launch 2: fold + dx (+ d_residual) ---------------------------------------------------
template <typename T, int VEC, bool RELU, bool HAVE_Y, bool DRES>
__global__ __launch_bounds__(BLOCK) void bn_bwd_apply_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ dx,
    T* __restrict__ dres, Geo g, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ weight, const float* __restrict__ bias, const float* __restrict__ ws,
    float inv_m, float* dweight, float* dbias, int accumulate) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    float sum_g, sum_gx;
    fold_sums(ws, c, S, sum_g, sum_gx);
    if (s == 0 && threadIdx.x == 0) {
        if (dweight) dweight[c] = accumulate ? dweight[c] + sum_gx : sum_gx;
        if (dbias) dbias[c] = accumulate ? dbias[c] + sum_g : sum_g;
    }
    const float mu = mean[c], is = invstd[c];
    float alpha, beta;
    affine_coeffs(mu, is, weight, bias, c, alpha, beta);
    const float k1 = sum_g * inv_m, k2 = sum_gx * inv_m;
#pragma unroll 2
    for (uint32_t v = s * BLOCK + threadIdx.x; v < g.V; v += S * BLOCK) {
        const int64_t off = vec_offset(g, v, c, VEC);
        float d[VEC], e[VEC], o[VEC];
        Ld<T, VEC>::ld(dy + off, d);
        Ld<T, VEC>::ld(x + off, e);
        if (RELU && HAVE_Y) Ld<T, VEC>::ld(y + off, o);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float gk = d[k];
            if (RELU) {
                const float act = HAVE_Y ? o[k] : fmaf(e[k], alpha, beta);
                gk = (act > 0.f) ? gk : 0.f;
            }
            d[k] = gk;
            const float xh = (e[k] - mu) * is;
            e[k] = (gk - k1 - xh * k2) * alpha;
        }
        Ld<T, VEC>::st(dx + off, e);
        if (DRES) Ld<T, VEC>::st(dres + off, d);
    }
}

// ---- per-channel input normalisation (resnet_s.py:87) ----------------------------------------------
// i indexes the OUTPUT; NHWC output gathers from the NCHW input image
template <typename TO, bool NHWC_OUT>
__global__ __launch_bounds__(BLOCK) void normalize_kernel(const float* __restrict__ x, TO* __restrict__ y,
                                                          int64_t total, int64_t c, int64_t hw,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ std) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        int64_t ch, src;
        if (NHWC_OUT) {
            ch = i % c;
            const int64_t pix = i / c;            // n*hw + p
            src = (pix / hw) * c * hw + ch * hw + (pix % hw);
        } else {
            ch = (i / hw) % c;
            src = i;
        }
        Elt<TO>::st(y + i, (x[src] - mean[ch]) / std[ch]);
    }
}

// ---- host side ------------------------------------------------------------------------------------
struct Plan {
    Geo g;
    int vec;  // elements per access
    dim3 grid;
    double tensor_bytes;  // bytes of one full activation tensor (algorithmic-traffic unit for the profiler)
};

template <typename T>
bool make_plan(int64_t n, int64_t c, int64_t hw, std::initializer_list<const void*> ptrs, Plan& p) {
    constexpr int NV = Elt<T>::VEC;
    bool vec_ok = (hw % NV == 0);
    for (const void* q : ptrs)
        if (q && !aligned(q, 16)) vec_ok = false;
    p.vec = vec_ok ? NV : 1;
    const int64_t hwv = hw / p.vec;
    const int64_t V = n * hwv;
    if (V <= 0 || V > 0x7fffffffLL || c > 65535) return false;
    p.g.hwv = (uint32_t)hwv;
    p.g.V = (uint32_t)V;
    p.g.plane = hw;
    p.g.nstride = c * hw;
    p.g.shift = ((hwv & (hwv - 1)) == 0) ? __builtin_ctzll((unsigned long long)hwv) : -1;
    // enough workgroups to fill 256 CUs x 8, at most MAX_SLICES per channel, >= 1 chunk each
    int64_t S = (2048 + c - 1) / c;
    const int64_t chunks = (V + BLOCK - 1) / BLOCK;
    if (S > chunks) S = chunks;
    if (S > MAX_SLICES) S = MAX_SLICES;
    if (S < 1) S = 1;
    p.grid = dim3((unsigned)S, (unsigned)c);
    p.tensor_bytes = (double)n * (double)c * (double)hw * sizeof(T);
    return true;
}

template <typename T>
int bn_stats_impl(const void* x, int64_t n, int64_t c, int64_t hw, float eps, float momentum, float* ws,
                  float* mean, float* invstd, float* rmean, float* rvar, int64_t* nbt, hipStream_t st) {
    Plan p;
    if (!make_plan<T>(n, c, hw, {x}, p)) return AFAN_ESHAPE;
    {
        AFAN_PROF("bn_stats_kernel", p.tensor_bytes, st);
        if (p.vec == 1)
            bn_stats_kernel<T, 1><<<p.grid, BLOCK, 0, st>>>((const T*)x, p.g, ws);
        else
            bn_stats_kernel<T, Elt<T>::VEC><<<p.grid, BLOCK, 0, st>>>((const T*)x, p.g, ws);
    }
    AFAN_LAUNCH_CHECK();
    AFAN_PROF("bn_finalize_kernel", 16.0 * c * p.grid.x, st);
    bn_finalize_kernel<<<(unsigned)c, AFAN_WAVE, 0, st>>>(ws, (int)p.grid.x, eps, momentum, mean, invstd,
                                                          rmean, rvar, nbt);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <typename T, int VEC, bool TRAIN>
void launch_apply(const Plan& p, const void* x, const void* res, void* y, const float* ws, float eps,
                  float momentum, const float* weight, const float* bias, float* mean, float* invstd,
                  float* rmean, float* rvar, int64_t* nbt, int relu, hipStream_t st) {
    AFAN_PROF("bn_apply_kernel", p.tensor_bytes * (res ? 3 : 2), st);
#define AFAN_GO(RES, RELU)                                                                          \
    bn_apply_kernel<T, VEC, RES, RELU, TRAIN><<<p.grid, BLOCK, 0, st>>>(                            \
        (const T*)x, (const T*)res, (T*)y, p.g, ws, eps, momentum, weight, bias, mean, invstd, rmean, \
        rvar, nbt)
    if (res) {
        if (relu) AFAN_GO(true, true); else AFAN_GO(true, false);
    } else {
        if (relu) AFAN_GO(false, true); else AFAN_GO(false, false);
    }
#undef AFAN_GO
    ++afan::prof::g_launches;       // (the caller checks the launch; the profiling scope above counts it here)
}

template <typename T>
int bn_forward_impl(const void* x, const void* res, void* y, int64_t n, int64_t c, int64_t hw, float eps,
                    float momentum, const float* weight, const float* bias, int relu, float* ws,
                    float* mean, float* invstd, float* rmean, float* rvar, int64_t* nbt, bool train,
                    hipStream_t st) {
    Plan p;
    if (!make_plan<T>(n, c, hw, {x, res, y}, p)) return AFAN_ESHAPE;
    constexpr int NV = Elt<T>::VEC;
    if (train) {
        {
            AFAN_PROF("bn_stats_kernel", p.tensor_bytes, st);
            if (p.vec == 1) bn_stats_kernel<T, 1><<<p.grid, BLOCK, 0, st>>>((const T*)x, p.g, ws);
            else bn_stats_kernel<T, NV><<<p.grid, BLOCK, 0, st>>>((const T*)x, p.g, ws);
        }
        AFAN_LAUNCH_CHECK();
        if (p.vec == 1) launch_apply<T, 1, true>(p, x, res, y, ws, eps, momentum, weight, bias, mean, invstd, rmean, rvar, nbt, relu, st);
        else launch_apply<T, NV, true>(p, x, res, y, ws, eps, momentum, weight, bias, mean, invstd, rmean, rvar, nbt, relu, st);
    } else {
        if (p.vec == 1) launch_apply<T, 1, false>(p, x, res, y, ws, eps, momentum, weight, bias, mean, invstd, rmean, rvar, nbt, relu, st);
        else launch_apply<T, NV, false>(p, x, res, y, ws, eps, momentum, weight, bias, mean, invstd, rmean, rvar, nbt, relu, st);
    }
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <typename T, int VEC>
int bn_backward_vec(const Plan& p, const void* dy, const void* x, const void* y, void* dx, void* dres,
                    const float* mean, const float* invstd, const float* weight, const float* bias,
                    int relu, float* ws, float inv_m, float* dweight, float* dbias, int accumulate,
                    hipStream_t st) {
    const T* dy_ = (const T*)dy; const T* x_ = (const T*)x; const T* y_ = (const T*)y;
#define AFAN_RED(RELU, HY) \
    bn_bwd_reduce_kernel<T, VEC, RELU, HY><<<p.grid, BLOCK, 0, st>>>(dy_, x_, y_, p.g, mean, invstd, weight, bias, ws)
    {
        AFAN_PROF("bn_bwd_reduce_kernel", p.tensor_bytes * ((relu && y) ? 3 : 2), st);
        if (!relu) AFAN_RED(false, false);
        else if (y) AFAN_RED(true, true);
        else AFAN_RED(true, false);
    }
#undef AFAN_RED
    AFAN_LAUNCH_CHECK();
#define AFAN_APP(RELU, HY, DR)                                                                         \
    bn_bwd_apply_kernel<T, VEC, RELU, HY, DR><<<p.grid, BLOCK, 0, st>>>(dy_, x_, y_, (T*)dx, (T*)dres, p.g, \
        mean, invstd, weight, bias, ws, inv_m, dweight, dbias, accumulate)
    AFAN_PROF("bn_bwd_apply_kernel", p.tensor_bytes * (3 + ((relu && y) ? 1 : 0) + (dres ? 1 : 0)), st);
    if (!relu) { if (dres) AFAN_APP(false, false, true); else AFAN_APP(false, false, false); }
    else if (y) { if (dres) AFAN_APP(true, true, true); else AFAN_APP(true, true, false); }
    else { if (dres) AFAN_APP(true, false, true); else AFAN_APP(true, false, false); }
#undef AFAN_APP
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <typename T>
int bn_backward_impl(const void* dy, const void* x, const void* y, void* dx, void* dres, int64_t n,
                     int64_t c, int64_t hw, const float* mean, const float* invstd, const float* weight,
                     const float* bias, int relu, float* ws, float* dweight, float* dbias, int accumulate,
                     hipStream_t st) {
    Plan p;
    if (!make_plan<T>(n, c, hw, {dy, x, y, dx, dres}, p)) return AFAN_ESHAPE;
    const float inv_m = 1.0f / (float)((double)n * (double)hw);
    if (p.vec == 1)
        return bn_backward_vec<T, 1>(p, dy, x, y, dx, dres, mean, invstd, weight, bias, relu, ws, inv_m, dweight, dbias, accumulate, st);
    return bn_backward_vec<T, Elt<T>::VEC>(p, dy, x, y, dx, dres, mean, invstd, weight, bias, relu, ws, inv_m, dweight, dbias, accumulate, st);
}

int check_layout(int layout) { return (layout == AFAN_NCHW || layout == AFAN_NHWC) ? AFAN_OK : AFAN_ELAYOUT; }

int check_common(int dtype, int64_t n, int64_t c, int64_t hw) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (n <= 0 || c <= 0 || hw <= 0) return AFAN_ESHAPE;
    return AFAN_OK;
}

// One more running-statistics update of a train-mode BatchNorm from the moments of an EARLIER forward pass (its saved
// statistics: mean | invstd): the step runs the clean tail pass once where main_perturb.py runs it twice (first PGD pass and
// the final clean forward, attack_algo.py:50 / main_perturb.py:196) — the second pass's only other effect is this update,
// which comes last in the reference's order.  The biased variance is recovered as 1/invstd^2 - eps (relative error ~1e-6).
// the same for up to RU_MAX layers in one launch (block = layer): a network's tail has 15-50 BatchNorms
struct RUEntry { const float* stats; float* rm; float* rv; int64_t* nbt; int c; float unbias, eps, momentum; };
constexpr int RU_MAX = 64;
struct RUBatch { RUEntry e[RU_MAX]; int n; };
__global__ __launch_bounds__(256) void running_update_batched_kernel(const RUBatch b) {
    const RUEntry& e = b.e[blockIdx.x];
    for (int c = threadIdx.x; c < e.c; c += blockDim.x) {
        const float mean = e.stats[c], is = e.stats[e.c + c];
        // biased variance back from the saved 1/sqrt(var + eps), in f64 (no second rounding on top of invstd's own: what is
        // left is invstd's fp32 ulp, ~1.2e-7 * (var + eps) absolute — relevant only for channels with var << eps)
        const float varb = (float)fmax(1.0 / ((double)is * (double)is) - (double)e.eps, 0.0);
        e.rm[c] = (1.0f - e.momentum) * e.rm[c] + e.momentum * mean;
        e.rv[c] = (1.0f - e.momentum) * e.rv[c] + e.momentum * (varb * e.unbias);
    }
    if (threadIdx.x == 0 && e.nbt) *e.nbt += 1;
}

__global__ void running_update_kernel(const float* __restrict__ stats, int C, float eps, float momentum, float unbias,
                                      float* __restrict__ rmean, float* __restrict__ rvar, int64_t* nbt) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float mean = stats[c], is = stats[C + c];
        const float varb = (float)fmax(1.0 / ((double)is * (double)is) - (double)eps, 0.0);
        rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (varb * unbias);
    }
    if (c == 0 && nbt) *nbt += 1;
}

}  // namespace

extern "C" {

int64_t afan_bn_workspace_floats(int64_t c) {
    if (c <= 0) return 0;
    const int64_t a = c * MAX_SLICES * WS_STRIDE, b = afan_nhwc::workspace_floats(c);
    return a > b ? a : b;
}

int afan_bn_stats(const void* x, int dtype, int layout, int64_t n, int64_t c, int64_t hw, float eps, float momentum,
                  float* workspace, float* stats, float* running_mean, float* running_var, int64_t* num_batches,
                  afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if ((e = check_layout(layout))) return e;
    if (!x || !workspace || !stats) return AFAN_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return AFAN_ENULL;
    if (!aligned(x, dtype == AFAN_F32 ? 4 : 2)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (layout == AFAN_NHWC)
        return afan_nhwc::stats(dtype, x, n * hw, c, eps, momentum, workspace, stats, running_mean, running_var, num_batches, st);
    if (dtype == AFAN_F32)
        return bn_stats_impl<float>(x, n, c, hw, eps, momentum, workspace, stats, stats + c, running_mean, running_var, num_batches, st);
    return bn_stats_impl<uint16_t>(x, n, c, hw, eps, momentum, workspace, stats, stats + c, running_mean, running_var, num_batches, st);
}

int afan_bn_train_forward(const void* x, const void* residual, void* y, int dtype, int layout, int64_t n, int64_t c,
                          int64_t hw, float eps, float momentum, const float* weight, const float* bias, int relu,
                          float* workspace, float* save_stats, float* running_mean, float* running_var,
                          int64_t* num_batches, afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if ((e = check_layout(layout))) return e;
    if (!x || !y || !workspace || !save_stats) return AFAN_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return AFAN_ENULL;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, a) || !aligned(y, a) || (residual && !aligned(residual, a))) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (layout == AFAN_NHWC)
        return afan_nhwc::fwd(dtype, x, residual, y, n * hw, c, eps, momentum, weight, bias, relu, workspace, save_stats,
                              nullptr, nullptr, running_mean, running_var, num_batches, true, st, nullptr, 0, nullptr);
    {   // the NCHW kernels apply one running-statistics update per pass: refuse a pending repeat count loudly
        const int pending = afan_nhwc::set_running_updates(1);
        afan_nhwc::set_running_updates(pending);
        if (pending != 1) return AFAN_ESHAPE;
    }
    if (dtype == AFAN_F32)
        return bn_forward_impl<float>(x, residual, y, n, c, hw, eps, momentum, weight, bias, relu, workspace, save_stats, save_stats + c, running_mean, running_var, num_batches, true, st);
    return bn_forward_impl<uint16_t>(x, residual, y, n, c, hw, eps, momentum, weight, bias, relu, workspace, save_stats, save_stats + c, running_mean, running_var, num_batches, true, st);
}

int afan_bn_train_forward_partials(const void* x, const void* residual, void* y, int dtype, int64_t n, int64_t c,
                                   int64_t hw, float eps, float momentum, const float* weight, const float* bias,
                                   int relu, const float* partials, int64_t partials_g, const float* partials_shift,
                                   float* save_stats, float* running_mean, float* running_var, int64_t* num_batches,
                                   afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if (!x || !y || !partials || !save_stats) return AFAN_ENULL;
    if (partials_g <= 0) return AFAN_ESHAPE;
    if ((running_mean == nullptr) != (running_var == nullptr)) return AFAN_ENULL;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, a) || !aligned(y, a) || (residual && !aligned(residual, a))) return AFAN_EALIGN;
    return afan_nhwc::fwd(dtype, x, residual, y, n * hw, c, eps, momentum, weight, bias, relu, nullptr, save_stats, nullptr,
                          nullptr, running_mean, running_var, num_batches, true, (hipStream_t)stream, partials,
                          partials_g, partials_shift);
}

int afan_bn_apply(const void* x, const void* residual, void* y, int dtype, int layout, int64_t n, int64_t c, int64_t hw,
                  const float* mean, const float* invstd, const float* weight, const float* bias, int relu,
                  float* workspace, afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if ((e = check_layout(layout))) return e;
    if (!x || !y || !mean || !invstd) return AFAN_ENULL;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, a) || !aligned(y, a) || (residual && !aligned(residual, a))) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    float* m = const_cast<float*>(mean);
    float* is = const_cast<float*>(invstd);
    if (layout == AFAN_NHWC) {
        if (!workspace) return AFAN_ENULL;
        float* stats = workspace + afan_nhwc::workspace_floats(c) - 4 * c;   // tail of the workspace
        return afan_nhwc::fwd(dtype, x, residual, y, n * hw, c, 0.f, 0.f, weight, bias, relu, workspace, stats, mean, invstd,
                              nullptr, nullptr, nullptr, false, st, nullptr, 0, nullptr);
    }
    if (dtype == AFAN_F32)
        return bn_forward_impl<float>(x, residual, y, n, c, hw, 0.f, 0.f, weight, bias, relu, nullptr, m, is, nullptr, nullptr, nullptr, false, st);
    return bn_forward_impl<uint16_t>(x, residual, y, n, c, hw, 0.f, 0.f, weight, bias, relu, nullptr, m, is, nullptr, nullptr, nullptr, false, st);
}

int afan_affine_coefs(const float* mean, const float* invstd, const float* weight, const float* bias, int64_t c,
                      float* coefs, afan_stream_t stream) {
    if (c <= 0) return AFAN_ESHAPE;
    if (!mean || !invstd || !coefs) return AFAN_ENULL;
    if (!aligned(coefs, 16)) return AFAN_EALIGN;
    return afan_nhwc::coefs(c, mean, invstd, weight, bias, coefs, (hipStream_t)stream);
}

int afan_affine_apply(const void* x, const void* residual, void* y, int dtype, int64_t n, int64_t c, int64_t hw,
                      const float* coefs, int relu, afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if (!x || !y || !coefs) return AFAN_ENULL;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, a) || !aligned(y, a) || (residual && !aligned(residual, a))) return AFAN_EALIGN;
    return afan_nhwc::fwd(dtype, x, residual, y, n * hw, c, 0.f, 0.f, nullptr, nullptr, relu, nullptr, const_cast<float*>(coefs),
                          nullptr, nullptr, nullptr, nullptr, nullptr, false, (hipStream_t)stream, nullptr, 0, nullptr);
}

int afan_bn_backward(const void* dy, const void* x, const void* y, void* dx, void* d_residual, int dtype, int layout,
                     int64_t n, int64_t c, int64_t hw, const float* save_stats, const float* weight, const float* bias,
                     int relu, float* workspace, float* dweight, float* dbias, int accumulate, const float* partials,
                     int64_t partials_g, afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if ((e = check_layout(layout))) return e;
    if (!dy || !x || !dx || !save_stats || !workspace) return AFAN_ENULL;
    if (partials && (layout != AFAN_NHWC || partials_g <= 0)) return AFAN_ESHAPE;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(dy, a) || !aligned(x, a) || !aligned(dx, a) || (y && !aligned(y, a)) ||
        (d_residual && !aligned(d_residual, a)))
        return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (layout == AFAN_NHWC)
        return afan_nhwc::bwd(dtype, dy, x, y, dx, d_residual, n * hw, c, save_stats, relu, workspace, dweight, dbias, accumulate, st,
                              partials, partials_g);
    const float* mean = save_stats;
    const float* invstd = save_stats + c;
    if (dtype == AFAN_F32)
        return bn_backward_impl<float>(dy, x, y, dx, d_residual, n, c, hw, mean, invstd, weight, bias, relu, workspace, dweight, dbias, accumulate, st);
    return bn_backward_impl<uint16_t>(dy, x, y, dx, d_residual, n, c, hw, mean, invstd, weight, bias, relu, workspace, dweight, dbias, accumulate, st);
}

int64_t afan_bn_acc_doubles(int64_t c) { return afan_nhwc::acc_doubles(c); }

int afan_bn_running_update_batched(const float* const* stats, float* const* running_mean, float* const* running_var,
                                   int64_t* const* num_batches, const int64_t* c, const double* m_count, const float* eps,
                                   const float* momentum, int n, afan_stream_t stream) {
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!stats || !running_mean || !running_var || !num_batches || !c || !m_count || !eps || !momentum) return AFAN_ENULL;
    for (int i0 = 0; i0 < n; i0 += RU_MAX) {
        RUBatch b{};
        b.n = n - i0 < RU_MAX ? n - i0 : RU_MAX;
        for (int i = 0; i < b.n; ++i) {
            const int j = i0 + i;
            if (c[j] <= 0 || m_count[j] < 1.0) return AFAN_ESHAPE;
            if (!stats[j] || !running_mean[j] || !running_var[j]) return AFAN_ENULL;
            b.e[i] = RUEntry{stats[j], running_mean[j], running_var[j], num_batches[j], (int)c[j],
                             m_count[j] > 1.0 ? (float)(m_count[j] / (m_count[j] - 1.0)) : 1.0f, eps[j], momentum[j]};
        }
        running_update_batched_kernel<<<(unsigned)b.n, 256, 0, (hipStream_t)stream>>>(b);
        AFAN_LAUNCH_CHECK();
    }
    return AFAN_OK;
}

int afan_bn_set_running_updates(int n) { return afan_nhwc::set_running_updates(n); }

int afan_bn_running_update(const float* stats, int64_t c, double m_count, float eps, float momentum, float* running_mean,
                           float* running_var, int64_t* num_batches, afan_stream_t stream) {
    if (c <= 0 || m_count < 1.0) return AFAN_ESHAPE;
    if (!stats || !running_mean || !running_var) return AFAN_ENULL;
    const float unbias = m_count > 1.0 ? (float)(m_count / (m_count - 1.0)) : 1.0f;
    running_update_kernel<<<(unsigned)((c + 255) / 256), 256, 0, (hipStream_t)stream>>>(stats, (int)c, eps, momentum, unbias,
                                                                                       running_mean, running_var, num_batches);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_bn_acc_supported(int dtype, int64_t c) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return 0;
    return afan_nhwc::acc_supported(dtype, c);
}

int afan_bn_train_forward_acc(const void* x, const void* residual, void* y, int dtype, int64_t n, int64_t c,
                              int64_t hw, float eps, float momentum, const float* weight, const float* bias, int relu,
                              double* acc, int acc_ready, float* save_stats, float* running_mean, float* running_var,
                              int64_t* num_batches, int groups, afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if (!x || !y || !acc || !save_stats) return AFAN_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return AFAN_ENULL;
    if (!afan_nhwc::acc_supported(dtype, c)) return AFAN_ESHAPE;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x, a) || !aligned(y, a) || (residual && !aligned(residual, a)) || !aligned(acc, 16)) return AFAN_EALIGN;
    if (groups < 1 || n % groups != 0) return AFAN_ESHAPE;
    return afan_nhwc::fwd_acc(dtype, x, residual, y, (n / groups) * hw, c, eps, momentum, weight, bias, relu, acc, acc_ready,
                              save_stats, running_mean, running_var, num_batches, (hipStream_t)stream, groups);
}

int afan_bn_train_forward_acc_dual(const void* x_a, const void* x_b, void* y, int dtype, int64_t n, int64_t c, int64_t hw,
                                   float eps_a, float momentum_a, const float* weight_a, const float* bias_a, double* acc_a,
                                   float* save_stats_a, float* running_mean_a, float* running_var_a, int64_t* num_batches_a,
                                   float eps_b, float momentum_b, const float* weight_b, const float* bias_b, double* acc_b,
                                   float* save_stats_b, float* running_mean_b, float* running_var_b, int64_t* num_batches_b,
                                   afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if (!x_a || !x_b || !y || !acc_a || !acc_b || !save_stats_a || !save_stats_b) return AFAN_ENULL;
    if ((running_mean_a == nullptr) != (running_var_a == nullptr) || (running_mean_b == nullptr) != (running_var_b == nullptr))
        return AFAN_ENULL;
    if (!afan_nhwc::acc_supported(dtype, c)) return AFAN_ESHAPE;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(x_a, a) || !aligned(x_b, a) || !aligned(y, a) || !aligned(acc_a, 16) || !aligned(acc_b, 16)) return AFAN_EALIGN;
    return afan_nhwc::fwd_acc_dual(dtype, x_a, x_b, y, n * hw, c, eps_a, momentum_a, weight_a, bias_a, acc_a, save_stats_a,
                                   running_mean_a, running_var_a, num_batches_a, eps_b, momentum_b, weight_b, bias_b, acc_b,
                                   save_stats_b, running_mean_b, running_var_b, num_batches_b, (hipStream_t)stream);
}

int afan_bn_backward_acc(const void* dy, const void* x, const void* y, void* dx, void* d_residual, int dtype,
                         int64_t n, int64_t c, int64_t hw, const float* save_stats, int relu, double* acc,
                         int acc_ready, float* dweight, float* dbias, int accumulate, int groups,
                         afan_stream_t stream) {
    int e = check_common(dtype, n, c, hw);
    if (e) return e;
    if (!dy || !x || !dx || !save_stats || !acc) return AFAN_ENULL;
    if (!afan_nhwc::acc_supported(dtype, c)) return AFAN_ESHAPE;
    const size_t a = dtype == AFAN_F32 ? 4 : 2;
    if (!aligned(dy, a) || !aligned(x, a) || !aligned(dx, a) || (y && !aligned(y, a)) ||
        (d_residual && !aligned(d_residual, a)) || !aligned(acc, 16))
        return AFAN_EALIGN;
    if (groups < 1 || n % groups != 0) return AFAN_ESHAPE;
    return afan_nhwc::bwd_acc(dtype, dy, x, y, dx, d_residual, (n / groups) * hw, c, save_stats, relu, acc, acc_ready, dweight,
                              dbias, accumulate, (hipStream_t)stream, groups);
}

int afan_normalize_nchw(const float* x, void* y, int out_dtype, int out_layout, int64_t n, int64_t c, int64_t hw,
                        const float* mean, const float* std, afan_stream_t stream) {
    int e = check_common(out_dtype, n, c, hw);
    if (e) return e;
    if ((e = check_layout(out_layout))) return e;
    if (!x || !y || !mean || !std) return AFAN_ENULL;
    const int64_t total = n * c * hw;
    const int grid = grid_for(total, BLOCK);
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("normalize_kernel", total * (4.0 + (out_dtype == AFAN_F32 ? 4 : 2)), st);
    if (out_dtype == AFAN_F32) {
        if (out_layout == AFAN_NHWC) normalize_kernel<float, true><<<grid, BLOCK, 0, st>>>(x, (float*)y, total, c, hw, mean, std);
        else normalize_kernel<float, false><<<grid, BLOCK, 0, st>>>(x, (float*)y, total, c, hw, mean, std);
    } else {
        if (out_layout == AFAN_NHWC) normalize_kernel<uint16_t, true><<<grid, BLOCK, 0, st>>>(x, (uint16_t*)y, total, c, hw, mean, std);
        else normalize_kernel<uint16_t, false><<<grid, BLOCK, 0, st>>>(x, (uint16_t*)y, total, c, hw, mean, std);
    }
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
