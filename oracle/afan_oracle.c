/*
 * afan_oracle.c — plain-C CPU ORACLE for the element-wise / reduction kernels of the A-FAN hot path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle (built into oracle/_ref/ by oracle/Makefile).
 *
 * Each function restates the reference behaviour cited next to it (paths relative to the reference
 * root).  Parity status: PINNED through tests/test_oracle_golden.py, which checks these functions
 * against tests/golden/*.npz (vectors produced by the reference's own Python via oracle/gen_golden.py).
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: every op rounds on its own, like ATen's
 * element-wise CPU kernels for these expressions).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static float sign_f(float g) { return (g != g) ? g : (float)((g > 0.f) - (g < 0.f)); }

/* Classification/attack_algo.py:53 (+ :55-56 -> :35-36 -> :9-19 when clip) */
void oracle_pgd_step(float* x_adv, const float* grad, const float* x_clean, int64_t n, float gamma,
                     float eps, int clip) {
    for (int64_t i = 0; i < n; ++i) {
        float d = gamma * sign_f(grad[i]);
        float t = x_adv[i] + d;
        if (clip) {
            float lo = x_clean[i] - eps, hi = x_clean[i] + eps;
            if (t < lo) t = lo;
            if (t > hi) t = hi;
        }
        x_adv[i] = t;
    }
}

/* Classification/attack_algo.py:42-44 with the host-drawn uniform u */
void oracle_axpy_noise(float* x_adv, const float* u, int64_t n, float eps) {
    for (int64_t i = 0; i < n; ++i) {
        float t = 2.0f * u[i];
        t = t - 1.0f;
        t = t * eps;
        x_adv[i] = x_adv[i] + t;
    }
}

/* Classification/main_perturb.py:188-192 (double accumulation: a tighter reference than fp32 torch.norm) */
void oracle_perturb_norms(const float* x_adv, const float* x_clean, int64_t batch, int64_t per_sample,
                          float* l2, float* linf) {
    for (int64_t b = 0; b < batch; ++b) {
        double ss = 0.0;
        float mx = 0.f;
        int nan = 0;
        for (int64_t j = 0; j < per_sample; ++j) {
            float d = x_adv[b * per_sample + j] - x_clean[b * per_sample + j];
            ss += (double)d * (double)d;
            float a = fabsf(d);
            if (a != a) nan = 1;
            if (a > mx) mx = a;
        }
        l2[b] = (float)sqrt(ss);
        linf[b] = nan ? NAN : mx;
    }
}

/* Segmentation/attack_algo.py:121-130: statistics over the channel dim per pixel, unbiased variance */
void oracle_mix_feature(const float* clean, const float* adv, float* out, int64_t n, int64_t c, int64_t hw,
                        float eps) {
    for (int64_t b = 0; b < n; ++b)
        for (int64_t p = 0; p < hw; ++p) {
            const float* pc = clean + b * c * hw + p;
            const float* pa = adv + b * c * hw + p;
            double mc = 0, ma = 0;
            for (int64_t k = 0; k < c; ++k) { mc += pc[k * hw]; ma += pa[k * hw]; }
            mc /= (double)c; ma /= (double)c;
            double vc = 0, va = 0;
            for (int64_t k = 0; k < c; ++k) {
                double dc = pc[k * hw] - mc, da = pa[k * hw] - ma;
                vc += dc * dc; va += da * da;
            }
            vc /= (double)(c - 1); va /= (double)(c - 1);
            float fmc = (float)mc, fma_ = (float)ma;
            float sc = sqrtf((float)vc + eps), sa = sqrtf((float)va + eps);
            for (int64_t k = 0; k < c; ++k) {
                float t = (pc[k * hw] - fmc) / sc;
                t = t * sa;
                t = t + fma_;
                out[b * c * hw + k * hw + p] = t;
            }
        }
}

/* Segmentation/attack_algo.py:108-118 with ATen's lerp: |w|<0.5 ? x + w*(y-x) : y - (y-x)*(1-w), as one
 * fused multiply-add fma(coeff, y-x, base) — what ATen's vectorised CPU kernel executes. */
void oracle_lerp_points(const float* x, const float* y, float* out, int64_t n, const float* w, int k_int) {
    for (int k = 0; k < k_int; ++k) {
        int small = fabsf(w[k]) < 0.5f;
        float coeff = small ? w[k] : w[k] - 1.0f;
        for (int64_t i = 0; i < n; ++i)
            out[(int64_t)k * n + i] = fmaf(coeff, y[i] - x[i], small ? x[i] : y[i]);
    }
}

/* nn.BatchNorm2d training forward (+ residual, + ReLU) as the backbone applies it (resnet_s.py:72-77),
 * double accumulation.  Updates running stats with the unbiased variance (momentum m). */
void oracle_bn_train_forward(const float* x, const float* res, float* y, int64_t n, int64_t c, int64_t hw,
                             float eps, float momentum, const float* weight, const float* bias, int relu,
                             float* save_mean, float* save_invstd, float* rmean, float* rvar) {
    const double M = (double)n * (double)hw;
    for (int64_t k = 0; k < c; ++k) {
        double s = 0;
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) s += x[(b * c + k) * hw + p];
        const double mu = s / M;
        double v = 0;
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) { double d = x[(b * c + k) * hw + p] - mu; v += d * d; }
        const double var_b = v / M;
        const float is = (float)(1.0 / sqrt(var_b + (double)eps));
        save_mean[k] = (float)mu;
        save_invstd[k] = is;
        if (rmean) {
            rmean[k] = (1.0f - momentum) * rmean[k] + momentum * (float)mu;
            rvar[k] = (1.0f - momentum) * rvar[k] + momentum * (float)(v / (M - 1.0));
        }
        const float w = weight ? weight[k] : 1.f, bb = bias ? bias[k] : 0.f;
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) {
                int64_t i = (b * c + k) * hw + p;
                float t = (float)(((double)x[i] - mu) * (double)is * (double)w + (double)bb);
                if (res) t += res[i];
                if (relu && !(t != t)) t = t > 0.f ? t : 0.f;
                y[i] = t;
            }
    }
}

/* autograd of the op above: g = dy*(y>0); dx = w*invstd*(g - mean(g) - xhat*mean(g*xhat)) */
void oracle_bn_backward(const float* dy, const float* x, const float* y, float* dx, float* dres, int64_t n,
                        int64_t c, int64_t hw, const float* mean, const float* invstd, const float* weight,
                        int relu, float* dweight, float* dbias) {
    const double M = (double)n * (double)hw;
    for (int64_t k = 0; k < c; ++k) {
        double sg = 0, sgx = 0;
        const double mu = mean[k], is = invstd[k];
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) {
                int64_t i = (b * c + k) * hw + p;
                double g = (relu && !(y[i] > 0.f)) ? 0.0 : dy[i];
                sg += g;
                sgx += g * (x[i] - mu) * is;
            }
        if (dweight) dweight[k] = (float)sgx;
        if (dbias) dbias[k] = (float)sg;
        const double w = weight ? weight[k] : 1.0;
        for (int64_t b = 0; b < n; ++b)
            for (int64_t p = 0; p < hw; ++p) {
                int64_t i = (b * c + k) * hw + p;
                double g = (relu && !(y[i] > 0.f)) ? 0.0 : dy[i];
                double xh = (x[i] - mu) * is;
                dx[i] = (float)((g - sg / M - xh * sgx / M) * w * is);
                if (dres) dres[i] = (float)g;
            }
    }
}

/* torch.optim.SGD step as configured at Classification/main_perturb.py:72-74 */
void oracle_sgd_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float wd,
                     float gscale) {
    for (int64_t i = 0; i < n; ++i) {
        float gg = g[i] * gscale;
        gg = gg + wd * p[i];
        m[i] = m[i] * momentum + gg;
        p[i] = p[i] - lr * m[i];
    }
}
