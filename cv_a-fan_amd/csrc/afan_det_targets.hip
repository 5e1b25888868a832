// Training targets, sampling lists and per-image losses of the Faster-RCNN step as a handful of launches.
// The reference writes them as ~100 small tensor operations per forward (bbox.py:41-92, rpn/region_proposal_network.py:58-105 and
// :163-185, model.py:256-282 and :343-367) — 16 forwards per A-FAN iteration, each operation a launch the host needs ~10 us to
// issue while the device needs 2-3 us to run it: the iteration waited for the host.  Same arithmetic here, operation by
// operation in fp32 (no contraction: the Makefile's -ffp-contract=off), so boxes, IoUs, labels and regression targets are the
// values the tensor operations give; the two loss sums are added in a fixed tree order (deterministic; the reference's order is
// the vendor reduction's — equal to rounding, tests state 1e-6).
//   afan_box_decode_clip      bbox.py:54-64 `apply_transformer` + :89-92 `clip`
//   afan_box_assign           IoU (bbox.py:66-82) -> best ground truth per box, labels by the RPN's rule (:66-82) or the head's (model.py:256-264)
//   afan_sample_lists         the foreground / background index lists `nonzero()` gives, and their lengths (ONE host read follows)
//   afan_sample_gather        the sampled rows: box, label, batch index, regression target (bbox.py:41-52 `calc_transformer`)
//   afan_det_loss_fwd / _bwd  per-image cross-entropy + beta-smooth-L1 (extension/functional.py:6-10) and their gradients
#include "afan_common.h"

using namespace afan;

namespace {

constexpr int TB = 256;

// float -> unsigned whose order is the floats' (negative values below positive; a positive NaN above everything, as
// torch.max propagates it)
__device__ __forceinline__ unsigned ordered(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// bbox.py:66-82 on one pair, the tensor operations' order: areas, clamp(min(right) - max(left), min=0), inter / (a + b - inter)
__device__ __forceinline__ float iou_cont(const float4 a, const float4 b) {
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float area_b = (b.z - b.x) * (b.w - b.y);
    float w = (a.z < b.z ? a.z : b.z) - (a.x > b.x ? a.x : b.x);
    float h = (a.w < b.w ? a.w : b.w) - (a.y > b.y ? a.y : b.y);
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float inter = w * h;
    return inter / (area_a + area_b - inter);
}

// a > b the way torch.max walks a row: the first maximum stays, a NaN replaces anything that is not one
__device__ __forceinline__ bool takes_over(float v, float best) { return v > best || (v != v && best == best); }

// ---- decode + clip ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void box_decode_clip_kernel(const float4* __restrict__ src, const float4* __restrict__ t,
                                                             float4* __restrict__ out, int64_t n, float right, float bottom) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const float4 s = src[i], d = t[i];
    const float scx = (s.x + s.z) / 2.f, scy = (s.y + s.w) / 2.f, sw = s.z - s.x, sh = s.w - s.y;
    const float cx = d.x * sw + scx, cy = d.y * sh + scy, w = expf(d.z) * sw, h = expf(d.w) * sh;
    float4 o = make_float4(cx - w / 2.f, cy - h / 2.f, cx + w / 2.f, cy + h / 2.f);
    o.x = fminf(fmaxf(o.x, 0.f), right);
    o.z = fminf(fmaxf(o.z, 0.f), right);
    o.y = fminf(fmaxf(o.y, 0.f), bottom);
    o.w = fminf(fmaxf(o.w, 0.f), bottom);
    out[i] = o;
}

// ---- assignment -------------------------------------------------------------------------------------------------------------
constexpr int MAX_G_LDS = 1024;

// pass 1: per box the best ground truth (first maximum) and the label its IoU gives; mode 0 also the per-ground-truth maximum
// over all boxes (ordered-unsigned atomicMax: workgroup-local in LDS first, one global atomic per workgroup and ground truth)
template <int MODE>
__global__ __launch_bounds__(TB) void box_assign_kernel(const float4* __restrict__ boxes, const float4* __restrict__ gt, int64_t N, int64_t G,
                                                        float lo, float hi, const int64_t* __restrict__ gt_classes,
                                                        int64_t* __restrict__ labels, int64_t* __restrict__ assign,
                                                        unsigned* __restrict__ gt_max) {
    __shared__ unsigned smax[MODE == 0 ? MAX_G_LDS : 1];
    const int64_t b = blockIdx.y, n = (int64_t)blockIdx.x * TB + threadIdx.x;
    const bool lds = MODE == 0 && G <= MAX_G_LDS;
    if (lds) {
        for (int g = threadIdx.x; g < G; g += TB) smax[g] = 0u;
        __syncthreads();
    }
    if (n < N) {
        const float4 a = boxes[b * N + n];
        float best = 0.f;
        int64_t arg = 0;
        for (int64_t g = 0; g < G; ++g) {
            const float v = iou_cont(a, gt[b * G + g]);
            if (g == 0 || takes_over(v, best)) { best = v; arg = g; }
            if (MODE == 0) {
                if (lds) atomicMax(&smax[g], ordered(v));
                else atomicMax(&gt_max[b * G + g], ordered(v));
            }
        }
        int64_t lab = -1;
        if (MODE == 0) {            // region_proposal_network.py:78-80 (the ties with a ground truth's best: pass 2)
            if (best < lo) lab = 0;
            if (best >= hi) lab = 1;
        } else {                    // model.py:259-264
            if (best < lo) lab = 0;
            if (best >= lo) lab = gt_classes[b * G + arg];
        }
        labels[b * N + n] = G > 0 ? lab : -1;
        assign[b * N + n] = arg;
    }
    if (lds) {
        __syncthreads();
        for (int g = threadIdx.x; g < G; g += TB)
            if (smax[g]) atomicMax(&gt_max[b * G + g], smax[g]);
    }
}

// pass 2 (mode 0): a box whose IoU with some ground truth is positive and equals that ground truth's best over all boxes is
// foreground (:76-79), unless... nothing: `labels[anchor_max >= 0.7] = 1` after it only adds ones
__global__ __launch_bounds__(TB) void box_assign_ties_kernel(const float4* __restrict__ boxes, const float4* __restrict__ gt, int64_t N, int64_t G,
                                                             const unsigned* __restrict__ gt_max, int64_t* __restrict__ labels) {
    const int64_t b = blockIdx.y, n = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (n >= N) return;
    const float4 a = boxes[b * N + n];
    bool add = false;
    for (int64_t g = 0; g < G; ++g) {
        const float v = iou_cont(a, gt[b * G + g]);
        add = add || (v > 0.f && v == unordered(gt_max[b * G + g]));
    }
    if (add) labels[b * N + n] = 1;
}

// ---- sampling lists ---------------------------------------------------------------------------------------------------------
// one workgroup: ascending positions of labels > 0 into fg, of labels == 0 into bg, their counts into counts[0..1]
constexpr int SL_THREADS = 1024;
__global__ __launch_bounds__(SL_THREADS) void sample_lists_kernel(const int64_t* __restrict__ labels, int64_t M, int64_t* __restrict__ fg,
                                                                  int64_t* __restrict__ bg, int64_t* __restrict__ counts) {
    __shared__ int sf[SL_THREADS], sb[SL_THREADS];
    const int64_t per = (M + SL_THREADS - 1) / SL_THREADS;
    const int64_t lo = threadIdx.x * per, hi = lo + per < M ? lo + per : M;
    int cf = 0, cb = 0;
    for (int64_t i = lo; i < hi; ++i) {
        const int64_t l = labels[i];
        cf += l > 0;
        cb += l == 0;
    }
    sf[threadIdx.x] = cf;
    sb[threadIdx.x] = cb;
    __syncthreads();
    for (int o = 1; o < SL_THREADS; o <<= 1) {             // Hillis-Steele inclusive scans
        const int vf = (int)threadIdx.x >= o ? sf[threadIdx.x - o] : 0, vb = (int)threadIdx.x >= o ? sb[threadIdx.x - o] : 0;
        __syncthreads();
        sf[threadIdx.x] += vf;
        sb[threadIdx.x] += vb;
        __syncthreads();
    }
    int64_t pf = sf[threadIdx.x] - cf, pb = sb[threadIdx.x] - cb;
    for (int64_t i = lo; i < hi; ++i) {
        const int64_t l = labels[i];
        if (l > 0) fg[pf++] = i;
        if (l == 0) bg[pb++] = i;
    }
    if (threadIdx.x == SL_THREADS - 1) { counts[0] = sf[threadIdx.x]; counts[1] = sb[threadIdx.x]; }
}

// ---- the sampled rows -------------------------------------------------------------------------------------------------------
// pos[i] >= 0: foreground list entry pos[i]; < 0: background list entry -pos[i] - 1 (the host's three randperm draws composed)
__global__ __launch_bounds__(TB) void sample_gather_kernel(const int64_t* __restrict__ fg, const int64_t* __restrict__ bg,
                                                           const int64_t* __restrict__ pos, int64_t S, const float4* __restrict__ boxes,
                                                           const float4* __restrict__ gt, const int64_t* __restrict__ assign,
                                                           const int64_t* __restrict__ labels, int64_t N, int64_t G,
                                                           int64_t* __restrict__ sel, float4* __restrict__ out_boxes,
                                                           int64_t* __restrict__ out_labels, float4* __restrict__ out_deltas,
                                                           int64_t* __restrict__ out_batch) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= S) return;
    const int64_t p = pos[i];
    const int64_t flat = p >= 0 ? fg[p] : bg[-p - 1];
    const int64_t b = flat / N;
    const float4 s = boxes[flat], d = gt[b * G + assign[flat]];
    sel[i] = flat;
    out_boxes[i] = s;
    out_labels[i] = labels[flat];
    out_batch[i] = b;
    // bbox.py:41-52: centres and extents of both, (dc - sc) / s_extent, log(d_extent / s_extent)
    const float scx = (s.x + s.z) / 2.f, scy = (s.y + s.w) / 2.f, sw = s.z - s.x, sh = s.w - s.y;
    const float dcx = (d.x + d.z) / 2.f, dcy = (d.y + d.w) / 2.f, dw = d.z - d.x, dh = d.w - d.y;
    out_deltas[i] = make_float4((dcx - scx) / sw, (dcy - scy) / sh, logf(dw / sw), logf(dh / sh));
}

// ---- losses -----------------------------------------------------------------------------------------------------------------
// fixed-order sum of one value per thread over the workgroup (tree in LDS): the same bits whatever the scheduling
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    scratch[threadIdx.x] = v;
    __syncthreads();
    for (int o = TB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) scratch[threadIdx.x] += scratch[threadIdx.x + o];
        __syncthreads();
    }
    const float r = scratch[0];
    __syncthreads();
    return r;
}

// one workgroup.  Sample s: row r = rows ? rows[s] : s of logits [R, C] and deltas [R, K, 4] (K == 1, or C: the sample's own
// class); unit gradients saved for the backward: u_logits [S, C] = softmax - onehot, u_deltas [S, 4] = d smooth-L1 / d input
// (times the foreground flag); per-image denominators in denom [B][2] = (samples, 4 x foreground + 1e-8).
__global__ __launch_bounds__(TB) void det_loss_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ deltas,
                                                          const int64_t* __restrict__ rows, const int64_t* __restrict__ gt_labels,
                                                          const float4* __restrict__ gt_deltas, const int64_t* __restrict__ batch,
                                                          int64_t S, int B, int C, int K, float beta, float4 mean, float4 stdv, int normalize,
                                                          float* __restrict__ ce_out, float* __restrict__ sl_out,
                                                          float* __restrict__ u_logits, float4* __restrict__ u_deltas, float* __restrict__ denom) {
    __shared__ float scratch[TB];
    for (int b = 0; b < B; ++b) {
        float ce = 0.f, cnt = 0.f, sl = 0.f, nfg = 0.f;
        for (int64_t s = threadIdx.x; s < S; s += TB) {
            if ((int)batch[s] != b) continue;
            const int64_t r = rows ? rows[s] : s, lab = gt_labels[s];
            const float* x = logits + r * C;
            float m = x[0];
            for (int c = 1; c < C; ++c) m = x[c] > m ? x[c] : m;
            float sum = 0.f;
            for (int c = 0; c < C; ++c) sum += expf(x[c] - m);
            const float lse = logf(sum);
            ce += -(x[lab] - m - lse);                                          // -log_softmax(x)[label]
            for (int c = 0; c < C; ++c) u_logits[s * C + c] = expf(x[c] - m - lse) - (c == lab ? 1.f : 0.f);
            cnt += 1.f;
            const bool is_fg = lab != 0;
            float4 t = gt_deltas[s];
            if (normalize) t = make_float4((t.x - mean.x) / stdv.x, (t.y - mean.y) / stdv.y, (t.z - mean.z) / stdv.z, (t.w - mean.w) / stdv.w);
            const float* p = deltas + (r * K + (K == 1 ? 0 : lab)) * 4;
            const float in[4] = {p[0], p[1], p[2], p[3]}, tg[4] = {t.x, t.y, t.z, t.w};
            float acc = 0.f, u[4];
            for (int j = 0; j < 4; ++j) {
                const float e = in[j] - tg[j], d = fabsf(e);
                acc += d < beta ? 0.5f * (d * d) / beta : d - 0.5f * beta;
                u[j] = !is_fg ? 0.f : (d < beta ? e / beta : (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)));
            }
            u_deltas[s] = make_float4(u[0], u[1], u[2], u[3]);
            if (is_fg) { sl += acc; nfg += 1.f; }
        }
        ce = block_sum(ce, scratch);
        cnt = block_sum(cnt, scratch);
        sl = block_sum(sl, scratch);
        nfg = block_sum(nfg, scratch);
        if (threadIdx.x == 0) {
            const float dn = 4.0f * nfg + 1e-8f;
            ce_out[b] = ce / cnt;               // (no samples: 0 / 0 = nan, the mean of nothing, like the reference)
            sl_out[b] = sl / dn;
            denom[2 * b] = cnt;
            denom[2 * b + 1] = dn;
        }
    }
}

// dense gradients (zeroed by the caller): row r of d_logits [R, C] and of d_deltas [R, K, 4] for every sample
__global__ __launch_bounds__(TB) void det_loss_bwd_kernel(const float* __restrict__ g_ce, const float* __restrict__ g_sl,
                                                          const float* __restrict__ u_logits, const float4* __restrict__ u_deltas,
                                                          const float* __restrict__ denom, const int64_t* __restrict__ rows,
                                                          const int64_t* __restrict__ gt_labels, const int64_t* __restrict__ batch, int64_t S, int C,
                                                          int K, float* __restrict__ d_logits, float* __restrict__ d_deltas) {
    const int64_t s = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (s >= S) return;
    const int b = (int)batch[s];
    const int64_t r = rows ? rows[s] : s, lab = gt_labels[s];
    const float kc = g_ce[b] / denom[2 * b], ks = g_sl[b] / denom[2 * b + 1];
    for (int c = 0; c < C; ++c) d_logits[r * C + c] = u_logits[s * C + c] * kc;
    const float4 u = u_deltas[s];
    float* q = d_deltas + (r * K + (K == 1 ? 0 : lab)) * 4;
    q[0] = u.x * ks; q[1] = u.y * ks; q[2] = u.z * ks; q[3] = u.w * ks;
}

}  // namespace

namespace {
__global__ void sum_scalars_kernel(const float* a, const float* b, const float* c, const float* d, float* out) {
    float s = a[0] + b[0];
    if (c) s = s + c[0];
    if (d) s = s + d[0];
    out[0] = s;
}
}  // namespace

namespace {
// rows[j] = j < min(count, P) ? cand[clamp(keep[j])] : 0 — the kept boxes of one image in their padded row block
__global__ void proposal_rows_kernel(const float4* cand, int64_t n_cand, const int64_t* keep, int64_t n_keep, const int64_t* count,
                                     int64_t P, float4* rows, int64_t* kept) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t c = count[0];
    c = c < 0 ? 0 : (c > P ? P : c);
    if (c > n_keep) c = n_keep;
    if (j == 0) kept[0] = c;
    if (j >= P) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < c && n_cand > 0) {
        int64_t k = keep[j];
        k = k < 0 ? 0 : (k >= n_cand ? n_cand - 1 : k);
        v = cand[k];
    }
    rows[j] = v;
}

// labels[b][j] = -1 for j >= max over images of kept[]
__global__ void labels_limit_kernel(int64_t* labels, int64_t B, int64_t N, const int64_t* kept, int64_t n_kept) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    int64_t lim = kept[0];
    for (int64_t b = 1; b < n_kept; ++b) lim = kept[b] > lim ? kept[b] : lim;
    if (i % N >= lim) labels[i] = -1;
}
}  // namespace

extern "C" {

// out[n] = clip(apply_transformer(src[n], t[n])): boxes as (left, top, right, bottom) fp32, clipped to [0, right] x [0, bottom]
int afan_box_decode_clip(const float* src, const float* t, float* out, int64_t n, float right, float bottom, afan_stream_t stream) {
    if (n < 0) return AFAN_ESHAPE;
    if (n == 0) return AFAN_OK;
    if (!src || !t || !out) return AFAN_ENULL;
    if (!aligned(src, 16) || !aligned(t, 16) || !aligned(out, 16)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("box_decode_clip_kernel", 48.0 * n, st);
    box_decode_clip_kernel<<<(unsigned)((n + TB - 1) / TB), TB, 0, st>>>((const float4*)src, (const float4*)t, (float4*)out, n, right, bottom);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// boxes [B, N, 4], gt [B, G, 4] -> assign [B, N] (index of the best ground truth, first maximum), labels [B, N]:
//   mode 0 (anchors): -1; 0 where the best IoU < lo; 1 where it is >= hi or where the box ties some ground truth's best positive IoU
//   mode 1 (proposals): -1; 0 where the best IoU < lo; gt_classes[b, assign] where it is >= lo        (hi unused)
// workspace: B * G * 4 bytes (mode 0)
int afan_box_assign(const float* boxes, const float* gt, int64_t B, int64_t N, int64_t G, int mode, float lo, float hi,
                    const int64_t* gt_classes, int64_t* labels, int64_t* assign, void* workspace, afan_stream_t stream) {
    if (B < 0 || N < 0 || G < 0 || B > 65535 || (mode != 0 && mode != 1)) return AFAN_ESHAPE;
    if (B == 0 || N == 0) return AFAN_OK;
    if (!boxes || !labels || !assign || (G > 0 && !gt) || (mode == 1 && G > 0 && !gt_classes) || (mode == 0 && G > 0 && !workspace)) return AFAN_ENULL;
    if (!aligned(boxes, 16) || (gt && !aligned(gt, 16))) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((N + TB - 1) / TB), (unsigned)B);
    AFAN_PROF("box_assign_kernel", 32.0 * B * N, st);
    if (mode == 0) {
        if (G > 0) {
            hipError_t e = hipMemsetAsync(workspace, 0, (size_t)(B * G * 4), st);
            if (e != hipSuccess) return (int)e;
        }
        box_assign_kernel<0><<<grid, TB, 0, st>>>((const float4*)boxes, (const float4*)gt, N, G, lo, hi, nullptr, labels, assign, (unsigned*)workspace);
        AFAN_LAUNCH_CHECK();
        if (G > 0) {
            box_assign_ties_kernel<<<grid, TB, 0, st>>>((const float4*)boxes, (const float4*)gt, N, G, (const unsigned*)workspace, labels);
            AFAN_LAUNCH_CHECK();
        }
    } else {
        box_assign_kernel<1><<<grid, TB, 0, st>>>((const float4*)boxes, (const float4*)gt, N, G, lo, hi, gt_classes, labels, assign, nullptr);
        AFAN_LAUNCH_CHECK();
    }
    return AFAN_OK;
}

// ascending positions of labels > 0 -> fg, of labels == 0 -> bg (both [M] int64), counts[0..1] their lengths
int afan_sample_lists(const int64_t* labels, int64_t M, int64_t* fg, int64_t* bg, int64_t* counts, afan_stream_t stream) {
    if (M < 0 || M > (int64_t)1 << 30) return AFAN_ESHAPE;
    if (!counts || (M > 0 && (!labels || !fg || !bg))) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("sample_lists_kernel", 16.0 * M, st);
    sample_lists_kernel<<<1, SL_THREADS, 0, st>>>(labels, M, fg, bg, counts);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// the S sampled rows (pos: see sample_gather_kernel) of boxes [B, N, 4] with their labels, batch indices, flat positions and
// regression targets towards gt[b, assign]
int afan_sample_gather(const int64_t* fg, const int64_t* bg, const int64_t* pos, int64_t S, const float* boxes, const float* gt,
                       const int64_t* assign, const int64_t* labels, int64_t N, int64_t G, int64_t* sel, float* out_boxes,
                       int64_t* out_labels, float* out_deltas, int64_t* out_batch, afan_stream_t stream) {
    if (S < 0 || N <= 0 || G <= 0) return AFAN_ESHAPE;
    if (S == 0) return AFAN_OK;
    if (!fg || !bg || !pos || !boxes || !gt || !assign || !labels || !sel || !out_boxes || !out_labels || !out_deltas || !out_batch) return AFAN_ENULL;
    if (!aligned(boxes, 16) || !aligned(gt, 16) || !aligned(out_boxes, 16) || !aligned(out_deltas, 16)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("sample_gather_kernel", 96.0 * S, st);
    sample_gather_kernel<<<(unsigned)((S + TB - 1) / TB), TB, 0, st>>>(fg, bg, pos, S, (const float4*)boxes, (const float4*)gt, assign, labels, N, G,
                                                                       sel, (float4*)out_boxes, out_labels, (float4*)out_deltas, out_batch);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// per image b < B: ce[b] = mean over its samples of -log_softmax(logits[row])[label]; sl1[b] = sum over its foreground samples
// (label != 0) and 4 coordinates of smooth-L1_beta(deltas[row, class] - target) / (4 x foreground + 1e-8).  norm: NULL, or 8
// floats (mean[4], std[4]): targets are (gt_deltas - mean) / std (model.py:352-354).  save: S * 4 + S * C + 2 * B floats, 16-byte aligned.
int afan_det_loss_fwd(const float* logits, const float* deltas, const int64_t* rows, const int64_t* gt_labels, const float* gt_deltas,
                      const int64_t* batch, int64_t S, int64_t B, int64_t C, int64_t K, float beta, const float* norm, float* ce, float* sl1,
                      float* save, afan_stream_t stream) {
    if (S < 0 || B <= 0 || B > 4096 || C <= 0 || C > 65536 || (K != 1 && K != C)) return AFAN_ESHAPE;
    if (!ce || !sl1 || !save || (S > 0 && (!logits || !deltas || !gt_labels || !gt_deltas || !batch))) return AFAN_ENULL;
    if ((gt_deltas && !aligned(gt_deltas, 16)) || !aligned(save, 16)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    float4 mean = make_float4(0, 0, 0, 0), stdv = make_float4(1, 1, 1, 1);
    if (norm) { mean = make_float4(norm[0], norm[1], norm[2], norm[3]); stdv = make_float4(norm[4], norm[5], norm[6], norm[7]); }
    AFAN_PROF("det_loss_fwd_kernel", 8.0 * S * C + 64.0 * S, st);
    det_loss_fwd_kernel<<<1, TB, 0, st>>>(logits, deltas, rows, gt_labels, (const float4*)gt_deltas, batch, S, (int)B, (int)C, (int)K, beta, mean,
                                          stdv, norm != nullptr, ce, sl1, save + S * 4, (float4*)save, save + S * 4 + S * C);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// d_logits [R, C] and d_deltas [R, K, 4]: zero, except the sampled rows: the saved unit gradients times g_ce[b] / samples and
// g_sl1[b] / (4 x foreground + 1e-8)
int afan_det_loss_bwd(const float* g_ce, const float* g_sl1, const float* save, const int64_t* rows, const int64_t* gt_labels,
                      const int64_t* batch, int64_t S, int64_t B, int64_t C, int64_t K, int64_t R, float* d_logits, float* d_deltas,
                      afan_stream_t stream) {
    if (S < 0 || B <= 0 || C <= 0 || (K != 1 && K != C) || R < 0) return AFAN_ESHAPE;
    if (R == 0) return AFAN_OK;
    if (!d_logits || !d_deltas || (S > 0 && (!g_ce || !g_sl1 || !save || !gt_labels || !batch))) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (d_deltas == d_logits + R * C) {           // (one allocation, d_deltas behind d_logits: one fill)
        e = hipMemsetAsync(d_logits, 0, (size_t)(R * C * 4 + R * K * 16), st);
    } else {
        e = hipMemsetAsync(d_logits, 0, (size_t)(R * C * 4), st);
        if (e == hipSuccess) e = hipMemsetAsync(d_deltas, 0, (size_t)(R * K * 16), st);
    }
    if (e != hipSuccess) return (int)e;
    if (S == 0) return AFAN_OK;
    AFAN_PROF("det_loss_bwd_kernel", 8.0 * S * C + 64.0 * S, st);
    det_loss_bwd_kernel<<<(unsigned)((S + TB - 1) / TB), TB, 0, st>>>(g_ce, g_sl1, save + S * 4, (const float4*)save, save + S * 4 + S * C, rows,
                                                                      gt_labels, batch, S, (int)C, (int)K, d_logits, d_deltas);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// out[0] = ((a[0] + b[0]) + c[0]) + d[0] (c, d optional): the reference's `loss1.mean() + loss2.mean() + loss3.mean() + loss4.mean()`
// (Detection/train_aug_sat_muti_advt.py:21-27, attack_algo.py:62) for per-image loss vectors of ONE image — the mean of one element is
// that element, the additions keep their order; seven launches forward and as many backward become one and none
int afan_sum_scalars_f32(const float* a, const float* b, const float* c, const float* d, float* out, afan_stream_t stream) {
    if (!a || !b || !out || (d && !c)) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("det_loss_sum_kernel", 20.0, st);
    sum_scalars_kernel<<<1, 1, 0, st>>>(a, b, c, d, out);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// The proposal layer's hand-over to the ROI head's sampling without a host read (region_proposal_network.py:255-270 ->
// model.py:256-264): rows [P, 4] = the first min(count[0], P) survivors `cand[keep[j]]` of ONE image (NMS output `keep` over the
// score-ordered candidates `cand`, `count` on the device), zero rows behind them like the reference's padding of shorter images;
// kept[0] = that number.  One launch for the reference's index / slice / cat / stack.
int afan_proposal_rows(const float* cand, int64_t n_cand, const int64_t* keep, int64_t n_keep, const int64_t* count, int64_t P,
                       float* rows, int64_t* kept, afan_stream_t stream) {
    if (!rows || !kept || !count || (n_cand > 0 && !cand) || (n_keep > 0 && !keep)) return AFAN_ENULL;
    if (P < 0 || n_cand < 0 || n_keep < 0) return AFAN_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("det_proposal_rows_kernel", 40.0 * P, st);
    proposal_rows_kernel<<<(unsigned)((P > 0 ? P : 1) + TB - 1) / TB, TB, 0, st>>>((const float4*)cand, n_cand, keep, n_keep, count, P,
                                                                                  (float4*)rows, kept);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// labels [B, N] int64: every column at and beyond max(kept[0..n_kept)) becomes -1 ("no candidate": neither list of the sampling
// takes it) — the padded rows beyond the batch's longest survivor list, which the reference never builds
int afan_labels_limit(int64_t* labels, int64_t B, int64_t N, const int64_t* kept, int64_t n_kept, afan_stream_t stream) {
    if (!labels || !kept) return AFAN_ENULL;
    if (B < 0 || N < 0 || n_kept < 1) return AFAN_ESHAPE;
    if (B * N == 0) return AFAN_OK;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("det_labels_limit_kernel", 8.0 * B * N, st);
    labels_limit_kernel<<<(unsigned)((B * N + TB - 1) / TB), TB, 0, st>>>(labels, B, N, kept, n_kept);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
