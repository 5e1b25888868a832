// Two small linear layers on one input, fp32: y1 = x w1^T + b1, y2 = x w2^T + b2 with N1 + N2 <= 128 output features — the Faster-RCNN
// heads of the A-FAN detection step: `_proposal_class` / `_proposal_transformer` (Detection/model.py:235-236, called at :255-256,
// :290-291, :343-344: 128 ROIs x 2048 -> 21 | 84) and the RPN's 1x1 convolutions `_anchor_objectness` / `_anchor_transformer`
// (Detection/rpn/region_proposal_network.py:35-36, called at :53-54, :120-121: 38 x 57 pixels x 512 -> 18 | 36, a 1x1 convolution on a
// channels-last map IS this product), forward, input gradient and parameter gradients.
//
// Why a kernel family of its own: as 1x1 convolutions on the general fp32 kernels these products are 16-17 workgroups walking a long
// reduction each (41 + 33 us forward, 2 x 20 us input gradient per RPN pass; 46 + 28 us forward — through the weight-gradient kernel, whose
// operand order their row-major x does not have — and 2 x 32 us input gradient per ROI-head pass; profiles/r06_frcnn_trace_by_grid.txt):
// ~4.7 ms of a 50 ms iteration.  Here the two layers share one launch (one read of x; the input gradient's sum of both layers' shares is
// the one reduction over N1 + N2 columns), and the work is cut so that >= 64 workgroups run: the forward splits K, the parameter gradient
// splits the rows, partial tiles go through a workspace and are added IN ORDER by a second launch (deterministic: no float atomics).
// fp32 FMA on the vector ALU: 128 x 2048 x 105 is 55 MFLOP.
#include "afan_common.h"

using namespace afan;

namespace {

constexpr int TB = 256;
constexpr int RT = 32;      // rows per workgroup tile (forward, input gradient)
constexpr int KT = 64;      // columns of x / gx per workgroup (input gradient, parameter gradient)
constexpr int NMAX = 128;   // N1 + N2

struct Pair {
    const float* w1; const float* w2;     // [N1, K], [N2, K]
    const float* b1; const float* b2;     // optional
    int n1, n2;
};
__device__ __forceinline__ const float* wrow(const Pair& p, int c, int64_t K) { return c < p.n1 ? p.w1 + (int64_t)c * K : p.w2 + (int64_t)(c - p.n1) * K; }

// ---- forward: partial[s][m][c] = sum over this slice's k of x[m][k] w[c][k]; slices == 1: y = partial + bias directly.
// The slice is walked in chunks of KCH reduction steps staged in LDS; the next chunk's operands are requested (into registers) before
// this chunk's arithmetic: one memory latency per workgroup is exposed, not one per chunk.
template <int NC, int KCH>   // column groups of 32 per thread: NC = ceil((N1 + N2) / 32); KCH: 128 (NC <= 2) or 64
__global__ __launch_bounds__(TB) void skinny_fwd_kernel(const float* __restrict__ x, Pair p, int64_t M, int64_t K, int kslice, float* __restrict__ part,
                                                        float* __restrict__ y1, float* __restrict__ y2) {
    constexpr int XV = RT * KCH / 4 / TB;            // float4 of x per thread and chunk
    constexpr int WV = NC * 32 * KCH / 4 / TB;       // float4 of the weights
    constexpr int Q = KCH / 4;                       // float4 per row
    __shared__ float xs[RT][KCH + 4];
    __shared__ float ws[KCH][NC * 32 + 1];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int NT = p.n1 + p.n2;
    const int64_t m0 = (int64_t)blockIdx.x * RT;
    const int64_t k_lo = (int64_t)blockIdx.y * kslice, k_hi = k_lo + kslice < K ? k_lo + kslice : K;
    float acc[4][NC];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[i][j] = 0.f;
    float4 xr[XV], wr[WV];
    auto request = [&](int64_t k0) {
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int idx = tid + TB * u, r = idx / Q, k4 = (idx % Q) * 4;
            xr[u] = (m0 + r < M && k0 + k4 < k_hi) ? *reinterpret_cast<const float4*>(x + (m0 + r) * K + k0 + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < WV; ++u) {
            const int idx = tid + TB * u, c = idx / Q, k4 = (idx % Q) * 4;
            wr[u] = (c < NT && k0 + k4 < k_hi) ? *reinterpret_cast<const float4*>(wrow(p, c, K) + k0 + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    request(k_lo);
    for (int64_t k0 = k_lo; k0 < k_hi; k0 += KCH) {
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int idx = tid + TB * u, r = idx / Q, k4 = (idx % Q) * 4;
            *reinterpret_cast<float4*>(&xs[r][k4]) = xr[u];
        }
#pragma unroll
        for (int u = 0; u < WV; ++u) {               // transposed on the way into LDS
            const int idx = tid + TB * u, c = idx / Q, k4 = (idx % Q) * 4;
            ws[k4 + 0][c] = wr[u].x; ws[k4 + 1][c] = wr[u].y; ws[k4 + 2][c] = wr[u].z; ws[k4 + 3][c] = wr[u].w;
        }
        __syncthreads();
        if (k0 + KCH < k_hi) request(k0 + KCH);
#pragma unroll 4
        for (int k4 = 0; k4 < KCH; k4 += 4) {
            float4 xv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4*>(&xs[4 * ty + i][k4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float wv[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) wv[j] = ws[k4 + e][tx + 32 * j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xe = e == 0 ? xv[i].x : e == 1 ? xv[i].y : e == 2 ? xv[i].z : xv[i].w;
#pragma unroll
                    for (int j = 0; j < NC; ++j) acc[i][j] = fmaf(xe, wv[j], acc[i][j]);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + 4 * ty + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = tx + 32 * j;
            if (c >= NT) continue;
            if (part) part[((int64_t)blockIdx.y * M + m) * NT + c] = acc[i][j];
            else if (c < p.n1) y1[m * p.n1 + c] = acc[i][j] + (p.b1 ? p.b1[c] : 0.f);
            else y2[m * p.n2 + c - p.n1] = acc[i][j] + (p.b2 ? p.b2[c - p.n1] : 0.f);
        }
    }
}

// y = bias + the slices' partials in slice order
__global__ __launch_bounds__(TB) void skinny_fwd_reduce_kernel(const float* __restrict__ part, int slices, Pair p, int64_t M, float* __restrict__ y1,
                                                               float* __restrict__ y2) {
    const int NT = p.n1 + p.n2;
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x, total = M * NT;
    if (i >= total) return;
    const int64_t m = i / NT;
    const int c = (int)(i - m * NT);
    float s = part[i];
    int z = 1;
    for (; z + 3 < slices; z += 4) {                 // (four requests in flight; the additions keep the slice order)
        const float a = part[(int64_t)z * total + i], b = part[(int64_t)(z + 1) * total + i], c2 = part[(int64_t)(z + 2) * total + i],
                    d = part[(int64_t)(z + 3) * total + i];
        s += a; s += b; s += c2; s += d;
    }
    for (; z < slices; ++z) s += part[(int64_t)z * total + i];
    if (c < p.n1) y1[m * p.n1 + c] = s + (p.b1 ? p.b1[c] : 0.f);
    else y2[m * p.n2 + c - p.n1] = s + (p.b2 ? p.b2[c - p.n1] : 0.f);
}

// ---- input gradient: gx[m][k] = sum_c g1[m][c] w1[c][k] + sum_c g2[m][c] w2[c][k] (columns of layer 1 first, ascending)
__global__ __launch_bounds__(TB) void skinny_dgrad_kernel(const float* __restrict__ g1, const float* __restrict__ g2, Pair p, int64_t M, int64_t K,
                                                          float* __restrict__ gx) {
    __shared__ float gs[RT][NMAX + 1];
    __shared__ float ws[NMAX][KT + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;      // 16 x float4 of k, 16 groups of 2 rows
    const int NT = p.n1 + p.n2;
    const int64_t m0 = (int64_t)blockIdx.x * RT, k0 = (int64_t)blockIdx.y * KT;
    for (int idx = tid; idx < RT * NT; idx += TB) {
        const int r = idx / NT, c = idx - r * NT;
        float v = 0.f;
        if (m0 + r < M) v = c < p.n1 ? g1[(m0 + r) * p.n1 + c] : g2[(m0 + r) * p.n2 + c - p.n1];
        gs[r][c] = v;
    }
    for (int idx = tid; idx < NT * (KT / 4); idx += TB) {
        const int c = idx / (KT / 4), k4 = (idx - c * (KT / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + k4 < K) v = *reinterpret_cast<const float4*>(wrow(p, c, K) + k0 + k4);
        *reinterpret_cast<float4*>(&ws[c][k4]) = v;
    }
    __syncthreads();
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    for (int c = 0; c < NT; ++c) {
        const float4 w = *reinterpret_cast<const float4*>(&ws[c][tx * 4]);
        const float u = gs[2 * ty][c], v = gs[2 * ty + 1][c];
        a0.x = fmaf(u, w.x, a0.x); a0.y = fmaf(u, w.y, a0.y); a0.z = fmaf(u, w.z, a0.z); a0.w = fmaf(u, w.w, a0.w);
        a1.x = fmaf(v, w.x, a1.x); a1.y = fmaf(v, w.y, a1.y); a1.z = fmaf(v, w.z, a1.z); a1.w = fmaf(v, w.w, a1.w);
    }
    const int64_t k = k0 + tx * 4;
    if (k < K) {
        if (m0 + 2 * ty < M) *reinterpret_cast<float4*>(gx + (m0 + 2 * ty) * K + k) = a0;
        if (m0 + 2 * ty + 1 < M) *reinterpret_cast<float4*>(gx + (m0 + 2 * ty + 1) * K + k) = a1;
    }
}

// ---- parameter gradients: part[s][c][k] = sum over this slice's rows of g[m][c] x[m][k]; the k-tile 0 workgroups also sum g's columns.
// Rows are walked 32 at a time through LDS, the next 32 requested (into registers) before this chunk's arithmetic.
__global__ __launch_bounds__(TB) void skinny_wgrad_kernel(const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ x, Pair p,
                                                          int64_t M, int64_t K, int rslice, float* __restrict__ part, float* __restrict__ part_b) {
    constexpr int GV = RT * NMAX / TB;               // elements of g per thread and chunk (at most)
    constexpr int XV = RT * KT / 4 / TB;             // float4 of x
    __shared__ float gs[RT][NMAX + 1];
    __shared__ float xs[RT][KT + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;      // 16 x float4 of k; columns ty, ty + 16, ..., ty + 112
    const int NT = p.n1 + p.n2;
    const int64_t k0 = (int64_t)blockIdx.x * KT;
    const int64_t m_lo = (int64_t)blockIdx.y * rslice, m_hi = m_lo + rslice < M ? m_lo + rslice : M;
    float4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    float colsum = 0.f;                                             // (k-tile 0: thread c < NT sums column c)
    int grow[GV], gcol[GV];                                         // this thread's (row, column) of the g tile, the same every chunk
#pragma unroll
    for (int u = 0; u < GV; ++u) {
        const int idx = tid + TB * u;
        grow[u] = idx < RT * NT ? idx / NT : -1;
        gcol[u] = idx < RT * NT ? idx - grow[u] * NT : 0;
    }
    float gr[GV];
    float4 xr[XV];
    auto request = [&](int64_t m0) {
#pragma unroll
        for (int u = 0; u < GV; ++u) {
            const int r = grow[u], c = gcol[u];
            float v = 0.f;
            if (r >= 0 && m0 + r < m_hi) v = c < p.n1 ? g1[(m0 + r) * p.n1 + c] : g2[(m0 + r) * p.n2 + c - p.n1];
            gr[u] = v;
        }
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int idx = tid + TB * u, r = idx / (KT / 4), k4 = (idx % (KT / 4)) * 4;
            xr[u] = (m0 + r < m_hi && k0 + k4 < K) ? *reinterpret_cast<const float4*>(x + (m0 + r) * K + k0 + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    request(m_lo);
    for (int64_t m0 = m_lo; m0 < m_hi; m0 += RT) {
#pragma unroll
        for (int u = 0; u < GV; ++u)
            if (grow[u] >= 0) gs[grow[u]][gcol[u]] = gr[u];
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int idx = tid + TB * u, r = idx / (KT / 4), k4 = (idx % (KT / 4)) * 4;
            *reinterpret_cast<float4*>(&xs[r][k4]) = xr[u];
        }
        __syncthreads();
        if (m0 + RT < m_hi) request(m0 + RT);
#pragma unroll 4
        for (int r = 0; r < RT; ++r) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[r][tx * 4]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float u = ty + 16 * j < NT ? gs[r][ty + 16 * j] : 0.f;
                acc[j].x = fmaf(u, xv.x, acc[j].x); acc[j].y = fmaf(u, xv.y, acc[j].y);
                acc[j].z = fmaf(u, xv.z, acc[j].z); acc[j].w = fmaf(u, xv.w, acc[j].w);
            }
        }
        if (blockIdx.x == 0 && tid < NT)
            for (int r = 0; r < RT; ++r) colsum += gs[r][tid];
        __syncthreads();
    }
    const int64_t k = k0 + tx * 4;
    if (k < K)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ty + 16 * j;
            if (c < NT) *reinterpret_cast<float4*>(part + ((int64_t)blockIdx.y * NT + c) * K + k) = acc[j];
        }
    if (blockIdx.x == 0 && tid < NT) part_b[(int64_t)blockIdx.y * NT + tid] = colsum;
}

// gw (+)= the row slices' partials in slice order; gb likewise
__global__ __launch_bounds__(TB) void skinny_wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_b, int slices, int n1, int n2,
                                                                 int64_t K, float* __restrict__ gw1, float* __restrict__ gw2, float* __restrict__ gb1,
                                                                 float* __restrict__ gb2, int accumulate) {
    const int NT = n1 + n2;
    const int64_t total = (int64_t)NT * K, i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < total) {
        const int c = (int)(i / K);
        const int64_t k = i - (int64_t)c * K;
        float s = part[i];
        int z = 1;
        for (; z + 3 < slices; z += 4) {
            const float a = part[(int64_t)z * total + i], b = part[(int64_t)(z + 1) * total + i], c2 = part[(int64_t)(z + 2) * total + i],
                        d = part[(int64_t)(z + 3) * total + i];
            s += a; s += b; s += c2; s += d;
        }
        for (; z < slices; ++z) s += part[(int64_t)z * total + i];
        float* dst = c < n1 ? gw1 + (int64_t)c * K + k : gw2 + (int64_t)(c - n1) * K + k;
        *dst = accumulate ? *dst + s : s;
    } else if (i < total + NT) {
        const int c = (int)(i - total);
        float* dst = c < n1 ? (gb1 ? gb1 + c : nullptr) : (gb2 ? gb2 + c - n1 : nullptr);
        if (dst) {
            float s = part_b[c];
            for (int z = 1; z < slices; ++z) s += part_b[(int64_t)z * NT + c];
            *dst = accumulate ? *dst + s : s;
        }
    }
}

// forward K slices / parameter-gradient row slices: enough workgroups for the chip, whole staging chunks
int fwd_chunk(int NT) { return NT <= 64 ? 128 : 64; }       // reduction steps per LDS stage: 128 up to 64 features, else 64 (LDS)
// One slice per staging chunk, whatever M is: a row's result is then the same sum — chunk partials in chunk order — in a batch of 128 rows
// and in a batch of 896 (the Faster-RCNN passes run their heads pass by pass or several passes at once: the same bits either way), and
// few rows still spread over K / chunk workgroups per row tile.  (Up to 64 slices; longer reductions take several chunks per slice.)
int fwd_slices(int64_t M, int64_t K, int NT) {
    (void)M;
    const int64_t kc = fwd_chunk(NT), chunks = (K + kc - 1) / kc;
    return (int)(chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks));
}
int fwd_kslice(int64_t K, int slices, int NT) {
    const int64_t kc = fwd_chunk(NT), chunks = (K + kc - 1) / kc;
    return (int)(((chunks + slices - 1) / slices) * kc);
}
int wgrad_slices(int64_t M, int64_t K) {
    const int64_t ktiles = (K + KT - 1) / KT, rows = (M + RT - 1) / RT;
    int64_t s = ktiles >= 256 ? 1 : (256 + ktiles - 1) / ktiles;      // ~256 workgroups: short row walks (each chunk is one memory latency)
    if (s > rows) s = rows;
    return (int)(s < 1 ? 1 : s);
}
int wgrad_rslice(int64_t M, int slices) {
    const int64_t rows = (M + RT - 1) / RT;
    return (int)(((rows + slices - 1) / slices) * RT);
}

int check_pair(int64_t M, int64_t n1, int64_t n2, int64_t K) {
    if (M < 0 || n1 < 1 || n2 < 0 || K < 4 || (K & 3) || n1 + n2 > NMAX || M * K > 0x7fffffffLL * 4) return AFAN_ESHAPE;
    return AFAN_OK;
}

}  // namespace

extern "C" {

// floats of workspace the forward (op 0) / the parameter gradients (op 1) of this shape need (0: none)
int64_t afan_linear_pair_workspace_floats(int op, int64_t M, int64_t n1, int64_t n2, int64_t K) {
    if (check_pair(M, n1, n2, K) != AFAN_OK) return -1;
    const int64_t NT = n1 + n2;
    if (op == 0) {
        const int s = fwd_slices(M, K, (int)NT);
        return s > 1 ? (int64_t)s * M * NT : 0;
    }
    const int s = wgrad_slices(M, K);
    return (int64_t)s * NT * (K + 1);
}

int afan_linear_pair_fwd_f32(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y1, float* y2, int64_t M, int64_t n1,
                             int64_t n2, int64_t K, float* ws, afan_stream_t stream) {
    int e = check_pair(M, n1, n2, K);
    if (e) return e;
    if (!x || !w1 || !y1 || (n2 > 0 && (!w2 || !y2))) return AFAN_ENULL;
    if (!aligned(x, 16) || !aligned(w1, 16) || (n2 > 0 && !aligned(w2, 16))) return AFAN_EALIGN;
    if (M == 0) return AFAN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int NT = (int)(n1 + n2), slices = fwd_slices(M, K, NT), kslice = fwd_kslice(K, slices, NT), nc = (NT + 31) / 32;
    if (slices > 1 && !ws) return AFAN_ENULL;
    Pair p{w1, w2, b1, b2, (int)n1, (int)n2};
    dim3 grid((unsigned)((M + RT - 1) / RT), (unsigned)slices);
    float* part = slices > 1 ? ws : nullptr;
    AFAN_PROF("linear_pair_fwd_kernel", 4.0 * (M * K + NT * K + M * NT), st);
    switch (nc) {
        case 1: skinny_fwd_kernel<1, 128><<<grid, TB, 0, st>>>(x, p, M, K, kslice, part, y1, y2); break;
        case 2: skinny_fwd_kernel<2, 128><<<grid, TB, 0, st>>>(x, p, M, K, kslice, part, y1, y2); break;
        case 3: skinny_fwd_kernel<3, 64><<<grid, TB, 0, st>>>(x, p, M, K, kslice, part, y1, y2); break;
        default: skinny_fwd_kernel<4, 64><<<grid, TB, 0, st>>>(x, p, M, K, kslice, part, y1, y2); break;
    }
    AFAN_LAUNCH_CHECK();
    if (slices > 1) {
        skinny_fwd_reduce_kernel<<<(unsigned)((M * NT + TB - 1) / TB), TB, 0, st>>>(ws, slices, p, M, y1, y2);
        AFAN_LAUNCH_CHECK();
    }
    return AFAN_OK;
}

int afan_linear_pair_dgrad_f32(const float* g1, const float* g2, const float* w1, const float* w2, float* gx, int64_t M, int64_t n1, int64_t n2, int64_t K,
                               afan_stream_t stream) {
    int e = check_pair(M, n1, n2, K);
    if (e) return e;
    if (!g1 || !w1 || !gx || (n2 > 0 && (!g2 || !w2))) return AFAN_ENULL;
    if (!aligned(gx, 16) || !aligned(w1, 16) || (n2 > 0 && !aligned(w2, 16))) return AFAN_EALIGN;
    if (M == 0) return AFAN_OK;
    hipStream_t st = (hipStream_t)stream;
    Pair p{w1, w2, nullptr, nullptr, (int)n1, (int)n2};
    dim3 grid((unsigned)((M + RT - 1) / RT), (unsigned)((K + KT - 1) / KT));
    AFAN_PROF("linear_pair_dgrad_kernel", 4.0 * (M * K + (n1 + n2) * K + M * (n1 + n2)), st);
    skinny_dgrad_kernel<<<grid, TB, 0, st>>>(g1, g2, p, M, K, gx);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

int afan_linear_pair_wgrad_f32(const float* g1, const float* g2, const float* x, float* gw1, float* gb1, float* gw2, float* gb2, int accumulate, int64_t M,
                               int64_t n1, int64_t n2, int64_t K, float* ws, afan_stream_t stream) {
    int e = check_pair(M, n1, n2, K);
    if (e) return e;
    if (!g1 || !x || !gw1 || !ws || (n2 > 0 && (!g2 || !gw2))) return AFAN_ENULL;
    if (!aligned(x, 16) || !aligned(ws, 16)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int NT = (int)(n1 + n2), slices = M > 0 ? wgrad_slices(M, K) : 1, rslice = M > 0 ? wgrad_rslice(M, slices) : RT;
    Pair p{nullptr, nullptr, nullptr, nullptr, (int)n1, (int)n2};
    dim3 grid((unsigned)((K + KT - 1) / KT), (unsigned)slices);
    float* part_b = ws + (int64_t)slices * NT * K;
    AFAN_PROF("linear_pair_wgrad_kernel", 4.0 * (M * K + NT * K + M * NT), st);
    skinny_wgrad_kernel<<<grid, TB, 0, st>>>(g1, g2, x, p, M, K, rslice, ws, part_b);
    AFAN_LAUNCH_CHECK();
    skinny_wgrad_reduce_kernel<<<(unsigned)(((int64_t)NT * K + NT + TB - 1) / TB), TB, 0, st>>>(ws, part_b, slices, (int)n1, (int)n2, K, gw1, gw2, gb1, gb2,
                                                                                             accumulate);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
