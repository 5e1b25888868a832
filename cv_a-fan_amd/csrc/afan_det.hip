// Detection operators of the A-FAN Detection step: the only native code the reference ships —
//   NMS        Detection/support/src/cuda/nms.cu:23-67 (mask kernel) + :70-130 (host-side greedy scan)
//   ROIAlign   Detection/support/src/cuda/ROIAlign_cuda.cu:65-122 (forward), :178-254 (backward)
// re-done for gfx950.
//
// NMS: boxes are visited in descending-score order (`order`, computed by the caller's sort).  The 64 x 64 overlap tiles map
// one to one onto wave64: one wave per tile, lane = row box, bit = column box, a 64-bit word per (row, column block).  Only
// tiles on or above the diagonal are computed (the scan never reads the others).  The reference then copies the whole mask
// to the HOST and scans it there (12 MB and a device sync per call at 9 770 boxes); here the greedy scan stays on the
// device: one workgroup walks the column blocks in order — a scanner wave resolves each diagonal tile and the band of tiles
// right of it, worker waves OR the kept rows' words of the far column blocks into the running `removed` bit set (LDS); see
// nms_scan_kernel.  Kept boxes are flagged at their ORIGINAL index and compacted by a prefix sum — ascending original
// indices, what nms_cuda returns after its sort.
// Suppression test: IoU > threshold like nms.cu:49 (inclusive = 0) or IoU >= threshold like nms_cpu.cpp:62 (inclusive = 1);
// areas and intersections with the reference's +1 (pixel-inclusive corners).
#include "afan_common.h"

using namespace afan;

namespace {

constexpr int W64 = 64;

// "IoU(a, b) > thresh" (inclusive: >=) with the reference's arithmetic — fp32, corners inclusive (+1), inter / (sa + sb -
// inter) by an IEEE division (nms.cu:36-49) — WITHOUT the division for all but the closest calls: with u = sa + sb - inter
// and q = fl(thresh * u), inter outside q (1 -+ 2e-6) decides the comparison of the ROUNDED quotient as well (the quotient,
// q and its two bounds are each within 2^-24 of their exact values, 30 times less than the margin); if any lane of the wave
// sits inside the margin, or has a degenerate union, the wave divides.  sa / sb: the boxes' areas, computed once per box;
// t_sane: the threshold is in [1e-3, 1] (wave-uniform).  ~28 instructions per pair against ~50 with the division's branches
// (the mask kernel is VALU-bound: 72 M pairs for 12 000 boxes).
__device__ __forceinline__ float box_area_incl(const float* a) { return (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f); }
// v_max / v_min without the canonicalisation fmaxf() adds for operands straight from memory (a NaN box is nobody's input)
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ bool iou_over(const float* a, float sa, const float* b, float sb, float thresh, int inclusive, bool t_sane) {
    const float left = vmax(a[0], b[0]), right = vmin(a[2], b[2]);
    const float top = vmax(a[1], b[1]), bottom = vmin(a[3], b[3]);
    const float w = fmaxf(right - left + 1.f, 0.f), h = fmaxf(bottom - top + 1.f, 0.f);
    const float inter = w * h;
    const float u = sa + sb - inter;
    const float q = thresh * u;
    const bool sane = t_sane && u >= 1.f && u <= 1e30f;
    bool hit = sane && inter > q * 1.000002f;
    const bool decided = hit || (sane && inter < q * 0.999998f);
    if (__ballot(!decided)) {                                      // wave-uniform, rare
        const float v = inter / u;
        hit = decided ? hit : (inclusive ? (v >= thresh) : (v > thresh));
    }
    return hit;
}

// The scan (below) resolves, per 64-box block, the diagonal tile AND the SC_D column blocks right of it by itself; for those
// "band" tiles it wants the TRANSPOSED words — one 64-bit word per COLUMN box, bit = row box — because "is column c struck
// by a kept row of this block" is then (word[c] & keep) != 0: one AND per column, lane-parallel, no reduction over rows.
#ifndef AFAN_SC_D
#define AFAN_SC_D 12
#endif
constexpr int SC_D = AFAN_SC_D;

// grid (col_blocks, col_blocks), one wave per tile.  The diagonal tile and the SC_D tiles right of it store the TRANSPOSED
// tile into bandT[COLUMN block][k = column block - row block: 0 = diagonal, 1..SC_D][column] — the comparison's wave-wide ballot
// IS the column's word; everything the scan needs to settle column block j against the SC_D blocks before it is one contiguous
// (SC_D + 1) x 512 bytes — and no row form: the scan reads rows only of the tiles further right.
__global__ __launch_bounds__(W64) void nms_mask_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ order, int n,
                                                       float thresh, int inclusive, unsigned long long* __restrict__ mask,
                                                       unsigned long long* __restrict__ bandT, int col_blocks) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;
    const int row_size = min(n - rb * W64, W64), col_size = min(n - cb * W64, W64);
    __shared__ __attribute__((aligned(16))) float cbox[W64 * 4];
    __shared__ float carea[W64];
    if ((int)threadIdx.x < col_size) {
        const float* src = boxes + order[cb * W64 + threadIdx.x] * 4;
        cbox[threadIdx.x * 4 + 0] = src[0]; cbox[threadIdx.x * 4 + 1] = src[1];
        cbox[threadIdx.x * 4 + 2] = src[2]; cbox[threadIdx.x * 4 + 3] = src[3];
        carea[threadIdx.x] = box_area_incl(src);
    }
    __syncthreads();
    const bool t_sane = thresh >= 1e-3f && thresh <= 1.f;
    const bool valid = (int)threadIdx.x < row_size;
    const int i = rb * W64 + threadIdx.x;
    float cur[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        const float* p = boxes + order[i] * 4;
        cur[0] = p[0]; cur[1] = p[1]; cur[2] = p[2]; cur[3] = p[3];
    }
    const float area = box_area_incl(cur);
    if (cb - rb <= SC_D) {
        const int first = (rb == cb) ? (int)threadIdx.x + 1 : 0;   // the diagonal tile: boxes after this row's only
        unsigned long long tw = 0;                                 // lane c: column c's word
        for (int j = 0; j < col_size; ++j) {
            const bool hit = valid && j >= first && iou_over(cur, area, cbox + j * 4, carea[j], thresh, inclusive, t_sane);
            const unsigned long long col = __ballot(hit);
            if ((int)threadIdx.x == j) tw = col;
        }
        bandT[((int64_t)cb * (SC_D + 1) + (cb - rb)) * W64 + threadIdx.x] = tw;
    } else if (valid) {
        unsigned long long t = 0;
        for (int j = 0; j < col_size; ++j) {
            if (iou_over(cur, area, cbox + j * 4, carea[j], thresh, inclusive, t_sane)) t |= 1ULL << j;
        }
        mask[(int64_t)i * col_blocks + cb] = t;                    // (row form: read by the scan's workers, far tiles only)
    }
}

// ---- the scan ---------------------------------------------------------------------------------------------------------------
// One workgroup of 16 waves in three roles that meet only through LDS words (no workgroup barrier inside the loop; every
// wave of the one workgroup is resident, so a spin always has someone to wait for):
//   SCANNER (wave 0), per block b: removed[b] and the block's SC_D + 1 transposed tiles in ONE LDS round trip (flags first,
//     the data they guard behind them), requested one block ahead -> lane c asks whether a kept row of the SC_D blocks before
//     b strikes column c (tile_k[c] & keep[b - k], keep words in scalar registers) -> the greedy recurrence on the diagonal
//     tile, by rounds -> the keep word.  It touches no global memory and writes nothing but its keep word and position.
//   PREFETCHERS (waves 4, 8, 12 — the scanner's SIMD, where they mostly sleep on memory): the transposed tiles of the
//     blocks up to SC_R ahead of the scanner into a ring.
//   WORKERS (the other 12 waves), block p to worker p mod 12: the kept rows' words of the column blocks from p + SC_D + 1
//     on, OR-ed into removed[] — due only when the scanner asks for block p + SC_D + 1, so a worker's memory round trips
//     (1.5 - 3 us) hide behind SC_D scanner iterations.  The owner of block p also writes its kept flags.
// removed[j] is complete when the scanner reads it: blocks j - SC_D .. j - 1 are the scanner's own band, block j - SC_D - 1
// is the worker it waits for, older blocks the waits of earlier iterations.  ORs commute: the result does not depend on timing.
// History (12 000 boxes, the headline step's proposals, ~1 800 survivors, all 188 blocks scanned): one wave doing everything
// 1.42 ms; 16 waves between two barriers per block, a scalar step per kept box, 283 us; barrier-free with the band pushed
// forward by LDS atomics 117 us; this form (band pulled, no atomics, reads one block ahead) 116 us with a third fewer
// instructions — what is left is ~1 300 cycles per block of DEPENDENT hops of one wave (LDS -> VALU -> ballot -> SALU -> VALU
// ...), not instruction count, not waiting (the scanner's spins: ~10 per call).  SC_D 6 .. 12, SC_R 8 .. 12: within 3 %.
constexpr int SCAN_WAVES = 16;
#ifndef AFAN_SC_R
#define AFAN_SC_R 8
#endif
constexpr int SC_R = AFAN_SC_R;         // ring slots
constexpr int SC_P = 3;                 // prefetch waves
constexpr int SC_G = 2;                 // blocks a prefetch wave requests per round trip
constexpr int SC_NW = SCAN_WAVES - 1 - SC_P;
constexpr int SC_KW = 32;               // ring of keep words (a worker is never more than SC_D + 1 blocks behind)
constexpr int SC_RC = 12;               // kept rows a worker requests per round trip (x 3 stretches of 64 column blocks)
static_assert(SC_R >= SC_P * SC_G && SC_KW > SC_D + 2, "ring sizes");

struct ScanShared {
    unsigned long long bt[SC_R][SC_D + 1][W64];                   // [slot][0]: the diagonal tile, [1..]: the band, by column
    unsigned long long keepw[SC_KW];
    int pf_ready[SC_R];
    int work_done[SC_NW];
    int scan_pos, stop;
};

// wave-uniform by construction (every lane reads the same word): handed back in a scalar register so the loops on it are
// scalar branches, not exec-mask loops
__device__ __forceinline__ int lds_acquire(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_release(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned long long to_scalar(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// The LDS unit executes one wave's instructions in issue order; where both sides of a hand-over are LDS accesses of the SAME
// wave order (flag read before data reads; data writes before flag write) a compiler barrier is all the fence it takes —
// no s_waitcnt between them, the round trips overlap.
#define LDS_ORDER() asm volatile("" ::: "memory")

__global__ __launch_bounds__(W64 * SCAN_WAVES) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                                    const unsigned long long* __restrict__ bandT,
                                                                    const int64_t* __restrict__ order, int n, int col_blocks,
                                                                    uint8_t* __restrict__ kept_flag, int max_keep) {
    extern __shared__ unsigned long long removed[];
    __shared__ ScanShared S;
    const int lane = threadIdx.x & (W64 - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / W64);
    for (int j = threadIdx.x; j < col_blocks; j += W64 * SCAN_WAVES) removed[j] = 0;
    if (threadIdx.x < SC_R) S.pf_ready[threadIdx.x] = 0;
    if (threadIdx.x < SC_NW) S.work_done[threadIdx.x] = 0;
    if (threadIdx.x == 0) { S.scan_pos = 0; S.stop = 0; }
    __syncthreads();

    if (wave == 0) {
        // ---- scanner ----
        // Block b's verdict needs: removed[b] (the workers' share: kept rows of blocks <= b - SC_D - 1), the keep words of the SC_D
        // blocks before it (kept here, in scalar registers) against the band tiles (b - k, b) in column form — lane c asks
        // OR_k (tile_k[c] & keep[b - k]) != 0, two v_and_or per tile — and the diagonal tile for the recurrence.  The scanner
        // writes nothing the next block reads from LDS, so block b + 1's words are requested BEFORE block b's work (two
        // register sets, the loop unrolled by two) and the LDS round trip hides behind it.
        struct Regs { unsigned long long bw[SC_D + 1], rem_v; int r1, r2; };
        auto issue = [&](int b, Regs& R) {
            const int need = b - SC_D - 1;
            LDS_ORDER();                                           // flags first, then what they guard: the LDS unit keeps the order
            R.r1 = S.pf_ready[b % SC_R];
            R.r2 = S.work_done[need >= 0 ? need % SC_NW : 0];
            LDS_ORDER();
            R.rem_v = removed[b];
#pragma unroll
            for (int k = 0; k <= SC_D; ++k) R.bw[k] = S.bt[b % SC_R][k][lane];
            LDS_ORDER();
        };
        unsigned long long kh[SC_D + 1];                           // kh[k]: keep word of block b - k (k = 1 .. SC_D)
#pragma unroll
        for (int k = 0; k <= SC_D; ++k) kh[k] = 0;
        int kept_total = 0;
        auto step = [&](int b, Regs& cur, Regs& nxt) -> bool {
            const int need = b - SC_D - 1;
            while (!(__builtin_amdgcn_readfirstlane(cur.r1) == b + 1 && (need < 0 || __builtin_amdgcn_readfirstlane(cur.r2) > need))) {
                __builtin_amdgcn_s_sleep(1);
                issue(b, cur);
            }
            if (b + 1 < col_blocks) issue(b + 1, nxt);
            unsigned acc_lo = 0, acc_hi = 0;
#pragma unroll
            for (int k = 1; k <= SC_D; ++k) {
                acc_lo |= (unsigned)cur.bw[k] & (unsigned)kh[k];
                acc_hi |= (unsigned)(cur.bw[k] >> 32) & (unsigned)(kh[k] >> 32);
            }
            const int size = min(n - b * W64, W64);
            unsigned long long alive = ~(to_scalar(cur.rem_v) | __ballot((acc_lo | acc_hi) != 0u));
            if (size < W64) alive &= ~(~0ULL << size);             // rows beyond n: never kept
            // the greedy recurrence by ROUNDS on the diagonal tile in column form (bw[0]: bit r = earlier box r strikes this
            // lane's box): an undecided box no undecided earlier box overlaps is kept — every earlier overlapping box is
            // decided, and a kept one would have struck it — so all such boxes are kept at once and strike theirs.  The
            // first undecided box always qualifies; a block takes as many rounds as its longest chain of overlaps (1 - 4),
            // not one scalar step per kept box.
            unsigned long long keep = 0;
            while (alive) {
                const unsigned long long newly = __ballot((cur.bw[0] & alive) == 0ULL) & alive;
                keep |= newly;
                alive &= ~(newly | __ballot((cur.bw[0] & newly) != 0ULL));
            }
#pragma unroll
            for (int k = SC_D; k > 1; --k) kh[k] = kh[k - 1];
            kh[1] = keep;
            kept_total += __popcll(keep);
            const bool fin = max_keep > 0 && kept_total >= max_keep;
            LDS_ORDER();
            if (lane == 0) {
                S.keepw[b % SC_KW] = keep;
                LDS_ORDER();
                S.scan_pos = b + 1;
                LDS_ORDER();
                if (fin) S.stop = 1;                               // after scan_pos: who sees stop sees the last block too
            }
            LDS_ORDER();
            return fin;
        };
        Regs ra, rb;
        issue(0, ra);
        for (int b = 0; b < col_blocks; b += 2) {
            if (step(b, ra, rb)) break;
            if (b + 1 < col_blocks && step(b + 1, rb, ra)) break;
        }
    } else if ((wave & 3) == 0) {
        // ---- prefetchers ----
        const int q = wave / 4 - 1;
        for (int b0 = q * SC_G; b0 < col_blocks; b0 += SC_P * SC_G) {
            bool quit = false;
            while (lds_acquire(&S.scan_pos) < b0 + SC_G - SC_R) {  // the slots of b0 .. b0 + G - 1 are free
                if (lds_acquire(&S.stop)) { quit = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (quit) break;
            unsigned long long w[SC_G][SC_D + 1];
#pragma unroll
            for (int t = 0; t < SC_G; ++t) {
                const int blk = b0 + t;
#pragma unroll
                for (int k = 0; k <= SC_D; ++k)
                    w[t][k] = (blk < col_blocks && k <= blk) ? bandT[((int64_t)blk * (SC_D + 1) + k) * W64 + lane] : 0ULL;
            }
#pragma unroll
            for (int t = 0; t < SC_G; ++t) {
                const int blk = b0 + t;
                if (blk >= col_blocks) break;
                const int slot = blk % SC_R;
#pragma unroll
                for (int k = 0; k <= SC_D; ++k) S.bt[slot][k][lane] = w[t][k];
                if (lane == 0) lds_release(&S.pf_ready[slot], blk + 1);
            }
        }
    } else {
        // ---- workers ----
        const int wi = wave - 1 - wave / 4;                        // waves 1,2,3,5,6,7,... -> 0..11
        for (int p = wi; p < col_blocks; p += SC_NW) {
            int sp;
            bool stopped = false;
            for (;;) {
                sp = lds_acquire(&S.scan_pos);
                if (sp > p) break;
                if (lds_acquire(&S.stop)) { sp = lds_acquire(&S.scan_pos); stopped = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (sp <= p) break;                                    // the scan ended before this block
            stopped = stopped || lds_acquire(&S.stop) != 0;
            const unsigned long long keep = to_scalar(S.keepw[p % SC_KW]);
            const int jf = p + SC_D + 1;
            if (!stopped && jf < col_blocks) {
                unsigned long long kk = keep;
                while (kk) {
                    int rsel[SC_RC];
#pragma unroll
                    for (int q = 0; q < SC_RC; ++q) {
                        rsel[q] = kk ? __ffsll((long long)kk) - 1 : -1;
                        kk &= kk - 1;
                    }
                    for (int j0 = jf; j0 < col_blocks; j0 += 3 * W64) {
                        unsigned long long v[SC_RC][3];
#pragma unroll
                        for (int q = 0; q < SC_RC; ++q)
#pragma unroll
                            for (int t = 0; t < 3; ++t) {
                                const int j = j0 + t * W64 + lane;
                                v[q][t] = (rsel[q] >= 0 && j < col_blocks) ? mask[((int64_t)p * W64 + rsel[q]) * col_blocks + j] : 0ULL;
                            }
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            unsigned long long o = 0;
#pragma unroll
                            for (int q = 0; q < SC_RC; ++q) o |= v[q][t];
                            const int j = j0 + t * W64 + lane;
                            if (o) atomicOr(&removed[j], o);
                        }
                    }
                }
            }
            if (lane == 0) lds_release(&S.work_done[wi], p + 1);
            // the flags last: the release above must not wait for these stores
            const int size = min(n - p * W64, W64);
            if (lane < size && ((keep >> lane) & 1ULL)) kept_flag[order[p * W64 + lane]] = 1;
        }
    }
}

// ascending original indices of the flagged boxes; one workgroup (prefix sum over chunks)
constexpr int CP_THREADS = 1024;
__global__ __launch_bounds__(CP_THREADS) void nms_compact_kernel(const uint8_t* __restrict__ kept_flag, int n, int64_t* __restrict__ keep_out,
                                                                 int64_t* __restrict__ count_out) {
    __shared__ int sums[CP_THREADS];
    const int per = (n + CP_THREADS - 1) / CP_THREADS;
    const int lo = threadIdx.x * per, hi = min(lo + per, n);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += kept_flag[i] ? 1 : 0;
    sums[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < CP_THREADS; o <<= 1) {             // Hillis-Steele inclusive scan
        const int v = (int)threadIdx.x >= o ? sums[threadIdx.x - o] : 0;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    int pos = sums[threadIdx.x] - c;
    for (int i = lo; i < hi; ++i)
        if (kept_flag[i]) keep_out[pos++] = i;
    if (threadIdx.x == CP_THREADS - 1) count_out[0] = sums[threadIdx.x];
}

// ---- ROIAlign ---------------------------------------------------------------------------------------------------------------
struct Bilin {
    int y_low, x_low, y_high, x_high;
    float w1, w2, w3, w4;
    bool empty;
};
// ROIAlign_cuda.cu:16-62 / :125-170: sample points outside [-1, size] contribute nothing; clamped to the border otherwise
__device__ __forceinline__ Bilin bilin_prep(int height, int width, float y, float x) {
    Bilin b;
    b.empty = (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width);
    if (b.empty) { b.y_low = b.x_low = b.y_high = b.x_high = -1; b.w1 = b.w2 = b.w3 = b.w4 = 0.f; return b; }
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    b.y_low = (int)y;
    b.x_low = (int)x;
    if (b.y_low >= height - 1) { b.y_high = b.y_low = height - 1; y = (float)b.y_low; } else b.y_high = b.y_low + 1;
    if (b.x_low >= width - 1) { b.x_high = b.x_low = width - 1; x = (float)b.x_low; } else b.x_high = b.x_low + 1;
    const float ly = y - (float)b.y_low, lx = x - (float)b.x_low;
    const float hy = 1.f - ly, hx = 1.f - lx;
    b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
    return b;
}

struct RoiGeom {
    int batch;
    float start_w, start_h, bin_h, bin_w;
    int grid_h, grid_w;
    float count;
};
__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int ph_n, int pw_n, int sampling_ratio) {
    RoiGeom g;
    g.batch = (int)roi[0];
    g.start_w = roi[1] * scale; g.start_h = roi[2] * scale;          // "Do not using rounding; this implementation detail is critical"
    const float end_w = roi[3] * scale, end_h = roi[4] * scale;
    const float rw = fmaxf(end_w - g.start_w, 1.f), rh = fmaxf(end_h - g.start_h, 1.f);   // malformed ROIs forced to 1x1
    g.bin_h = rh / (float)ph_n; g.bin_w = rw / (float)pw_n;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph_n);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw_n);
    g.count = (float)(g.grid_h * g.grid_w);
    return g;
}

// index over the pooled output.  NCHW: (n, c, ph, pw) with pw fastest (the reference's layout); NHWC: (n, ph, pw, c) with c
// fastest — lanes run along channels, so the four corner reads of a sample point are coalesced 256-byte rows.
template <typename T, bool NHWC>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const T* __restrict__ x, const float* __restrict__ rois, T* __restrict__ y,
                                                            int64_t total, float scale, int C, int H, int Wd, int PH, int PW,
                                                            int sampling_ratio) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int pw, ph, c;
        int64_t n;
        if (NHWC) { c = (int)(idx % C); pw = (int)((idx / C) % PW); ph = (int)((idx / C / PW) % PH); n = idx / C / PW / PH; }
        else { pw = (int)(idx % PW); ph = (int)((idx / PW) % PH); c = (int)((idx / PW / PH) % C); n = idx / PW / PH / C; }
        const RoiGeom g = roi_geom(rois + n * 5, scale, PH, PW, sampling_ratio);
        float out = 0.f;
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float yy = g.start_h + ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float xx = g.start_w + pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                const Bilin b = bilin_prep(H, Wd, yy, xx);
                if (b.empty) continue;
                float v1, v2, v3, v4;
                if (NHWC) {
                    const T* base = x + (int64_t)g.batch * H * Wd * C + c;
                    v1 = Elt<T>::ld(base + ((int64_t)b.y_low * Wd + b.x_low) * C); v2 = Elt<T>::ld(base + ((int64_t)b.y_low * Wd + b.x_high) * C);
                    v3 = Elt<T>::ld(base + ((int64_t)b.y_high * Wd + b.x_low) * C); v4 = Elt<T>::ld(base + ((int64_t)b.y_high * Wd + b.x_high) * C);
                } else {
                    const T* base = x + ((int64_t)g.batch * C + c) * H * Wd;
                    v1 = Elt<T>::ld(base + b.y_low * Wd + b.x_low); v2 = Elt<T>::ld(base + b.y_low * Wd + b.x_high);
                    v3 = Elt<T>::ld(base + b.y_high * Wd + b.x_low); v4 = Elt<T>::ld(base + b.y_high * Wd + b.x_high);
                }
                out += b.w1 * v1 + b.w2 * v2 + b.w3 * v3 + b.w4 * v4;
            }
        }
        Elt<T>::st(y + idx, out / g.count);
    }
}

// scatter with hardware fp32 atomics into a zeroed fp32 gradient map (the reference's atomicAdd, ROIAlign_cuda.cu:238-241:
// the summation order — and so the last bits — varies from run to run there too)
template <typename T, bool NHWC>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ rois, float* __restrict__ dx,
                                                            int64_t total, float scale, int C, int H, int Wd, int PH, int PW,
                                                            int sampling_ratio) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int pw, ph, c;
        int64_t n;
        if (NHWC) { c = (int)(idx % C); pw = (int)((idx / C) % PW); ph = (int)((idx / C / PW) % PH); n = idx / C / PW / PH; }
        else { pw = (int)(idx % PW); ph = (int)((idx / PW) % PH); c = (int)((idx / PW / PH) % C); n = idx / PW / PH / C; }
        const RoiGeom g = roi_geom(rois + n * 5, scale, PH, PW, sampling_ratio);
        const float gtop = Elt<T>::ld(dy + idx);
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float yy = g.start_h + ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float xx = g.start_w + pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                const Bilin b = bilin_prep(H, Wd, yy, xx);
                if (b.empty) continue;
                const float g1 = gtop * b.w1 / g.count, g2 = gtop * b.w2 / g.count, g3 = gtop * b.w3 / g.count, g4 = gtop * b.w4 / g.count;
                if (NHWC) {
                    float* base = dx + (int64_t)g.batch * H * Wd * C + c;
                    unsafeAtomicAdd(base + ((int64_t)b.y_low * Wd + b.x_low) * C, g1); unsafeAtomicAdd(base + ((int64_t)b.y_low * Wd + b.x_high) * C, g2);
                    unsafeAtomicAdd(base + ((int64_t)b.y_high * Wd + b.x_low) * C, g3); unsafeAtomicAdd(base + ((int64_t)b.y_high * Wd + b.x_high) * C, g4);
                } else {
                    float* base = dx + ((int64_t)g.batch * C + c) * H * Wd;
                    unsafeAtomicAdd(base + b.y_low * Wd + b.x_low, g1); unsafeAtomicAdd(base + b.y_low * Wd + b.x_high, g2);
                    unsafeAtomicAdd(base + b.y_high * Wd + b.x_low, g3); unsafeAtomicAdd(base + b.y_high * Wd + b.x_high, g4);
                }
            }
        }
    }
}

// Channels-last forms with the GEOMETRY hoisted: one workgroup per output bin (roi, ph, pw).  The element-per-thread kernels
// above redo the index split, the ROI geometry and every sample point's bilinear setup for each of the bin's C channels
// (~300 instructions per element: 222 us for 128 rois x 14 x 14 x 1024 channels); here a thread
// does them once per bin and then runs over 16-byte channel vectors.  Same arithmetic per element as above.  (Forward only:
// the backward is bound by its fp32 atomics, not by this arithmetic — two re-mappings measured slower, NOTES.md 10.8.)
template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_bins_kernel(const T* __restrict__ x, const float* __restrict__ rois, T* __restrict__ y,
                                                                 int64_t bins, float scale, int C, int H, int Wd, int PH, int PW,
                                                                 int sampling_ratio) {
    constexpr int VEC = Elt<T>::VEC;
    const int CV = C / VEC;
    for (int64_t bin = blockIdx.x; bin < bins; bin += gridDim.x) {
        const int pw = (int)(bin % PW), ph = (int)((bin / PW) % PH);
        const int64_t n = bin / PW / PH;
        const RoiGeom g = roi_geom(rois + n * 5, scale, PH, PW, sampling_ratio);
        for (int cv = threadIdx.x; cv < CV; cv += blockDim.x) {
            float out[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) out[k] = 0.f;
            const T* base = x + (int64_t)g.batch * H * Wd * C + cv * VEC;
            for (int iy = 0; iy < g.grid_h; ++iy) {
                const float yy = g.start_h + ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                for (int ix = 0; ix < g.grid_w; ++ix) {
                    const float xx = g.start_w + pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                    const Bilin b = bilin_prep(H, Wd, yy, xx);
                    if (b.empty) continue;
                    float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
                    Elt<T>::ldv(base + ((int64_t)b.y_low * Wd + b.x_low) * C, v1);
                    Elt<T>::ldv(base + ((int64_t)b.y_low * Wd + b.x_high) * C, v2);
                    Elt<T>::ldv(base + ((int64_t)b.y_high * Wd + b.x_low) * C, v3);
                    Elt<T>::ldv(base + ((int64_t)b.y_high * Wd + b.x_high) * C, v4);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) out[k] += b.w1 * v1[k] + b.w2 * v2[k] + b.w3 * v3[k] + b.w4 * v4[k];
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) out[k] = out[k] / g.count;
            Elt<T>::stv(y + bin * C + cv * VEC, out);
        }
    }
}

// Channels-last backward WITHOUT atomics: gather by map tile.  The scatter above issues one fp32 atomic per (sample corner,
// channel) — 8.6 per pooled element on the bench's proposals, 220 M per launch: 664 us at the chip's ~325 G atomics/s.  Two facts
// remove them.  (1) ROIAlign is separable: a sample's weight is wy(y) * wx(x), its validity (ROIAlign_cuda.cu:134-139) is
// valid_y && valid_x and the clamps (:141-163) act per axis, so the gradient one ROI leaves on the map is
//     dx[Y, X, c] = sum_ph sum_pw  Wy[Y][ph] * Wx[X][pw] * dy[ph, pw, c] / count,
//     Wy[Y][ph] = sum over the bin's iy samples of (hy if y_low == Y) + (ly if y_high == Y)            (Wx alike)
// — the same samples, weights and 1/count as :178-254, summed in a fixed order.  (2) A workgroup that owns a 4 x 4 cell x
// 16-vector tile of the map can visit the ROIs overlapping it, accumulate in registers and store once: deterministic, no
// memset, no atomics.  One wave = one map row of the tile (lane = column x channel vector); the waves run their ROI loops
// independently (no barrier after the culling): per ROI, lanes 0..PH-1 build the row's Wy, lanes (col, pw) the four columns'
// Wx, ballots give the non-zero bin ranges and shuffles hand the weights out.  A row meets at most 3 bins of an ROI whose bins
// are at least one cell tall (samples within (Y-1, Y+1)): that case is 9 unconditional, independent 16-byte loads; ROIs smaller
// than 14 cells (which meet few tiles) take the bin-by-bin loop.
constexpr int RG_TH = 4, RG_TW = 4, RG_NV = 16, RG_MAXP = 16, RG_THREADS = 256, RG_LIST = 1024;

struct AxisW { int lo, hi; float l, h; bool valid; };
// one axis of bilinear_interpolate_gradient (ROIAlign_cuda.cu:125-170)
__device__ __forceinline__ AxisW axis_prep(int size, float t) {
    AxisW a;
    a.valid = !(t < -1.0f || t > (float)size);
    if (t <= 0) t = 0;
    a.lo = (int)t;
    if (a.lo >= size - 1) { a.hi = a.lo = size - 1; t = (float)a.lo; } else a.hi = a.lo + 1;
    a.l = t - (float)a.lo;
    a.h = 1.f - a.l;
    return a;
}

// The per-(ROI, row, bin) and per-(ROI, column, bin) weights of the gather below, once per launch instead of once per wave
// (each (tile row, channel slab) wave would rebuild them: 8 slabs x 4 rows of redundant divisions — measured 4.3 us per ROI and
// wave on 800-pixel boxes, VALU-bound).  One thread per (roi, row) / (roi, column): wy[16] / wx[16] (zero where the bin does not
// reach the line) and the non-zero bin range packed lo | hi << 8 (hi < lo: none).
__global__ __launch_bounds__(256) void roi_bwd_tables_kernel(const float* __restrict__ rois, int R, float scale, int H, int Wd, int PH, int PW,
                                                             int sampling_ratio, float* __restrict__ Wy, float* __restrict__ Wx,
                                                             int* __restrict__ Ry, int* __restrict__ Rx) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)R * (H + Wd)) return;
    const int r = (int)(t / (H + Wd)), pos = (int)(t - (int64_t)r * (H + Wd));
    const RoiGeom g = roi_geom(rois + (int64_t)r * 5, scale, PH, PW, sampling_ratio);
    const bool isy = pos < H;
    const int line = isy ? pos : pos - H, size = isy ? H : Wd, P = isy ? PH : PW, grid = isy ? g.grid_h : g.grid_w;
    const float start = isy ? g.start_h : g.start_w, bin = isy ? g.bin_h : g.bin_w;
    float* out = isy ? Wy + ((int64_t)r * H + line) * RG_MAXP : Wx + ((int64_t)r * Wd + line) * RG_MAXP;
    int lo = 1, hi = 0;
    for (int p = 0; p < RG_MAXP; ++p) {
        float w = 0.f;
        if (p < P) {
            for (int i = 0; i < grid; ++i) {
                const AxisW a = axis_prep(size, start + p * bin + ((float)i + .5f) * bin / (float)grid);
                if (a.valid) w += (a.lo == line ? a.h : 0.f) + (a.hi == line ? a.l : 0.f);
            }
            if (isy) w = w / g.count;
        }
        out[p] = w;
        if (w != 0.f) { if (hi < lo) lo = p; hi = p; }
    }
    (isy ? Ry + (int64_t)r * H : Rx + (int64_t)r * Wd)[line] = lo | (hi << 8);
}

template <typename T, bool TAB>
__global__ __launch_bounds__(RG_THREADS) void roi_align_bwd_gather_kernel(const T* __restrict__ dy, const float* __restrict__ rois,
                                                                          float* __restrict__ dx, int R, float scale, int C, int H, int Wd,
                                                                          int PH, int PW, int sampling_ratio, int tiles_x, int tiles_y,
                                                                          int slabs, const float* __restrict__ Wy, const float* __restrict__ Wx,
                                                                          const int* __restrict__ Ry, const int* __restrict__ Rx) {
    constexpr int VEC = Elt<T>::VEC, CS = RG_NV * VEC;
    static_assert(RG_TH * 64 == RG_THREADS && RG_TW * RG_NV == 64 && RG_TW * RG_MAXP <= 64, "one wave per tile row");
    __shared__ int list[RG_LIST];
    __shared__ int wcnt[RG_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int slab = b % slabs; b /= slabs;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int y0 = ty * RG_TH, x0 = tx * RG_TW, c0 = slab * CS;
    const int Y = y0 + wave, col = lane >> 4, X = x0 + col, ch = c0 + (lane & 15) * VEC;
    const bool lane_ok = X < Wd && ch < C;
    const int ch_safe = ch < C ? ch : c0;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;

    for (int rbase = 0; rbase < R; rbase += RG_LIST) {
        // ---- which ROIs of this chunk touch the tile (conservative footprint: every sample lies in [start, start + extent]),
        //      compacted in ascending ROI index: a fixed summation order
        __syncthreads();
        int total = 0;
        for (int k = 0; k < RG_LIST / RG_THREADS && rbase + k * RG_THREADS < R; ++k) {
            const int r = rbase + k * RG_THREADS + tid;
            bool hit = false;
            if (r < R) {
                const float* roi = rois + (int64_t)r * 5;
                if ((int)roi[0] == n) {
                    const float sw = roi[1] * scale, sh = roi[2] * scale;
                    const float rw = fmaxf(roi[3] * scale - sw, 1.f), rh = fmaxf(roi[4] * scale - sh, 1.f);
                    const int ylo = sh < 0.f ? 0 : (int)floorf(sh), yhi = min(H - 1, (int)floorf(sh + rh) + 1);
                    const int xlo = sw < 0.f ? 0 : (int)floorf(sw), xhi = min(Wd - 1, (int)floorf(sw + rw) + 1);
                    hit = ylo <= y0 + RG_TH - 1 && yhi >= y0 && xlo <= x0 + RG_TW - 1 && xhi >= x0;
                }
            }
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wcnt[wave] = __popcll(m);
            __syncthreads();
            int off = total;
#pragma unroll
            for (int w = 0; w < RG_THREADS / 64; ++w) { if (w < wave) off += wcnt[w]; total += wcnt[w]; }
            if (hit) list[off + __popcll(m & ((1ULL << lane) - 1ULL))] = r;
            __syncthreads();
        }
        if (Y >= H) continue;                              // (wave-uniform; the wave still takes part in the next chunk's barriers)

        // TAB: the lane-held weights (lane = ph for the row, lane = column * 16 + pw for the columns) come from the tables, two
        // coalesced loads per ROI issued one ROI ahead; !TAB: the wave builds them
        const int bc = lane >> 4, bpw = lane & 15, Xc = x0 + bc;
        float wy_n = 0.f, wx_n = 0.f;
        auto fetch = [&](int i) {
            if (TAB && i < total) {
                const int r2 = list[i];
                wy_n = lane < RG_MAXP ? Wy[((int64_t)r2 * H + Y) * RG_MAXP + lane] : 0.f;
                wx_n = Xc < Wd ? Wx[((int64_t)r2 * Wd + Xc) * RG_MAXP + bpw] : 0.f;
            }
        };
        fetch(0);
        for (int i = 0; i < total; ++i) {
            const int rr = list[i];
            float wy = wy_n, wx = wx_n;
            if (TAB) {
                fetch(i + 1);
            } else {
                wy = 0.f; wx = 0.f;
                const RoiGeom g = roi_geom(rois + (int64_t)rr * 5, scale, PH, PW, sampling_ratio);
                // the row's weights per bin: lane = ph
                if (lane < PH) {
                    for (int iy = 0; iy < g.grid_h; ++iy) {
                        const AxisW a = axis_prep(H, g.start_h + lane * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h);
                        if (a.valid) wy += (a.lo == Y ? a.h : 0.f) + (a.hi == Y ? a.l : 0.f);
                    }
                    wy = wy / g.count;
                }
                // the four columns' weights per bin: lane = column * 16 + pw
                if (bpw < PW && Xc < Wd)
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        const AxisW a = axis_prep(Wd, g.start_w + bpw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w);
                        if (a.valid) wx += (a.lo == Xc ? a.h : 0.f) + (a.hi == Xc ? a.l : 0.f);
                    }
            }
            const unsigned long long my = __ballot(wy != 0.f);
            if (my == 0) continue;                         // the ROI does not reach this row
            const unsigned long long mxa = __ballot(wx != 0.f);
            if (mxa == 0) continue;
            const int plo = __ffsll((long long)my) - 1, phi = 63 - __clzll((long long)my);
            const unsigned mx = (unsigned)((mxa >> (col * 16)) & 0xffffULL);          // this lane's column
            const int qlo = mx ? __ffs((int)mx) - 1 : 0, qhi = mx ? 31 - __clz((int)mx) : -1;
            const T* base = dy + (int64_t)rr * PH * PW * C + ch_safe;
            const bool small = (phi - plo) < 3 && __ballot(qhi - qlo >= 3) == 0;
            if (small) {
                float wyi[3], wxj[3];
                int phs[3], pws[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    phs[a] = min(plo + a, phi);
                    const float w = __shfl(wy, phs[a], 64);
                    wyi[a] = (plo + a <= phi) ? w : 0.f;
                    pws[a] = qhi < 0 ? 0 : min(qlo + a, qhi);
                    const float u = __shfl(wx, col * 16 + pws[a], 64);
                    wxj[a] = (qlo + a <= qhi) ? u : 0.f;
                }
                float d[9][VEC];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) Elt<T>::ldv(base + (int64_t)(phs[a] * PW + pws[c]) * C, d[a * 3 + c]);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float w = wyi[a] * wxj[c];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) acc[q] = fmaf(w, d[a * 3 + c][q], acc[q]);
                    }
            } else {
                const int qmin = __ffsll((long long)(mxa | (mxa >> 16) | (mxa >> 32) | (mxa >> 48)) & 0xffff) - 1;
                const int qmax = 63 - __clzll((long long)((mxa | (mxa >> 16) | (mxa >> 32) | (mxa >> 48)) & 0xffffULL));
                for (int ph = plo; ph <= phi; ++ph) {
                    const float wyv = __shfl(wy, ph, 64);
                    for (int pw = qmin; pw <= qmax; ++pw) {
                        const float w = wyv * __shfl(wx, col * 16 + pw, 64);
                        if (w != 0.f) {
                            float d[VEC];
                            Elt<T>::ldv(base + (int64_t)(ph * PW + pw) * C, d);
#pragma unroll
                            for (int q = 0; q < VEC; ++q) acc[q] = fmaf(w, d[q], acc[q]);
                        }
                    }
                }
            }
        }
    }
    if (Y < H && lane_ok) {
        float* o = dx + (((int64_t)n * H + Y) * Wd + X) * C + ch;
#pragma unroll
        for (int q = 0; q < VEC; q += 4) {
            const f32x4 t = {acc[q], acc[q + 1], acc[q + 2], acc[q + 3]};
            *reinterpret_cast<f32x4*>(o + q) = t;
        }
    }
}

}  // namespace

// AFAN_ROI_BWD_GATHER=0: the scatter-with-atomics backward everywhere (A/B measurements, tools/probe/roi_bwd_time.py)
static bool afan_roi_bwd_gather_enabled() {
    static const int on = [] { const char* e = getenv("AFAN_ROI_BWD_GATHER"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

extern "C" {

// bytes of scratch afan_nms needs for n boxes: the overlap mask (n x ceil(n/64) 64-bit words) + one flag per box + the
// band tiles in transposed form (SC_D per row block)
int64_t afan_nms_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    const int64_t cb = (n + W64 - 1) / W64;
    return n * cb * 8 + ((n + 7) & ~(int64_t)7) + cb * (SC_D + 1) * W64 * 8;      // + the band tiles transposed
}

// boxes [n,4] fp32 (left, top, right, bottom; corners inclusive), order [n] int64 = indices by DESCENDING score;
// keep_out [n] int64: the kept boxes' original indices, ascending, count_out[0] of them valid.
static int nms_impl(const float* boxes, const int64_t* order, int64_t n, float threshold, int inclusive, void* workspace,
                    int64_t* keep_out, int64_t* count_out, int64_t max_keep, afan_stream_t stream);

int afan_nms(const float* boxes, const int64_t* order, int64_t n, float threshold, int inclusive, void* workspace,
             int64_t* keep_out, int64_t* count_out, afan_stream_t stream) {
    return nms_impl(boxes, order, n, threshold, inclusive, workspace, keep_out, count_out, 0, stream);
}

// The same with an upper bound on the survivors the caller will look at: the scan stops after the 64-box block in which the
// kept count reaches max_keep (count_out[0] may exceed max_keep by up to 63; the first max_keep entries of keep_out — in
// SCORE order they are the first max_keep survivors — are those of the unbounded call).  keep_out is ascending by ORIGINAL
// index, so a caller that slices must rank by score again or, like the proposal layer, pass boxes already sorted by score.
int afan_nms_top(const float* boxes, const int64_t* order, int64_t n, float threshold, int inclusive, void* workspace,
                 int64_t* keep_out, int64_t* count_out, int64_t max_keep, afan_stream_t stream) {
    if (max_keep < 0) return AFAN_ESHAPE;
    return nms_impl(boxes, order, n, threshold, inclusive, workspace, keep_out, count_out, max_keep, stream);
}

static int nms_impl(const float* boxes, const int64_t* order, int64_t n, float threshold, int inclusive, void* workspace,
                    int64_t* keep_out, int64_t* count_out, int64_t max_keep, afan_stream_t stream) {
    if (n < 0 || n > (1 << 24)) return AFAN_ESHAPE;
    if (!count_out) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return (int)hipMemsetAsync(count_out, 0, 8, st);
    if (!boxes || !order || !workspace || !keep_out) return AFAN_ENULL;
    if (!aligned(workspace, 8)) return AFAN_EALIGN;
    const int cb = (int)((n + W64 - 1) / W64);
    if ((size_t)cb * 8 > 64 * 1024) return AFAN_ESHAPE;          // the scan keeps one word per column block in LDS
    unsigned long long* mask = (unsigned long long*)workspace;
    uint8_t* flag = (uint8_t*)workspace + n * (int64_t)cb * 8;
    hipError_t e = hipMemsetAsync(flag, 0, (size_t)n, st);
    if (e != hipSuccess) return (int)e;
    AFAN_PROF("nms_kernel", 16.0 * n + 8.0 * n * cb, st);
    unsigned long long* bandT = (unsigned long long*)(flag + ((n + 7) & ~(int64_t)7));
    nms_mask_kernel<<<dim3(cb, cb), W64, 0, st>>>(boxes, order, (int)n, threshold, inclusive, mask, bandT, cb);
    AFAN_LAUNCH_CHECK();
    const int mk = (int)(max_keep > 0x7fffffffLL ? 0 : max_keep);
    if ((size_t)cb * 8 > 8 * 1024) {                               // the static ring + removed[] pass the default 64 KB of LDS
        static const hipError_t attr = hipFuncSetAttribute((const void*)nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (attr != hipSuccess) return (int)attr;
    }
    nms_scan_kernel<<<1, W64 * SCAN_WAVES, (size_t)cb * 8, st>>>(mask, bandT, order, (int)n, cb, flag, mk);
    AFAN_LAUNCH_CHECK();
    nms_compact_kernel<<<1, CP_THREADS, 0, st>>>(flag, (int)n, keep_out, count_out);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

static int roi_check(int dtype, int layout, int64_t num_rois, int64_t c, int64_t h, int64_t w, int ph, int pw) {
    if (dtype != AFAN_F32 && dtype != AFAN_BF16) return AFAN_EDTYPE;
    if (layout != AFAN_NCHW && layout != AFAN_NHWC) return AFAN_ELAYOUT;
    if (num_rois < 0 || c <= 0 || h <= 0 || w <= 0 || ph <= 0 || pw <= 0) return AFAN_ESHAPE;
    return AFAN_OK;
}

// y[num_rois, C, PH, PW] (in `layout`: NCHW, or [num_rois, PH, PW, C]) from x[N, C, H, W]; rois [num_rois, 5] fp32 =
// (batch index, x1, y1, x2, y2) in image coordinates; sampling_ratio <= 0: ceil(roi extent / pooled extent) samples per bin.
int afan_roi_align_fwd(const void* x, const float* rois, void* y, int dtype, int layout, int64_t num_rois, int64_t c, int64_t h,
                       int64_t w, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, afan_stream_t stream) {
    int e = roi_check(dtype, layout, num_rois, c, h, w, pooled_h, pooled_w);
    if (e) return e;
    if (num_rois == 0) return AFAN_OK;
    if (!x || !rois || !y) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = num_rois * c * pooled_h * pooled_w;
    const int grid = grid_for(total, 256, 8192);
    AFAN_PROF("roi_align_fwd_kernel", (double)total * (dtype == AFAN_F32 ? 4 : 2) * 5, st);
    const int vec = dtype == AFAN_F32 ? 4 : 8;
    if (layout == AFAN_NHWC && c % vec == 0 && c / vec >= 16 && aligned(x, 16) && aligned(y, 16)) {      // geometry once per bin
        const int64_t bins = num_rois * pooled_h * pooled_w;
        const int cv = (int)(c / vec), threads = cv >= 256 ? 256 : ((cv + 63) / 64) * 64;
        const unsigned g2 = (unsigned)(bins < (1 << 20) ? bins : (1 << 20));
        if (dtype == AFAN_F32) roi_align_fwd_bins_kernel<float><<<g2, threads, 0, st>>>((const float*)x, rois, (float*)y, bins, spatial_scale, (int)c, (int)h, (int)w, pooled_h, pooled_w, sampling_ratio);
        else roi_align_fwd_bins_kernel<uint16_t><<<g2, threads, 0, st>>>((const uint16_t*)x, rois, (uint16_t*)y, bins, spatial_scale, (int)c, (int)h, (int)w, pooled_h, pooled_w, sampling_ratio);
        AFAN_LAUNCH_CHECK();
        return AFAN_OK;
    }
#define RA_(T, L) roi_align_fwd_kernel<T, L><<<grid, 256, 0, st>>>((const T*)x, rois, (T*)y, total, spatial_scale, (int)c, (int)h, (int)w, pooled_h, pooled_w, sampling_ratio)
    if (dtype == AFAN_F32) { if (layout == AFAN_NHWC) RA_(float, true); else RA_(float, false); }
    else { if (layout == AFAN_NHWC) RA_(uint16_t, true); else RA_(uint16_t, false); }
#undef RA_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

// dx[N, C, H, W] fp32 (zeroed here, in `layout`) += scatter of dy[num_rois, C, PH, PW] (`dtype`, same layout convention)
// bytes of scratch afan_roi_align_bwd_ws takes for the per-(roi, row) / (roi, column) weight tables (0: shape outside the gather form)
int64_t afan_roi_align_bwd_workspace_bytes(int64_t num_rois, int64_t h, int64_t w) {
    if (num_rois <= 0 || h <= 0 || w <= 0) return 0;
    return num_rois * (h + w) * (RG_MAXP * 4 + 4);
}

int afan_roi_align_bwd(const void* dy, const float* rois, float* dx, int dtype, int layout, int64_t num_rois, int64_t n, int64_t c,
                       int64_t h, int64_t w, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                       afan_stream_t stream) {
    return afan_roi_align_bwd_ws(dy, rois, dx, dtype, layout, num_rois, n, c, h, w, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                 nullptr, stream);
}

// The same with caller-owned scratch (afan_roi_align_bwd_workspace_bytes, 16-byte aligned; NULL: the waves build the weights
// themselves): the channels-last gather reads its bilinear weights from tables one small launch fills.
int afan_roi_align_bwd_ws(const void* dy, const float* rois, float* dx, int dtype, int layout, int64_t num_rois, int64_t n, int64_t c,
                          int64_t h, int64_t w, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, void* workspace,
                          afan_stream_t stream) {
    int e = roi_check(dtype, layout, num_rois, c, h, w, pooled_h, pooled_w);
    if (e) return e;
    if (n <= 0) return AFAN_ESHAPE;
    if (!dx) return AFAN_ENULL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = num_rois * c * pooled_h * pooled_w;
    const int vec = dtype == AFAN_F32 ? 4 : 8;
    // channels-last with 16-byte channel vectors: the atomic-free gather by map tile (deterministic; writes every cell itself)
    if (layout == AFAN_NHWC && num_rois > 0 && c % vec == 0 && pooled_h <= RG_MAXP && pooled_w <= RG_MAXP && aligned(dx, 16) &&
        dy && aligned(dy, 16) && num_rois < (1 << 24) && afan_roi_bwd_gather_enabled()) {
        if (!rois) return AFAN_ENULL;
        const int cs = RG_NV * vec, tiles_x = (int)((w + RG_TW - 1) / RG_TW), tiles_y = (int)((h + RG_TH - 1) / RG_TH), slabs = (int)((c + cs - 1) / cs);
        const int64_t blocks = n * tiles_y * tiles_x * slabs;
        if (blocks < (1LL << 31)) {
            AFAN_PROF("roi_align_bwd_kernel", (double)total * (dtype == AFAN_F32 ? 4 : 2) + 4.0 * n * c * h * w, st);
            float *Wy = nullptr, *Wx = nullptr;
            int *Ry = nullptr, *Rx = nullptr;
            if (workspace) {
                if (!aligned(workspace, 16)) return AFAN_EALIGN;
                Wy = (float*)workspace; Wx = Wy + num_rois * h * RG_MAXP;
                Ry = (int*)(Wx + num_rois * w * RG_MAXP); Rx = Ry + num_rois * h;
                const int64_t tt = num_rois * (h + w);
                roi_bwd_tables_kernel<<<(unsigned)((tt + 255) / 256), 256, 0, st>>>(rois, (int)num_rois, spatial_scale, (int)h, (int)w, pooled_h, pooled_w, sampling_ratio, Wy, Wx, Ry, Rx);
                AFAN_LAUNCH_CHECK();
            }
#define RGK_(T, TAB) roi_align_bwd_gather_kernel<T, TAB><<<(unsigned)blocks, RG_THREADS, 0, st>>>((const T*)dy, rois, dx, (int)num_rois, spatial_scale, (int)c, (int)h, (int)w, pooled_h, pooled_w, sampling_ratio, tiles_x, tiles_y, slabs, Wy, Wx, Ry, Rx)
            if (dtype == AFAN_F32) { if (workspace) RGK_(float, true); else RGK_(float, false); }
            else { if (workspace) RGK_(uint16_t, true); else RGK_(uint16_t, false); }
#undef RGK_
            AFAN_LAUNCH_CHECK();
            return AFAN_OK;
        }
    }
    hipError_t he = hipMemsetAsync(dx, 0, (size_t)(n * c * h * w) * 4, st);
    if (he != hipSuccess) return (int)he;
    if (num_rois == 0) return AFAN_OK;
    if (!dy || !rois) return AFAN_ENULL;
    const int grid = grid_for(total, 256, 8192);
    AFAN_PROF("roi_align_bwd_kernel", (double)total * ((dtype == AFAN_F32 ? 4 : 2) + 16.0) + 4.0 * n * c * h * w, st);
#define RB_(T, L) roi_align_bwd_kernel<T, L><<<grid, 256, 0, st>>>((const T*)dy, rois, dx, total, spatial_scale, (int)c, (int)h, (int)w, pooled_h, pooled_w, sampling_ratio)
    if (dtype == AFAN_F32) { if (layout == AFAN_NHWC) RB_(float, true); else RB_(float, false); }
    else { if (layout == AFAN_NHWC) RB_(uint16_t, true); else RB_(uint16_t, false); }
#undef RB_
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
