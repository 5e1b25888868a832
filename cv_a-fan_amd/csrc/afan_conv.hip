// Implicit-GEMM convolution for the bf16 channels-last backbone on gfx950 (MFMA 32x32x16 bf16, fp32 accumulate).
// The A-FAN step is >95 % convolution FLOPs — 4H + (2K+6)T forward-equivalents per image — and the
// PGD inner loop (Classification/attack_algo.py:50-52) runs the tail forward and its input-gradient K times per
// iteration, so forward and dgrad are the two kernels that matter; both are THIS kernel with different tap lists.
//
// GEMM view:  Y[m, co] = sum_t sum_c  X[pixel(m) + (dh_t, dw_t), c] * W_t[co, c]
//   m  -> (n, h', w') over an N x Hg x Wg grid of output positions; input pixel = (h'*in_s + dh_t, w'*in_s + dw_t)
//         (zero outside the image), output pixel = (h'*out_s + out_h0, w'*out_s + out_w0)
//   W_t -> tap t of a [rows][taps][c] weight tensor whose innermost dimension is the reduction channel
// which covers: forward 3x3/1x1 at stride 1 or 2 (KRSC weights), dgrad at stride 1 (CRSK = transposed weights,
// mirrored tap offsets), dgrad at stride 2 as four output-parity classes with 1/2/2/4 taps each (no multiplications
// by the zeros a "dilated" formulation would insert).
//
// Tile: BM x BN outputs per workgroup of 8 waves (2x4 or 4x2; 2x2 with 4 waves as a tuning variant), BK = 64 channels
// of one tap per step, 32x32 MFMA tiles.  Default staging is LDS-DMA (global_load_lds: no VGPR round trip; unpadded
// 128-byte rows XOR-swizzled for conflict-free ds_read_b128; a zero page feeds the padding rows), with 2 LDS stages and
// __syncthreads() where two workgroups share a CU, and 4 stages with counted s_waitcnt vmcnt(N) + bare s_barrier where
// the launch is about one workgroup per CU (see dispatch()).  Register-staged variants (PF 1 / 2: 144-byte padded rows)
// remain as A/B references (AFAN_CONV_MODE).  The epilogue rounds to bf16 through LDS so that every store is a 16-byte
// piece of a channels-last row, and carries the optional fusions (addend, BatchNorm sums per image group).
// Knobs (tuning / A-B only): AFAN_CONV_MODE, AFAN_CONV_BM, AFAN_CONV_NW, AFAN_CONV_DEEP, AFAN_CONV_TALL, AFAN_CONV_C64,
// AFAN_CONV_HALO.
// (Measured-and-lost variants — 256-row tiles, producer waves on the two-stage launches, a half-step software pipeline,
// streaming stores, register-streamed weights — are described in NOTES.md 9.5 and no longer compiled in.)
#include "afan_common.h"
#include "afan_conv_c64.h"
#include "afan_conv_stem.h"
#include "afan_conv_params.h"
#include <stdlib.h>

#ifndef AFAN_CONV_DRES_EARLY
#define AFAN_CONV_DRES_EARLY 1   // in-launch BatchNorm backward: the masked gradient leaves between the barrier's arrival and its wait (0: behind it, A/B)
#endif
#ifndef AFAN_CONV_FRAG_BATCH
#define AFAN_CONV_FRAG_BATCH 4   // k16-slices of operand fragments in flight before their MFMAs (1: the compiler's order)
#endif

using namespace afan;
using namespace afan_conv;

namespace afan_nhwc {   // afan_bn_nhwc.hip: accumulator copies per channel / doubles per accumulator block
int acc_slot_count(int64_t C);
int64_t acc_doubles(int64_t C);
int set_running_updates(int n);
}

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TRANSPOSE_THREADS = 256;
constexpr int BK = 64;           // reduction channels per step
constexpr int LDK = BK + 8;      // padded LDS row (elements): 144 bytes (register-staged variants)

// landing zone for the padding rows of the LDS-DMA variant (a DMA lane has to read SOMETHING: zeros)
// (|tap displacement| <= ConvP::max_pad pixels in both directions: 1 for 3x3 / 1x1, the dilation for atrous 3x3)

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }

#ifdef AFAN_CONV_STAMP
// diagnostic build (tools/build_stamp.sh): per-tap cycle stamps of workgroup (0,0,0)'s first MFMA wave and first producer wave in
// the halo form, parked in LDS during the loop (a global store would join the producers' counted vmcnt) and written out once
__device__ unsigned long long afan_stamps[2][96][3];
#define AFAN_STAMP(role, idx, k) do { if ((role) == 0 && stamp_on && (idx) < 72) st_lds[idx][k] = __builtin_readcyclecounter(); } while (0)   /* (72 taps fit the stamped build's LDS) */
// phases of the launch as thread 0 of workgroup (0,0,0) passes them: entry | K loop starts | K loop done | output tile in LDS | first
// epilogue pass done | sums added | (in-launch BatchNorm: barrier passed | second pass done) | exit  -> afan_stamps[1][p][0]
// ([p][1]: the same for the LAST workgroup of the grid — in a launch of several rounds per CU one that starts with the kernel's code and
// arguments already in the caches, which workgroup (0,0,0) never does)
#define AFAN_PHASE(p) do { if (threadIdx.x == 0 && blockIdx.z == 0) { \
        if (blockIdx.x == 0 && blockIdx.y == 0) afan_stamps[1][p][0] = __builtin_readcyclecounter(); \
        if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) afan_stamps[1][p][1] = __builtin_readcyclecounter(); } } while (0)
#else
#define AFAN_STAMP(role, idx, k) do { } while (0)
#define AFAN_PHASE(p) do { } while (0)
#endif

// ---- grid-wide barrier for the in-launch BatchNorm (ConvP::bnf): every workgroup of the launch is resident (dispatch_bnf checks),
// what the barrier orders — the f64 column sums — travels through memory-side atomics that each workgroup has drained
// (s_waitcnt vmcnt(0)) before it arrives, so RELAXED agent-scope atomics suffice: no L2 write-back / invalidate (release +
// acquire at agent scope cost 8-24 us here, tools/probe/grid_barrier.hip; this form 2-3 us, tools/probe/grid_barrier2.hip).
// Sense-reversing, sharded over 8 arrival counters (workgroup id & 7), self-resetting: reusable by the next launch without a
// memset node.  The spin is BOUNDED (0.2 s of the 100 MHz clock): a launch whose workgroups are not all resident (another
// process's kernels holding CUs) flags bar[ERR] and goes on with wrong totals instead of hanging the GPU; ops.py checks the flag.
struct GridBar {                       // words 64 bytes apart
    unsigned shard_cnt[8][16];
    unsigned global_cnt[16];
    unsigned flag[8][16];
    unsigned err[16];
};
static_assert(sizeof(GridBar) <= 2048, "afan_grid_barrier_bytes()");
// arrive / wait are separate so that a workgroup can store its raw tile while the others still arrive.  Per-shard arrival counters
// (reset by the shard's last arriver, fire and forget: the next episode is a later launch, ordered behind this one by the stream)
// and ONE monotonic release counter that every shard's last arriver bumps: a workgroup's episode number is that counter / 8 read
// BEFORE its own arrival (its shard cannot have completed yet), it leaves when the counter reaches 8 * (episode + 1).  The last
// arriver's path is two dependent atomics, a waiter's one poll round trip behind them.
__device__ __forceinline__ unsigned grid_episode(unsigned* words) {       // any time before the workgroup's own arrival (early: off the critical path)
    GridBar* b = reinterpret_cast<GridBar*>(words);
    return (__hip_atomic_load(&b->global_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / 8u + 1u) * 8u;
}
__device__ __forceinline__ void grid_arrive(unsigned* words, unsigned id, unsigned nwg) {
    GridBar* b = reinterpret_cast<GridBar*>(words);
    const unsigned sh = id & 7u;
    const unsigned per = nwg / 8u + (sh < (nwg & 7u) ? 1u : 0u);            // (nwg >= 8: launch_gs checks)
    const unsigned a = __hip_atomic_fetch_add(&b->shard_cnt[sh][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a + 1 == per) {
        __hip_atomic_store(&b->shard_cnt[sh][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&b->global_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void grid_wait(unsigned* words, unsigned target) {
    GridBar* b = reinterpret_cast<GridBar*>(words);
    if (__hip_atomic_load(&b->err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // a spin already gave up: do not stack 0.2 s waits
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(&b->global_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > 20000000LL) {
            __hip_atomic_store(&b->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
// a memory-side total (the accumulators are only ever touched by device-scope atomics): a load that passes this XCD's L2 (sc1)
__device__ __forceinline__ double ld_total(const __amdgpu_buffer_rsrc_t& r, int byte_off) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(double, (u32x2)__builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16));
}

// halo form, producer side: the next chunk's KG halo groups per producer wave ride with the first seven taps; BR weight DMAs per tap
template <int KG, int BR> struct HaloSched {
    static constexpr int s(int t) { return (t < 0 || t >= 7) ? 0 : KG / 7 + (t < KG % 7 ? 1 : 0); }          // groups behind tap t
    static constexpr int g0(int t) { int g = 0; for (int i = 0; i < t; ++i) g += s(i); return g; }            // first group of tap t
    // instructions of the issue slot behind tap tp's barrier (tp < 0: a slot of the chunk before, or of the prologue: weights only)
    static constexpr int slot(int tp, bool more) { return tp < 0 ? BR : ((tp + 3 < 9 || more) ? BR : 0) + (more ? s(tp) : 0); }
    static constexpr int wait(int t, bool more) { return slot(t - 1, more) + slot(t - 2, more); }             // may stay in flight at tap t's wait
};

// halo form, FIVE tiles ahead (PF = 7: six weight buffers; the MFMA waves request a tap's fragments during the tap before): the next
// chunk's KG halo groups per producer wave ride with the first FIVE taps (all of it must be in LDS at tap 8's barrier: tap 8's MFMA
// waves already read the next chunk's first tap), BR weight DMAs per tap
template <int KG, int BR> struct HaloSched5 {
    static constexpr int s(int t) { return (t < 0 || t >= 5) ? 0 : KG / 5 + (t < KG % 5 ? 1 : 0); }
    static constexpr int g0(int t) { int g = 0; for (int i = 0; i < t; ++i) g += s(i); return g; }
    // instructions of the issue slot behind tap tp's barrier (tp < 0: a slot of the chunk before — its halo part, if any, came with its
    // own first five taps — or of the prologue: weights only)
    static constexpr int slot(int tp, bool more) { return tp < 0 ? BR : ((tp + 5 < 9 || more) ? BR : 0) + (more ? s(tp) : 0); }
    // may stay in flight at tap t's wait: what was issued behind tile t + 1's own DMAs (slot t - 4): the three slots since
    static constexpr int wait(int t, bool more) { return slot(t - 1, more) + slot(t - 2, more) + slot(t - 3, more); }
};

// PF: 1/2 = register-staged operands (1 or 2 register sets), 3 = LDS-DMA.  NW: waves per workgroup (4 = 2x2, 8 = 2x4):
// the tile is the same, 8 waves halve the per-wave work so twice as many waves per SIMD cover each other's waits.
// PW > 0: PW extra PRODUCER waves issue every operand DMA and the WM x WN waves only read LDS and issue MFMAs.  An LDS-DMA
// instruction holds its wave's issue slot for 60-185 cycles (tools/conv_ablate_bench.py: MFMAs on constant fragments
// 12.0 us, the same plus the DMA 19.8 us, DMA alone 13.8 us on the 256-channel 8x8 layer — the two serialise inside a
// wave); in a producer wave that stall costs no MFMA slot.
// HL > 0 (capacity in pixels): the activation HALO of the row tile stays in LDS across the nine taps of a 3x3 / stride 1
// problem — per 64-channel chunk the producers fetch the tile's pixels plus their one-pixel border ONCE (a contiguous range
// of the zero-padded raster, HL pixels at most) and only weight tiles stream per K-step: 9 x 16 KB + ~25 KB of L2 requests
// per chunk instead of 9 x 32 KB.  The K loop runs chunk-outer / tap-inner; an MFMA wave reads its fragment rows at the
// tap's displacement inside the halo.  Swizzle: pixel j keeps 16-byte piece q at q ^ ((v >> 1) & 7) with v = the pixel's
// linear position in the UNPADDED raster (n*H*W + y*W + x, also for border pixels): 16 consecutive output pixels at any
// tap are 16 consecutive v, and j - v is even everywhere, so (j & 1, (v >> 1) & 7) takes 16 distinct values — conflict-free
// ds_read_b128 for image rows of 4, 8, 16 or 33 pixels alike (round 1's halo kernel lost to exactly those conflicts).
// GS: the instantiation for launches whose image groups (two half-batches with separate BatchNorm sums) are NOT whole row
// tiles: the one tile holding rows of both halves runs the epilogue once per group.  A separate instantiation because the
// pass loop costs the 768- and 1 024-thread variants their last free registers (scratch in the K loop: ResNet-18's step
// 8.8 -> 11.6 ms when every launch carried it); launch<> picks it only for such launches.
// BF: the instantiation with the in-launch BatchNorm behind a grid barrier (ConvP::bnf; afan_conv_bnf.hip).
template <int BM, int BN, int PF, int WM, int WN, int PW = 0, int FBT = AFAN_CONV_FRAG_BATCH, int HL = 0, bool GS = false, bool BF = false>   // WM x WN waves: pixels x channels
__device__ __forceinline__ void conv_igemm_body(const ConvP& pp) {
    static_assert(!(BF && GS), "in-launch BatchNorm: one image group");
    AFAN_PHASE(0);
    constexpr int NW = WM * WN;
    constexpr int THREADS = 64 * (NW + PW);
    constexpr int STG = PW ? 64 * PW : THREADS;       // threads that stage operands
    constexpr int RPP = STG / 8;                      // rows staged per pass of the workgroup
    const ConvClass& cc = pp.cls[blockIdx.z];
    const uint32_t Wg = (uint32_t)cc.Wg, Hg = (uint32_t)cc.Hg;
    const uint32_t M = (uint32_t)pp.N * Hg * Wg;
    const uint32_t m0 = blockIdx.y * BM;
    if (m0 >= M) {         // smaller class than the grid's tallest: nothing to compute, but its partial slots must read 0
        if (pp.stats && threadIdx.x < BN && (int)(blockIdx.x * BN + threadIdx.x) < pp.Co) {
            const int64_t G = (int64_t)gridDim.y * gridDim.z, slot = (int64_t)blockIdx.z * gridDim.y + blockIdx.y;
            const int c = blockIdx.x * BN + threadIdx.x;
            pp.stats[((int64_t)0 * pp.Co + c) * G + slot] = 0.f;
            pp.stats[((int64_t)1 * pp.Co + c) * G + slot] = 0.f;
        }
        return;
    }
    constexpr int TM = BM / WM, TN = BN / WN;        // wave tile (pixels x channels)
    constexpr int MI = TM / 32, NI = TN / 32;        // 32x32 MFMA tiles per wave
    static_assert(MI >= 1 && NI >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows staged per pass");
    constexpr int A_ROWS = BM / RPP, B_ROWS = BN / RPP;  // 16-byte pieces per thread per step (rows t/8 + RPP*i)
    constexpr bool GLDS = (PF >= 3);                 // operands go global -> LDS by DMA (no VGPR staging, no ds_write)
    constexpr int NS = PF <= 3 ? 2 : PF - 1;         // LDS stages: PF 3 -> 2, PF 4 -> 3, PF 5 -> 4 (tiles in flight: NS - 1)
    constexpr int LDR = GLDS ? BK : LDK;             // LDS row length: DMA rows are unpadded 128 B, XOR-swizzled instead
    constexpr int STAGE = HL ? BN * LDR : (BM + BN) * LDR;   // elements per buffer (halo form: weight tiles only)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    __shared__ int out_off[BM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = PW ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const bool producer = PW && wave >= NW;           // wave-uniform
    const int stid = PW ? tid - 64 * NW : tid, swave = PW ? wave - NW : wave;   // index among the staging threads / waves
    const int n0 = blockIdx.x * BN;
    const int T = cc.T, Ci = pp.Ci, Hi = pp.Hi, Wi = pp.Wi;
    // Kernel arguments come through scalar loads of ~100 cycles each, and a load the compiler finds behind a condition is issued
    // THERE, waited for, and the next condition's only after it: the tap-validity loop (`t < T && hi >= 0 && ...`, two loads per tap
    // and row) and the coefficient block's pointer tests made 57 such load -> wait pairs of this prologue — 5 600 of a 1x1 launch's
    // 14 800 cycles per workgroup (launch-phase stamps, profiles/r05h_conv_launch_phases.txt).  Everything the prologue tests is
    // therefore read HERE, unconditionally and side by side (adjacent fields merge into s_load_dwordx4/x8/x16), and the conditions
    // below are written without short circuits.
    int tdh[MAX_TAPS], tdw[MAX_TAPS];
#pragma unroll
    for (int t = 0; t < MAX_TAPS; ++t) {
        tdh[t] = HL ? 0 : cc.dh[t];
        tdw[t] = HL ? 0 : cc.dw[t];
    }
    const int tap0_w = cc.wofs[0], tap0_a = cc.aofs[0], tap0_h = cc.dh[0], tap0_d = cc.dw[0];

    // Buffer descriptors: out-of-range offsets read as zero in hardware, so the zero padding of the convolution (and
    // rows beyond M) costs one v_cndmask per load instead of a branch around it.  All offsets are 32-bit bytes.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(pp.x), 0, (int)((int64_t)pp.N * Hi * Wi * Ci * 2 + pp.a_extra), 0x00020000);
    const int w_row_stride = pp.multi ? pp.w_rs[blockIdx.z] : pp.w_row_stride;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(pp.w + pp.w_off[blockIdx.z]), 0, (int)((int64_t)pp.Co * w_row_stride * 2), 0x00020000);
    constexpr uint32_t OOB = 0x80000000u;

    // ---- per-thread gather bookkeeping: A_ROWS rows, one 16-byte channel piece each (32-bit index math) -------
    // LDS-DMA: the destination of a wave instruction is 1 KiB of consecutive LDS (8 unpadded rows), so the bank-conflict
    // fix is a swizzle instead of padding: row r keeps its 16-byte piece q at position q ^ ((r >> 1) & 7).  Seen from
    // the source side, thread t (linear LDS position: row t/8 + 32 i, slot t & 7) fetches logical piece
    // (t & 7) ^ ((t >> 4) & 7) — the same for all its rows.
    const int piece = GLDS ? ((stid & 7) ^ ((stid >> 4) & 7)) : (stid & 7);
    const int row0 = stid >> 3;
    uint32_t a_off[A_ROWS];   // byte offset of (n, hi0, wi0, piece*8)
    uint32_t a_valid[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const uint32_t m = m0 + row0 + RPP * i;
        a_off[i] = 0;
        a_valid[i] = 0;
        if (!HL && m < M && (!PW || producer)) {
            const uint32_t t1 = m / Wg, wg = m - t1 * Wg;
            const uint32_t n = t1 / Hg, hg = t1 - n * Hg;
            const int hi0 = (int)hg * pp.in_s, wi0 = (int)wg * pp.in_s;
            a_off[i] = (((n * Hi + hi0) * Wi + wi0) * Ci + piece * 8) * 2u;
            uint32_t v = 0;
#pragma unroll
            for (int t = 0; t < MAX_TAPS; ++t)      // (unsigned compare: 0 <= hi < Hi in one)
                if ((t < T) & ((uint32_t)(hi0 + tdh[t]) < (uint32_t)Hi) & ((uint32_t)(wi0 + tdw[t]) < (uint32_t)Wi)) v |= 1u << t;
            a_valid[i] = v;
        }
    }
    // The epilogue's per-channel coefficients (a BatchNorm's shift / mean, alpha, beta; a frozen BatchNorm's alpha, beta) are fetched
    // HERE, into LDS: asked for at the top of the epilogue their ~1 us of load latency stood in front of its first pass in every
    // launch that carries one of the fusions (launch-phase stamps, profiles/r05g_conv_launch_phases.txt: a 64-row 1x1 launch's first
    // pass 3.5 k -> 6.9 k ticks).  (The one tile that straddles two image groups reads the second group's directly.)
    // Which rows, and whether at all, the launcher has worked out (fill_coef): three unconditional loads that leave together
    // (chosen here from the arguments' pointers they were three branches, each with its own load and its own wait).
    __shared__ float coef_s[3][BN];
    float cv0 = 0.f, cv1 = 0.f, cv2 = 0.f;
    if (tid < BN) {
        const int c = n0 + tid < pp.Co ? n0 + tid : 0;
        const uint32_t half_ = (uint32_t)(pp.N / 2) * Hg * Wg;
        const int64_t o_ = c + (m0 >= half_ ? pp.coef_gofs : 0);
        const int64_t o0_ = o_ + ((pp.coef_mask & 8) ? pp.shift_off[blockIdx.z] : 0);
        cv0 = pp.coef[0][o0_];
        cv1 = pp.coef[1][o_];
        cv2 = pp.coef[2][o_];
    }
    // What only the EPILOGUE needs of the prologue — the output row offsets (-1 = row outside M) and the coefficients' way into
    // LDS — runs behind the first operand requests: in the variants with producer waves on the MFMA waves while the producers
    // compute their gather offsets and issue, in the others between the first tile's DMAs and the wait for them.
    auto late_prologue = [&]() {
        for (int r = tid; r < BM; r += THREADS) {
            const uint32_t m = m0 + r;
            int off = -1;
            if (m < M) {
                const uint32_t t1 = m / Wg, wg = m - t1 * Wg;
                const uint32_t n = t1 / Hg, hg = t1 - n * Hg;
                off = (int)(((n * pp.Ho + hg * pp.out_s + cc.out_h0) * pp.Wo + wg * pp.out_s + cc.out_w0) * pp.Co);
            }
            out_off[r] = off;
        }
        if (tid < BN) {
            const int mask_ = pp.coef_mask;
            coef_s[0][tid] = (mask_ & 1) ? cv0 : 0.f;
            coef_s[1][tid] = (mask_ & 2) ? cv1 : 0.f;
            coef_s[2][tid] = (mask_ & 4) ? cv2 : 0.f;
        }
    };
    const uint32_t b_off = ((uint32_t)(n0 + row0) * w_row_stride + piece * 8) * 2u;
    const int b_rows_ok = pp.Co - n0 - row0;                      // weight row (row0 + RPP * i) exists iff RPP * i < b_rows_ok
    const uint32_t b_row32 = (uint32_t)RPP * w_row_stride * 2u;

    // accumulators: acc[j][i] = channels tile j x pixels tile i (weights are the MFMA A operand, so a lane ends up
    // with 4 consecutive CHANNELS of one pixel per register quad: 8-byte packed bf16 on the way out)
    f32x16 acc[NI][MI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    // Ragged channel counts (multiples of 8, e.g. DeepLab's 304-channel decoder input and 48-channel projection): the last
    // 64-channel chunk of a tap is partly beyond Ci — those 16-byte pieces are requested at an out-of-range offset (zeros
    // from the hardware range check) for BOTH operands, and weight rows / output channels >= Co are masked the same way.
    const int chunks = (Ci + BK - 1) / BK;
    const int KS = T * chunks;
    const bool last_ok = (chunks - 1) * BK + piece * 8 < Ci;      // this thread's piece exists in the last chunk
    u32x4 ra0[A_ROWS], rb0[B_ROWS], ra1[A_ROWS], rb1[B_ROWS];

    auto gload = [&](int ks, u32x4 (&ra)[A_ROWS], u32x4 (&rb)[B_ROWS]) {
        const int q = ks / T, t = ks - q * T;            // chunk-outer, tap-inner: the one K order of every tiled variant
        const uint32_t a_tap = (uint32_t)(((cc.dh[t] * Wi + cc.dw[t]) * Ci + q * BK + cc.aofs[t]) * 2);   // may be "negative": wraps
        const uint32_t b_tap = (uint32_t)((cc.wofs[t] + q * BK) * 2);
        const bool pk = q + 1 < chunks || last_ok;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const uint32_t off = (((a_valid[i] >> t) & 1u) && pk) ? a_off[i] + a_tap : OOB;
            ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i)
            rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (pk && RPP * i < b_rows_ok) ? (int)(b_off + i * b_row32) : (int)OOB, (int)b_tap, 0));
    };
    auto lstore = [&](int buf, const u32x4 (&ra)[A_ROWS], const u32x4 (&rb)[B_ROWS]) {
        uint16_t* A = lds + buf * STAGE;
        uint16_t* B = A + BM * LDK;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            *reinterpret_cast<u32x4*>(A + (row0 + RPP * i) * LDK + piece * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i)
            *reinterpret_cast<u32x4*>(B + (row0 + RPP * i) * LDK + piece * 8) = rb[i];
    };
    auto compute = [&](int buf) {
        const uint16_t* A = lds + buf * STAGE;
        const uint16_t* B = A + BM * LDR;
        const int frow = lane & 31, fk = (lane >> 5) * 8;
        const int sw = (frow >> 1) & 7;     // GLDS swizzle of this lane's row (tile bases are multiples of 32 rows)
        // Fragments of FB k16-slices are requested before the first MFMA that uses them (the compiler's own order reuses
        // four fragment registers and waits for every read: one exposed LDS latency per MFMA pair — measured 1 150 cycles
        // per K-step with NO operand traffic at all, against 512 of MFMA).
        constexpr int FB = FBT;
#pragma unroll
        for (int k0 = 0; k0 < BK / 16; k0 += FB) {
            bf16x8 fx[FB][MI], fw[FB][NI];
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int kk = k0 + b;
                const int koff = GLDS ? (((kk * 2 + (lane >> 5)) ^ sw) * 8) : (kk * 16 + fk);
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fx[b][i] = *reinterpret_cast<const bf16x8*>(A + (wr * TM + i * 32 + frow) * LDR + koff);
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    fw[b][j] = *reinterpret_cast<const bf16x8*>(B + (wc * TN + j * 32 + frow) * LDR + koff);
            }
            if (FB > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < FB; ++b)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b][j], fx[b][i], acc[j][i], 0, 0, 0);
            if (FB > 1) __builtin_amdgcn_sched_barrier(0);
        }
    };


    // LDS-DMA issue of one K-step into buffer `buf`: A_ROWS + B_ROWS wave instructions, each 64 lanes x 16 B = 1 KiB of
    // consecutive LDS; padding rows read the zero page
    // Buffer form of the DMA: per-thread byte offsets (a_off / b_voff, constant over the K loop) in the VGPR operand, the
    // K-step's tap and channel-chunk displacement in the SCALAR offset — no per-lane 64-bit address arithmetic per load
    // (the flat form cost ~8 VALU instructions per load: measured 78 VALU against 8 MFMA per wave and K-step).  The
    // activation descriptor is based A_BIAS bytes below the tensor to keep the scalar offset non-negative; gfx950 range-checks
    // VGPR + scalar offset against num_records (measured: with num_records = tensor bytes the last rows of the tensor read
    // as zero), so num_records = tensor bytes + A_BIAS; an invalid row is VGPR offset 2^31 (LDS gets zeros).
    const uint32_t A_BIAS = (uint32_t)(((pp.max_pad * Wi + pp.max_pad) * Ci) * 2);
    const __amdgpu_buffer_rsrc_t xdma = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(const_cast<uint16_t*>(pp.x)) - A_BIAS, 0, (int)((int64_t)pp.N * Hi * Wi * Ci * 2 + A_BIAS + pp.a_extra), 0x00020000);
    uint32_t b_voff[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_voff[i] = RPP * i < b_rows_ok ? b_off + i * b_row32 : OOB;
    // K-steps are issued strictly in order 0, 1, 2, ... = (chunk 0: taps 0..T-1), (chunk 1: taps 0..T-1), ...: the order of
    // the halo form, so that every tiled variant adds an output's products in the same sequence (a half-batch launched
    // alone may take another variant than the concatenated batch; the two must still agree bit for bit).  The (tap,
    // chunk) position is carried from call to call (no integer division per step); the next tap's table entries are
    // requested at the end of a call and turned into the two scalar offsets at the start of the next, a K-step later.
    int dma_t = 0, dma_q = 0;
    int tap_a = T > 0 ? (tap0_h * Wi + tap0_d) * Ci + tap0_a : 0;
    int tap_b = T > 0 ? tap0_w : 0;
    auto gdma = [&](int /*ks*/, int buf) {
        const int t = dma_t;
        const int a_tap = (tap_a + dma_q * BK) * 2 + (int)A_BIAS, b_tap = (tap_b + dma_q * BK) * 2;
        const bool pk = dma_q + 1 < chunks || last_ok;
        uint16_t* A = lds + buf * STAGE + swave * 512;          // wave-uniform: M0 base; hardware adds lane * 16 B
        uint16_t* B = A + BM * LDR;
        typedef __attribute__((address_space(3))) void* lptr;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const uint32_t voff = (((a_valid[i] >> t) & 1u) && pk) ? a_off[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xdma, (lptr)(A + i * (RPP * 64)), 16, (int)voff, a_tap, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr)(B + i * (RPP * 64)), 16, pk ? (int)b_voff[i] : (int)OOB, b_tap, 0, 0);
        // advance to the next K-step
        if (++dma_t == T) {
            dma_t = 0;
            ++dma_q;
        }
        tap_a = (cc.dh[dma_t] * Wi + cc.dw[dma_t]) * Ci + cc.aofs[dma_t];
        tap_b = cc.wofs[dma_t];
    };

    if constexpr (HL > 0) {
        static_assert(PW == 4 && GLDS && (NS == 4 || NS == 6), "halo form: FOUR producer waves (hdma's group index is k * 4 + swave), four weight stages (or six: the one-tap-ahead form)");
        // a chunk's halo is issued two groups (of 8 pixels per producer wave: 64 pixels) per K-step from t = 0 and must be complete
        // before the next chunk's first tile waits with vmcnt(2 * LPT): all groups issued by t <= 6, i.e. at most 7 x 64 pixels
        static_assert(HL <= 448, "halo form: the next chunk's halo must be issued within 7 K-steps (at most 2 groups x 4 waves x 8 pixels each)");
        constexpr int HPM = HL;                              // pixels per halo buffer
        uint16_t* const Hbase = lds + NS * STAGE;            // two halo buffers of HPM x 64 channels
        uint16_t* const pad_zone = Hbase + 2 * HPM * BK;     // 1 KiB landing zone for the padding DMAs (below)
        typedef __attribute__((address_space(3))) void* lptr;
        const uint32_t W2 = Wg + 2, H2 = Hg + 2;
        auto p0 = [&](uint32_t m) -> int {                   // position of output pixel m's top-left tap in the padded raster
            const uint32_t t1 = m / Wg, x = m - t1 * Wg;
            const uint32_t n = t1 / Hg, y = t1 - n * Hg;
            return (int)((n * H2 + y) * W2 + x);
        };
        const uint32_t m_last = (m0 + BM < M ? m0 + BM : M) - 1;
        const int Pb = p0(m0);
        const int HPn = p0(m_last) + 2 * (int)W2 + 2 - Pb + 1;     // halo pixels of this tile (host: <= HPM)
        const int G = (HPn + 7) >> 3;                              // DMA groups of 8 pixels (1 KiB of LDS)
        __shared__ int2 htab[HPM];                                 // per halo pixel: byte offset of its channel row (or OOB), swizzle key
        for (int j = tid; j < HPM; j += THREADS) {
            int2 e = {(int)OOB, 0};
            if (j < HPn) {
                const uint32_t P = (uint32_t)(Pb + j), t1 = P / W2, hx = P - t1 * W2;
                const uint32_t n = t1 / H2, hy = t1 - n * H2;
                if (n < (uint32_t)pp.N && hy >= 1 && hy <= Hg && hx >= 1 && hx <= Wg)
                    e.x = (int)((((n * Hg + hy - 1) * Wg + hx - 1) * Ci) * 2u);
                const int v = (int)(n * Hg * Wg) + ((int)hy - 1) * (int)Wg + (int)hx - 1;
                e.y = (v >> 1) & 7;
            }
            htab[j] = e;
        }
        int tj[9], tv[9], two[9];                             // per tap: displacement in the padded raster / in v, weight offset
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            tj[t] = (cc.dh[t] + 1) * (int)W2 + cc.dw[t] + 1;
            tv[t] = cc.dh[t] * (int)Wg + cc.dw[t];
            two[t] = cc.wofs[t] * 2;
        }
        int jrow[MI], vrow[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            uint32_t m = m0 + wr * TM + i * 32 + (lane & 31);
            if (m > m_last) m = m_last;
            jrow[i] = p0(m) - Pb;
            vrow[i] = (int)m;
        }
#ifdef AFAN_CONV_STAMP
        __shared__ unsigned long long st_lds[72][3];
        const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && wave == 0;
#endif
        __syncthreads();                                      // htab is written
        AFAN_PHASE(1);

        auto compute_h = [&](int buf, int hb, int tjt, int tvt) {
            const uint16_t* B = lds + buf * STAGE;
            const uint16_t* Hh = Hbase + hb * (HPM * BK);
            const int frow = lane & 31;
            const int sw = (frow >> 1) & 7;
            int arow[MI], akey[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                arow[i] = (jrow[i] + tjt) * BK;
                akey[i] = ((vrow[i] + tvt) >> 1) & 7;
            }
            constexpr int FB = FBT;
#pragma unroll
            for (int k0 = 0; k0 < BK / 16; k0 += FB) {
                bf16x8 fx[FB][MI], fw[FB][NI];
#pragma unroll
                for (int b = 0; b < FB; ++b) {
                    const int c2 = (k0 + b) * 2 + (lane >> 5);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        fx[b][i] = *reinterpret_cast<const bf16x8*>(Hh + arow[i] + ((c2 ^ akey[i]) * 8));
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        fw[b][j] = *reinterpret_cast<const bf16x8*>(B + (wc * TN + j * 32 + frow) * LDR + ((c2 ^ sw) * 8));
                }
                if (FB > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < FB; ++b)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int i = 0; i < MI; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[b][j], fx[b][i], acc[j][i], 0, 0, 0);
                if (FB > 1) __builtin_amdgcn_sched_barrier(0);
            }
        };

        // (Round 5 tried the fragments one group AHEAD of their MFMAs, across the barrier — the next weight tile in LDS one barrier
        // earlier, address arithmetic and requests interleaved into the MFMA shadows with sched_group_barrier: per-tap stamps,
        // tools/probe/conv_stamps.py, profiles/r05b_conv_stamps_baseline.txt / r05c_conv_stamps_hpipe2.txt, showed the MFMA waves'
        // share of a tap falling 1010 -> 835, 724 -> 600, 610 -> 529 cycles while the launches kept their time: the tap period of the
        // 64-column tiles is set by the producers' DMA issue, and with one tile less in flight the step's cold weights cost more
        // than the waves gained: 9.0-9.2 ms against 8.66.  Not kept; what the stamps did pay for is the schedule below.)
        //
        // Issue slots: behind tap t's barrier a producer wave issues the weight tile three taps ahead (B_ROWS DMAs) and
        // HaloSched::s(t) halo groups of the NEXT chunk; a group index beyond this tile's groups is a DMA of zeros into the landing
        // zone (the counted waits need compile-time instruction counts).
        uint32_t b_vo[B_ROWS];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) b_vo[i] = RPP * i < b_rows_ok ? b_off + i * b_row32 : OOB;
        auto hdma = [&](int hb, int k, int q) {               // group k*4 + swave of chunk q -> halo buffer hb
            const int gi = k * 4 + swave;
            if (gi < G) {
                const int2 e = htab[gi * 8 + (lane >> 3)];
                const uint32_t voff = (uint32_t)e.x + (uint32_t)((((lane & 7) ^ e.y)) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr)(Hbase + hb * (HPM * BK) + gi * 512), 16, (int)voff, q * (BK * 2), 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr)pad_zone, 16, (int)OOB, 0, 0, 0);
            }
        };
        auto bdma = [&](int buf, int wo2, int q) {
#ifdef AFAN_EXP_NOW
            return;
#endif
            uint16_t* Bd = lds + buf * STAGE + swave * 512;
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr)(Bd + i * (RPP * 64)), 16, (int)b_vo[i], wo2 + q * (BK * 2), 0, 0);
        };
        // The next chunk's halo rides with the first seven taps of this one: KG = ceil(HL / 32) groups per producer wave, s(t) of
        // them behind tap t's weight tile (round 4 issued two per tap on all nine taps and filled the unused slots with padding DMAs
        // of zeros to keep the counted waits fixed: 36 DMA instructions per wave and chunk for 26-31 of payload — and the per-tap
        // stamps of round 5, profiles/r05b_conv_stamps_baseline.txt, show the producers' DMA issue, ~130 cycles per instruction
        // through the CU's one address path, setting the tap period of the 64-column tiles).  The counted waits take each tap's own
        // instruction count — compile-time constants of the unrolled tap index.
        if constexpr (NS == 6) {
        // ---- round 6: the MFMA waves one tap AHEAD.  In the form below (NS == 4) a tap is, for an MFMA wave: barrier -> 12-16 fragment
        // requests -> their latency -> 8-16 MFMAs, with one MFMA wave per SIMD and nothing to cover the requests (stamps: 833 / 1 039
        // ticks per tap at the 512- / 256-channel stages for 256 / 512 of matrix-pipe time), and for a producer wave: wait -> barrier
        // -> 3-5 DMA instructions of ~130 ticks each.  Round 5 moved the fragments one group ahead with four weight buffers — one tile
        // less in flight — and the launches kept their time (NOTES 12.3); round 6 measured that neither the weights' traffic nor the
        // barrier count is what a tap waits for (profiles/r06_kloop_experiments.txt).  Here BOTH chains get slack: six weight
        // buffers, tile t + 1 required at tap t's barrier and tiles t + 2 .. t + 5 in flight behind it (four instead of three), and an
        // MFMA wave requests tap t + 1's fragments slice by slice between tap t's MFMAs (two register sets, the set index a literal
        // of the tap unrolled over a double chunk: 18 taps).  The products are added in the same order as in every other form.
        // LDS hazards: tile t + 1's buffer is re-filled behind barrier t + 2 (slot t + 2 issues tile t + 7): every wave's requests
        // of tile t + 1 were consumed by tap t + 1's MFMAs by then; the halo buffer of chunk q - 1 is re-filled from slot (q, 0) on:
        // its last requests (for tap (q - 1, 8)) were consumed before barrier (q, 0).
        typedef HaloSched5<(HL + 31) / 32, B_ROWS> HS5;
        if (producer) {
            for (int k = 0; k * 4 < G; ++k) hdma(0, k, 0);     // chunk 0's halo (older than every counted instruction)
#pragma unroll
            for (int u = 0; u < 5; ++u) bdma(u, two[u], 0);    // tiles 0 .. 4
            __builtin_amdgcn_s_waitcnt(vmcnt_imm(4 * B_ROWS)); // tile 0 (and the halo in front of it) has landed
            __builtin_amdgcn_s_barrier();                      // the MFMA waves request tap 0's fragments behind this one
            for (int q = 0; q < chunks; ++q) {
                const bool more = q + 1 < chunks;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    switch (t) {                               // tile (q, t) + 1 has landed: the three newest slots may be in flight
#define AFAN_HS_WAIT(T) case T: if (more) __builtin_amdgcn_s_waitcnt(vmcnt_imm(HS5::wait(T, true))); else __builtin_amdgcn_s_waitcnt(vmcnt_imm(HS5::wait(T, false))); break;
                        AFAN_HS_WAIT(0) AFAN_HS_WAIT(1) AFAN_HS_WAIT(2) AFAN_HS_WAIT(3) AFAN_HS_WAIT(4) AFAN_HS_WAIT(5) AFAN_HS_WAIT(6) AFAN_HS_WAIT(7)
                        AFAN_HS_WAIT(8)
#undef AFAN_HS_WAIT
                    }
                    __builtin_amdgcn_s_barrier();
                    if (more) {                                // the next chunk's halo part first: it is waited for one slot earlier than the tile behind it
#pragma unroll
                        for (int j = 0; j < HS5::s(t); ++j) hdma((q + 1) & 1, HS5::g0(t) + j, q + 1);
                    }
                    const int g5 = 9 * q + t + 5;              // tile five taps ahead -> the buffer tile (q, t) - 1 used
                    if (t + 5 < 9 || more) bdma(g5 % 6, two[(t + 5) % 9], t + 5 < 9 ? q : q + 1);
                }
            }
        } else {
            late_prologue();
            const int frow = lane & 31, sw = (frow >> 1) & 7;
            bf16x8 fxs[2][BK / 16][MI], fws[2][BK / 16][NI];
            // requests of one k16-slice of a tap's fragments into register set SET; the MFMAs of one slice from set SET
#define AFAN_RD(SET, WBUF, HB, TJ, TV, KK)                                                                                        \
            {                                                                                                                     \
                const uint16_t* B_ = lds + (WBUF) * STAGE;                                                                        \
                const uint16_t* H_ = Hbase + (HB) * (HPM * BK);                                                                   \
                const int c2_ = (KK) * 2 + (lane >> 5);                                                                           \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                    \
                    fxs[SET][KK][i] = *reinterpret_cast<const bf16x8*>(H_ + (jrow[i] + (TJ)) * BK + ((c2_ ^ (((vrow[i] + (TV)) >> 1) & 7)) * 8)); \
                _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                                    \
                    fws[SET][KK][j] = AFAN_EXP_W(B_ + (wc * TN + j * 32 + frow) * LDR + ((c2_ ^ sw) * 8), c2_ + j);               \
            }
#ifdef AFAN_EXP_NOW      /* timing experiment only: the weight operand out of thin air (no DMA, no LDS reads) */
#define AFAN_EXP_W(PTR, X) __builtin_bit_cast(bf16x8, u32x4{(uint32_t)jrow[0], (uint32_t)(X), 0x3c003c00u, (uint32_t)vrow[0]})
#else
#define AFAN_EXP_W(PTR, X) (*reinterpret_cast<const bf16x8*>(PTR))
#endif
#define AFAN_MF(SET, KK)                                                                                                          \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                                        \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                    \
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fws[SET][KK][j], fxs[SET][KK][i], acc[j][i], 0, 0, 0);
            // tap U of the double chunk (global tap gbase + U): its MFMAs from set U & 1, the next tap's requests into the other set —
            // weight buffer (U + 1) % 6 (18 is a multiple of 6), halo buffer = parity of the next tap's chunk
#define AFAN_TAP(U)                                                                                                               \
            {   /* (no condition anywhere in a tap: a branch makes the compiler copy the accumulators around it and wait for every  \
                   outstanding request at the join; the last tap's requests read a stale buffer and are never used) */              \
                __builtin_amdgcn_s_barrier();                                                                                     \
                _Pragma("unroll") for (int kk = 0; kk < BK / 16; ++kk) {                                                          \
                    AFAN_RD(((U) + 1) & 1, ((U) + 1) % 6, ((((U) + 1) % 18) >= 9 ? 1 : 0), tj[((U) + 1) % 9], tv[((U) + 1) % 9], kk) \
                    __builtin_amdgcn_sched_barrier(0);                                                                            \
                    AFAN_MF((U) & 1, kk)                                                                                          \
                    __builtin_amdgcn_sched_barrier(0);                                                                            \
                }                                                                                                                 \
            }
            __builtin_amdgcn_s_barrier();                      // tile 0 and chunk 0's halo are in LDS
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) AFAN_RD(0, 0, 0, tj[0], tv[0], kk)
            for (int d = 0; d < (chunks >> 1); ++d) {          // whole double chunks: 18 taps, the register sets' parity comes back to 0
#pragma unroll
                for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(jrow[i]), "+v"(vrow[i]));
                AFAN_TAP(0) AFAN_TAP(1) AFAN_TAP(2) AFAN_TAP(3) AFAN_TAP(4) AFAN_TAP(5) AFAN_TAP(6) AFAN_TAP(7) AFAN_TAP(8)
                AFAN_TAP(9) AFAN_TAP(10) AFAN_TAP(11) AFAN_TAP(12) AFAN_TAP(13) AFAN_TAP(14) AFAN_TAP(15) AFAN_TAP(16) AFAN_TAP(17)
            }
            if (chunks & 1) {                                  // a last single chunk
#pragma unroll
                for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(jrow[i]), "+v"(vrow[i]));
                AFAN_TAP(0) AFAN_TAP(1) AFAN_TAP(2) AFAN_TAP(3) AFAN_TAP(4) AFAN_TAP(5) AFAN_TAP(6) AFAN_TAP(7) AFAN_TAP(8)
            }
#undef AFAN_RD
#undef AFAN_MF
#undef AFAN_TAP
        }
        } else {
        typedef HaloSched<(HL + 31) / 32, B_ROWS> HS;
        if (producer) {
            for (int k = 0; k * 4 < G; ++k) hdma(0, k, 0);     // chunk 0's halo (older than every counted instruction)
#pragma unroll
            for (int s = 0; s < NS - 1; ++s) bdma(s, two[s], 0);
            for (int q = 0; q < chunks; ++q) {
                const bool more = q + 1 < chunks;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    // tiles after (q, t): the two newest issue slots may still be in flight behind the wait
                    switch (t) {                               // (the builtin wants literal constants: t is one after unrolling)
#define AFAN_HS_WAIT(T) case T: if (more) __builtin_amdgcn_s_waitcnt(vmcnt_imm(HS::wait(T, true))); else __builtin_amdgcn_s_waitcnt(vmcnt_imm(HS::wait(T, false))); break;
                        AFAN_HS_WAIT(0) AFAN_HS_WAIT(1) AFAN_HS_WAIT(2) AFAN_HS_WAIT(3) AFAN_HS_WAIT(4) AFAN_HS_WAIT(5) AFAN_HS_WAIT(6) AFAN_HS_WAIT(7)
                        AFAN_HS_WAIT(8)
#undef AFAN_HS_WAIT
                    }
                    __builtin_amdgcn_s_barrier();              // tile (q, t) is in LDS for everyone; the buffer of the tile before is free
                    const int nb = (q + t + 3) & 3;            // 9 = 1 (mod 4): tile 9 q + t lives in weight buffer (q + t) & 3
                    if (t + 3 < 9 || more) bdma(nb, two[(t + 3) % 9], t + 3 < 9 ? q : q + 1);
                    // chunk q + 1's halo goes into the buffer chunk q - 1 used (free since this chunk's first barrier); all of it is
                    // issued by tap 6, i.e. complete at chunk q + 1's first wait (which leaves the slots of taps 7 and 8 in flight)
                    if (more) {
#pragma unroll
                        for (int j = 0; j < HS::s(t); ++j) hdma((q + 1) & 1, HS::g0(t) + j, q + 1);
                    }
                }
            }
        } else
        {
            late_prologue();
            for (int q = 0; q < chunks; ++q) {
                // (opaque per chunk: otherwise the nine taps' fragment addresses are hoisted out of the chunk loop as
                // 40-odd loop-invariant registers, which the 168-register variant spills)
#pragma unroll
                for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(jrow[i]), "+v"(vrow[i]));
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    AFAN_STAMP(0, q * 9 + t, 0);
                    __builtin_amdgcn_s_barrier();
                    AFAN_STAMP(0, q * 9 + t, 1);
                    compute_h((q + t) & 3, q & 1, tj[t], tv[t]);
                    AFAN_STAMP(0, q * 9 + t, 2);
                }
            }
        }
        }
        __syncthreads();
#ifdef AFAN_CONV_STAMP
        if (stamp_on)
            for (int i = 0; i < 96; ++i)
                for (int k = 0; k < 3; ++k) afan_stamps[0][i][k] = (i < chunks * 9 && i < 72) ? st_lds[i][k] : 0ull;
        __syncthreads();
#endif
    } else if constexpr (PW > 0) {
        static_assert(GLDS, "producer waves: LDS-DMA only");
        AFAN_PHASE(1);
        constexpr int LPT = A_ROWS + B_ROWS;
        if constexpr (NS == 2) {
            // two stages (two workgroups per CU cover each other's DMA latency): the producers issue tile ks+1 while the
            // MFMA waves work on tile ks; one barrier per K-step ends both
            if (producer) {
                gdma(0, 0);
                __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
                __builtin_amdgcn_s_barrier();
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) gdma(ks + 1, (ks & 1) ^ 1);
                    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
                    __builtin_amdgcn_s_barrier();
                }
            } else {
                late_prologue();
                __builtin_amdgcn_s_barrier();
                for (int ks = 0; ks < KS; ++ks) {
                    compute(ks & 1);
                    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this tile's fragment reads have returned
                    __builtin_amdgcn_s_barrier();
                }
            }
            __syncthreads();
        } else
        {
        // Two separate loops with the same barrier count (one loop with a role branch inside made the compiler copy all
        // 64 accumulator registers around the branch on every K-step).
        if (producer) {
#pragma unroll
            for (int s = 0; s < NS - 1; ++s)
                if (s < KS) gdma(s, s);
            int buf = 0;
            for (int ks = 0; ks < KS; ++ks) {
                const int rem = KS - 1 - ks;             // this wave's pieces of tile ks have landed:
                if (rem >= NS - 2) __builtin_amdgcn_s_waitcnt(vmcnt_imm(LPT * (NS - 2)));
                else if (NS == 4 && rem == 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(LPT));
                else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
                __builtin_amdgcn_s_barrier();            // tile ks is in LDS for everyone; buffer of tile ks-1 is free
                if (ks + NS - 1 < KS) gdma(ks + NS - 1, buf == 0 ? NS - 1 : buf - 1);
                buf = buf + 1 == NS ? 0 : buf + 1;
            }
        } else {
#ifdef AFAN_CONV_PRIO
            __builtin_amdgcn_s_setprio(AFAN_CONV_PRIO);
#endif
            late_prologue();
            int buf = 0;
            for (int ks = 0; ks < KS; ++ks) {
                __builtin_amdgcn_s_barrier();
                compute(buf);
                buf = buf + 1 == NS ? 0 : buf + 1;
            }
        }
        __syncthreads();
        }
    } else if constexpr (GLDS && NS > 2) {
        AFAN_PHASE(1);
        // Deep pipeline for launches of about one workgroup per CU (the 8x8 and 4x4 stages: few, long K loops): with two
        // buffers and __syncthreads() every DMA has to land within ONE step's MFMAs (a few hundred cycles against
        // 1-2 k of memory latency).  Here NS - 1 tiles are in flight; the wait is counted (vmcnt(N) leaves the younger
        // tiles outstanding) and the barrier is the bare instruction — __syncthreads() would drain the DMA queue.
        constexpr int LPT = A_ROWS + B_ROWS;             // DMA instructions per thread per tile
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < KS) gdma(s, s);
        late_prologue();
        int buf = 0;
        for (int ks = 0; ks < KS; ++ks) {
            const int rem = KS - 1 - ks;                 // tiles after this one; min(rem, NS - 2) of them are in flight
            if (rem >= NS - 2) __builtin_amdgcn_s_waitcnt(vmcnt_imm(LPT * (NS - 2)));
            else if (NS == 4 && rem == 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(LPT));
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
            __builtin_amdgcn_s_barrier();                // tile ks is in LDS for everyone; buffer of tile ks-1 is free
            if (ks + NS - 1 < KS) gdma(ks + NS - 1, buf == 0 ? NS - 1 : buf - 1);
            compute(buf);
            buf = buf + 1 == NS ? 0 : buf + 1;
        }
        __syncthreads();
    } else if constexpr (GLDS) {
        AFAN_PHASE(1);
        gdma(0, 0);
        late_prologue();
        __syncthreads();                     // (the compiler drains vmcnt before the barrier)
        for (int ks = 0; ks < KS; ++ks) {
            const int buf = ks & 1;
            if (ks + 1 < KS) gdma(ks + 1, buf ^ 1);   // lands while this step's MFMAs run
            compute(buf);
            __syncthreads();
        }
    } else if constexpr (PF == 1) {
        late_prologue();
        gload(0, ra0, rb0);
        lstore(0, ra0, rb0);
        __syncthreads();
        for (int ks = 0; ks < KS; ++ks) {
            const int buf = ks & 1;
            if (ks + 1 < KS) gload(ks + 1, ra0, rb0);  // in flight under this step's MFMAs
            compute(buf);
            if (ks + 1 < KS) lstore(buf ^ 1, ra0, rb0);
            __syncthreads();
        }
    } else {
        // two register sets: the loads of step k+2 are issued at the top of step k and written to LDS at the bottom of
        // step k+1, so they have a whole step (MFMAs + barrier) more to land.  Unrolled by 2: static register sets.
        late_prologue();
        gload(0, ra0, rb0);
        lstore(0, ra0, rb0);
        if (KS > 1) gload(1, ra1, rb1);
        __syncthreads();
        for (int ks = 0; ks < KS; ks += 2) {
            if (ks + 2 < KS) gload(ks + 2, ra0, rb0);
            compute(0);
            if (ks + 1 < KS) lstore(1, ra1, rb1);
            __syncthreads();
            if (ks + 1 < KS) {
                if (ks + 3 < KS) gload(ks + 3, ra1, rb1);
                compute(1);
                if (ks + 2 < KS) lstore(0, ra0, rb0);
                __syncthreads();
            }
        }
    }

    AFAN_PHASE(2);
    // ---- epilogue: fp32 accumulators -> packed bf16 tile [pixel][channel] in LDS -> 16-byte channels-last stores ------
    constexpr int LDC = BN + 8;
    uint16_t* C = lds;  // BM x LDC elements <= 2 * STAGE
    // (in-launch BatchNorm backward with the projection shortcut's: that BatchNorm's input tile is staged in LDS behind the output
    // tile by DMA — no registers, in flight while the accumulators are rounded into C; the barrier below drains it)
    uint16_t* const Z = lds + BM * LDC;
    // (not in the 768-thread variant: at its 168-register cap the third sum's state spills in the prefetch phase; launch_gs refuses)
    constexpr bool DRES_EARLY = AFAN_CONV_DRES_EARLY != 0;
    constexpr bool BSC_OK = BF && HL > 0;        // (and not in the per-tap variants: their LDS has no room for the staged tile; launch_gs refuses)
    bool have_bsc = false;
    __shared__ float zmean_s[BF ? BN : 1];       // (BF) the projection BatchNorm's mean per tile column (registers are short in the row loop)
    unsigned bar_target = 0;                     // (BF) this launch's barrier episode, asked for here: long before thread 0 needs it
    if constexpr (BF) {
        if (tid == 0) bar_target = grid_episode(pp.bar);
        have_bsc = BSC_OK && pp.bnf == 2 && pp.bsc.x != nullptr;
        if (have_bsc && tid < BN) zmean_s[tid] = n0 + tid < pp.Co ? pp.bsc.stats[n0 + tid] : 0.f;
        if (have_bsc) {
            constexpr int PIECES_Z = BN / 8, NZ = BM * BN * 2 / 1024;
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint16_t*>(pp.bsc.x), 0, (int)((int64_t)pp.N * pp.Ho * pp.Wo * pp.Co * 2), 0x00020000);
            typedef __attribute__((address_space(3))) void* lptr_z;
            for (int k = wave; k < NZ; k += THREADS / 64) {
                const int pos = k * 64 + lane, r = pos / PIECES_Z, pcz = pos % PIECES_Z;
                const int off = out_off[r];
                const uint32_t vo = (off >= 0 && n0 + pcz * 8 < pp.Co) ? (uint32_t)(off + n0 + pcz * 8) * 2u : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(zr, (lptr_z)(Z + k * 512), 16, (int)vo, 0, 0, 0);
            }
        }
    }
    if (!producer)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int pix = wr * TM + i * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = wc * TN + j * 32 + 8 * g + 4 * (lane >> 5);
                u16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f2bf(acc[j][i][4 * g + e]);
                *reinterpret_cast<u16x4*>(C + pix * LDC + ch) = v;
            }
        }
    __syncthreads();
    AFAN_PHASE(3);
    // The epilogue's own global reads (residual addend, BN input, BN output) are requested for all of this thread's
    // output rows before the first is used.  Buffer loads: a row outside the tensor, or a fusion that is off (zero
    // records), is an out-of-range offset — no divergent branch to serialise the requests.  (Requesting them before
    // the K loop instead measured slower: they sit in front of the first operand tiles in the in-order return queue.)
    constexpr int PIECES = BN / 8;               // 16-byte pieces per output row
    constexpr int EPI_THREADS = (THREADS & (THREADS - 1)) == 0 ? THREADS : 512;   // 768-thread workgroups store with their first 512
    constexpr int ROWS_PER_PASS = EPI_THREADS / PIECES;
    constexpr int EPI_ROWS = BM / ROWS_PER_PASS;
    static_assert(BM % ROWS_PER_PASS == 0 && EPI_THREADS <= THREADS, "epilogue passes must cover the tile exactly");
    const bool epi_on = tid < EPI_THREADS;
    const int pc = tid % PIECES, pr = tid / PIECES;
    const bool ch_ok = n0 + pc * 8 < pp.Co;      // this thread's 8 output channels exist (Co % 8 == 0)
    const bool want_stats = pp.stats != nullptr || pp.acc != nullptr;
    const bool bn_bwd = want_stats && pp.bnx != nullptr;
    // (multi-problem forward launch, ConvP::multi: the BatchNorm belongs to problem 0; the other problems' workgroups run the plain
    // forward with sums — they store their raw tile and leave without touching the barrier, which counts problem 0's workgroups only)
    const bool bf_other = BF && pp.multi && blockIdx.z != 0;
    const bool bf_fwd = BF && pp.bnf == 1 && !bf_other, bf_bwd = BF && pp.bnf == 2;
    const uint16_t* addp = bf_fwd ? pp.bnf_res : (pp.aff ? pp.aff_res : pp.addend);     // the one extra output-shaped operand of either fusion
    // Image groups (two half-batches with their own BatchNorm sums): a tile belongs to the half its rows are in; the ONE tile
    // of a launch that holds rows of both (half-batches of any size: the row count need not be a multiple of the tile) walks
    // its rows twice, once per group, each row fetched, stored and summed in the pass of its own group (the whole epilogue,
    // operand requests included, sits inside the pass: nothing but a few scalars lives across it — kept in registers for a
    // second pass, the requests' 128 registers spill in the 768-thread variants).
    const uint32_t half = pp.groups == 2 ? (uint32_t)(pp.N / 2) * Hg * Wg : 0u;
    const bool straddle = GS && pp.groups == 2 && m0 < half && m0 + BM > half;      // (GS = false: n_pass is the constant 1)
    const int n_pass = straddle ? 2 : 1;
    const float* shift_p = pp.shift ? pp.shift + pp.shift_off[blockIdx.z] : nullptr;
    uint16_t* y_p = pp.y + pp.y_off[blockIdx.z];
    for (int gpass = 0; gpass < n_pass; ++gpass) {
    const int grp = straddle ? gpass : ((pp.groups == 2 && m0 >= half) ? 1 : 0);
    const bool grp_first = grp == 0 ? m0 == 0 : (straddle || m0 == half);     // this tile holds the group's first row
    const bool dual = pp.aff && pp.aff_bwd == 2;
    u32x4 pre_a[EPI_ROWS], pre_x[EPI_ROWS], pre_y[EPI_ROWS];      // (pre_x: the BN input — or, dual form (never both), the addend)
    uint32_t ymask[(EPI_ROWS + 3) / 4];
    {
        const int out_bytes = (int)((int64_t)pp.N * pp.Ho * pp.Wo * pp.Co * 2);
        const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(addp), 0, addp ? out_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pp.addend), 0, (dual && pp.addend) ? out_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t bxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pp.bnx), 0, bn_bwd ? out_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t byr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pp.bny), 0, (bn_bwd && pp.bny) ? out_bytes : 0, 0x00020000);
#pragma unroll
        for (int q = 0; q < EPI_ROWS; ++q) {
            const int r_ = pr + q * ROWS_PER_PASS;
            const int off = (epi_on && (!straddle || ((m0 + (uint32_t)r_ >= half) == (grp == 1)))) ? out_off[r_] : -1;
            const uint32_t bo = (off >= 0 && ch_ok) ? (uint32_t)(off + n0 + pc * 8) * 2u : OOB;
            if (addp && !(BF && EPI_ROWS >= 8)) pre_a[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ar, (int)bo, 0, 0));
            if (dual && pp.addend) pre_x[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, (int)bo, 0, 0));
            else if (bn_bwd) pre_x[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bxr, (int)bo, 0, 0));
            if (bn_bwd && pp.bny) pre_y[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(byr, (int)bo, 0, 0));
        }
        // (BF: the stored-output mask as bits.  The 8-rows-per-thread tile (768 threads at their 168-register cap) cannot keep three
        // prefetched operand tiles through the row loop: the output tile and the BatchNorm input are requested first, the mask is
        // taken from the output tile as soon as it is in, and only then the other branch's gradient is requested into its place)
#pragma unroll
        for (int q4 = 0; q4 < (EPI_ROWS + 3) / 4; ++q4) ymask[q4] = 0u;
        if constexpr (BF) {
            if (bn_bwd && pp.bny) {
                if (EPI_ROWS >= 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < EPI_ROWS; ++q) {
                    const u16x8 yv = __builtin_bit_cast(u16x8, pre_y[q]);
                    uint32_t m = 0u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) m |= (bf2f(yv[j]) > 0.f ? 1u : 0u) << j;
                    ymask[q / 4] |= m << (8 * (q % 4));
                }
                if (EPI_ROWS >= 8) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (EPI_ROWS >= 8) {
                if (addp) {
#pragma unroll
                    for (int q = 0; q < EPI_ROWS; ++q) {
                        const int r_ = pr + q * ROWS_PER_PASS;
                        const int off = epi_on ? out_off[r_] : -1;
                        const uint32_t bo = (off >= 0 && ch_ok) ? (uint32_t)(off + n0 + pc * 8) * 2u : OOB;
                        pre_a[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ar, (int)bo, 0, 0));
                    }
                }
            }
        }
    }
    // Optional fusions on the way out (same LDS reads as the stores):
    //  * addend: y += addend (the other branch of a residual gradient), rounded to bf16 like a separate add would;
    //  * moments of y for a following train-mode BN forward (one partial per (tile, channel), summed by its finalize);
    //  * or, when y is the gradient entering a BN backward, that backward's reduction pass (sum g, sum g*(x - mean)).
    const float* bn_stats = pp.bn_stats ? pp.bn_stats + (int64_t)grp * 4 * pp.Co : nullptr;
    double* acc_blk = pp.acc ? pp.acc + pp.acc_off[blockIdx.z] + (int64_t)grp * pp.acc_stride : nullptr;
    float s1[8], s2[8], sh[8], al[8], be[8];
    float s3[8];                                                  // (BF, projection shortcut's BatchNorm: third sum; its mean: zmean_s)
#pragma unroll
    for (int j = 0; j < 8; ++j) s3[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = s2[j] = 0.f;
        if (GS && straddle && gpass > 0) {                          // (the second group's statistics: not what the prologue fetched)
            const int c = ch_ok ? n0 + pc * 8 + j : 0;
            sh[j] = bn_bwd ? bn_stats[c] : ((want_stats && shift_p) ? shift_p[c] : 0.f);   // mean or shift
            al[j] = bn_bwd ? bn_stats[2 * pp.Co + c] : (pp.aff ? pp.aff[(pp.aff_bwd ? 0 : 2 * pp.Co) + c] : 0.f);
            be[j] = bn_bwd ? bn_stats[3 * pp.Co + c] : ((pp.aff && !pp.aff_bwd) ? pp.aff[3 * pp.Co + c] : 0.f);
        } else {
            sh[j] = coef_s[0][pc * 8 + j];
            al[j] = coef_s[1][pc * 8 + j];
            be[j] = coef_s[2][pc * 8 + j];
        }
    }
#pragma unroll
    for (int q = 0; q < EPI_ROWS; ++q) {
        const int r = pr + q * ROWS_PER_PASS;
        const int off = (epi_on && (!straddle || ((m0 + (uint32_t)r >= half) == (grp == 1)))) ? out_off[r] : -1;
        if (off >= 0 && ch_ok) {
            u16x8 v = *reinterpret_cast<const u16x8*>(C + r * LDC + pc * 8);
            const int64_t go = (int64_t)off + n0 + pc * 8;
            if (dual) {
                // a residual block's output, backwards: the gradient (this launch's + the other branch's, rounded like the
                // dgrad-with-addend launch rounds it), masked by the block's stored output, leaves twice — as it is for the
                // shortcut, times alpha for the last convolution's frozen BatchNorm (affine_bwd_kernel's two outputs)
                const u16x8 a = __builtin_bit_cast(u16x8, pre_a[q]);
                const u16x8 d = __builtin_bit_cast(u16x8, pre_x[q]);
                u16x8 m;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float g = bf2f(v[j]);
                    if (pp.addend) g = bf2f(f2bf(g + bf2f(d[j])));
                    const float t = (bf2f(a[j]) > 0.f) ? g : 0.f;
                    m[j] = f2bf(t);
                    v[j] = f2bf(t * al[j]);
                }
                *reinterpret_cast<u16x8*>(pp.y2 + go) = m;
            } else if (pp.aff && pp.aff_bwd) {
                // the backward of the frozen BatchNorm + ReLU in front of this convolution, on the bf16-rounded input gradient:
                // affine_bwd_kernel's expression (mask from the stored activation, then the plain multiply)
                const u16x8 a = __builtin_bit_cast(u16x8, pre_a[q]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = (bf2f(a[j]) > 0.f) ? bf2f(v[j]) : 0.f;
                    t *= al[j];
                    v[j] = f2bf(t);
                }
            } else if (pp.aff) {
                // a frozen BatchNorm (+ residual) (+ ReLU) on the bf16-rounded convolution output: afan_nhwc::apply_kernel's
                // expression, term for term (fmaf, then the residual, then the NaN-passing ReLU)
                const u16x8 a = __builtin_bit_cast(u16x8, pre_a[q]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = fmaf(bf2f(v[j]), al[j], be[j]);
                    if (pp.aff_res) t += bf2f(a[j]);
                    if (pp.aff_relu) t = (t > 0.f) ? t : ((t != t) ? t : 0.f);
                    v[j] = f2bf(t);
                }
            } else if (pp.addend) {
                const u16x8 a = __builtin_bit_cast(u16x8, pre_a[q]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) + bf2f(a[j]));
            }
            // (in-launch BatchNorm: the backward never writes the unnormalised gradient; the forward stores its raw tile later,
            // between its arrival at the grid barrier and its wait there)
            if (!BF) *reinterpret_cast<u16x8*>(y_p + go) = v;
            if (bn_bwd) {
                const u16x8 xv = __builtin_bit_cast(u16x8, pre_x[q]);
                u16x8 yv = xv;
                if constexpr (!BF) if (pp.bny) yv = __builtin_bit_cast(u16x8, pre_y[q]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xf = bf2f(xv[j]);
                    float g = bf2f(v[j]);
                    if (pp.bny) g = (BF ? ((ymask[q / 4] >> (8 * (q % 4) + j)) & 1u) != 0u : bf2f(yv[j]) > 0.f) ? g : 0.f;   // mask from the stored activation
                    else if (pp.bn_relu) g = (fmaf(xf, al[j], be[j]) > 0.f) ? g : 0.f;   // mask recomputed exactly as the forward
                    s1[j] += g;
                    s2[j] += g * (xf - sh[j]);
                    if constexpr (BF) v[j] = f2bf(g);                                    // (exact: g is v[j] or 0)
                }
                if constexpr (BF) {     // the masked gradient waits in the tile (this thread's own elements) for the totals
                    if (bf_bwd) *reinterpret_cast<u16x8*>(C + r * LDC + pc * 8) = v;
                    if (have_bsc) {
                        const u16x8 zv = *reinterpret_cast<const u16x8*>(Z + r * BN + pc * 8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) s3[j] += bf2f(v[j]) * (bf2f(zv[j]) - zmean_s[pc * 8 + j]);
                    }
                }
            } else if (want_stats) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f(v[j]) - sh[j];
                    s1[j] += f;
                    s2[j] += f * f;
                }
            }
        }
    }
    AFAN_PHASE(4);
    float bf_shift = 0.f;
    if (want_stats) {
        // lanes l, l+PIECES, l+2*PIECES, ... of a wave hold the same 8 columns: butterfly over those, then over waves
#pragma unroll
        for (int o = PIECES; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
        // per-wave partial sums: static LDS, or (halo form: the static budget is spent) the second halo buffer, idle by now
        __shared__ float red_static[HL ? 1 : THREADS / 64][2][BN];
        float (*red)[2][BN] = red_static;
        if constexpr (HL > 0) red = reinterpret_cast<float (*)[2][BN]>(lds + ((PF - 1) * BN * BK + HL * BK));
        __shared__ float red3[BSC_OK ? THREADS / 64 : 1][BSC_OK ? BN : 1];      // the projection BatchNorm's third sum, per wave
        if constexpr (BF && HL > 0) {                                    // (behind the staged Z tile, which the halo-form spot overlaps)
            static_assert((size_t)(BM * LDC + BM * BN) * 2 + (size_t)(THREADS / 64) * 2 * BN * 4 <=
                          (size_t)(PF - 1) * BN * BK * 2 + (size_t)2 * HL * BK * 2 + 1024, "in-launch BatchNorm: LDS behind the two tiles");
            red = reinterpret_cast<float (*)[2][BN]>(lds + (BM * LDC + BM * BN));
        }
        if constexpr (BF) {
            if (have_bsc) {
#pragma unroll
                for (int o = PIECES; o < 64; o <<= 1)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s3[j] += __shfl_xor(s3[j], o, 64);
                if (lane < PIECES)
#pragma unroll
                    for (int j = 0; j < 8; ++j) red3[wave][lane * 8 + j] = s3[j];
            }
        }
        if (lane < PIECES) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[wave][0][lane * 8 + j] = s1[j];
                red[wave][1][lane * 8 + j] = s2[j];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < pp.Co) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) {
                a += red[w][0][tid];
                b += red[w][1][tid];
            }
            if (pp.acc) {
                double* dst = acc_blk + (int64_t)((blockIdx.y + blockIdx.z) & (pp.acc_ns - 1)) * 2 * pp.Co;
                unsafeAtomicAdd(dst + n0 + tid, (double)a);
                unsafeAtomicAdd(dst + pp.Co + n0 + tid, (double)b);
                if (!bn_bwd && grp_first && (blockIdx.z == 0 || pp.multi))   // snapshot of the shift for the BN that consumes the sums
                    reinterpret_cast<float*>(acc_blk + (int64_t)2 * pp.acc_ns * pp.Co)[n0 + tid] = shift_p ? shift_p[n0 + tid] : 0.f;
                if constexpr (BF) {
                    bf_shift = shift_p ? shift_p[n0 + tid] : 0.f;   // read BEFORE the barrier: tile 0 updates the running mean behind it
                    if (have_bsc) {
                        float b3 = 0.f;
#pragma unroll
                        for (int w = 0; w < THREADS / 64; ++w) b3 += red3[w][tid];
                        double* dz = pp.bsc.acc + (int64_t)((blockIdx.y + blockIdx.z) & (pp.acc_ns - 1)) * 2 * pp.Co;
                        unsafeAtomicAdd(dz + n0 + tid, (double)a);
                        unsafeAtomicAdd(dz + pp.Co + n0 + tid, (double)b3);
                    }
                }
            } else {
                const int64_t G = (int64_t)gridDim.y * gridDim.z, slot = (int64_t)blockIdx.z * gridDim.y + blockIdx.y;
                pp.stats[((int64_t)0 * pp.Co + n0 + tid) * G + slot] = a;
                pp.stats[((int64_t)1 * pp.Co + n0 + tid) * G + slot] = b;
            }
        }
        if (n_pass > 1) __syncthreads();         // the second pass reuses the per-wave partial sums' LDS
        AFAN_PHASE(5);
        if constexpr (BF) {
            // ---- the BatchNorm itself, inside this launch: every workgroup's sums are in the accumulators once all have passed the
            // barrier; each then derives the coefficients of its own BN channels from the totals (the expressions of apply_acc_kernel /
            // apply_acc_dual_kernel / bwd_apply_acc_kernel, term for term) and finishes its tile from LDS.
            if (bf_other) {
#pragma unroll
                for (int q = 0; q < EPI_ROWS; ++q) {
                    const int r = pr + q * ROWS_PER_PASS;
                    const int off = epi_on ? out_off[r] : -1;
                    if (off >= 0 && ch_ok)
                        *reinterpret_cast<u16x8*>(y_p + (int64_t)off + n0 + pc * 8) = *reinterpret_cast<const u16x8*>(C + r * LDC + pc * 8);
                }
                return;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's accumulator atomics have been performed
            __syncthreads();
            const unsigned wg_id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            if (tid == 0) grid_arrive(pp.bar, wg_id, gridDim.x * gridDim.y * (pp.multi ? 1u : gridDim.z));
            // what is final BEFORE the totals leaves while the other workgroups arrive: the forward's raw tile; the backward's masked
            // gradient (the shortcut's share, y2 — it waits in the tile since the first pass; round 6: was stored behind the barrier,
            // in front of the second pass's own stores)
            uint16_t* const early = bf_fwd ? y_p : ((bf_bwd && DRES_EARLY) ? pp.y2 : nullptr);
            if (early) {
#pragma unroll
                for (int q = 0; q < EPI_ROWS; ++q) {
                    const int r = pr + q * ROWS_PER_PASS;
                    const int off = epi_on ? out_off[r] : -1;
                    if (off >= 0 && ch_ok)
                        *reinterpret_cast<u16x8*>(early + (int64_t)off + n0 + pc * 8) = *reinterpret_cast<const u16x8*>(C + r * LDC + pc * 8);
                }
            }
            if (tid == 0) grid_wait(pp.bar, bar_target);
            __syncthreads();
            AFAN_PHASE(6);
            float* cf = &red[0][0][0];                                // [4][BN] coefficients of the second pass (the partial sums are consumed)
            const bool have_sc = bf_fwd && pp.bnf_sc.acc != nullptr;
            if (tid < BN && n0 + tid < pp.Co) {
                const int c = n0 + tid, NSl = pp.acc_ns;
                const bool pub = blockIdx.y == 0 && blockIdx.z == 0;  // the row tile that publishes statistics / parameter gradients
                double av[8], bv[8];                                  // (>= 128 channels: at most 8 accumulator copies; launch_gs checks)
                const __amdgpu_buffer_rsrc_t accr = __builtin_amdgcn_make_buffer_rsrc(acc_blk, 0, (int)(2 * NSl * pp.Co * 8), 0x00020000);
#pragma unroll
                for (int s = 0; s < 8; ++s) {                         // fold_slots' order; all requests in flight together
                    av[s] = s < NSl ? ld_total(accr, ((2 * s) * pp.Co + c) * 8) : 0.0;
                    bv[s] = s < NSl ? ld_total(accr, ((2 * s + 1) * pp.Co + c) * 8) : 0.0;
                }
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    a += av[s];
                    b += bv[s];
                }
                if (bf_fwd) {
                    const float wv = pp.bnf_w ? pp.bnf_w[c] : 1.f, bsv = pp.bnf_b ? pp.bnf_b[c] : 0.f;
                    const double dm = a * pp.bnf_inv_m;
                    const double m2 = fmax(b - a * dm, 0.0);
                    const float mean = (float)((double)bf_shift + dm);
                    const float varb = (float)(m2 * pp.bnf_inv_m);
                    const float is = 1.0f / sqrtf(varb + pp.bnf_eps);
                    const float alpha = is * wv, beta = fmaf(-mean, alpha, bsv);
                    cf[tid] = alpha;
                    cf[BN + tid] = beta;
                    if (pub) {
                        pp.bnf_stats[c] = mean;
                        pp.bnf_stats[pp.Co + c] = is;
                        pp.bnf_stats[2 * pp.Co + c] = alpha;
                        pp.bnf_stats[3 * pp.Co + c] = beta;
                        if (pp.bnf_rmean)
                            for (int u = 0; u < pp.bnf_updates; ++u) {
                                pp.bnf_rmean[c] = (1.0f - pp.bnf_mom) * pp.bnf_rmean[c] + pp.bnf_mom * mean;
                                pp.bnf_rvar[c] = (1.0f - pp.bnf_mom) * pp.bnf_rvar[c] + pp.bnf_mom * (varb * pp.bnf_unbias);
                            }
                        if (c == 0 && pp.bnf_nbt) *pp.bnf_nbt += pp.bnf_updates;
                    }
                    if (have_sc) {                                    // the projection shortcut's BatchNorm: sums of an earlier launch
                        const ConvP::Sc& P = pp.bnf_sc;
                        double a2 = 0.0, b2 = 0.0;
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            av[s] = s < NSl ? P.acc[(int64_t)(2 * s) * pp.Co + c] : 0.0;
                            bv[s] = s < NSl ? P.acc[(int64_t)(2 * s + 1) * pp.Co + c] : 0.0;
                        }
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            a2 += av[s];
                            b2 += bv[s];
                        }
                        const float w2 = P.w ? P.w[c] : 1.f, bs2 = P.b ? P.b[c] : 0.f;
                        const float sh2 = reinterpret_cast<const float*>(P.acc + (int64_t)2 * NSl * pp.Co)[c];
                        const double dm2 = a2 * pp.bnf_inv_m;
                        const double m22 = fmax(b2 - a2 * dm2, 0.0);
                        const float mean2 = (float)((double)sh2 + dm2);
                        const float var2 = (float)(m22 * pp.bnf_inv_m);
                        const float is2 = 1.0f / sqrtf(var2 + P.eps);
                        const float alpha2 = is2 * w2, beta2 = fmaf(-mean2, alpha2, bs2);
                        cf[2 * BN + tid] = alpha2;
                        cf[3 * BN + tid] = beta2;
                        if (pub) {
                            P.stats[c] = mean2;
                            P.stats[pp.Co + c] = is2;
                            P.stats[2 * pp.Co + c] = alpha2;
                            P.stats[3 * pp.Co + c] = beta2;
                            if (P.rmean)
                                for (int u = 0; u < pp.bnf_updates; ++u) {
                                    P.rmean[c] = (1.0f - P.mom) * P.rmean[c] + P.mom * mean2;
                                    P.rvar[c] = (1.0f - P.mom) * P.rvar[c] + P.mom * (var2 * pp.bnf_unbias);
                                }
                            if (c == 0 && P.nbt) *P.nbt += pp.bnf_updates;
                        }
                    }
                } else {
                    const float is = bn_stats[pp.Co + c], alpha = bn_stats[2 * pp.Co + c];
                    const float sum_g = (float)a;
                    const float sum_gx = (float)b * is;
                    cf[tid] = -alpha * is * (float)((double)sum_gx * pp.bnf_inv_m);
                    cf[BN + tid] = -alpha * (float)((double)sum_g * pp.bnf_inv_m);
                    if (pub) {
                        if (pp.bnf_db) pp.bnf_db[c] = pp.bnf_accum ? pp.bnf_db[c] + sum_g : sum_g;
                        if (pp.bnf_dw) pp.bnf_dw[c] = pp.bnf_accum ? pp.bnf_dw[c] + sum_gx : sum_gx;
                    }
                    if (have_bsc) {
                        const __amdgpu_buffer_rsrc_t zr2 = __builtin_amdgcn_make_buffer_rsrc(pp.bsc.acc, 0, (int)(2 * NSl * pp.Co * 8), 0x00020000);
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            av[s] = s < NSl ? ld_total(zr2, ((2 * s) * pp.Co + c) * 8) : 0.0;
                            bv[s] = s < NSl ? ld_total(zr2, ((2 * s + 1) * pp.Co + c) * 8) : 0.0;
                        }
                        double a2 = 0.0, b2 = 0.0;
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            a2 += av[s];
                            b2 += bv[s];
                        }
                        const float is2 = pp.bsc.stats[pp.Co + c], alpha2 = pp.bsc.stats[2 * pp.Co + c];
                        const float sg2 = (float)a2;
                        const float sgx2 = (float)b2 * is2;
                        cf[2 * BN + tid] = -alpha2 * is2 * (float)((double)sgx2 * pp.bnf_inv_m);
                        cf[3 * BN + tid] = -alpha2 * (float)((double)sg2 * pp.bnf_inv_m);
                        if (pub) {
                            if (pp.bsc.db) pp.bsc.db[c] = pp.bnf_accum ? pp.bsc.db[c] + sg2 : sg2;
                            if (pp.bsc.dw) pp.bsc.dw[c] = pp.bnf_accum ? pp.bsc.dw[c] + sgx2 : sgx2;
                        }
                    }
                }
            }
            __syncthreads();
            float k0[8], k1[8], k2[8], k3[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                k0[j] = cf[pc * 8 + j];
                k1[j] = cf[BN + pc * 8 + j];
                k2[j] = (have_sc || have_bsc) ? cf[2 * BN + pc * 8 + j] : 0.f;
                k3[j] = (have_sc || have_bsc) ? cf[3 * BN + pc * 8 + j] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < EPI_ROWS; ++q) {
                const int r = pr + q * ROWS_PER_PASS;
                const int off = epi_on ? out_off[r] : -1;
                if (off >= 0 && ch_ok) {
                    u16x8 v = *reinterpret_cast<const u16x8*>(C + r * LDC + pc * 8);
                    const int64_t go = (int64_t)off + n0 + pc * 8;
                    if (bf_fwd) {
                        const u16x8 rv = __builtin_bit_cast(u16x8, pre_a[q]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float t = fmaf(bf2f(v[j]), k0[j], k1[j]);
                            if (have_sc) t = t + fmaf(bf2f(rv[j]), k2[j], k3[j]);
                            else if (pp.bnf_res) t += bf2f(rv[j]);
                            if (pp.bnf_relu) t = (t > 0.f) ? t : ((t != t) ? t : 0.f);
                            v[j] = f2bf(t);
                        }
                        *reinterpret_cast<u16x8*>(pp.y2 + go) = v;
                    } else {
                        const u16x8 xv = __builtin_bit_cast(u16x8, pre_x[q]);
                        u16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            o[j] = f2bf(fmaf(bf2f(v[j]), al[j], fmaf(bf2f(xv[j]) - sh[j], k0[j], k1[j])));
                        *reinterpret_cast<u16x8*>(y_p + go) = o;
                        if (!DRES_EARLY && pp.y2) *reinterpret_cast<u16x8*>(pp.y2 + go) = v;
                        if (have_bsc) {
                            const u16x8 zv = *reinterpret_cast<const u16x8*>(Z + r * BN + pc * 8);
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                o[j] = f2bf(fmaf(bf2f(v[j]), pp.bsc.stats[2 * pp.Co + n0 + pc * 8 + j], fmaf(bf2f(zv[j]) - zmean_s[pc * 8 + j], k2[j], k3[j])));
                            *reinterpret_cast<u16x8*>(pp.bsc.y3 + go) = o;
                        }
                    }
                }
            }
            AFAN_PHASE(7);
        }
    }
    }   // group passes
    AFAN_PHASE(8);
}

// The one body under two entry points, so that a kernel trace (rocprofv3 --kernel-trace --stats) separates the forward
// launches from the input-gradient launches: bench.py's roofline names whichever is the larger and profiles/ can be
// checked against it symbol by symbol.
template <int BM, int BN, int PF, int WM, int WN, int PW = 0, int FBT = AFAN_CONV_FRAG_BATCH, int HL = 0, bool GS = false, bool BF = false>
__global__ __launch_bounds__(64 * (WM * WN + PW)) void conv_igemm_fwd_kernel(const ConvP pp) {
    conv_igemm_body<BM, BN, PF, WM, WN, PW, FBT, HL, GS, BF>(pp);
}
template <int BM, int BN, int PF, int WM, int WN, int PW = 0, int FBT = AFAN_CONV_FRAG_BATCH, int HL = 0, bool GS = false, bool BF = false>
__global__ __launch_bounds__(64 * (WM * WN + PW)) void conv_igemm_dgrad_kernel(const ConvP pp) {
    conv_igemm_body<BM, BN, PF, WM, WN, PW, FBT, HL, GS, BF>(pp);
}

static int64_t max_rows(const ConvP& p) {
    int64_t m = 0;
    for (int c = 0; c < p.n_classes; ++c) {
        const int64_t v = (int64_t)p.N * p.cls[c].Hg * p.cls[c].Wg;
        if (v > m) m = v;
    }
    return m;
}

// ConvP::coef: the rows the kernel's prologue prefetches for its epilogue — a fused BatchNorm backward's mean | alpha | beta
// (bn_stats rows 0, 2, 3; the second image group's block 4 * Co further), else the sums' shift in slot 0 and a frozen BatchNorm's
// alpha | beta (forward) or alpha (backward) in slots 1 | 2
inline void fill_coef(ConvP& p) {
    const float* const dummy = reinterpret_cast<const float*>(p.w);
    const bool ws = p.stats != nullptr || p.acc != nullptr, bb = ws && p.bnx != nullptr;
    p.coef[0] = p.coef[1] = p.coef[2] = dummy;
    p.coef_mask = 0;
    p.coef_gofs = 0;
    if (bb) {
        if (!p.bn_stats) return;
        p.coef[0] = p.bn_stats;
        p.coef[1] = p.bn_stats + 2 * (int64_t)p.Co;
        p.coef[2] = p.bn_stats + 3 * (int64_t)p.Co;
        p.coef_mask = 7;
        p.coef_gofs = p.groups == 2 ? 4 * p.Co : 0;
        return;
    }
    if (ws && p.shift) {
        p.coef[0] = p.shift;
        p.coef_mask |= 1 | 8;
    }
    if (p.aff) {
        p.coef[1] = p.aff + (p.aff_bwd ? 0 : 2 * (int64_t)p.Co);
        p.coef_mask |= 2;
        if (!p.aff_bwd) {
            p.coef[2] = p.aff + 3 * (int64_t)p.Co;
            p.coef_mask |= 4;
        }
    }
}

template <int BM, int BN, int PF, int WM, int WN, int PW, int FBT, int HL, bool GS, bool BF = false>
int launch_gs(const ConvP& p_in, hipStream_t st, bool dgrad) {
    ConvP p = p_in;
    fill_coef(p);
    constexpr int THREADS = 64 * (WM * WN + PW);
    const int64_t M = max_rows(p);
    dim3 grid((unsigned)((p.Co + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), (unsigned)p.n_classes);
    // halo form: (PF - 1) weight stages + two halo buffers of HL pixels + the 1 KiB landing zone of the padding DMAs
    constexpr size_t stage_bytes = HL ? (size_t)(PF - 1) * BN * BK * 2 + (size_t)2 * HL * BK * 2 + 1024
                                      : (size_t)(PF <= 3 ? 2 : PF - 1) * (BM + BN) * (PF >= 3 ? BK : LDK) * 2;
    constexpr size_t epi_bytes = (size_t)BM * (BN + 8) * 2;
    constexpr size_t lds = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    static bool attr_done = false;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_fwd_kernel<BM, BN, PF, WM, WN, PW, FBT, HL, GS, BF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)conv_igemm_dgrad_kernel<BM, BN, PF, WM, WN, PW, FBT, HL, GS, BF>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if constexpr (BF) {
        // the in-launch BatchNorm's grid barrier needs EVERY workgroup of the launch resident at once: CUs x the occupancy of
        // this instantiation (the smaller of the two entry points'), asked once; a launch beyond it is refused before anything runs
        static int max_resident = -1, n_cus = 0;
        if (max_resident < 0) {
            int dev = 0, cus = 0, of = 0, od = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&of, (const void*)conv_igemm_fwd_kernel<BM, BN, PF, WM, WN, PW, FBT, HL, GS, BF>, THREADS, lds) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&od, (const void*)conv_igemm_dgrad_kernel<BM, BN, PF, WM, WN, PW, FBT, HL, GS, BF>, THREADS, lds) != hipSuccess)
                return AFAN_ESHAPE;
            n_cus = cus;
            max_resident = cus * (of < od ? of : od);
        }
        const int64_t cap = afan_conv::g_bnf_one_per_cu ? (max_resident < n_cus ? max_resident : n_cus) : max_resident;
        if (p.acc_ns > 8) return AFAN_ESHAPE;                        // (the epilogue folds at most 8 accumulator copies)
        if (HL == 0 && p.bsc.x) return AFAN_ESHAPE;                  // (the projection BatchNorm's backward: halo-form launches only)
        // (multi-problem forward: only problem 0's workgroups meet at the barrier, and they are dispatched first — z is the slowest grid
        // dimension — so they alone must fit; the other problems' workgroups wait for nothing)
        const int64_t meet = (int64_t)grid.x * grid.y * (p.multi ? 1 : grid.z);
        if (meet > cap || meet < 8 || !p.bar || !p.acc || p.groups != 1 || (p.multi && (p.bnf != 1 || p.bnf_sc.acc || p.bnf_res)))
            return AFAN_ESHAPE;
        // several output-parity classes (a stride-2 input gradient): one set of sums over all of them; every tile of the grid must
        // be a real one (a workgroup of a smaller class would leave at the top, before the barrier): classes of equal size only
        for (int z = 1; z < p.n_classes; ++z)
            if (p.cls[z].Hg != p.cls[0].Hg || p.cls[z].Wg != p.cls[0].Wg) return AFAN_ESHAPE;
        if (p.n_classes != 1 && !p.multi && (p.bnf != 2 || p.bsc.x)) return AFAN_ESHAPE;
    }
    if (dgrad) conv_igemm_dgrad_kernel<BM, BN, PF, WM, WN, PW, FBT, HL, GS, BF><<<grid, THREADS, lds, st>>>(p);
    else conv_igemm_fwd_kernel<BM, BN, PF, WM, WN, PW, FBT, HL, GS, BF><<<grid, THREADS, lds, st>>>(p);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

template <int BM, int BN, int PF, int WM, int WN, int PW = 0, int FBT = AFAN_CONV_FRAG_BATCH, int HL = 0>
int launch(const ConvP& p, hipStream_t st, bool dgrad) {
    // image groups whose halves are not whole row tiles of THIS variant: the instantiation whose epilogue walks the straddling
    // tile once per group (not built for the 256-row and the 16-wave variants: no register to spare; dispatch() never sends them
    // such a launch — its halo256 / 256 x 64 guards ask for whole 256-row tiles per half)
    bool gs = false;
    if (p.groups == 2)
        for (int z = 0; z < p.n_classes; ++z) gs = gs || ((int64_t)(p.N / 2) * p.cls[z].Hg * p.cls[z].Wg) % BM != 0;
    constexpr bool GS_BUILT = BM <= 128 && WM * WN + PW <= 12;
    if (gs) {
        if constexpr (GS_BUILT) return launch_gs<BM, BN, PF, WM, WN, PW, FBT, HL, true>(p, st, dgrad);
        else return AFAN_ESHAPE;
    }
    return launch_gs<BM, BN, PF, WM, WN, PW, FBT, HL, false>(p, st, dgrad);
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// tile choice: fill >= ~256 workgroups when the problem allows it
int choose_bm(int64_t M, int co, int n_classes, bool equal_classes = false) {
    const bool n128 = (co % 128 == 0);
    const int64_t wg_128 = ((M + 127) / 128) * ((co + (n128 ? 127 : 63)) / (n128 ? 128 : 64)) * n_classes;
    static const int thr = env_int("AFAN_CONV_THR128", 256);   // (256 vs 384: +0.3 % of the step)
    if (wg_128 >= thr) return 128;
    // 385..768 workgroups of 64 rows (the 8x8 stage) would be two per CU on the two-stage pipeline; 128-row tiles bring
    // the launch back to one workgroup per CU on the four-stage one (measured +3.6 % step rate)
    static const int tall = env_int("AFAN_CONV_TALL", 1);
    const int64_t wg_64 = ((M + 63) / 64) * ((co + (n128 ? 127 : 63)) / (n128 ? 128 : 64)) * n_classes;
    // (round 6) from 257 on: 257..384 workgroups of 64 rows ran the four-stage form, ONE workgroup per CU (108 KB of LDS), in two rounds —
    // Faster-RCNN's layer3 256 -> 1024 at 38 x 57 (272 workgroups): 12.3 us for four K-steps, workgroup 0 done after 5.6
    // (profiles/r06_det_conv_shapes.txt); 128-row tiles are one round.  DeepLab 19.08 -> 18.77 ms with AFAN_CONV_DEEPMAX's new default
    static const int tall_lo = env_int("AFAN_CONV_TALL_LO", 256);
    // (equal_classes: independent problems of ONE size in the grid's z — ASPP's three atrous branches; at four 513 x 513 images they
    // are 414 workgroups of 64 rows on the two-stage form, 468 us; 210 of 128 rows on the four-stage form)
    static const int tall_multi = env_int("AFAN_CONV_TALL_MULTI", 1);
    if (tall && n128 && (n_classes == 1 || (equal_classes && tall_multi)) && wg_64 > tall_lo && wg_64 <= 768) return 128;
    return 64;
}


// LDS-resident halo (conv_igemm_body's HL form): one dense 3x3 / stride 1 problem on whole 64-channel chunks whose every
// row tile's halo — the contiguous range of the zero-padded raster between its first pixel's top-left tap and its last
// pixel's bottom-right tap — fits the buffer
constexpr int HALO_PIXELS = 320;       // 128- and 64-row tiles
constexpr int HALO_PIXELS_256 = 328;   // 256-row tiles: one 16x16 image with its border is 18 x 18 = 324 pixels
constexpr int HALO_PIXELS_256N = 400;  // 256-row x 64-channel tiles (8 KB weight stages leave the room): four 8x8 images, 4 x 100
static bool halo_ok(const ConvP& p, int bm, int cap = HALO_PIXELS) {
    const ConvClass& c = p.cls[0];
    if (p.n_classes != 1 || p.multi || p.a_extra || p.in_s != 1 || p.out_s != 1 || p.max_pad != 1 || c.T != 9) return false;
    if (c.Hg != p.Hi || c.Wg != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi || p.Ci % 64 || p.Co % 128) return false;
    if (c.out_h0 || c.out_w0) return false;
    for (int t = 0; t < 9; ++t)
        if (c.dh[t] < -1 || c.dh[t] > 1 || c.dw[t] < -1 || c.dw[t] > 1 || c.aofs[t]) return false;
    const int64_t H = p.Hi, W = p.Wi, M = (int64_t)p.N * H * W;
    // the geometry test walks the row tiles (a few hundred integer divisions): remembered per thread for the shapes a
    // network keeps asking about (an eager iteration of Faster-RCNN dispatches ~3 000 such convolutions)
    struct Seen { int64_t n, h, w; int bm, cap; bool ok; };
    constexpr int NSEEN = 32;
    thread_local Seen seen[NSEEN];
    thread_local int n_seen = 0, next_seen = 0;
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].n == p.N && seen[i].h == H && seen[i].w == W && seen[i].bm == bm && seen[i].cap == cap) return seen[i].ok;
    auto p0 = [&](int64_t m) { const int64_t t1 = m / W, x = m % W, n = t1 / H, y = t1 % H; return (n * (H + 2) + y) * (W + 2) + x; };
    const int64_t tiles = (M + bm - 1) / bm, walk = tiles < 4096 ? tiles : 4096;   // (callers come with <= 768 tiles)
    bool ok = tiles <= 4096;
    for (int64_t i = 0; ok && i < walk; ++i) {
        const int64_t m0 = i * bm, m1 = (m0 + bm < M ? m0 + bm : M) - 1;
        if (p0(m1) + 2 * (W + 2) + 2 - p0(m0) + 1 > cap) ok = false;
    }
    seen[next_seen] = Seen{p.N, H, W, bm, cap, ok};
    next_seen = (next_seen + 1) % NSEEN;
    if (n_seen < NSEEN) ++n_seen;
    return ok;
}

#ifndef AFAN_CONV_BNF_TU      // (afan_conv_bnf.hip includes everything above for its own instantiations)
int dispatch(const ConvP& p, hipStream_t st, bool dgrad) {
    // staging variant: 3 = LDS-DMA (global_load_lds), 1 = global -> VGPR -> LDS, 2 = same with two register sets
    static const int mode = env_int("AFAN_CONV_MODE", 3);
    static const int force_bm = env_int("AFAN_CONV_BM", 0);   // tuning knobs (tools/conv_bench.py A/B)
    static const int nw = env_int("AFAN_CONV_NW", 8);          // waves per workgroup where the tile allows it
    const int bm = force_bm ? force_bm : choose_bm(max_rows(p), p.Co, p.n_classes, p.multi != 0);
    const bool n128 = p.Co % 128 == 0;
    // launches of about one workgroup per CU (8x8 / 4x4 stages): deep DMA pipeline (3 or 4 LDS stages, counted vmcnt)
    // In the training step every layer's weights are cold (44 MB of bf16 weights cycle through a 32 MB L2 between two
    // uses), so each K-step's weight tile is a first-touch miss that all row tiles of the launch wait for together:
    // measured 70 us in the step against 37 us back-to-back for the 512-channel layers.  Tiles in flight hide that.
    // Only for launches of about one workgroup per CU: with two or more resident workgroups the second one already
    // covers the first one's wait, and the deeper pipeline measured slower (8x8 stage, 512 workgroups: -5 % step rate).
    static const int deep = env_int("AFAN_CONV_DEEP", 1);     // 0: two LDS stages everywhere
    if (mode == 3 && nw >= 8 && n128 && deep >= 1) {
        const int64_t wgs = (int64_t)(p.Co / 128) * ((max_rows(p) + bm - 1) / bm) * p.n_classes;
        // (round 6: 384 -> 256.  One workgroup per CU: a launch of 257..384 is two rounds of the four-stage form; the two-stage forms hold
        // two workgroups per CU and finish in one — Faster-RCNN's layer2 128 -> 512 at 75 x 113, 268 workgroups: 14.0 us.  Its convolution
        // kernel time per iteration 32.9 -> 30.9 ms together with AFAN_CONV_TALL_LO, same box, A/B twice)
        static const int deep_max = env_int("AFAN_CONV_DEEPMAX", 256);
        static const int spec = env_int("AFAN_CONV_SPEC", 1);   // 1: four producer waves + four 64x64 (32x64) MFMA waves
        static const int halo = env_int("AFAN_CONV_HALO", 1);   // 0: per-tap operand tiles everywhere (A/B)
        // (round 6) launches that leave half the chip idle — Faster-RCNN's layer3 at one 600 x 904 image: 2 166 rows, 1024 -> 256 on 68
        // workgroups of 64 x 128, 256 -> 256 3x3 on 68 of 128 x 64 — run at ONE CU's LDS-DMA intake per workgroup (24.6 KB per
        // 64-channel step of a 64 x 128 tile: ~770 cycles of the ~850 measured per step, profiles/r06_det_conv_shapes.txt) while
        // three CUs in four do nothing: 64 x 64 tiles are twice the workgroups at 16 KB per step, the same K order (same bits)
        static const int t64 = env_int("AFAN_CONV_T64", 1);
        // (with BatchNorm sums too: dispatch_bnf() has the same rule — a launch pair must add a column's rows in the order of the fused
        // launch it stands in for; DeepLab's layer3 at two 513 x 513 images: 1024 -> 256 on 70, 256 -> 256 3x3 on 72 workgroups)
        if (t64 && spec && !p.stats && p.n_classes == 1 && p.Co % 64 == 0 && p.Ci >= 128) {
            const int64_t w64 = (int64_t)(p.Co / 64) * ((max_rows(p) + 63) / 64);
            if (w64 <= 256 && w64 >= 16)
            {
                static const int ahead64 = env_int("AFAN_CONV_AHEAD64", 1);
                if (halo && halo_ok(p, 64))
                    return ahead64 ? launch<64, 64, 7, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS>(p, st, dgrad)
                                   : launch<64, 64, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS>(p, st, dgrad);
                return launch<64, 64, 5, 2, 2, 4>(p, st, dgrad);
            }
        }
        // 385..768 workgroups of 128 rows (ResNet-18's 16x16 stage: two per CU on the two-stage kernel): 256-row tiles
        // bring the launch to one workgroup per CU on the halo form — eight MFMA waves + four producers, a whole 16x16
        // image and its border per tile (the partial-slab statistics' slot count is tied to choose_bm(): not with those)
        static const int halo256 = env_int("AFAN_CONV_HALO256", 1);
        if (halo && halo256 && spec && bm == 128 && wgs > deep_max && wgs <= 2 * deep_max && !p.stats
            && (p.groups != 2 || ((int64_t)(p.N / 2) * p.cls[0].Hg * p.cls[0].Wg) % 256 == 0) && halo_ok(p, 256, HALO_PIXELS_256))
            return launch<256, 128, 5, 4, 2, 4, 2, HALO_PIXELS_256>(p, st, dgrad);   // (two k16-slices of fragments in flight: 168 registers per wave at three waves per SIMD)
        // 64-row launches whose weights dominate the L2 traffic (ResNet-18's 4x4 stage: 64 row tiles each stream the 4.7 MB of
        // a 512 -> 512 layer's weights, 300 MB of L2 requests per launch): the same workgroup count as 128-row x 64-channel
        // tiles — half as many row tiles read the weights, each tile's halo is read by twice as many channel tiles (it is
        // the small operand) — a third less L2 traffic per launch
        static const int n64 = env_int("AFAN_CONV_HALO_N64", 1);
        static const int ahead = env_int("AFAN_CONV_AHEAD", 1);    // 1: the 64-column halo tiles one tap ahead on six weight buffers (PF = 7); 0: the four-buffer form (A/B)
        if (halo && n64 && spec && bm == 64 && !p.stats && p.Co % 64 == 0 && p.Ci >= 256) {
            const int64_t wgs_n64 = (int64_t)(p.Co / 64) * ((max_rows(p) + 127) / 128) * p.n_classes;
            if (wgs_n64 <= deep_max && wgs_n64 >= wgs && halo_ok(p, 128))
                return ahead ? launch<128, 64, 7, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS>(p, st, dgrad)
                             : launch<128, 64, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS>(p, st, dgrad);
        }
        // and the 128-row launches of the same kind (the 8x8 stage) as 256-row x 64-channel tiles: 4 images and their borders
        // per tile (400 pixels), 8 KB weight stages
        if (halo && n64 && spec && bm == 128 && wgs <= deep_max && !p.stats && p.Co % 64 == 0 && p.Ci >= 256 &&
            (p.groups != 2 || ((int64_t)(p.N / 2) * p.cls[0].Hg * p.cls[0].Wg) % 256 == 0)) {
            const int64_t wgs_n64 = (int64_t)(p.Co / 64) * ((max_rows(p) + 255) / 256) * p.n_classes;
            // (four MFMA waves of 64 x 64 instead of eight of 64 x 32: the eight re-read 96 KB of fragments per tap from LDS for 40 KB of
            // tile — 768 cycles of the LDS port against 512 of the MFMA pipe; four read 64 KB.  ResNet-18 step 7.98 -> 7.94 ms, same box)
            static const int w41 = env_int("AFAN_CONV_W41", 1);
            if (wgs_n64 <= deep_max && wgs_n64 >= wgs && halo_ok(p, 256, HALO_PIXELS_256N))
                return !w41 ? launch<256, 64, 5, 4, 2, 4, 2, HALO_PIXELS_256N>(p, st, dgrad)
                            : (ahead ? launch<256, 64, 7, 4, 1, 4, 2, HALO_PIXELS_256N>(p, st, dgrad) : launch<256, 64, 5, 4, 1, 4, 2, HALO_PIXELS_256N>(p, st, dgrad));
        }
        if (wgs <= deep_max && spec && halo && halo_ok(p, bm))
            return bm == 64 ? launch<64, 128, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS>(p, st, dgrad)
                            : launch<128, 128, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS>(p, st, dgrad);
        if (wgs <= deep_max && spec) return bm == 64 ? launch<64, 128, 5, 2, 2, 4>(p, st, dgrad) : launch<128, 128, 5, 2, 2, 4>(p, st, dgrad);
        if (wgs <= deep_max) return bm == 64 ? launch<64, 128, 5, 2, 4>(p, st, dgrad) : launch<128, 128, 5, 2, 4>(p, st, dgrad);   // 4 stages
    }
    if (mode == 3 && nw == 16 && n128 && bm == 128) return launch<128, 128, 3, 4, 4>(p, st, dgrad);
    if (mode == 3 && nw >= 8) {
        if (n128) return bm == 128 ? launch<128, 128, 3, 2, 4>(p, st, dgrad) : launch<64, 128, 3, 2, 4>(p, st, dgrad);
        if (bm == 128) return launch<128, 64, 3, 4, 2>(p, st, dgrad);
    }
#define AFAN_CONV_GO(M)                                                                                      \
    do {                                                                                                     \
        if (n128) return bm == 128 ? launch<128, 128, M, 2, 2>(p, st, dgrad) : launch<64, 128, M, 2, 2>(p, st, dgrad);     \
        return bm == 128 ? launch<128, 64, M, 2, 2>(p, st, dgrad) : launch<64, 64, M, 2, 2>(p, st, dgrad);                 \
    } while (0)
    if (mode == 1) AFAN_CONV_GO(1);
    if (mode == 2) AFAN_CONV_GO(2);
    AFAN_CONV_GO(3);
#undef AFAN_CONV_GO
}

// groups = 2: the sums go to per-group f64 accumulator blocks (the slab form has no group dimension)
static bool small_groups_ok(const ConvP& p) {
    return p.groups != 2 || ((int64_t)(p.N / 2) * p.cls[0].Hg * p.cls[0].Wg) % 128 == 0;
}
int set_groups(ConvP& p, int groups, int64_t n, int64_t positions_per_image, int64_t channels, bool have_acc) {
    p.groups = 1;
    p.acc_stride = 0;
    if (groups <= 1) return AFAN_OK;
    if (groups != 2 || (n & 1) || !have_acc) return AFAN_ESHAPE;
    // (the tiled kernel takes half-batches of any size: its one straddling tile sums per row; the small-channel kernel walks whole
    // 128-row tiles per group: small_groups_ok)
    p.groups = 2;
    p.acc_stride = (int)((afan_nhwc::acc_doubles(channels) + 1) & ~(int64_t)1);
    return AFAN_OK;
}

// reduction channels `ci` / output channels `co` of the GEMM: multiples of 64 (the tiled kernels), or the small-channel
// kernel's set (afan_conv_small.hip): ci in {16, 32, 64}, co a multiple of 16 up to 64
bool channels_ok(int64_t ci, int64_t co) {
    if (ci % 8 == 0 && co % 8 == 0 && ci >= 40 && co >= 40) return true;     // tiled kernel (ragged last chunk / tile masked)
    return (ci == 16 || ci == 32 || ci == 64) && co % 16 == 0 && co <= 64;
}

int check_dims(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride, int dilation = 1) {
    if (n <= 0 || hi <= 0 || wi <= 0 || ci <= 0 || co <= 0) return AFAN_ESHAPE;
    if (!channels_ok(ci, co)) return AFAN_ESHAPE;                // caller falls back for the 3-channel stem
    if (!(k == 1 || k == 3) || !(stride == 1 || stride == 2)) return AFAN_ESHAPE;
    if (dilation < 1 || dilation > 64 || (dilation > 1 && (stride != 1 || k != 3))) return AFAN_ESHAPE;   // atrous: 3x3, stride 1
    if (n * hi * wi * (ci > co ? ci : co) * 2 + 2 * dilation * (wi + 1) * (ci > co ? ci : co) > 0x7fffffffLL) return AFAN_ESHAPE;  // 32-bit byte offsets / buffer descriptors (+ the DMA descriptor's bias)
    return AFAN_OK;
}

static int dgrad_impl(const void* dy, const void* dy_sc, const void* wt, void* dx, int64_t n, int64_t hi, int64_t wi,
                      int64_t ci, int64_t co, int k, int stride, int dilation, const void* addend, const void* bn_x,
                      const float* bn_stats, int bn_relu, const void* bn_y, float* bn_partials, double* bn_acc, int groups,
                      afan_stream_t stream, const float* aff_alpha = nullptr, const void* aff_act = nullptr, void* dx2 = nullptr,
                      const ConvP* bnf = nullptr) {
    int e = check_dims(n, hi, wi, co, ci, k, stride, dilation);   // reduction runs over co here
    if (e) return e;
    if (!dy || !wt || !dx) return AFAN_ENULL;
    if (!aligned(dy, 16) || !aligned(wt, 16) || !aligned(dx, 16)) return AFAN_EALIGN;
    const int pad = k / 2;
    const int ho = (int)((hi + 2 * pad - k) / stride + 1), wo = (int)((wi + 2 * pad - k) / stride + 1);
    hipStream_t st = (hipStream_t)stream;
    ConvP p{};
    p.x = (const uint16_t*)dy; p.w = (const uint16_t*)wt; p.y = (uint16_t*)dx;
    p.N = (int)n; p.Hi = ho; p.Wi = wo; p.Ci = (int)co;      // GEMM input = dy
    p.Ho = (int)hi; p.Wo = (int)wi; p.Co = (int)ci;          // GEMM output = dx
    p.w_row_stride = (int)(k * k * co);
    int64_t sc_off = 0;                                      // elements from dy to dy_sc
    if (dy_sc) {
        if (k != 3 || stride != 2 || dilation != 1 || groups > 1 || co % 8 || ci % 8 || co < 40 || ci < 40) return AFAN_ESHAPE;
        if (!aligned(dy_sc, 16)) return AFAN_EALIGN;
        sc_off = (const uint16_t*)dy_sc - (const uint16_t*)dy;
        const int64_t tensor = n * ho * wo * co;
        if (sc_off < tensor || (sc_off + tensor) * 2 > 0x7fffffffLL) return AFAN_ESHAPE;   // behind dy, inside 32-bit byte offsets
        p.w_row_stride = (int)(10 * co);
        p.a_extra = sc_off * 2;
    }
    p.max_pad = dilation;
    p.addend = (const uint16_t*)addend;
    if (aff_alpha) {                                         // the layer in front's frozen BatchNorm + ReLU backward in the epilogue
        if (!aff_act) return AFAN_ENULL;
        if ((addend && !dx2) || bn_partials || bn_acc || dy_sc || groups > 1 || !aligned(aff_act, 16)) return AFAN_ESHAPE;
        if (dx2 && (!aligned(dx2, 16) || (addend && !aligned(addend, 16)))) return AFAN_EALIGN;
        if (afan_c64::eligible(n, hi, wi, co, ci, k, stride)) return AFAN_ESHAPE;      // (that kernel's epilogue has no such form)
        p.aff = aff_alpha; p.aff_res = (const uint16_t*)aff_act; p.aff_bwd = dx2 ? 2 : 1; p.y2 = (uint16_t*)dx2;
    }
    if (bn_partials && bn_acc) return AFAN_ESHAPE;
    if (bn_partials || bn_acc) {
        if (!bn_x || !bn_stats) return AFAN_ENULL;
        if (bn_acc && !aligned(bn_acc, 16)) return AFAN_EALIGN;
        // image groups: every parity class of a stride-2 launch has at least (hi/2)*(wi/2) positions per image, a
        // multiple of that class's own count is what matters; the smallest class decides
        if ((e = set_groups(p, groups, n, stride == 1 ? hi * wi : (hi / 2) * (wi / 2), ci, bn_acc != nullptr))) return e;
        if (groups == 2 && stride == 2 && ((hi | wi) & 1)) return AFAN_ESHAPE;
        p.stats = bn_partials; p.acc = bn_acc; p.acc_ns = afan_nhwc::acc_slot_count(ci);
        p.bnx = (const uint16_t*)bn_x; p.bn_stats = bn_stats; p.bn_relu = bn_relu; p.bny = (const uint16_t*)bn_y;
    }
    if (bnf) {      // the BatchNorm backward itself behind a grid barrier in this launch (afan_conv_dgrad_bn_nhwc_bf16)
        // stride 1 (3x3 incl. atrous, 1x1), or the stride-2 pair form (a block's first 3x3 + its projection, four output-parity classes)
        if (!bn_acc || (k != 3 && k != 1) || aff_alpha || groups > 1) return AFAN_ESHAPE;
        if (!((stride == 1 && !dy_sc) || (stride == 2 && dy_sc && k == 3 && dilation == 1 && !((hi | wi) & 1)))) return AFAN_ESHAPE;
        if (afan_c64::eligible(n, hi, wi, co, ci, k, stride)) return AFAN_ESHAPE;          // (that kernel has no such epilogue)
        p.bnf = 2; p.bar = bnf->bar; p.y2 = bnf->y2; p.bnf_dw = bnf->bnf_dw; p.bnf_db = bnf->bnf_db; p.bnf_accum = bnf->bnf_accum;
        p.bsc = bnf->bsc;
        p.bnf_inv_m = 1.0 / ((double)n * hi * wi);
    }
    const double bytes = 2.0 * ((double)n * ho * wo * co + (double)n * hi * wi * ci + (double)co * k * k * ci);
    AFAN_PROF_FLOPS(bnf ? "conv_bn_dgrad_kernel" : "conv_igemm_dgrad_kernel", bytes + (dy_sc ? 2.0 * ((double)n * ho * wo * co + (double)co * ci) : 0.0),
                    2.0 * (double)n * ho * wo * co * (k * k + (dy_sc ? 1 : 0)) * ci, st);
    p.in_s = 1;
    if (!dy_sc && !bn_partials && groups <= 1 && dilation == 1 && afan_c64::eligible(n, hi, wi, co, ci, k, stride)) {
        afan_c64::Params q{};
        q.x = p.x; q.w = p.w; q.y = p.y; q.N = p.N; q.H = (int)hi; q.W = (int)wi; q.flip = 1;
        q.acc = bn_acc; q.acc_ns = p.acc_ns; q.bnx = p.bnx; q.bn_stats = p.bn_stats; q.bn_relu = p.bn_relu; q.bny = p.bny;
        q.addend = p.addend;
        return afan_c64::launch(q, st);
    }
    if (stride == 1) {
        // dx[h,w] = sum_{r,s} dy[h + pad - r, w + pad - s] * w[., r, s, .]
        p.out_s = 1; p.n_classes = 1;
        ConvClass& c0 = p.cls[0];
        c0.Hg = (int)hi; c0.Wg = (int)wi; c0.out_h0 = 0; c0.out_w0 = 0; c0.T = k * k;
        for (int r = 0; r < k; ++r)
            for (int s = 0; s < k; ++s) {
                const int t = r * k + s;
                c0.dh[t] = (pad - r) * dilation; c0.dw[t] = (pad - s) * dilation; c0.wofs[t] = (int)(t * co);
            }
        if (bnf) return small_eligible(p) ? AFAN_ESHAPE : dispatch_bnf(p, st, true);
        if (small_eligible(p)) return (p.aff || !small_groups_ok(p)) ? AFAN_ESHAPE : small_launch(p, st);
        if (co % 8 != 0 || ci % 8 != 0 || co < 40 || ci < 40) return AFAN_ESHAPE;   // (small shape asked for the partial-slab sums)
        return dispatch(p, st, true);
    }
    // stride 2: output pixel (2h'+ph, 2w'+pw) receives tap (r,s) iff (ph + pad - r) and (pw + pad - s) are even;
    // then the dy pixel is (h' + (ph+pad-r)/2, w' + (pw+pad-s)/2).  Four classes (1/2/2/4 taps for k = 3) in ONE launch.
    p.out_s = 2;
    int nc = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            ConvClass& c = p.cls[nc];
            c.Hg = (int)((hi - ph + 1) / 2); c.Wg = (int)((wi - pw + 1) / 2);
            if (c.Hg <= 0 || c.Wg <= 0) continue;
            c.out_h0 = ph; c.out_w0 = pw;
            int T = 0;
            for (int r = 0; r < k; ++r)
                for (int s = 0; s < k; ++s) {
                    const int a = ph + pad - r, b = pw + pad - s;
                    if ((a & 1) || (b & 1)) continue;
                    c.dh[T] = a / 2; c.dw[T] = b / 2; c.wofs[T] = (int)((r * k + s) * co);
                    ++T;
                }
            if (dy_sc && ph == 0 && pw == 0) {               // the 1x1 / 2 projection: one more tap of the even/even class
                c.dh[T] = 0; c.dw[T] = 0; c.wofs[T] = (int)(9 * co); c.aofs[T] = (int)sc_off;
                ++T;
            }
            if (T == 0) {
                // no tap reaches this parity class (1x1 stride 2): its gradient is zero.  One all-invalid tap makes
                // the kernel write zeros through its normal path.
                T = 1; c.dh[0] = -(ho + 2); c.dw[0] = 0; c.wofs[0] = 0;   // (small enough for the 32-bit tap offset)
            }
            c.T = T;
            ++nc;
        }
    p.n_classes = nc;
    if (bnf) return nc == 4 ? dispatch_bnf(p, st, true) : AFAN_ESHAPE;      // (even sizes: four classes of equal size, no idle tiles)
    if (!dy_sc && small_eligible(p)) return (p.aff || !small_groups_ok(p)) ? AFAN_ESHAPE : small_launch(p, st);
    if (co % 8 != 0 || ci % 8 != 0 || co < 40 || ci < 40) return AFAN_ESHAPE;
    return dispatch(p, st, true);
}

}  // namespace

extern "C" {

#ifdef AFAN_CONV_STAMP
// diagnostic build only: the last halo-form launch's per-tap stamps, [2 roles][96 taps][3] cycle counters
int afan_conv_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(afan_stamps), sizeof(unsigned long long) * 2 * 96 * 3);
}
#endif

int afan_conv_supported(int64_t ci, int64_t co, int k, int stride) {
    if (ci == 3) return afan_stem::eligible(1, 1, 32, ci, co, k, stride) ? 1 : 0;   // image stem (forward only; width % 32 == 0)
    // forward needs channels_ok(ci, co), the input gradient channels_ok(co, ci): both must hold for a layer to be taken
    return (channels_ok(ci, co) && channels_ok(co, ci) && (k == 1 || k == 3) && (stride == 1 || stride == 2)) ? 1 : 0;
}

// number of row tiles (= BN-statistics partials per channel) the forward launch of this problem uses
int64_t afan_conv_fwd_tiles(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride) {
    if (check_dims(n, hi, wi, ci, co, k, stride)) return 0;
    const int pad = k / 2;     // (a dilated 3x3 at stride 1 keeps the spatial size, like the plain one)
    const int64_t M = n * ((hi + 2 * pad - k) / stride + 1) * ((wi + 2 * pad - k) / stride + 1);
    const int bm = choose_bm(M, (int)co, 1);
    return (M + bm - 1) / bm;
}

// y[N,Ho,Wo,Co] = conv(x[N,Hi,Wi,Ci], w[Co,k,k,Ci]) with padding dilation*(k/2), stride 1 or 2 (dilation > 1: 3x3 at
// stride 1, the atrous convolutions of Segmentation/network/backbone/resnet.py:29-32 and _deeplab.py:146-153); all bf16,
// channels-last.
int afan_conv_fwd_nhwc_bf16(const void* x, const void* w, void* y, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                            int64_t co, int k, int stride, int dilation, float* stats_partials, const float* stats_shift,
                            double* stats_acc, int groups, afan_stream_t stream) {
    if (ci == 3) {                                              // the image stem has its own kernel
        if (!afan_stem::eligible(n, hi, wi, ci, co, k, stride) || stats_partials || groups > 1 || dilation != 1) return AFAN_ESHAPE;
        if (!x || !w || !y) return AFAN_ENULL;
        if (!aligned(x, 2) || !aligned(w, 2) || !aligned(y, 16) || (stats_acc && !aligned(stats_acc, 16))) return AFAN_EALIGN;
        hipStream_t st = (hipStream_t)stream;
        const double M = (double)n * hi * wi;
        AFAN_PROF_FLOPS("conv_stem_fwd_kernel", 2.0 * (M * co + M * 3 + 27.0 * co), 2.0 * M * co * 27, st);
        return afan_stem::fwd_launch(x, w, y, n, hi, wi, co, stats_acc, afan_nhwc::acc_slot_count(co), stats_shift, st);
    }
    int e = check_dims(n, hi, wi, ci, co, k, stride, dilation);
    if (e) return e;
    if (!x || !w || !y) return AFAN_ENULL;
    if (!aligned(x, 16) || !aligned(w, 16) || !aligned(y, 16)) return AFAN_EALIGN;
    const int pad = k / 2;
    ConvP p{};
    p.max_pad = dilation;
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w; p.y = (uint16_t*)y;
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci;
    p.Ho = (int)((hi + 2 * pad - k) / stride + 1); p.Wo = (int)((wi + 2 * pad - k) / stride + 1); p.Co = (int)co;
    p.in_s = stride; p.out_s = 1; p.w_row_stride = (int)(k * k * ci); p.n_classes = 1;
    if (stats_partials && stats_acc) return AFAN_ESHAPE;   // one destination for the moments, not both
    if (stats_acc && !aligned(stats_acc, 16)) return AFAN_EALIGN;
    p.stats = stats_partials; p.shift = stats_shift; p.acc = stats_acc; p.acc_ns = afan_nhwc::acc_slot_count(co);
    if ((e = set_groups(p, groups, n, (int64_t)p.Ho * p.Wo, co, stats_acc != nullptr))) return e;
    ConvClass& c0 = p.cls[0];
    c0.Hg = p.Ho; c0.Wg = p.Wo; c0.out_h0 = 0; c0.out_w0 = 0; c0.T = k * k;
    for (int r = 0; r < k; ++r)
        for (int s = 0; s < k; ++s) {
            const int t = r * k + s;
            c0.dh[t] = (r - pad) * dilation; c0.dw[t] = (s - pad) * dilation; c0.wofs[t] = (int)(t * ci);
        }
    hipStream_t st = (hipStream_t)stream;
    const double M = (double)n * p.Ho * p.Wo;
    AFAN_PROF_FLOPS("conv_igemm_fwd_kernel", 2.0 * (M * co + (double)n * hi * wi * ci + (double)co * k * k * ci),
                    2.0 * M * co * k * k * ci, st);
    if (small_eligible(p)) return small_groups_ok(p) ? small_launch(p, st) : AFAN_ESHAPE;
    if (ci % 8 != 0 || co % 8 != 0 || ci < 40 || co < 40) return AFAN_ESHAPE;   // (small shape asked for the partial-slab statistics)
    if (!stats_partials && groups <= 1 && dilation == 1 && afan_c64::eligible(n, hi, wi, ci, co, k, stride)) {   // weights-in-registers kernel
        afan_c64::Params q{};
        q.x = p.x; q.w = p.w; q.y = p.y; q.N = p.N; q.H = p.Hi; q.W = p.Wi; q.flip = 0;
        q.acc = stats_acc; q.acc_ns = p.acc_ns; q.shift = stats_shift;
        return afan_c64::launch(q, st);
    }
    return dispatch(p, st, false);
}

// The same convolution with a FROZEN BatchNorm (+ residual) (+ ReLU) applied in its epilogue: y = [relu](bf16(conv(x, w)) * alpha
// + beta [+ residual]), coefs = an afan_affine_coefs block — bit for bit the convolution launch followed by afan_affine_apply,
// without the raw tensor's round trip and the second launch (Detection's frozen bottlenecks: seven launches of 5-13 us per block
// become three or four).  Shapes of the tiled kernel only: AFAN_ESHAPE for the stem, the small-channel and the 64 -> 64
// weights-in-registers kernels (their epilogues have no such form; the caller issues the two launches).
int afan_conv_fwd_affine_nhwc_bf16(const void* x, const void* w, void* y, int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co,
                                   int k, int stride, const float* coefs, const void* residual, int relu, afan_stream_t stream) {
    int e = check_dims(n, hi, wi, ci, co, k, stride, 1);
    if (e) return e;
    if (!x || !w || !y || !coefs) return AFAN_ENULL;
    if (!aligned(x, 16) || !aligned(w, 16) || !aligned(y, 16) || (residual && !aligned(residual, 16))) return AFAN_EALIGN;
    if (ci % 8 != 0 || co % 8 != 0 || ci < 40 || co < 40) return AFAN_ESHAPE;
    const int pad = k / 2;
    ConvP p{};
    p.max_pad = 1;
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w; p.y = (uint16_t*)y;
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci;
    p.Ho = (int)((hi + 2 * pad - k) / stride + 1); p.Wo = (int)((wi + 2 * pad - k) / stride + 1); p.Co = (int)co;
    p.in_s = stride; p.out_s = 1; p.w_row_stride = (int)(k * k * ci); p.n_classes = 1;
    p.acc_ns = afan_nhwc::acc_slot_count(co);
    if ((e = set_groups(p, 1, n, (int64_t)p.Ho * p.Wo, co, false))) return e;
    p.aff = coefs; p.aff_res = (const uint16_t*)residual; p.aff_relu = relu ? 1 : 0;
    ConvClass& c0 = p.cls[0];
    c0.Hg = p.Ho; c0.Wg = p.Wo; c0.out_h0 = 0; c0.out_w0 = 0; c0.T = k * k;
    for (int r = 0; r < k; ++r)
        for (int s = 0; s < k; ++s) {
            const int t = r * k + s;
            c0.dh[t] = r - pad; c0.dw[t] = s - pad; c0.wofs[t] = (int)(t * ci);
        }
    if (small_eligible(p) || afan_c64::eligible(n, hi, wi, ci, co, k, stride)) return AFAN_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const double M = (double)n * p.Ho * p.Wo;
    AFAN_PROF_FLOPS("conv_igemm_fwd_kernel", 2.0 * (M * co * (residual ? 2 : 1) + (double)n * hi * wi * ci + (double)co * k * k * ci),
                    2.0 * M * co * k * k * ci, st);
    return dispatch(p, st, false);
}

// nb <= 4 forward problems on the SAME input with the SAME output shape in one launch (grid.z = problem): ASPP's atrous 3x3
// branches (Segmentation/network/_deeplab.py:143-150,173-176: 2048 -> 256 at three dilations) — at 2 images per GPU one
// branch is 70 workgroups with a 288-step K loop (92 us on 256 CUs), three branches in one grid take the same 92 us — and
// a BasicBlock's first 3x3 convolution with its 1x1 projection shortcut (Classification/resnet_s.py:52-77 option B: both
// read x at stride 2 and write [N, planes, H/2, W/2]).  ksize[b] in {1, 3}, padding dilation[b] * (ksize[b] / 2).
// BatchNorm moments go to f64 accumulator blocks (one per problem; groups = 2: two consecutive ones, one per half-batch) or nowhere.
static int fwd_multi_impl(const void* x, const void* const* w, void* const* y, int nb, int64_t n, int64_t hi, int64_t wi,
                          int64_t ci, int64_t co, const int* ksize, int stride, const int* dilation,
                          const float* const* stats_shift, double* const* stats_acc, int groups, afan_stream_t stream, const ConvP* bnf) {
    if (nb < 1 || nb > 4) return AFAN_ESHAPE;
    if (groups != 1 && (groups != 2 || !stats_acc || (n & 1))) return AFAN_ESHAPE;
    if (!w || !y || !dilation || !ksize) return AFAN_ENULL;
    if ((stats_acc != nullptr) != (stats_shift != nullptr)) return AFAN_ESHAPE;
    int dmax = 1;
    int64_t ho = -1, wo = -1;
    for (int b = 0; b < nb; ++b) {
        int e = check_dims(n, hi, wi, ci, co, ksize[b], stride, dilation[b]);
        if (e) return e;
        if (!w[b] || !y[b]) return AFAN_ENULL;
        if (!aligned(w[b], 16) || !aligned(y[b], 16)) return AFAN_EALIGN;
        if (stats_acc && (!stats_acc[b] || !aligned(stats_acc[b], 16))) return stats_acc[b] ? AFAN_EALIGN : AFAN_ENULL;
        if (dilation[b] > dmax) dmax = dilation[b];
        const int pad = ksize[b] / 2;                 // (a dilated 3x3 at stride 1 keeps the spatial size, like the plain one)
        const int64_t h_b = (hi + 2 * pad - ksize[b]) / stride + 1, w_b = (wi + 2 * pad - ksize[b]) / stride + 1;
        if (b && (h_b != ho || w_b != wo)) return AFAN_ESHAPE;
        ho = h_b; wo = w_b;
    }
    if (!x) return AFAN_ENULL;
    if (!aligned(x, 16)) return AFAN_EALIGN;
    if (ci % 8 != 0 || co % 8 != 0 || ci < 40 || co < 40 || ci == 3) return AFAN_ESHAPE;
    ConvP p{};
    p.max_pad = dmax;
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w[0]; p.y = (uint16_t*)y[0];
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci;
    p.Ho = (int)ho; p.Wo = (int)wo; p.Co = (int)co;
    p.in_s = stride; p.out_s = 1; p.w_row_stride = (int)(ksize[0] * ksize[0] * ci); p.n_classes = nb;
    p.multi = 1;
    p.acc = stats_acc ? stats_acc[0] : nullptr; p.shift = stats_shift ? stats_shift[0] : nullptr;
    p.acc_ns = afan_nhwc::acc_slot_count(co);
    {   // groups = 2: the batch is two half-batches, each problem's accumulator block is two consecutive blocks (one per half)
        const int e = set_groups(p, groups, n, ho * wo, co, stats_acc != nullptr);
        if (e) return e;
    }
    double flops = 0, wbytes = 0;
    for (int b = 0; b < nb; ++b) {
        const int k = ksize[b], pad = k / 2;
        p.w_off[b] = (const uint16_t*)w[b] - (const uint16_t*)w[0];
        p.y_off[b] = (uint16_t*)y[b] - (uint16_t*)y[0];
        p.w_rs[b] = (int)(k * k * ci);
        if (stats_acc) {
            if (!stats_shift[b] != !stats_shift[0]) return AFAN_ESHAPE;     // all problems with a shift, or none
            p.acc_off[b] = stats_acc[b] - stats_acc[0];
            p.shift_off[b] = stats_shift[0] ? stats_shift[b] - stats_shift[0] : 0;
        }
        ConvClass& c = p.cls[b];
        c.Hg = p.Ho; c.Wg = p.Wo; c.out_h0 = 0; c.out_w0 = 0; c.T = k * k;
        for (int r = 0; r < k; ++r)
            for (int s = 0; s < k; ++s) {
                const int t = r * k + s;
                c.dh[t] = (r - pad) * dilation[b]; c.dw[t] = (s - pad) * dilation[b]; c.wofs[t] = (int)(t * ci);
            }
        flops += 2.0 * n * ho * wo * co * k * k * ci;
        wbytes += 2.0 * co * k * k * ci;
    }
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF_FLOPS(bnf ? "conv_bn_fwd_kernel" : "conv_igemm_fwd_kernel", (nb + (bnf ? 1 : 0)) * 2.0 * n * ho * wo * co + wbytes + 2.0 * (double)n * hi * wi * ci, flops, st);
    if (bnf) {         // problem 0's train-mode BatchNorm (+ ReLU) inside the launch (afan_conv_fwd_multi_bn_nhwc_bf16)
        if (!stats_acc || groups != 1 || ci % 64 != 0 || co % 64 != 0) return AFAN_ESHAPE;
        const int64_t M = n * ho * wo;
        const int pending = afan_nhwc::set_running_updates(1);
        afan_nhwc::set_running_updates(pending);
        p.bnf = 1; p.bar = bnf->bar; p.y2 = bnf->y2;
        p.bnf_w = bnf->bnf_w; p.bnf_b = bnf->bnf_b; p.bnf_eps = bnf->bnf_eps; p.bnf_mom = bnf->bnf_mom;
        p.bnf_rmean = bnf->bnf_rmean; p.bnf_rvar = bnf->bnf_rvar; p.bnf_nbt = bnf->bnf_nbt; p.bnf_updates = pending;
        p.bnf_stats = bnf->bnf_stats; p.bnf_relu = bnf->bnf_relu;
        p.bnf_inv_m = 1.0 / (double)M;
        p.bnf_unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f;
        return dispatch_bnf(p, st, false);
    }
    return dispatch(p, st, false);
}

int afan_conv_fwd_multi_nhwc_bf16(const void* x, const void* const* w, void* const* y, int nb, int64_t n, int64_t hi, int64_t wi,
                                  int64_t ci, int64_t co, const int* ksize, int stride, const int* dilation,
                                  const float* const* stats_shift, double* const* stats_acc, int groups, afan_stream_t stream) {
    return fwd_multi_impl(x, w, y, nb, n, hi, wi, ci, co, ksize, stride, dilation, stats_shift, stats_acc, groups, stream, nullptr);
}

// The same launch with problem 0's train-mode BatchNorm (+ ReLU) applied inside it (y_act0 = relu(bn(y[0])), stats0 [4][co] out,
// running buffers updated like afan_bn_train_forward_acc would): a residual block's first 3x3 / stride-2 convolution, whose 1x1
// projection rides as problem 1 (its BatchNorm is applied by the block's last launch).  Only problem 0's workgroups meet at the
// grid barrier.  AFAN_ESHAPE: nothing has run (afan_conv_fwd_multi_nhwc_bf16 + afan_bn_train_forward_acc instead).
int afan_conv_fwd_multi_bn_nhwc_bf16(const void* x, const void* const* w, void* const* y, int nb, int64_t n, int64_t hi, int64_t wi,
                                     int64_t ci, int64_t co, const int* ksize, int stride, const int* dilation,
                                     const float* const* stats_shift, double* const* stats_acc, void* y_act0, const float* bn_weight,
                                     const float* bn_bias, float eps, float momentum, float* stats0, float* running_mean,
                                     float* running_var, int64_t* num_batches, int relu, void* barrier, afan_stream_t stream) {
    if (!y_act0 || !stats0 || !barrier || !stats_acc || !stats_shift) return AFAN_ENULL;
    if (!aligned(y_act0, 16) || !aligned(barrier, 64)) return AFAN_EALIGN;
    if (nb < 2) return AFAN_ESHAPE;
    ConvP b{};
    b.bar = (unsigned*)barrier; b.y2 = (uint16_t*)y_act0;
    b.bnf_w = bn_weight; b.bnf_b = bn_bias; b.bnf_eps = eps; b.bnf_mom = momentum;
    b.bnf_rmean = running_mean; b.bnf_rvar = running_var; b.bnf_nbt = num_batches; b.bnf_stats = stats0; b.bnf_relu = relu ? 1 : 0;
    return fwd_multi_impl(x, w, y, nb, n, hi, wi, ci, co, ksize, stride, dilation, stats_shift, stats_acc, 1, stream, &b);
}

// number of partial slots per channel the dgrad launch of this problem writes when asked for fused BN-backward sums
int64_t afan_conv_dgrad_tiles(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride) {
    if (check_dims(n, hi, wi, co, ci, k, stride)) return 0;
    if (stride == 1) {
        const int64_t M = n * hi * wi;
        const int bm = choose_bm(M, (int)ci, 1);
        return (M + bm - 1) / bm;
    }
    int nc = 0;
    int64_t mmax = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            const int64_t hg = (hi - ph + 1) / 2, wg = (wi - pw + 1) / 2;
            if (hg <= 0 || wg <= 0) continue;
            ++nc;
            if (n * hg * wg > mmax) mmax = n * hg * wg;
        }
    const int bm = choose_bm(mmax, (int)ci, nc);
    return ((mmax + bm - 1) / bm) * nc;
}

// dx[N,Hi,Wi,Ci] = conv_transpose(dy[N,Ho,Wo,Co], w) given wt[Ci,k,k,Co] = w[Co,k,k,Ci] transposed (CRSK).
// Optional epilogue fusions: dx += addend (same shape as dx);  bn_x/bn_stats/bn_partials: dx is the gradient entering
// the backward of the BatchNorm whose input was bn_x — its reduction sums are written to bn_partials[2][Ci][G], or
// added into the f64 accumulators bn_acc[2][Ci] (zeroed by the caller; see afan_bn_backward_acc).
int afan_conv_dgrad_nhwc_bf16(const void* dy, const void* wt, void* dx, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                              int64_t co, int k, int stride, int dilation, const void* addend, const void* bn_x,
                              const float* bn_stats, int bn_relu, const void* bn_y, float* bn_partials,
                              double* bn_acc, int groups, afan_stream_t stream) {
    return dgrad_impl(dy, nullptr, wt, dx, n, hi, wi, ci, co, k, stride, dilation, addend, bn_x, bn_stats, bn_relu, bn_y,
                      bn_partials, bn_acc, groups, stream);
}

// The input gradient of a convolution whose INPUT came out of a frozen BatchNorm + ReLU (Detection's bottlenecks), with that layer's
// backward applied on the way out: dx = bf16((act > 0 ? bf16(dgrad(dy)) : 0) * alpha[c]) — bit for bit afan_conv_dgrad_nhwc_bf16
// followed by afan_affine_relu_bwd(relu = 1), without the raw gradient's round trip and the second launch.  alpha = that layer's
// alpha row [ci], act = its stored output [n, ci, hi, wi].  AFAN_ESHAPE where another kernel owns the shape.
int afan_conv_dgrad_affine_nhwc_bf16(const void* dy, const void* wt, void* dx, int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co,
                                     int k, int stride, const float* alpha, const void* act, afan_stream_t stream) {
    if (!alpha || !act) return AFAN_ENULL;
    return dgrad_impl(dy, nullptr, wt, dx, n, hi, wi, ci, co, k, stride, 1, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 1, stream,
                      alpha, act);
}

// The input gradient that arrives at the OUTPUT of a frozen-BatchNorm residual block (Detection's bottlenecks: out = relu(bn3(conv3) +
// shortcut)), with that block's first backward step applied on the way out: g = bf16(bf16(dgrad(dy)) + addend) (addend optional:
// the consumer block's own shortcut share), m = act > 0 ? g : 0, dres = m, d3 = bf16(m * alpha[c]) — bit for bit
// afan_conv_dgrad_nhwc_bf16(addend) followed by afan_affine_relu_bwd(relu = 1, dx = d3, dres = dres).  alpha: the producing block's
// LAST BatchNorm's alpha row [ci]; act: its stored output (= this convolution's input tensor) [n, ci, hi, wi].  AFAN_ESHAPE where
// another kernel owns the shape.
int afan_conv_dgrad_dual_nhwc_bf16(const void* dy, const void* wt, void* d3, void* dres, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                                   int64_t co, int k, int stride, const void* addend, const float* alpha, const void* act,
                                   afan_stream_t stream) {
    if (!alpha || !act || !dres) return AFAN_ENULL;
    return dgrad_impl(dy, nullptr, wt, d3, n, hi, wi, ci, co, k, stride, 1, addend, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 1, stream,
                      alpha, act, dres);
}

// The input gradient of a residual block's two stride-2 branches in ONE launch (Classification/resnet_s.py:52-77, option B):
// dx = conv_transpose(dy, w1: 3x3 / 2) + conv_transpose(dy_sc, w_sc: 1x1 / 2).  dy and dy_sc are [N, Ho, Wo, Co] tensors in
// ONE allocation (dy_sc behind dy); wt10 is [Ci][10][Co]: the nine taps of w1 transposed, the tenth slot w_sc transposed
// (afan_transpose_weights writes both, see its descriptor).  The projection is one more tap of the even/even output-parity
// class, gathered from dy_sc: no second launch, no dx_sc tensor, no addend read in the epilogue.
int afan_conv_dgrad_sc_nhwc_bf16(const void* dy, const void* dy_sc, const void* wt10, void* dx, int64_t n, int64_t hi,
                                 int64_t wi, int64_t ci, int64_t co, const void* bn_x, const float* bn_stats, int bn_relu,
                                 const void* bn_y, float* bn_partials, double* bn_acc, afan_stream_t stream) {
    if (!dy_sc) return AFAN_ENULL;
    return dgrad_impl(dy, dy_sc, wt10, dx, n, hi, wi, ci, co, 3, 2, 1, nullptr, bn_x, bn_stats, bn_relu, bn_y, bn_partials,
                      bn_acc, 1, stream);
}
// ---- round 5: convolution + train-mode BatchNorm in ONE launch (ConvP::bnf; kernels in afan_conv_bnf.hip) -------------------------
// The BatchNorm's batch statistics need every tile of the launch, so the epilogue meets the launch's other workgroups at a grid-wide
// barrier between its sums and its second pass; only launches whose workgroups are all resident take it (3x3 / stride 1 on whole
// 64-channel chunks in the LDS-resident halo form, <= one workgroup per CU: the ResNet tails at batch 256).  AFAN_ESHAPE = "not this
// launch": nothing has run, the caller issues the two launches it replaces (same bits).  `barrier`: afan_grid_barrier_bytes() of
// device memory, zeroed once, shared by every such launch of one stream; word afan_grid_barrier_error_word() turns non-zero if a
// barrier's bounded spin gave up (workgroups not co-resident: another process's kernels on the GPU) — results are then invalid.
int afan_grid_barrier_bytes(void) { return 2048; }
// on != 0: kernels of other streams may run beside the launches below (weight gradients on a side stream): only launches of at
// most one workgroup per CU take the in-launch form (see g_bnf_one_per_cu).  Returns the previous setting.
int afan_grid_barrier_shared_gpu(int on) {
    const int old = afan_conv::g_bnf_one_per_cu;
    afan_conv::g_bnf_one_per_cu = on ? 1 : 0;
    return old;
}
int afan_grid_barrier_error_word(void) { return (int)(offsetof(GridBar, err) / sizeof(unsigned)); }

// y_raw = conv(x, w) (3x3, stride 1, padding 1) AND y_act = [relu](bn(y_raw) [+ residual]) with the batch statistics of y_raw —
// afan_conv_fwd_nhwc_bf16(stats_acc) followed by afan_bn_train_forward_acc (or, sc_raw != NULL, afan_bn_train_forward_acc_dual:
// y_act = relu(bn(y_raw) + bn_sc(sc_raw)), the projection shortcut's BatchNorm from ITS accumulators sc_acc, filled by an earlier
// launch), term for term.  acc: this launch's accumulator block (zeroed by the caller); shift: the running mean (moments are taken
// around it).  stats [4][co] out; running buffers updated afan_bn_set_running_updates() times like the stand-alone launches.
int afan_conv_fwd_bn_nhwc_bf16(const void* x, const void* w, void* y_raw, void* y_act, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                               int64_t co, int ksize, int dilation, double* acc, const float* shift, const float* bn_weight, const float* bn_bias, float eps,
                               float momentum, float* stats, float* running_mean, float* running_var, int64_t* num_batches,
                               const void* residual, int relu, const void* sc_raw, const double* sc_acc, const float* sc_weight,
                               const float* sc_bias, float sc_eps, float sc_momentum, float* sc_stats, float* sc_running_mean,
                               float* sc_running_var, int64_t* sc_num_batches, void* barrier, afan_stream_t stream) {
    if ((ksize != 1 && ksize != 3) || dilation < 1 || (ksize == 1 && dilation != 1)) return AFAN_ESHAPE;
    const int k = ksize, stride = 1, pad = k / 2;
    int e = check_dims(n, hi, wi, ci, co, k, stride, dilation);
    if (e) return e;
    if (!x || !w || !y_raw || !y_act || !acc || !stats || !barrier) return AFAN_ENULL;
    if (!aligned(x, 16) || !aligned(w, 16) || !aligned(y_raw, 16) || !aligned(y_act, 16) || !aligned(acc, 16) || !aligned(barrier, 64) ||
        (residual && !aligned(residual, 16)) || (sc_raw && !aligned(sc_raw, 16)))
        return AFAN_EALIGN;
    if (sc_raw && (residual || !sc_acc || !sc_stats || !relu)) return AFAN_ESHAPE;
    if (ci % 64 != 0 || co % 64 != 0) return AFAN_ESHAPE;
    ConvP p{};
    p.max_pad = dilation;
    p.x = (const uint16_t*)x; p.w = (const uint16_t*)w; p.y = (uint16_t*)y_raw; p.y2 = (uint16_t*)y_act;
    p.N = (int)n; p.Hi = (int)hi; p.Wi = (int)wi; p.Ci = (int)ci;
    p.Ho = (int)hi; p.Wo = (int)wi; p.Co = (int)co;
    p.in_s = 1; p.out_s = 1; p.w_row_stride = (int)(k * k * ci); p.n_classes = 1;
    p.shift = shift; p.acc = acc; p.acc_ns = afan_nhwc::acc_slot_count(co);
    p.groups = 1;
    ConvClass& c0 = p.cls[0];
    c0.Hg = p.Ho; c0.Wg = p.Wo; c0.T = k * k;
    for (int r = 0; r < k; ++r)
        for (int q = 0; q < k; ++q) {
            const int t = r * k + q;
            c0.dh[t] = (r - pad) * dilation; c0.dw[t] = (q - pad) * dilation; c0.wofs[t] = (int)(t * ci);
        }
    const int64_t M = n * hi * wi;
    const int pending = afan_nhwc::set_running_updates(1);
    afan_nhwc::set_running_updates(pending);
    p.bnf = 1; p.bar = (unsigned*)barrier;
    p.bnf_w = bn_weight; p.bnf_b = bn_bias; p.bnf_eps = eps; p.bnf_mom = momentum;
    p.bnf_rmean = running_mean; p.bnf_rvar = running_var; p.bnf_nbt = num_batches; p.bnf_updates = pending;
    p.bnf_stats = stats; p.bnf_relu = relu ? 1 : 0;
    p.bnf_res = (const uint16_t*)(sc_raw ? sc_raw : residual);
    p.bnf_inv_m = 1.0 / (double)M;
    p.bnf_unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f;
    if (sc_raw) {
        p.bnf_sc.acc = sc_acc; p.bnf_sc.w = sc_weight; p.bnf_sc.b = sc_bias; p.bnf_sc.eps = sc_eps; p.bnf_sc.mom = sc_momentum;
        p.bnf_sc.rmean = sc_running_mean; p.bnf_sc.rvar = sc_running_var; p.bnf_sc.nbt = sc_num_batches; p.bnf_sc.stats = sc_stats;
    }
    if (small_eligible(p) || afan_c64::eligible(n, hi, wi, ci, co, k, stride)) return AFAN_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    // (its own label: the launch does the convolution's FLOPs AND the BatchNorm's passes — bench.py prices it separately)
    AFAN_PROF_FLOPS("conv_bn_fwd_kernel", 2.0 * ((double)M * co * (p.bnf_res ? 3 : 2) + (double)M * ci + (double)co * k * k * ci),
                    2.0 * (double)M * co * k * k * ci, st);
    return dispatch_bnf(p, st, false);
}

// dx = the gradient entering the INPUT of the BatchNorm in front of this convolution (afan_conv_dgrad_nhwc_bf16 with its bn_x /
// bn_stats / bn_acc sums, followed by afan_bn_backward_acc on its result: the same bits), dres (optional) = that backward's second
// output, the ReLU-masked gradient (a residual block's shortcut share).  The unnormalised input gradient is never written.
// dweight / dbias [ci]: the BatchNorm's parameter gradients (NULL: not wanted), added to when accumulate != 0.
// sc_x != NULL (block-output form only): the producing block's projection shortcut's BatchNorm (no ReLU; input sc_x, statistics
// sc_stats [4][ci], zeroed accumulators sc_acc) receives the masked gradient too — d_sc = the gradient entering ITS input
// (afan_bn_backward_acc(dres, sc_x, relu = 0)'s result up to the summation order of its two sums), sc_dweight / sc_dbias optional.
int afan_conv_dgrad_bn_nhwc_bf16(const void* dy, const void* wt, void* dx, void* dres, int64_t n, int64_t hi, int64_t wi, int64_t ci,
                                 int64_t co, int ksize, int dilation, const void* addend, const void* bn_x, const float* bn_stats, int bn_relu, const void* bn_y,
                                 double* bn_acc, float* dweight, float* dbias, int accumulate, const void* sc_x, const float* sc_stats,
                                 double* sc_acc, void* d_sc, float* sc_dweight, float* sc_dbias, void* barrier, afan_stream_t stream) {
    if (!barrier || !bn_acc) return AFAN_ENULL;
    if (!aligned(barrier, 64) || (dres && !aligned(dres, 16))) return AFAN_EALIGN;
    if (ci % 64 != 0 || co % 64 != 0 || (ksize != 1 && ksize != 3) || dilation < 1 || (ksize == 1 && dilation != 1)) return AFAN_ESHAPE;
    ConvP b{};
    b.bar = (unsigned*)barrier; b.y2 = (uint16_t*)dres; b.bnf_dw = dweight; b.bnf_db = dbias; b.bnf_accum = accumulate;
    if (sc_x) {
        if (!sc_stats || !sc_acc || !d_sc) return AFAN_ENULL;
        if (!aligned(sc_x, 16) || !aligned(d_sc, 16) || !aligned(sc_acc, 16)) return AFAN_EALIGN;
        if (!bn_y) return AFAN_ESHAPE;                      // (the block-output form: the masked gradient is what the projection's BatchNorm receives)
        b.bsc.x = (const uint16_t*)sc_x; b.bsc.stats = sc_stats; b.bsc.acc = sc_acc; b.bsc.y3 = (uint16_t*)d_sc;
        b.bsc.dw = sc_dweight; b.bsc.db = sc_dbias;
    }
    return dgrad_impl(dy, nullptr, wt, dx, n, hi, wi, ci, co, ksize, 1, dilation, addend, bn_x, bn_stats, bn_relu, bn_y, nullptr, bn_acc, 1, stream,
                      nullptr, nullptr, nullptr, &b);
}

// The same for the stride-2 pair form (afan_conv_dgrad_sc_nhwc_bf16: a block's first 3x3 / 2 and its 1x1 / 2 projection, whose input
// gradient is the gradient leaving the PREVIOUS block's output): that block's last BatchNorm's backward inside the launch (block-
// output form: bn_y = the block's stored output, dres = the masked gradient for its shortcut).  hi, wi even.
int afan_conv_dgrad_sc_bn_nhwc_bf16(const void* dy, const void* dy_sc, const void* wt10, void* dx, void* dres, int64_t n, int64_t hi,
                                    int64_t wi, int64_t ci, int64_t co, const void* bn_x, const float* bn_stats, int bn_relu,
                                    const void* bn_y, double* bn_acc, float* dweight, float* dbias, int accumulate, void* barrier,
                                    afan_stream_t stream) {
    if (!dy_sc || !barrier || !bn_acc) return AFAN_ENULL;
    if (!aligned(barrier, 64) || (dres && !aligned(dres, 16))) return AFAN_EALIGN;
    if (ci % 64 != 0 || co % 64 != 0) return AFAN_ESHAPE;
    ConvP b{};
    b.bar = (unsigned*)barrier; b.y2 = (uint16_t*)dres; b.bnf_dw = dweight; b.bnf_db = dbias; b.bnf_accum = accumulate;
    return dgrad_impl(dy, dy_sc, wt10, dx, n, hi, wi, ci, co, 3, 2, 1, nullptr, bn_x, bn_stats, bn_relu, bn_y, nullptr, bn_acc, 1, stream,
                      nullptr, nullptr, nullptr, &b);
}

// ---- batched KRSC -> CRSK transpose of every convolution weight (dgrad operands), once per SGD step -------------------
// desc[i] = {src_off, dst_off, K, RS, C, first_tile, dst_RS, rs0}; tiles of 64(k) x 64(c) at fixed rs,
// ceil(K/64)*RS*ceil(C/64) of them per tensor (partial tiles are masked); K % 8 == 0, C % 8 == 0.  The destination rows hold
// dst_RS >= RS tap slots and this tensor's taps land in slots rs0 .. rs0 + RS - 1 (dst_RS = RS, rs0 = 0: the plain CRSK copy;
// a block's 3x3 weights with dst_RS = 10 and its 1x1 projection with rs0 = 9 share one [C][10][K] operand:
// afan_conv_dgrad_sc_nhwc_bf16).
__global__ __launch_bounds__(TRANSPOSE_THREADS) void transpose_weights_kernel(const uint16_t* __restrict__ src,
                                                                uint16_t* __restrict__ dst,
                                                                const int64_t* __restrict__ desc, int n_desc) {
    __shared__ uint16_t tile[64][64 + 8];
    int d = 0;
    while (d + 1 < n_desc && (int64_t)blockIdx.x >= desc[(d + 1) * 8 + 5]) ++d;
    const int64_t* D = desc + d * 8;
    const int64_t K = D[2], RS = D[3], Cc = D[4], DRS = D[6], rs0 = D[7];
    int64_t t = blockIdx.x - D[5];
    const int64_t ctiles = (Cc + 63) / 64;
    const int64_t ct = t % ctiles; t /= ctiles;
    const int64_t rs = t % RS;
    const int64_t kt = t / RS;
    const uint16_t* s = src + D[0];
    uint16_t* o = dst + D[1];
    const int piece = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = r0 + 32 * i;
        if (kt * 64 + k < K && ct * 64 + piece * 8 < Cc)
            *reinterpret_cast<u32x4*>(&tile[k][piece * 8]) =
                *reinterpret_cast<const u32x4*>(s + ((kt * 64 + k) * RS + rs) * Cc + ct * 64 + piece * 8);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = r0 + 32 * i;
        u16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[piece * 8 + j][c];
        if (ct * 64 + c < Cc && kt * 64 + piece * 8 < K)
            *reinterpret_cast<u16x8*>(o + ((ct * 64 + c) * DRS + rs0 + rs) * K + kt * 64 + piece * 8) = v;
    }
}

int afan_transpose_weights(const void* src_arena, void* dst_arena, const int64_t* desc_dev, int n_desc,
                           int64_t total_tiles, afan_stream_t stream) {
    if (n_desc < 0 || total_tiles < 0) return AFAN_ESHAPE;
    if (n_desc == 0 || total_tiles == 0) return AFAN_OK;
    if (!src_arena || !dst_arena || !desc_dev) return AFAN_ENULL;
    if (!aligned(src_arena, 16) || !aligned(dst_arena, 16)) return AFAN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    AFAN_PROF("transpose_weights_kernel", (double)total_tiles * 64 * 64 * 4, st);
    transpose_weights_kernel<<<(unsigned)total_tiles, 256, 0, st>>>((const uint16_t*)src_arena, (uint16_t*)dst_arena,
                                                                    desc_dev, n_desc);
    AFAN_LAUNCH_CHECK();
    return AFAN_OK;
}

}  // extern "C"
#endif  // AFAN_CONV_BNF_TU
