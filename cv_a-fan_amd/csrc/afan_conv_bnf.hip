// The tiled convolution's instantiations with a train-mode BatchNorm applied INSIDE the launch (ConvP::bnf, conv_igemm_body<..., BF>):
// forward  — raw output + batch sums -> grid barrier -> totals -> normalised (+ residual / projection BatchNorm) (+ ReLU) output;
// backward — input gradient (+ other branch) + the BatchNorm backward's two sums -> grid barrier -> totals -> the gradient entering
//            the BatchNorm's input (and the masked gradient for the shortcut).
// Reference chain: Classification/resnet_s.py:72-77 (conv - bn - relu - conv - bn - (+shortcut) - relu), PGD's K tail passes
// (attack_algo.py:48-56).  Round 4 ran each BatchNorm as its own 8-16 us launch behind the convolution that had already summed
// its moments: ~150 launches, 2.2 ms of an 8.6 ms step (profiles/r04f_r18_kernel_stats.csv).  A separate translation unit so that
// the two sets of instantiations compile in parallel; the kernel body is afan_conv.hip's (included below, up to its dispatch).
#define AFAN_CONV_BNF_TU 1
#include "afan_conv.hip"

}  // namespace   (afan_conv.hip's anonymous namespace: its tail, which closes it, is not compiled in this unit)

namespace afan_conv {

int g_bnf_one_per_cu = 0;

// dispatch()'s halo-form choices, with the BF instantiations; any other launch: AFAN_ESHAPE (the caller issues two launches)
int dispatch_bnf(const ConvP& p, hipStream_t st, bool dgrad) {
    static const int halo = env_int("AFAN_CONV_HALO", 1);
    if (!halo || p.Co % 128 != 0 || p.stats || !p.acc || p.groups != 1) return AFAN_ESHAPE;
    const int bm = choose_bm(max_rows(p), p.Co, p.n_classes, p.multi != 0);
    const int64_t wgs = (int64_t)(p.Co / 128) * ((max_rows(p) + bm - 1) / bm) * p.n_classes;
    constexpr int deep_max = 384;
    static const int ahead = env_int("AFAN_CONV_AHEAD", 1);    // (dispatch()'s switch: the same tile family in both forms — same bits)
    // (round 6) dispatch()'s 64 x 64 tiles for launches that would leave half the chip idle
    static const int t64 = env_int("AFAN_CONV_T64", 1);
    if (t64 && p.n_classes == 1 && p.Ci >= 128) {
        const int64_t w64 = (int64_t)(p.Co / 64) * ((max_rows(p) + 63) / 64);
        if (w64 <= 256 && w64 >= 16) {
            static const int ahead64 = env_int("AFAN_CONV_AHEAD64", 1);
            if (halo_ok(p, 64))
                return ahead64 ? launch_gs<64, 64, 7, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS, false, true>(p, st, dgrad)
                               : launch_gs<64, 64, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS, false, true>(p, st, dgrad);
            return launch_gs<64, 64, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, 0, false, true>(p, st, dgrad);
        }
    }
    if (bm == 128 && wgs > deep_max && wgs <= 2 * deep_max && halo_ok(p, 256, HALO_PIXELS_256))
        return launch_gs<256, 128, 5, 4, 2, 4, 2, HALO_PIXELS_256, false, true>(p, st, dgrad);
    if (bm == 64 && p.Ci >= 256) {
        const int64_t wgs_n64 = (int64_t)(p.Co / 64) * ((max_rows(p) + 127) / 128) * p.n_classes;
        if (wgs_n64 <= deep_max && wgs_n64 >= wgs && halo_ok(p, 128))
            return ahead ? launch_gs<128, 64, 7, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS, false, true>(p, st, dgrad)
                         : launch_gs<128, 64, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS, false, true>(p, st, dgrad);
    }
    if (bm == 128 && wgs <= deep_max && p.Ci >= 256) {
        const int64_t wgs_n64 = (int64_t)(p.Co / 64) * ((max_rows(p) + 255) / 256) * p.n_classes;
        static const int w41 = env_int("AFAN_CONV_W41", 1);
        if (wgs_n64 <= deep_max && wgs_n64 >= wgs && halo_ok(p, 256, HALO_PIXELS_256N))
            return !w41 ? launch_gs<256, 64, 5, 4, 2, 4, 2, HALO_PIXELS_256N, false, true>(p, st, dgrad)
                        : (ahead ? launch_gs<256, 64, 7, 4, 1, 4, 2, HALO_PIXELS_256N, false, true>(p, st, dgrad)
                                 : launch_gs<256, 64, 5, 4, 1, 4, 2, HALO_PIXELS_256N, false, true>(p, st, dgrad));
    }
    if (wgs <= deep_max && halo_ok(p, bm))
        return bm == 64 ? launch_gs<64, 128, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS, false, true>(p, st, dgrad)
                        : launch_gs<128, 128, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, HALO_PIXELS, false, true>(p, st, dgrad);
    // per-tap operand tiles (1x1 convolutions — the bottleneck blocks' first and last — and 3x3 ones whose halo does not fit): the
    // four-stage form with producer waves up to about one workgroup per CU, the two-stage form (two workgroups per CU) beyond;
    // launch_gs refuses what is not resident at once
    static const int tap3 = env_int("AFAN_BNF_TAP3", 1);       // 0: no 3x3 convolution on the per-tap variants (A/B)
    if (!tap3 && p.cls[0].T == 9) return AFAN_ESHAPE;
    if (wgs <= deep_max) {
        const int e = bm == 64 ? launch_gs<64, 128, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, 0, false, true>(p, st, dgrad)
                               : launch_gs<128, 128, 5, 2, 2, 4, AFAN_CONV_FRAG_BATCH, 0, false, true>(p, st, dgrad);
        if (e != AFAN_ESHAPE) return e;          // (one workgroup per CU there: 257 .. 384 of them are resident only on the two-stage form)
    }
    return bm == 64 ? launch_gs<64, 128, 3, 2, 4, 0, AFAN_CONV_FRAG_BATCH, 0, false, true>(p, st, dgrad)
                    : launch_gs<128, 128, 3, 2, 4, 0, AFAN_CONV_FRAG_BATCH, 0, false, true>(p, st, dgrad);
}

}  // namespace afan_conv

#ifdef AFAN_CONV_STAMP
// diagnostic build only: this translation unit's copy of the stamps (tools/probe/conv_stamps.py)
extern "C" int afan_conv_bnf_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(afan_stamps), sizeof(unsigned long long) * 2 * 96 * 3);
}
#endif
