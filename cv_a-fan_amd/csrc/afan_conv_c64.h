// 3x3 / stride 1 / 64 -> 64 channel convolution (ResNet-18's layer1 inside the PGD tail): see afan_conv_c64.hip.
#pragma once
#include "afan_common.h"

namespace afan_c64 {

struct Params {
    const uint16_t* x;       // [N, H, W, 64] bf16 channels-last
    const uint16_t* w;       // [64 out][9 taps][64 in] bf16 (KRSC for the forward, CRSK for the input gradient)
    uint16_t* y;             // [N, H, W, 64]
    int N, H, W;
    int flip;                // 0: tap (r, s) reads x[h + r - 1][w + s - 1]; 1: x[h + 1 - r][w + 1 - s] (input gradient)
    // epilogue fusions, same meaning as ConvP in afan_conv.hip
    double* acc;             // f64 accumulator block (BN moments, or BN-backward sums when bnx != NULL)
    int acc_ns;
    const float* shift;
    const uint16_t* bnx;
    const float* bn_stats;
    int bn_relu;
    const uint16_t* bny;
    const uint16_t* addend;
};

// shapes this kernel takes: ci == co == 64, k == 3, stride == 1, W a power of two in [4, 32], H a multiple of 128 / W
bool eligible(int64_t n, int64_t h, int64_t w, int64_t ci, int64_t co, int k, int stride);
int launch(const Params& p, hipStream_t st);

}  // namespace afan_c64
