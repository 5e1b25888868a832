// Weight gradient of 3x3 convolutions with a 16/32-channel side — see afan_wgrad_small.hip.
#pragma once
#include "afan_common.h"

namespace afan_wgrad_small {

// k == 3, stride 1 or 2, ci in {16, 32}, co in {16, 32, 64}; output width 8/16/32 (whole rows per 32-pixel tile) or a
// multiple of 32
bool eligible(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int k, int stride);
int64_t workspace_floats(int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int stride);
// grad [Co,3,3,Ci] fp32 (KRSC) (+)= wgrad(x [N,Hi,Wi,Ci], dy [N,Ho,Wo,Co]), bf16 channels-last operands
// optional second operand pair (x2, dy2, n2 images) of the same layer, summed in the same launch; workspace for n + n2
int launch(const void* x, const void* dy, float* grad, int64_t n, int64_t hi, int64_t wi, int64_t ci, int64_t co, int stride,
           float* ws, int accumulate, hipStream_t st, const void* x2 = nullptr, const void* dy2 = nullptr, int64_t n2 = 0);

}  // namespace afan_wgrad_small
