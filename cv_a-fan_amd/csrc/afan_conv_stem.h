// 3-channel image stem (3x3, stride 1): forward and weight gradient — see afan_conv_stem.hip.
#pragma once
#include "afan_common.h"

namespace afan_stem {

// ci == 3, k == 3, stride == 1, co in {16, 32, 64}, image width a multiple of 32
bool eligible(int64_t n, int64_t h, int64_t w, int64_t ci, int64_t co, int k, int stride);
// x [N,H,W,3], w [Co,3,3,3] (KRSC), y [N,H,W,Co], all bf16; acc / shift: optional BatchNorm moments (f64 accumulator block)
int fwd_launch(const void* x, const void* w, void* y, int64_t n, int64_t h, int64_t wd, int64_t co, double* acc, int acc_ns,
               const float* shift, hipStream_t st);
int64_t wgrad_workspace_floats(int64_t n, int64_t h, int64_t w, int64_t co);
// grad [Co,3,3,3] fp32 (KRSC) (+)= sum dy[N,H,W,Co] * window(x)
int wgrad_launch(const void* x, const void* dy, float* grad, int64_t n, int64_t h, int64_t wd, int64_t co, float* ws,
                 int accumulate, hipStream_t st);

}  // namespace afan_stem
