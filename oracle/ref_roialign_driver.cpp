// TEST INFRASTRUCTURE — never linked into, loaded by or shipped with the product path.
//
// Driver translation unit for the REFERENCE's own CPU ROIAlign forward.  The reference file
// Detection/support/src/cpu/ROIAlign_cpu.cpp holds, in lines 4-219, three plain templates with no ATen in them
// (PreCalc<T> :5-16, pre_calc_for_bilinear_interpolate<T> :18-108, ROIAlignForward_cpu_kernel<T> :110-219); line 2
// (#include "cpu/vision.h": ATen) and the tensor wrapper :221-257 (AT_DISPATCH_FLOATING_TYPES(input.type(), ...):
// does not compile against torch 2.10) are the only parts this image cannot build.  oracle/Makefile extracts exactly
// lines 4-219, unedited, into a scratch directory under /tmp (deleted after the compile; the text never enters the
// repository or oracle/_ref/), and compiles this file with -DREF_LINES=<that scratch file> against the real <cmath> /
// <vector> / <algorithm>: no stand-in headers, no edits.  Output: oracle/_ref/libref_roialign.so (git-ignored).
// What the wrapper :221-257 would have added is only the shape bookkeeping repeated below (output_size =
// num_rois * pooled_h * pooled_w * channels, :237).
#include <algorithm>
#include <cmath>
#include <vector>

#include REF_LINES

extern "C" {

// ROIAlign_forward_cpu, ROIAlign_cpu.cpp:221-257, minus the at::Tensor plumbing.  input [N, C, H, W] contiguous, rois [R, 5] =
// (batch index, x1, y1, x2, y2), out [R, C, PH, PW].
void ref_roi_align_forward_f32(const float* input, const float* rois, float* out, long num_rois, long channels, long height,
                               long width, int pooled_height, int pooled_width, float spatial_scale, int sampling_ratio) {
    const long output_size = num_rois * pooled_height * pooled_width * channels;
    if (output_size == 0) return;
    ROIAlignForward_cpu_kernel<float>((int)output_size, input, spatial_scale, (int)channels, (int)height, (int)width, pooled_height,
                                      pooled_width, sampling_ratio, rois, out);
}

void ref_roi_align_forward_f64(const double* input, const double* rois, double* out, long num_rois, long channels, long height,
                               long width, int pooled_height, int pooled_width, double spatial_scale, int sampling_ratio) {
    const long output_size = num_rois * pooled_height * pooled_width * channels;
    if (output_size == 0) return;
    ROIAlignForward_cpu_kernel<double>((int)output_size, input, spatial_scale, (int)channels, (int)height, (int)width, pooled_height,
                                       pooled_width, sampling_ratio, rois, out);
}

}  // extern "C"
