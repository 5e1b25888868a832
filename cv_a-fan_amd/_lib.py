"""ctypes binding of libafan_hip.so (the C-ABI declared in include/afan_hip.h).

The library is the product: there is NO fallback.  If it is missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AFAN_HIP_LIB: another build of the SAME library (diagnostic builds of tools/build_ablate.sh); never a different backend
LIB_PATH = os.environ.get("AFAN_HIP_LIB") or os.path.join(_HERE, "libafan_hip.so")

AFAN_F32, AFAN_BF16 = 0, 1
AFAN_NCHW, AFAN_NHWC = 0, 1
_ERRORS = {-1: "AFAN_EDTYPE (unknown dtype code)", -2: "AFAN_EALIGN (misaligned pointer)",
           -3: "AFAN_ESHAPE (bad sizes)", -4: "AFAN_ENULL (required pointer is NULL)",
           -5: "AFAN_ELAYOUT (unknown layout code)"}

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/afan_hip.h one to one
SIGNATURES = {
    "afan_version": (_i, []),
    "afan_arch": (C.c_char_p, []),
    "afan_pgd_step": (_i, [_p, _p, _i, _p, _p, _l, _f, _f, _i, _p]),
    "afan_tensor_clamp": (_i, [_p, _p, _p, _l, _p]),
    "afan_norms_workspace_floats": (_l, [_l, _l]),
    "afan_pgd_step_norms": (_i, [_p, _p, _i, _p, _p, _l, _l, _f, _f, _i, _p, _p, _p, _p]),
    "afan_perturb_norms": (_i, [_p, _p, _l, _l, _p, _p, _p, _p]),
    "afan_axpy_noise": (_i, [_p, _p, _l, _f, _p, _p]),
    "afan_mix_feature": (_i, [_p, _p, _p, _l, _l, _l, _f, _i, _p]),
    "afan_mix_feature_nhwc": (_i, [_p, _p, _p, _l, _l, _l, _f, _i, _p]),
    "afan_lerp_points": (_i, [_p, _p, _p, _l, C.POINTER(_f), _i, _p]),
    "afan_lerp_mix": (_i, [_p, _p, _p, _l, _l, _l, C.POINTER(_f), _i, C.c_uint, _f, _i, _p]),
    "afan_head_max_classes": (_i, []),
    "afan_head_forward": (_i, [_p, _i, _l, _l, _l, _p, _p, _l, _p, _p, _p]),
    "afan_head_backward": (_i, [_p, _p, _p, _l, _l, _l, _l, _p, _i, _p, _p, _i, _p]),
    "afan_cross_entropy": (_i, [_p, _p, _l, _l, _p, _p, _p]),
    "afan_mix_w_workspace_floats": (_l, []),
    "afan_mix_w": (_i, [_p, _p, _p, _p, _i, _l, _p]),
    "afan_mix_w_backward": (_i, [_p, _i, _p, _p, _l, _p, _p, _i, _p]),
    "afan_bn_workspace_floats": (_l, [_l]),
    "afan_bn_acc_doubles": (_l, [_l]),
    "afan_bn_acc_supported": (_i, [_i, _l]),
    "afan_bn_set_running_updates": (_i, [_i]),
    "afan_bn_running_update": (_i, [_p, _l, C.c_double, _f, _f, _p, _p, _p, _p]),
    "afan_bn_running_update_batched": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "afan_bn_train_forward_acc": (_i, [_p, _p, _p, _i, _l, _l, _l, _f, _f, _p, _p, _i, _p, _i, _p, _p, _p, _p, _i, _p]),
    "afan_bn_train_forward_acc_dual": (_i, [_p, _p, _p, _i, _l, _l, _l, _f, _f, _p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p,
                                            _p, _p, _p, _p]),
    "afan_bn_backward_acc": (_i, [_p, _p, _p, _p, _p, _i, _l, _l, _l, _p, _i, _p, _i, _p, _p, _i, _i, _p]),
    "afan_bn_stats": (_i, [_p, _i, _i, _l, _l, _l, _f, _f, _p, _p, _p, _p, _p, _p]),
    "afan_bn_train_forward": (_i, [_p, _p, _p, _i, _i, _l, _l, _l, _f, _f, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
    "afan_bn_apply": (_i, [_p, _p, _p, _i, _i, _l, _l, _l, _p, _p, _p, _p, _i, _p, _p]),
    "afan_bn_backward": (_i, [_p, _p, _p, _p, _p, _i, _i, _l, _l, _l, _p, _p, _p, _i, _p, _p, _p, _i, _p, _l, _p]),
    "afan_conv_supported": (_i, [_l, _l, _i, _i]),
    "afan_conv_fwd_nhwc_bf16": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _i, _i, _i, _p, _p, _p, _i, _p]),
    "afan_conv_fwd_affine_nhwc_bf16": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _i, _i, _p, _p, _i, _p]),
    "afan_conv_fwd_multi_nhwc_bf16": (_i, [_p, _p, _p, _i, _l, _l, _l, _l, _l, _p, _i, _p, _p, _p, _i, _p]),
    "afan_conv_fwd_tiles": (_l, [_l, _l, _l, _l, _l, _i, _i]),
    "afan_bn_train_forward_partials": (_i, [_p, _p, _p, _i, _l, _l, _l, _f, _f, _p, _p, _i, _p, _l, _p, _p, _p, _p, _p, _p]),
    "afan_conv_dgrad_nhwc_bf16": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _i, _p]),
    "afan_conv_dgrad_affine_nhwc_bf16": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _i, _i, _p, _p, _p]),
    "afan_conv_dgrad_dual_nhwc_bf16": (_i, [_p, _p, _p, _p, _l, _l, _l, _l, _l, _i, _i, _p, _p, _p, _p]),
    "afan_conv_dgrad_sc_nhwc_bf16": (_i, [_p, _p, _p, _p, _l, _l, _l, _l, _l, _p, _p, _i, _p, _p, _p, _p]),
    "afan_grid_barrier_bytes": (_i, []),
    "afan_grid_barrier_error_word": (_i, []),
    "afan_grid_barrier_shared_gpu": (_i, [_i]),
    "afan_conv_fwd_bn_nhwc_bf16": (_i, [_p, _p, _p, _p, _l, _l, _l, _l, _l, _i, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p,
                                        _f, _f, _p, _p, _p, _p, _p, _p]),
    "afan_conv_dgrad_bn_nhwc_bf16": (_i, [_p, _p, _p, _p, _l, _l, _l, _l, _l, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p,
                                          _p, _p]),
    "afan_conv_fwd_multi_bn_nhwc_bf16": (_i, [_p, _p, _p, _i, _l, _l, _l, _l, _l, _p, _i, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i, _p,
                                              _p]),
    "afan_conv_dgrad_sc_bn_nhwc_bf16": (_i, [_p, _p, _p, _p, _p, _l, _l, _l, _l, _l, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p]),
    "afan_conv_dgrad_tiles": (_l, [_l, _l, _l, _l, _l, _i, _i]),
    "afan_conv_wgrad_workspace_floats": (_l, [_l, _l, _l, _l, _l, _i, _i]),
    "afan_conv_wgrad_nhwc_bf16": (_i, [_p, _p, _p, _l, _l, _l, _l, _l, _i, _i, _i, _p, _i, _p]),
    "afan_conv_wgrad_plan": (_i, [_l, _l, _l, _l, _l, _i, _i]),
    "afan_conv_wgrad_multi_nhwc_bf16": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "afan_conv_wgrad2_nhwc_bf16": (_i, [_p, _p, _l, _p, _p, _l, _p, _l, _l, _l, _l, _i, _i, _i, _p, _i, _p]),
    "afan_conv_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _l, _l, _l, _l, _l, _i, _i, _i, _i, _p]),
    "afan_conv_dgrad": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _l, _l, _i, _i, _i, _i, _p]),
    "afan_conv_wgrad_f32_workspace_floats": (_l, [_l, _l, _l, _l, _l, _i, _i, _i, _i]),
    "afan_conv_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _l, _l, _i, _i, _i, _i, _p, _i, _p]),
    "afan_conv_stem7_supported": (_i, [_l, _l, _i, _i]),
    "afan_conv_stem7_fwd_nhwc_bf16": (_i, [_p, _p, _p, _l, _l, _l, _p]),
    "afan_conv_stem7_im2col_k": (_i, []),
    "afan_conv_stem7_im2col": (_i, [_p, _p, _l, _l, _l, _p]),
    "afan_conv_stem7_col2im": (_i, [_p, _p, _l, _l, _l, _p]),
    "afan_conv_stem7_wgrad_workspace_floats": (_l, [_l, _l, _l]),
    "afan_conv_stem7_wgrad_nhwc_bf16": (_i, [_p, _p, _p, _l, _l, _l, _p, _i, _p]),
    "afan_upsample_bilinear_fwd": (_i, [_p, _p, _i, _i, _l, _l, _l, _l, _l, _l, _p]),
    "afan_upsample_bilinear_bwd": (_i, [_p, _p, _i, _i, _l, _l, _l, _l, _l, _l, _p]),
    "afan_upsample_bilinear_fwd_slice": (_i, [_p, _p, _i, _l, _l, _l, _l, _l, _l, _l, _p]),
    "afan_upsample_bilinear_bwd_workspace_floats": (_l, [_l, _l, _l, _l]),
    "afan_upsample_bilinear_bwd_slice": (_i, [_p, _p, _i, _l, _l, _l, _l, _l, _l, _l, _p, _p]),
    "afan_ce2d_workspace_floats": (_l, [_l]),
    "afan_ce2d": (_i, [_p, _p, _i, _l, _l, _l, _l, _f, _p, _p, _p, _p]),
    "afan_ce2d_upsampled_workspace_floats": (_l, [_l, _l, _l, _l, _l, _l]),
    "afan_ce2d_upsampled": (_i, [_p, _p, _l, _l, _l, _l, _l, _l, _l, _f, _p, _p, _p, _p]),
    "afan_maxpool3x3s2_fwd": (_i, [_p, _p, _i, _i, _l, _l, _l, _l, _p]),
    "afan_maxpool3x3s2_bwd": (_i, [_p, _p, _p, _i, _i, _l, _l, _l, _l, _p]),
    "afan_maxpool2d_fwd": (_i, [_p, _p, _p, _i, _i, _l, _l, _l, _l, _i, _i, _i, _p]),
    "afan_maxpool2d_bwd": (_i, [_p, _p, _p, _p, _i, _i, _l, _l, _l, _l, _i, _i, _i, _p]),
    "afan_affine_relu_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _l, _l, _l, _i, _p]),
    "afan_affine_coefs": (_i, [_p, _p, _p, _p, _l, _p, _p]),
    "afan_frozen_bottleneck_fwd": (_i, [_p, _l, _l, _l, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "afan_frozen_bottleneck_bwd_scratch": (_l, [_l, _l, _l, _l, _l, _i]),
    "afan_frozen_bottleneck_bwd": (_i, [_p, _p, _p, _p, _p, _l, _l, _l, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                        _p, _p]),
    "afan_frozen_bottleneck_bwd_chain": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _l, _l, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                              _p, _p, _p, _p, _p, _p]),
    "afan_affine_apply": (_i, [_p, _p, _p, _i, _l, _l, _l, _p, _i, _p]),
    "afan_avgpool_fwd": (_i, [_p, _p, _i, _i, _l, _l, _l, _i, _p]),
    "afan_avgpool_bwd": (_i, [_p, _p, _i, _i, _l, _l, _l, _i, _p]),
    "afan_linear_small_max_rows": (_i, []),
    "afan_linear_small_fwd": (_i, [_p, _p, _p, _l, _l, _l, _p]),
    "afan_linear_small_bwd": (_i, [_p, _p, _p, _p, _p, _l, _l, _l, _i, _p]),
    "afan_pointwise_max_co": (_i, []),
    "afan_pointwise_fwd": (_i, [_p, _i, _p, _p, _p, _l, _l, _l, _p]),
    "afan_pointwise_bwd_dx": (_i, [_p, _p, _p, _i, _l, _l, _l, _p]),
    "afan_pointwise_workspace_floats": (_l, [_l, _l, _l]),
    "afan_pointwise_bwd_dw": (_i, [_p, _p, _i, _p, _p, _l, _l, _l, _p, _i, _p]),
    "afan_dropout": (_i, [_p, _p, _i, _l, _f, _p, _p, _p, _i, _p]),
    "afan_nms_workspace_bytes": (_l, [_l]),
    "afan_nms": (_i, [_p, _p, _l, _f, _i, _p, _p, _p, _p]),
    "afan_nms_top": (_i, [_p, _p, _l, _f, _i, _p, _p, _p, _l, _p]),
    "afan_box_decode_clip": (_i, [_p, _p, _p, _l, _f, _f, _p]),
    "afan_box_assign": (_i, [_p, _p, _l, _l, _l, _i, _f, _f, _p, _p, _p, _p, _p]),
    "afan_sample_lists": (_i, [_p, _l, _p, _p, _p, _p]),
    "afan_sample_gather": (_i, [_p, _p, _p, _l, _p, _p, _p, _p, _l, _l, _p, _p, _p, _p, _p, _p]),
    "afan_det_loss_fwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _l, _l, _l, _f, _p, _p, _p, _p, _p]),
    "afan_sum_scalars_f32": (_i, [_p, _p, _p, _p, _p, _p]),
    "afan_linear_pair_workspace_floats": (_l, [_i, _l, _l, _l, _l]),
    "afan_linear_pair_fwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _l, _l, _l, _p, _p]),
    "afan_linear_pair_dgrad_f32": (_i, [_p, _p, _p, _p, _p, _l, _l, _l, _l, _p]),
    "afan_linear_pair_wgrad_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _l, _l, _l, _l, _p, _p]),
    "afan_proposal_rows": (_i, [_p, _l, _p, _l, _p, _l, _p, _p, _p]),
    "afan_labels_limit": (_i, [_p, _l, _l, _p, _l, _p]),
    "afan_det_loss_bwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _l, _l, _l, _l, _p, _p, _p]),
    "afan_roi_align_fwd": (_i, [_p, _p, _p, _i, _i, _l, _l, _l, _l, _i, _i, _f, _i, _p]),
    "afan_roi_align_bwd": (_i, [_p, _p, _p, _i, _i, _l, _l, _l, _l, _l, _i, _i, _f, _i, _p]),
    "afan_roi_align_bwd_workspace_bytes": (_l, [_l, _l, _l]),
    "afan_roi_align_bwd_ws": (_i, [_p, _p, _p, _i, _i, _l, _l, _l, _l, _l, _i, _i, _f, _i, _p, _p]),
    "afan_transpose_weights": (_i, [_p, _p, _p, _i, _l, _p]),
    "afan_sgd_step": (_i, [_p, _p, _p, _p, _l, _p, _f, _f, _f, _i, _p]),
    "afan_sgd_step_guarded": (_i, [_p, _p, _p, _p, _l, _p, _f, _f, _f, _i, _p, _p]),
    "afan_guarded_copy": (_i, [_p, _p, _l, _p, _p, _p]),
    "afan_occupy_cus": (_i, [_i, _i, _i, _p]),
    "afan_cast_bf16": (_i, [_p, _p, _l, _p]),
    "afan_pgd_init": (_i, [_p, _i, _p, _p, _p, _l, _p]),
    "afan_normalize_nchw": (_i, [_p, _p, _i, _i, _l, _l, _l, _p, _p, _p]),
    "afan_profile_enable": (_i, [_i]),
    "afan_profile_collect": (_i, [C.c_char_p, C.POINTER(_l), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), _i]),
    "afan_profile_event_overhead": (_i, [_i, C.POINTER(_f), _p]),
}

_lib = None


class AfanLibraryError(RuntimeError):
    pass


def load():
    """Load libafan_hip.so once; raise loudly (no fallback) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AfanLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cv_a-fan_amd/csrc`). There is no CPU or PyTorch fallback for the A-FAN kernels.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise AfanLibraryError(f"{what}: {_ERRORS.get(rc, rc)}")
    raise AfanLibraryError(f"{what}: HIP launch failed with hipError_t={rc}")
