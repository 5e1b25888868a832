"""cv_a-fan_amd — the A-FAN training hot path (feature-space PGD + joint clean/adv step), MI355X-native.

The directory name carries a hyphen (it is fixed by the build contract), so import it with
    import importlib; afan = importlib.import_module("cv_a-fan_amd")
Sub-modules: attack_algo (PGD & friends, reference signatures), resnet_s (slice-protocol models),
arena (flat parameter arena + fused SGD), train_step (the joint step, data parallel), ops (tensor
wrappers over the C-ABI in include/afan_hip.h), main_perturb (entry point for cmd/run_perturb.sh), deeplab (the
DeepLabv3+ split-forward network), seg_attack_algo / seg_trainer (the Segmentation A-FAN operators and iteration), det_ops / det_attack_algo / det_model / det_trainer
(the Detection operators, iteration, the Faster-RCNN / ResNet-101 model and its data-parallel trainer).
"""
from . import _lib, ops  # noqa: F401
from ._lib import AfanLibraryError, LIB_PATH  # noqa: F401
from . import resnet_s, attack_algo, arena, grid_guard, train_step, learnable, seg_attack_algo, deeplab, seg_trainer, det_ops, det_attack_algo, det_model, det_trainer, host  # noqa: F401
from .attack_algo import PGD, get_sample_points, linfball_proj, mix_feature, tensor_clamp  # noqa: F401

__version__ = "0.1.0"
