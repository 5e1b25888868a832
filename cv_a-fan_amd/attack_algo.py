"""A-FAN operators, MI355X-native: drop-in for the reference's `attack_algo` modules.

Same names, argument meaning and return contract as
  Classification/attack_algo.py:9-58   tensor_clamp, linfball_proj, PGD
  Segmentation/attack_algo.py:108-130  get_sample_points, mix_feature   (== Detection/attack_algo.py:236-265)
but every arithmetic step is a hand-written HIP kernel from libafan_hip.so (sign-step + projection +
per-sample norms fused in one pass; channel-moment re-normalisation in one pass).  `model` may be any
nn.Module on the GPU that follows the slice protocol `model(t, end_point=, start_point=)`; for the
models of `resnet_s.py` the tail additionally runs its fused BN kernels with parameter gradients
switched off (the reference asks autograd for the feature gradient only, attack_algo.py:52).
"""
import torch

from . import ops
from .resnet_s import _dense, _like_layout, dgrad_only, fused_criterion

__all__ = ["PGD", "tensor_clamp", "linfball_proj", "mix_feature", "get_sample_points", "last_norms"]

_last = {"l2": None, "linf": None}


def last_norms():
    """Per-sample (L2, Linf) of the perturbation produced by the most recent PGD(..., with_norms=True) call:
    device tensors, no host sync (main_perturb.py:188-192 computes them on the host from a D2H copy)."""
    return _last["l2"], _last["linf"]


def tensor_clamp(t, min, max, in_place=True):
    """attack_algo.py:9-19: element-wise clamp between two tensors, lower bound first."""
    res = t if in_place else t.clone()
    ops.tensor_clamp_(res.data, min.contiguous(), max.contiguous())
    return res


def linfball_proj(center, radius, t, in_place=True):
    """attack_algo.py:35-36: project t onto the L-inf ball of `radius` around `center` (one HIP launch)."""
    res = t if in_place else t.clone()
    zero = torch.zeros_like(res.data)
    ops.pgd_step_(res.data, zero, 0.0, x_clean=center.contiguous(), eps=float(radius), clip=True)
    return res


def PGD(x, loss_fn, y=None, model=None, steps=3, gamma=None, start_idx=1, layer_number=16, eps=(2 / 255),
        randinit=False, clip=False, with_norms=False, grad0=None):
    """K-step sign-gradient ascent on the feature map `x` (attack_algo.py:38-58).

    Returns a NEW fp32 leaf tensor with requires_grad=True; `x` is not modified.  `with_norms=True`
    (an addition) fuses the per-sample L2/Linf perturbation norms into the last step; read them with
    `last_norms()`.  `grad0` (an addition; not with randinit): a positive multiple of d(loss)/d(x) at x itself, already
    computed by the caller — the first ascent step uses it instead of running the tail on x (sign() ignores the scale).
    """
    if x.device.type != "cuda":
        raise ops.AfanLibraryError("PGD: x must live on the MI355X (no CPU path in this build)")
    # keep whatever dense layout the feature map has (the bf16 backbone hands over channels-last tensors: a
    # .contiguous() here would transpose 64 MB to NCHW and the tail would transpose it back every PGD step)
    lp = getattr(model, "compute_dtype", torch.float32) == torch.bfloat16
    if grad0 is not None and (randinit or steps < 1):
        raise ValueError("grad0 is the gradient at x: not with randinit, and only when there is a first step")
    # x (bf16 from the product's backbone, or fp32) -> fp32 x, its clone x_adv and, where the first tail pass needs it, the
    # bf16 shadow: one launch (with grad0 the first step's kernel writes the shadow; with randinit the noise kernel does)
    x, x_adv, shadow = ops.pgd_init(_dense(x.detach()), want_shadow=lp and grad0 is None and not randinit)
    if lp and shadow is None:
        shadow = torch.empty_like(x, dtype=torch.bfloat16)
    if randinit:
        # the reference draws the noise on the CPU default generator (attack_algo.py:44); same stream here
        u = _like_layout(torch.rand(x_adv.shape).to(x.device, non_blocking=True), x_adv)
        ops.axpy_noise_(x_adv, u, eps, shadow)
    l2 = linf = None
    loss_fn = fused_criterion(loss_fn, model)
    for t in range(steps):
        if t == 0 and grad0 is not None:
            grad = grad0.detach()
        else:
            # the tail consumes the bf16 shadow written by the previous step's kernel (no separate cast)
            xin = (shadow if lp else x_adv).detach().requires_grad_(True)
            with dgrad_only():
                out = model(xin, end_point=layer_number, start_point=start_idx)
                loss = loss_fn(out, y)
                root = ops.one(loss.device) if (loss.dim() == 0 and loss.dtype == torch.float32) else None
                grad = torch.autograd.grad(loss, xin, grad_outputs=root, only_inputs=True)[0]
        grad = _like_layout(grad, x_adv)
        if with_norms and t == steps - 1:
            l2, linf = ops.pgd_step_norms_(x_adv, grad, gamma, x, eps, clip, shadow)
        else:
            ops.pgd_step_(x_adv, grad, gamma, x, eps, clip, shadow)
    if with_norms:
        if l2 is None:
            l2, linf = ops.perturb_norms(x_adv, x)
        _last["l2"], _last["linf"] = l2, linf
    x_adv.requires_grad_(True)
    if lp:
        x_adv._afan_shadow = shadow  # bf16 copy of the final x_adv for the adv tail forward
    return x_adv


def mix_feature(clean_feature, adv_feature):
    """Segmentation/attack_algo.py:121-130: re-normalise clean features to the adversarial channel statistics."""
    return ops.mix_feature(_dense(clean_feature), _dense(adv_feature), 1e-5)      # (either dense layout; adv follows clean's)


def sample_points_mixed(pointx, pointy, number, mix):
    """get_sample_points followed by `points[j] = mix_feature(pointx, points[j])` for every j >= 1 with mix[j-1] set
    (Segmentation/main_aug_final.py:186-192), fused: clean and adv are read once."""
    px, py = _dense(pointx), _dense(pointy)
    if py.stride() != px.stride():
        py = _like_layout(py, px)
    return [pointx] + ops.lerp_mix(px, py, number, [bool(f) for f in mix])


def get_sample_points(pointx, pointy, number):
    """Segmentation/attack_algo.py:108-118: [x, lerp(x,y,1/(n-1)), ..., y]; interior points in one launch."""
    px, py = _dense(pointx), _dense(pointy)
    if py.stride() != px.stride():
        py = _like_layout(py, px)
    inner = ops.lerp_points(px, py, number)
    return [pointx] + inner + [pointy]
