"""DeepLabv3+ (atrous ResNet-50/101 backbone) with the reference's split-forward protocol, on the library's kernels
(second slice; BASELINE config 4).

Host-side mirror of `Segmentation/network/`:
  * `_SimpleSegmentationModel.forward(input_dict)` — the dict dispatch of network/utils.py:14-47 (flag head / tail / clean,
    integer or "aspp|concat"_"head|tail" out_idx) that `Segmentation/attack_algo.py:40-105` calls back into;
  * `ResNet` backbone with `replace_stride_with_dilation` (backbone/resnet.py:109-304) and `DeepLabHeadV3Plus` with the
    aspp / concat split (_deeplab.py:28-80), `ASPP` (_deeplab.py:143-193);
  * state_dict keys identical to the reference's (`backbone.conv1.weight`, `backbone.layer3.0.downsample.1.running_mean`,
    `classifier.aspp.convs.4.1.weight`, `classifier.classifier.3.bias`, ...), so checkpoints interchange.
What differs is execution: bf16 channels-last activations, every 1x1 / 3x3 / atrous convolution on the implicit-GEMM MFMA
kernels (a Bottleneck is one autograd node, `resnet_s._BlockFn`), BatchNorm (+ReLU, +residual) fused, and the resize / max
pool / average pool / classifier / dropout / per-pixel cross-entropy layers as hand-written HIP kernels (`afan_seg.hip`).
fp32 (parity mode) runs the library's general fp32-arithmetic convolutions (afan_conv_f32.hip: f32 MFMA), like the
classification path; everything else is the same code.  No vendor convolution in either mode.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .resnet_s import (BatchNorm2d, Conv2d, NormalizeByChannelMeanStd, _BlockFn, _ConvFn, _Flags, _WgradStream, _accumulates_in_place,
                       _block_fast_path_ok, _block_params, _dense, _like_layout, _to_compute)

__all__ = ["deeplabv3plus_resnet50", "deeplabv3plus_resnet101", "deeplabv3_resnet50", "deeplabv3_resnet101", "DeepLabV3",
           "seg_criterion", "set_bn_momentum", "PolyLR", "MODELS"]


# ------------------------------------------------------------------------------------------- autograd nodes
class _UpsampleFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) (afan_upsample_bilinear_*)."""

    @staticmethod
    def forward(ctx, x, size):
        x = _dense(x)
        ctx.in_hw, ctx.like = x.shape[2:], x
        return ops.upsample_bilinear(x, size)

    @staticmethod
    def backward(ctx, g):
        y_like = ctx.like
        cl = ops.layout_of(y_like) == ops.AFAN_NHWC
        g = g.contiguous(memory_format=torch.channels_last) if cl else g.contiguous()
        return ops.upsample_bilinear_backward(g, ctx.in_hw), None


class _UpsampleCatFn(torch.autograd.Function):
    """torch.cat([low, F.interpolate(hi, low's size)], dim=1) (_deeplab.py:54-56) as one node: the resized tensor is written
    into its channel slice of the result and its gradient is read from there (ops.upsample_concat): no concat pass over the
    256-channel half, no copy of the gradient slice."""

    @staticmethod
    def forward(ctx, low, hi):
        low, hi = _dense(low), _dense(hi)
        ctx.c0, ctx.in_hw = low.shape[1], tuple(hi.shape[2:])
        return ops.upsample_concat(low, hi)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous(memory_format=torch.channels_last)
        d_low = g[:, :ctx.c0] if ctx.needs_input_grad[0] else None
        d_hi = ops.upsample_concat_backward(g, ctx.c0, ctx.in_hw) if ctx.needs_input_grad[1] else None
        return d_low, d_hi


def interpolate(x, size):
    size = (int(size[0]), int(size[1]))
    if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
        if tuple(x.shape[2:]) == size:
            return x
        return _UpsampleFn.apply(x, size)
    return nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)


class _BroadcastFn(torch.autograd.Function):
    """Bilinear resize of a 1x1 map = broadcast over the pixels (the ASPP pooling branch, _deeplab.py:139-141), cast from
    the branch's fp32 to the network's compute dtype on the way; the backward sums the pixels of every (image, channel)
    in fp32 — through the average-pool kernel (sum = mean * HW)."""

    @staticmethod
    def forward(ctx, x, size, channels_last, dtype):
        n, c = x.shape[:2]
        ctx.hw = size
        y = x.reshape(n, c, 1, 1).to(dtype).expand(n, c, size[0], size[1])
        return y.contiguous(memory_format=torch.channels_last if channels_last else torch.contiguous_format)

    @staticmethod
    def backward(ctx, g):
        s = ops.avgpool(_dense(g), out_fp32=True)
        return s * float(ctx.hw[0] * ctx.hw[1]), None, None, None


class _LinearSmallFn(torch.autograd.Function):
    """The pooling branch's 1x1 convolution on one vector per image, fp32 with the master weights (afan_linear_small_*)."""

    @staticmethod
    def forward(ctx, x, weight, want_pgrad):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.pg = want_pgrad
        return ops.linear_small(x, weight.detach())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        want_p = ctx.pg and ctx.needs_input_grad[1]
        dw, direct = None, False
        if want_p:
            direct = _accumulates_in_place(weight)
            dw = weight.grad if direct else torch.empty_like(weight, memory_format=torch.contiguous_format)
        dx = ops.linear_small_backward(g, x, weight.detach(), ctx.needs_input_grad[0], dw, accumulate=direct)
        return dx, (None if direct else dw), None


class _MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(k, stride, pad): the forward leaves each winner's position (one byte per output element) and the backward
    gathers from it instead of re-scanning the windows."""

    @staticmethod
    def forward(ctx, x, k=3, stride=2, pad=1):
        x = _dense(x)
        y, idx = ops.maxpool2d(x, k, stride, pad, want_idx=ctx.needs_input_grad[0])
        ctx.geom, ctx.shape = (k, stride, pad), tuple(x.shape)
        ctx.save_for_backward(idx if idx is not None else x)
        ctx.have_idx = idx is not None
        return y

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        k, stride, pad = ctx.geom
        g = _match(g, t)
        if ctx.have_idx:
            return ops.maxpool2d_backward(g, t, ctx.shape, k, stride, pad), None, None, None
        return ops.maxpool2d_backward(g, None, ctx.shape, k, stride, pad, x=t), None, None, None


def _match(g, ref):
    """g in ref's memory format (channels-last or contiguous)."""
    if ref.dim() == 4 and ops.layout_of(ref) == ops.AFAN_NHWC:
        return g.contiguous(memory_format=torch.channels_last)
    return g.contiguous()


class _AvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(1) with an fp32 result (see afan_linear_small_fwd for why the pooled side is fp32)."""

    @staticmethod
    def forward(ctx, x):
        x = _dense(x)
        ctx.like = x
        return ops.avgpool(x, out_fp32=True)

    @staticmethod
    def backward(ctx, g):
        return ops.avgpool_backward(g.float().contiguous(), ctx.like)


class _PointwiseFn(torch.autograd.Function):
    """nn.Conv2d(ci, num_classes, 1) with bias on a channels-last map, fp32 logits (afan_pointwise_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_pgrad):
        x = _dense(x)
        ctx.save_for_backward(x, weight)
        ctx.bias, ctx.pg = bias, want_pgrad
        return ops.pointwise_forward(x, weight.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        bias = ctx.bias
        want_p = ctx.pg and ctx.needs_input_grad[1]
        dw = db = None
        direct = False
        if want_p:
            direct = _accumulates_in_place(weight) and (bias is None or _accumulates_in_place(bias))
            if direct:
                dw, db = weight.grad, (bias.grad if bias is not None else None)
            else:
                dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
                db = torch.empty_like(bias) if bias is not None else None
        g = g.contiguous(memory_format=torch.channels_last)
        wd = weight.detach()
        if direct:      # the 100-us parameter-gradient launch off the critical path (resnet_s._WgradStream)
            dx = ops.pointwise_backward(g, x, wd, True, None, None) if ctx.needs_input_grad[0] else None
            _WgradStream.run(lambda dw=dw, db=db: ops.pointwise_backward(g, x, wd, False, dw, db, accumulate=True), g, x)
            dw = db = None
        else:
            dx = ops.pointwise_backward(g, x, wd, ctx.needs_input_grad[0], dw, db, accumulate=False)
        return dx, dw, db, None


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, mask):
        x = _dense(x)
        y, used = ops.dropout(x, p, mask)
        ctx.p, ctx.mask, ctx.used = p, mask, used
        return y

    @staticmethod
    def backward(ctx, g):
        g = _dense(g)
        if ctx.mask is not None:
            g = _match(g, ctx.mask)
        return ops.dropout(g, ctx.p, ctx.mask, ctx.used)[0], None, None


class _CE2dFn(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index=I)(logits [N,C,H,W], target [N,H,W]) and grad_scale * its gradient in one pass
    (afan_ce2d).  backward with the library's cached scalar 1 (ops.one) hands the stored gradient over as it is."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index, grad_scale):
        logits = _dense(logits)
        loss, dl = ops.ce2d(logits, target, ignore_index, grad_scale)
        ctx.save_for_backward(dl)
        ctx.scale = grad_scale
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if g.data_ptr() == ops.one(g.device).data_ptr():
            return dl, None, None, None
        return dl * (g / ctx.scale), None, None, None


class _CE2dUpFn(torch.autograd.Function):
    """criterion(F.interpolate(logits, labels' size, 'bilinear'), labels) on the LOW-resolution logits (afan_ce2d_upsampled):
    resize, loss and the gradient back at low resolution in one kernel."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index, grad_scale):
        logits = _dense(logits)
        loss, dl = ops.ce2d_upsampled(logits, target, ignore_index, grad_scale)
        ctx.save_for_backward(dl)
        ctx.scale = grad_scale
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if g.data_ptr() == ops.one(g.device).data_ptr():
            return dl, None, None, None
        return dl * (g / ctx.scale), None, None, None


class LowResLogits:
    """What `model(input_dict)` returns instead of image-size logits when the dict carries "low_res": True: the classifier's
    own [N, C, h, w] output plus the size utils.py:30,45 would resize it to.  seg_criterion's callable takes it directly."""
    __slots__ = ("logits", "size")

    def __init__(self, logits, size):
        self.logits, self.size = logits, (int(size[0]), int(size[1]))


def seg_criterion(criterion):
    """`criterion` as the step applies it to the upsampled logits: a plain nn.CrossEntropyLoss(ignore_index=I,
    reduction='mean') becomes the one-pass HIP form on GPU fp32 logits of up to 32 classes; anything else is unchanged.
    The returned callable also takes grad_scale= (the weight of this term in the joint loss, main_aug_final.py:216):
    the gradient it stores is pre-scaled, for `torch.autograd.backward(loss, ops.one(dev))`."""
    if not (type(criterion) is nn.CrossEntropyLoss and criterion.weight is None and criterion.reduction == "mean"
            and getattr(criterion, "label_smoothing", 0.0) == 0.0):
        return criterion
    ign = criterion.ignore_index

    def ce(out, y, grad_scale=1.0):
        if isinstance(out, LowResLogits):
            if tuple(y.shape[1:]) == out.size and y.dtype == torch.int64 and ops.ce2d_upsampled_ok(out.logits, out.size):
                return _CE2dUpFn.apply(out.logits, y, ign, float(grad_scale))
            out = interpolate(out.logits, out.size)
        if (out.is_cuda and out.dtype == torch.float32 and out.dim() == 4 and out.shape[1] <= ops.CE2D_MAX_CLASSES
                and y.dtype == torch.int64 and y.dim() == 3):
            return _CE2dFn.apply(out, y, ign, float(grad_scale))
        return criterion(out, y)
    ce.fused = True
    ce.low_res = os.environ.get("AFAN_CE_LOWRES", "1") != "0"      # (0: always resize first — A/B)
    return ce


class _Stem7Fn(torch.autograd.Function):
    """conv1 (7x7 / 2, 3 -> 64) on the image.  im2col (afan_conv_stem7_im2col: 152 columns, the 147 taps in the weights'
    KRSC order + 5 zeros) turns forward and weight gradient into 1x1 problems for the MFMA kernels; the column tensor is
    kept for the backward.  An image that carries a gradient (Detection's image-level perturbation,
    train_aug_sat_muti_advt.py:82-95) takes the same forward — ~50 us instead of 169 on the general kernel at 600 x 904 —
    and gets its gradient as dy x W (a 1x1 problem on the MFMA kernel, 64 -> 152 columns) followed by
    afan_conv_stem7_col2im: ~65 us instead of 527 (a 3-channel output uses 3 of the general kernel's 32 MFMA columns).
    AFAN_STEM7_DIRECT=1 selects the direct FMA kernels (A/B)."""
    DIRECT = os.environ.get("AFAN_STEM7_DIRECT", "0") == "1"

    @staticmethod
    def forward(ctx, x, w_master, w_lp, want_wgrad):
        ctx.w_master, ctx.want = w_master, want_wgrad
        ctx.w_lp, ctx.in_hw = (w_lp, tuple(x.shape[2:])) if ctx.needs_input_grad[0] else (None, None)
        if _Stem7Fn.DIRECT:
            ctx.save_for_backward(x)
            return ops.conv_stem7_fwd(x, w_lp)
        cols = ops.conv_stem7_im2col(x)
        k = cols.shape[1]
        wp = torch.zeros((64, k, 1, 1), dtype=torch.bfloat16, device=x.device)
        wp.view(64, k)[:, :147] = w_lp.permute(0, 2, 3, 1).reshape(64, 147)        # KRSC memory: a plain row copy
        # (image gradient wanted: dcols = dy x W as a 1x1 problem with the transposed operand, then col2im)
        ctx.wp_t = wp.view(64, k).t().contiguous().view(k, 64, 1, 1).contiguous(memory_format=torch.channels_last) \
            if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(cols)
        return ops.conv_fwd(cols, wp.contiguous(memory_format=torch.channels_last), 1)

    @staticmethod
    def backward(ctx, gy):
        (saved,) = ctx.saved_tensors
        gw = None
        if ctx.want and ctx.needs_input_grad[1]:
            gy = gy.contiguous(memory_format=torch.channels_last)
            wm = ctx.w_master
            direct = _accumulates_in_place(wm) and wm.grad.is_contiguous(memory_format=torch.channels_last)
            if _Stem7Fn.DIRECT:
                if direct:
                    ops.conv_stem7_wgrad(saved, gy, wm.grad, accumulate=True)
                else:
                    gw = ops.conv_stem7_wgrad(saved, gy)
            else:
                g = ops.conv_wgrad(saved, gy, 1, 1).view(64, -1)[:, :147]               # fp32 [64, 152] -> the 147 taps
                if direct:
                    wm.grad.permute(0, 2, 3, 1).reshape(64, 147).add_(g)                 # (a view of the KRSC arena slice)
                else:
                    gw = g.reshape(64, 7, 7, 3).permute(0, 3, 1, 2)
        gx = None
        if ctx.needs_input_grad[0] and ctx.w_lp is not None:
            gy = gy.contiguous(memory_format=torch.channels_last)
            if getattr(ctx, "wp_t", None) is not None and gy.dtype == torch.bfloat16:
                gx = ops.conv_stem7_col2im(ops.conv_fwd(gy, ctx.wp_t, 1), ctx.in_hw)       # 1x1 on the MFMA kernel + gather
            else:
                gx = ops.conv_general_dgrad(gy, ctx.w_lp, ctx.in_hw, 2, 3, 1)
        return gx, gw, None, None


# ---------------------------------------------------------------------------------------------------- layers
class StemConv(Conv2d):
    """backbone/resnet.py:143: nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)."""

    def forward(self, x):
        x = _to_compute(x, self.compute_dtype)
        w = self.lp_weight()
        if ops.conv_stem7_ok(x, w, self.stride, self.padding) and w.is_contiguous(memory_format=torch.channels_last) \
                and (not x.requires_grad or not _Stem7Fn.DIRECT):
            return _Stem7Fn.apply(x, self.weight, w.detach(), _Flags.param_grads)
        return super().forward(x)

    def forward_with_stats(self, x, bn):
        return self.forward(x), None


class ClassifierConv(Conv2d):
    """_deeplab.py:45: nn.Conv2d(256, num_classes, 1) — with bias, fp32 logits."""

    @property
    def own_kernel(self):       # resnet_s.general_convs: the pointwise kernel takes it on channels-last maps
        return self.in_channels % 8 == 0 and self.out_channels <= 32 and self.in_channels * self.out_channels * 4 <= 64 * 1024

    def forward(self, x):
        x = _to_compute(x, self.compute_dtype)
        if ops.pointwise_supported(x, self.out_channels) and self.weight.dtype == torch.float32:
            return _PointwiseFn.apply(x, self.weight, self.bias, _Flags.param_grads)
        xf = x if x.dtype == torch.float32 else x.float()        # NCHW maps: the general fp32 kernel, bias in its epilogue
        return _ConvFn.apply(xf, self.weight, self.weight.detach(), None, self.stride, self.padding, _Flags.param_grads,
                             None, self.dilation, self.bias)


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and (self.kernel_size, self.stride, self.padding) == (3, 2, 1):
            return _MaxPoolFn.apply(x)
        return super().forward(x)


class Dropout(nn.Dropout):
    """nn.Dropout(p) with the mask drawn in-kernel (device generator) — or supplied by the caller through `.mask`
    (uint8 tensor, consumed by the next forward; parity tests)."""
    mask = None

    def forward(self, x):
        if not self.training or (self.p == 0 and self.mask is None):
            return x
        if not x.is_cuda:
            return super().forward(x)
        mask = self.mask
        if mask is not None:
            mask = _match(mask.to(x.device, torch.uint8), _dense(x))
        G = _Flags.bn_groups
        if G != 1 and mask is None:
            # two concatenated passes (resnet_s.bn_groups): each is its own model(...) call in the reference, with its own draw
            n = x.shape[0] // G
            return torch.cat([_DropoutFn.apply(x[g * n:(g + 1) * n], float(self.p), None) for g in range(G)], dim=0)
        return _DropoutFn.apply(x, float(self.p), mask)


def _enter(x, dtype, channels_last):
    """A caller's tensor (e.g. the fp32 x_adv of a PGD step) in the network's compute dtype and activation layout."""
    if channels_last and x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return _to_compute(x, dtype)


def _cbr(conv, bn, x, relu=True):
    """conv -> BatchNorm (moments from the convolution's epilogue where the kernels take the shape) -> ReLU."""
    out, st = conv.forward_with_stats(x, bn)
    return bn.fused(out, None, relu, st)


class Bottleneck(nn.Module):
    """backbone/resnet.py:76-119: 1x1 -> 3x3 (stride, dilation) -> 1x1 x4, `downsample` = 1x1 conv + BN when the shape
    changes.  One autograd node on the bf16 channels-last path (resnet_s._BlockFn)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, stride=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, kernel_size=1, stride=1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self._sc_kind = "conv" if downsample is not None else "identity"

    @property
    def shortcut(self):          # the name resnet_s._BlockFn uses for the projection branch
        return self.downsample

    def _chain(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]

    def forward(self, x):
        x = _to_compute(x, self.conv1.compute_dtype)
        if _Flags.block_fusion and _block_fast_path_ok(self, x):
            return _BlockFn.apply(x, self, _Flags.param_grads, *_block_params(self))
        out = _cbr(self.conv1, self.bn1, x)
        out = _cbr(self.conv2, self.bn2, out)
        out, st = self.conv3.forward_with_stats(out, self.bn3)
        res = x if self.downsample is None else _cbr(self.downsample[0], self.downsample[1], x, relu=False)
        return self.bn3.fused(out, res, True, st)


class ResNet(nn.Module):
    """backbone/resnet.py:109-304 (Bottleneck variants): atrous ResNet with the head / tail / clean dispatch."""

    def __init__(self, layers, replace_stride_with_dilation=(False, False, False)):
        super().__init__()
        self.normal = NormalizeByChannelMeanStd(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
        self.inplanes, self.dilation = 64, 1
        self.conv1 = StemConv(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(512, layers[3], stride=2, dilate=replace_stride_with_dilation[2])
        for m in self.modules():                               # backbone/resnet.py:160-165
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        downsample, previous_dilation = None, self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    def _stem(self, x):
        x = self.normal(x)
        x = self.bn1.fused(self.conv1(x), None, True)
        return self.maxpool(x)

    def forward(self, input_dict):
        stages = [self.layer1, self.layer2, self.layer3, self.layer4]
        flag, out = input_dict["flag"], {}
        if flag in ("head", "clean"):
            last = 4 if flag == "clean" else input_dict["out_idx"]
            assert last in (1, 2, 3, 4)
            x = self.layer1(self._stem(input_dict["x"]))
            out["low_level"] = x
            for st in stages[1:last]:
                x = st(x)
            out["out"] = x
            return out
        assert flag == "tail" and input_dict["out_idx"] in (1, 2, 3, 4)
        x = _enter(input_dict["adv"], self.conv1.compute_dtype, self.normal.channels_last)
        for st in stages[input_dict["out_idx"]:]:
            x = st(x)
        out["out"] = x
        out["low_level"] = input_dict["low_level_feat"]
        return out


class ASPPConv(nn.Sequential):
    """_deeplab.py:143-150."""

    def __init__(self, in_channels, out_channels, dilation):
        super().__init__(Conv2d(in_channels, out_channels, 3, padding=dilation, dilation=dilation, bias=False),
                         BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def forward(self, x):
        return _cbr(self[0], self[1], x)


class _ConvMultiFn(torch.autograd.Function):
    """The atrous 3x3 branches of ASPP (_deeplab.py:173-176) as ONE forward launch (afan_conv_fwd_multi_nhwc_bf16: same input,
    same shapes, their own weights / dilations / BatchNorm moment accumulators).  Backward: the input gradients chained
    through the dgrad epilogue's addend (no separate sums), each branch's weight gradient into its arena view."""

    @staticmethod
    def forward(ctx, x, convs, want_wgrad, reqs, *w_masters):
        ctx.convs, ctx.want_wgrad = convs, want_wgrad
        ws = [c.lp_weight().detach() for c in convs]
        ys, sts = ops.conv_fwd_multi(x, ws, convs[0].stride[0], [c.dilation[0] for c in convs],
                                     None if reqs is None else [r[0] for r in reqs], groups=_Flags.bn_groups if reqs is not None else 1)
        if reqs is not None:
            for r, st in zip(reqs, sts):
                r.append(st)
        ctx.save_for_backward(x, *ws)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        x, *ws = ctx.saved_tensors
        convs = ctx.convs
        gx, gws = None, []
        for i, (c, w_lp, gy) in enumerate(zip(convs, ws, gys)):
            gw = None
            if gy is not None:
                gy = _like_layout(gy, x)
                k, st, dil = w_lp.shape[2], c.stride[0], c.dilation[0]
                if ctx.needs_input_grad[0]:
                    gx = ops.conv_dgrad(gy, c.lp_weight_t(), x.shape[2:], st, addend=gx, dilation=dil)
                if ctx.want_wgrad and ctx.needs_input_grad[4 + i]:
                    wm = c.weight
                    if _accumulates_in_place(wm) and wm.grad.is_contiguous(memory_format=torch.channels_last):
                        g_ = wm.grad
                        _WgradStream.run(lambda gy=gy, k=k, st=st, g_=g_, dil=dil: ops.conv_wgrad(x, gy, k, st, g_, accumulate=True, dilation=dil), x, gy)
                    else:
                        gw = ops.conv_wgrad(x, gy, k, st, dilation=dil)
            gws.append(gw)
        return (gx, None, None, None, *gws)


def _multi_branch_ok(x, mods):
    """The bf16 channels-last training step, every branch a bias-free conv -> train-mode BatchNorm of the same shape."""
    convs, bns = [m[0] for m in mods], [m[1] for m in mods]
    c0 = convs[0]
    if not (x.is_cuda and c0.compute_dtype == torch.bfloat16 and x.dtype == torch.bfloat16):
        return False
    if not all(b.training and b.track_running_stats for b in bns):
        return False
    if any(c.bias is not None or c.stride != c0.stride or c.weight.shape != c0.weight.shape or c.kernel_size != (3, 3)
           or c.padding != c.dilation for c in convs):
        return False
    return ops.conv_fwd_multi_ok(x, [c.lp_weight() for c in convs], c0.stride[0]) and \
        ops.conv_wgrad_supported(x.shape[1], c0.out_channels, 3, c0.stride[0], (x.shape[0], x.shape[2], x.shape[3]))


class ASPPPooling(nn.Sequential):
    """_deeplab.py:152-163: global average pool -> 1x1 conv -> BN -> ReLU -> resize back (a broadcast)."""

    def __init__(self, in_channels, out_channels):
        super().__init__(nn.AdaptiveAvgPool2d(1), Conv2d(in_channels, out_channels, 1, bias=False), BatchNorm2d(out_channels),
                         nn.ReLU(inplace=True))

    def forward(self, x):
        size = tuple(x.shape[-2:])
        if not x.is_cuda:
            y = nn.functional.relu(self[2](self[1](self[0](x))))
            return nn.functional.interpolate(y, size=size, mode="bilinear", align_corners=False)
        cl = ops.layout_of(x) == ops.AFAN_NHWC
        p = _AvgPoolFn.apply(x)                                      # fp32 [N, C, 1, 1]
        n, conv, bn = p.shape[0], self[1], self[2]
        if ops.linear_small_ok(p, conv.weight):
            y = _LinearSmallFn.apply(p.reshape(n, -1), conv.weight, _Flags.param_grads).reshape(n, conv.out_channels, 1, 1)
            y = bn.fused(y, None, True)                              # BatchNorm over the N images, in fp32
        else:
            y = _cbr(conv, bn, p.to(x.dtype))
        return _BroadcastFn.apply(y, size, cl, x.dtype)


class ASPP(nn.Module):
    """_deeplab.py:165-193."""
    MULTI = os.environ.get("AFAN_ASPP_MULTI", "1") != "0"      # 0: one launch per atrous branch (A/B)

    def __init__(self, in_channels, atrous_rates):
        super().__init__()
        oc = 256
        mods = [nn.Sequential(Conv2d(in_channels, oc, 1, bias=False), BatchNorm2d(oc), nn.ReLU(inplace=True))]
        mods += [ASPPConv(in_channels, oc, r) for r in atrous_rates]
        mods.append(ASPPPooling(in_channels, oc))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(Conv2d(5 * oc, oc, 1, bias=False), BatchNorm2d(oc), nn.ReLU(inplace=True), Dropout(0.1))

    def forward(self, x, pre_dropout=False):
        x = _to_compute(x, self.convs[0][0].compute_dtype)
        atrous = [m for m in list(self.convs)[1:] if isinstance(m, ASPPConv)]
        if ASPP.MULTI and 2 <= len(atrous) <= 4 and len(atrous) == len(self.convs) - 2 and _multi_branch_ok(x, atrous):
            # the atrous branches in one launch: at 2 images per GPU each fills a quarter of the chip for 92 us
            convs, bns = [m[0] for m in atrous], [m[1] for m in atrous]
            reqs = [[b.running_mean, None] for b in bns]
            raws = _ConvMultiFn.apply(x, convs, _Flags.param_grads, reqs, *[c.weight for c in convs])
            mid = [b.fused(r, None, True, q[2]) for b, r, q in zip(bns, raws, reqs)]
            res = [_cbr(self.convs[0][0], self.convs[0][1], x)] + mid + [self.convs[-1](x)]
        else:
            res = [_cbr(self.convs[0][0], self.convs[0][1], x)] + [c(x) for c in list(self.convs)[1:]]
        res = torch.cat(res, dim=1)
        pre = _cbr(self.project[0], self.project[1], res)
        return pre if pre_dropout else self.project[3](pre)


class DeepLabHeadV3Plus(nn.Module):
    """_deeplab.py:28-90, with the aspp / concat split-forward return types."""

    def __init__(self, in_channels, low_level_channels, num_classes, aspp_dilate=(12, 24, 36)):
        super().__init__()
        self.project = nn.Sequential(Conv2d(low_level_channels, 48, 1, bias=False), BatchNorm2d(48), nn.ReLU(inplace=True))
        self.aspp = ASPP(in_channels, aspp_dilate)
        self.classifier = nn.Sequential(Conv2d(304, 256, 3, padding=1, bias=False), BatchNorm2d(256), nn.ReLU(inplace=True),
                                        ClassifierConv(256, num_classes, 1))
        _init_head(self)

    def _classify(self, cat):
        return self.classifier[3](_cbr(self.classifier[0], self.classifier[1], cat))

    channels_last = False     # set by DeepLabV3.set_channels_last
    CAT_FUSED = os.environ.get("AFAN_SEG_CAT_FUSED", "1") != "0"     # resize written into the concat's channel slice (A/B knob)

    def _concat(self, low, hi):
        hi = _enter(hi, self.project[0].compute_dtype, self.channels_last)
        if (self.CAT_FUSED and self.channels_last and hi.is_cuda and hi.dtype == low.dtype and hi.dtype in (torch.float32, torch.bfloat16)
                and tuple(hi.shape[2:]) != tuple(low.shape[2:]) and ops.layout_of(hi) == ops.AFAN_NHWC
                and ops.layout_of(low) == ops.AFAN_NHWC):
            return _UpsampleCatFn.apply(low, hi)
        return torch.cat([low, interpolate(hi, low.shape[2:])], dim=1)

    def forward(self, feature, return_type=None):
        dt, cl = self.project[0].compute_dtype, self.channels_last
        if return_type == "aspp_head":
            return self.aspp(_enter(feature["out"], dt, cl))
        if return_type == "concat_tail":
            return self._classify(_enter(feature["adv"], dt, cl))
        low = _cbr(self.project[0], self.project[1], _enter(feature["low_level"], dt, cl))
        if return_type == "aspp_tail":
            return self._classify(self._concat(low, feature["adv"]))
        cat = self._concat(low, self.aspp(_enter(feature["out"], dt, cl)))
        if return_type == "concat_head":
            return cat
        assert return_type is None
        return self._classify(cat)


class DeepLabHead(nn.Module):
    """_deeplab.py:92-115 (DeepLabv3: no decoder)."""

    def __init__(self, in_channels, num_classes, aspp_dilate=(12, 24, 36)):
        super().__init__()
        self.classifier = nn.Sequential(ASPP(in_channels, aspp_dilate), Conv2d(256, 256, 3, padding=1, bias=False),
                                        BatchNorm2d(256), nn.ReLU(inplace=True), ClassifierConv(256, num_classes, 1))
        _init_head(self)

    channels_last = False

    def forward(self, feature):
        c = self.classifier
        return c[4](_cbr(c[1], c[2], c[0](_enter(feature["out"], c[1].compute_dtype, self.channels_last))))


def _init_head(head):          # _deeplab.py:82-89
    for m in head.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class DeepLabV3(nn.Module):
    """network/utils.py:8-47 `_SimpleSegmentationModel`: the dict dispatch the A-FAN operators call back into."""

    def __init__(self, backbone, classifier):
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier
        self.compute_dtype = torch.float32
        self.channels_last = False

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, Conv2d):
                m.compute_dtype = dtype
            elif isinstance(m, NormalizeByChannelMeanStd):
                m.out_dtype = dtype
        return self

    def set_channels_last(self, on=True):
        self.channels_last = bool(on)
        for m in self.modules():
            if isinstance(m, NormalizeByChannelMeanStd):
                m.channels_last = self.channels_last
        self.classifier.channels_last = self.channels_last
        return self

    def _up(self, x, input_shape, low_res=False):
        if low_res and x.is_cuda and x.dtype == torch.float32:
            return LowResLogits(x, input_shape)        # the criterion resizes inside its own kernel (seg_criterion)
        return interpolate(x, input_shape)

    def forward_clean_folded(self, x, se_idx, sd_idx, pgd0=False, cut=False):
        return _folded_clean_forward(self, x, se_idx, sd_idx, pgd0, cut)

    def fold_ok(self, x):
        """Can the iteration's three clean forwards run as one (forward_clean_folded)?"""
        return bool(self.channels_last and x.is_cuda and self.training and isinstance(self.classifier, DeepLabHeadV3Plus))

    def forward(self, input_dict):
        flag = input_dict["flag"]
        if flag == "head":
            return self.backbone(input_dict)
        assert flag in ("tail", "clean")
        idx = input_dict["out_idx"]
        lr = bool(input_dict.get("low_res", False))
        if type(idx) == int:
            features = self.backbone(input_dict)
            return self._up(self.classifier(features), input_dict["x"].shape[-2:], lr)
        if idx in ("aspp_head", "concat_head"):
            features = self.backbone(input_dict)
            features["adv"] = self.classifier(features, return_type=idx)
            return features
        assert idx in ("aspp_tail", "concat_tail")
        return self._up(self.classifier(input_dict["adv"], return_type=idx), input_dict["x"].shape[-2:], lr)


class _FoldedClean:
    """What DeepLabV3.forward_clean_folded hands back (see there)."""
    __slots__ = ("low", "low_graph", "fm_se", "dec", "logits", "_recs", "se_in", "low_in", "sd_t")

    def replay_sd_pgd0_bn(self):
        """With the first PGD passes folded in as well: the decoder-PGD's first pass (main_aug_final.py:179-181) updates the
        decoder's BatchNorms AFTER the SE loop — call this between the two PGD loops."""
        self._recs[1].replay()

    def replay_deferred_bn(self):
        """Apply the running-statistics updates that belong to the reference's `o0` forward (main_aug_final.py:193), which
        comes AFTER the two PGD loops."""
        for r in self._recs:
            r.replay()


def _folded_clean_forward(self, x, se_idx, sd_idx, pgd0=False, cut=False):
    """The three clean forwards of a Segmentation A-FAN iteration as ONE pass (the fold of DESIGN section 4, for
    main_aug_final.py:166-193): `model(head, out_idx=se_idx)` (:166), `model(clean, out_idx=sd_idx + "_head")` (:167) and
    `model(clean, out_idx=0)` (:193) evaluate the same layers on the same images with the same weights — the backbone up
    to the SE point three times, layer4 + ASPP twice.  One pass with its autograd graph yields all of their values:
      * `fm_se`, `low`: the SE point's feature map and the low-level feature (graph tensors: the decoders of the
        perturbed forwards differentiate through `low`, exactly as they do through the head passes' graphs in the reference);
      * `dec`: the decoder-PGD input dict (`adv` = the SD point's clean feature, with its OWN dropout draw);
      * `logits`: the clean forward's output (a second dropout draw on the same pre-dropout ASPP output).
    BatchNorm side effects in the reference's order: stem .. SE point: three updates now (nothing else touches them);
    SE point .. SD point: one now (:167), one deferred (:193 comes after the PGD loops); after the SD point: deferred.

    pgd0=True (no random start, at least one PGD step): the FIRST pass of both PGD loops is this pass too — PGD starts at
    the clean feature (attack_algo.py:42-43), so its first tail forward / cross-entropy / input gradient are the clean
    forward's own logits, loss and d(loss)/d(feature).  The graph is then cut at the SE point and at the low-level feature
    (`se_in`, `low_in`: leaves whose gradients the caller feeds back into the head graph at the end) so that the clean
    loss can be back-propagated through the tail at once; `sd_t` is the SD point's tensor (retain_grad).  One dropout draw
    then serves :167, :193 and the SE loop's first pass, where the reference draws three masks — so seg_train_step only
    asks for pgd0 when no Dropout with p > 0 is active (seg_attack_algo._dropout_active); with p = 0 the passes are identical.
    BatchNorm: SE .. SD point two updates now (:167 and the SE loop's first pass) + :193 deferred; after the SD point one
    now (SE loop's first pass), one at `replay_sd_pgd0_bn()` (decoder loop's first pass), one at `replay_deferred_bn()`.
    Channels-last kernels only (the repeat count is a feature of those launches)."""
    bb, head = self.backbone, self.classifier
    assert isinstance(head, DeepLabHeadV3Plus) and sd_idx in ("aspp", "concat") and se_idx in (1, 2, 3, 4)
    stages = [bb.layer1, bb.layer2, bb.layer3, bb.layer4]
    dt, cl = head.project[0].compute_dtype, head.channels_last
    out = _FoldedClean()
    with ops.bn_running_updates(3):
        h = bb.layer1(bb._stem(x))
        low = h
        for st in stages[1:se_idx]:
            h = st(h)
    out.fm_se, out.low, out.low_graph = h, low, low
    out.se_in = out.low_in = out.sd_t = None
    low_dec = low
    if pgd0 or cut:
        # cut = True (seg_train_phases): the same cuts without the pgd0 fold, so that the joint backward can run in two
        # parts (tail, then head) and a data-parallel caller can exchange the tail's gradients in between.  out.low stays the
        # leaf the perturbed forwards and PGD passes read too (their gradients w.r.t. the low-level feature accumulate in
        # low_in.grad and enter the head with the rest); out.low_graph is the head-graph tensor the caller back-propagates from.
        out.se_in = h = h.detach().requires_grad_(True)
        out.low_in = low_dec = low.detach().requires_grad_(True)
        out.low = low_dec
    drop = head.aspp.project[3]
    with ops.bn_running_updates(2 if pgd0 else 1), ops.record_bn_updates() as rec_mid:
        for st in stages[se_idx:]:
            h = st(h)
        pre = head.aspp(_enter(h, dt, cl), pre_dropout=True)
        if sd_idx == "concat":
            low_p = _cbr(head.project[0], head.project[1], _enter(low_dec, dt, cl))
    with ops.bn_running_updates(1 if pgd0 else 0), ops.record_bn_updates() as rec_late:
        if sd_idx == "aspp":
            low_p = _cbr(head.project[0], head.project[1], _enter(low_dec, dt, cl))
        if pgd0:
            sd_t = drop(pre) if sd_idx == "aspp" else head._concat(low_p, drop(pre))
            sd_t.retain_grad()
            out.sd_t, adv = sd_t, sd_t.detach()
            logits = head._classify(head._concat(low_p, sd_t) if sd_idx == "aspp" else sd_t)
        else:
            if sd_idx == "aspp":
                adv = drop(pre.detach())                      # :167's dropout draw
            else:
                adv = head._concat(low_p.detach(), drop(pre.detach()))
            logits = head._classify(head._concat(low_p, drop(pre)))   # :193's own draw
    out.dec = {"out": h, "low_level": out.low, "adv": adv}
    out.logits = self._up(logits, x.shape[-2:])
    out._recs = (rec_mid, rec_late)
    return out


def set_bn_momentum(model, momentum=0.1):
    """utils/utils.py:26-29 (main_aug_final.py:77 sets the backbone's BatchNorm momentum to 0.01)."""
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = momentum


class PolyLR(torch.optim.lr_scheduler._LRScheduler):
    """utils/scheduler.py:3-12."""

    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1, min_lr=1e-6):
        self.power, self.max_iters, self.min_lr = power, max_iters, min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [max(base_lr * (1 - self.last_epoch / self.max_iters) ** self.power, self.min_lr) for base_lr in self.base_lrs]


def _segm_resnet(name, layers, num_classes, output_stride):
    """network/modeling.py:6-29."""
    if output_stride == 8:
        rswd, aspp_dilate = (False, True, True), (12, 24, 36)
    else:
        rswd, aspp_dilate = (False, False, True), (6, 12, 18)
    backbone = ResNet(layers, replace_stride_with_dilation=rswd)
    if name == "deeplabv3plus":
        classifier = DeepLabHeadV3Plus(2048, 256, num_classes, aspp_dilate)
    else:
        classifier = DeepLabHead(2048, num_classes, aspp_dilate)
    return DeepLabV3(backbone, classifier)


def deeplabv3plus_resnet101(num_classes=21, output_stride=8, pretrained_backbone=False):
    """network/modeling.py:124-132.  pretrained_backbone: there is no network here — load a state_dict instead."""
    if pretrained_backbone:
        raise NotImplementedError("no download in this build: load the ImageNet weights with load_state_dict")
    return _segm_resnet("deeplabv3plus", (3, 4, 23, 3), num_classes, output_stride)


def deeplabv3plus_resnet50(num_classes=21, output_stride=8, pretrained_backbone=False):
    if pretrained_backbone:
        raise NotImplementedError("no download in this build: load the ImageNet weights with load_state_dict")
    return _segm_resnet("deeplabv3plus", (3, 4, 6, 3), num_classes, output_stride)


def deeplabv3_resnet101(num_classes=21, output_stride=8, pretrained_backbone=False):
    if pretrained_backbone:
        raise NotImplementedError("no download in this build: load the ImageNet weights with load_state_dict")
    return _segm_resnet("deeplabv3", (3, 4, 23, 3), num_classes, output_stride)


def deeplabv3_resnet50(num_classes=21, output_stride=8, pretrained_backbone=False):
    if pretrained_backbone:
        raise NotImplementedError("no download in this build: load the ImageNet weights with load_state_dict")
    return _segm_resnet("deeplabv3", (3, 4, 6, 3), num_classes, output_stride)


MODELS = {"deeplabv3plus_resnet101": deeplabv3plus_resnet101, "deeplabv3plus_resnet50": deeplabv3plus_resnet50,
          "deeplabv3_resnet101": deeplabv3_resnet101, "deeplabv3_resnet50": deeplabv3_resnet50}
