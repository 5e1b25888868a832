"""Faster-RCNN on a frozen-BatchNorm ResNet-101 with the reference's split-forward dict protocol, on the library's kernels
(BASELINE configs[4]).

Host-side mirror of `Detection/model.py:18-185` (`Model`: flag head / tail / clean; out_idx 1-3, 'rpn_head' / 'rpn_tail',
'roi_head' / 'roi_tail'), `Detection/backbone/resnet101_ori.py:130-262` (the dict-dispatch ResNet whose `layer4` is the
detection head's `hidden` module, `backbone/resnet101.py:13-35`), `Detection/rpn/region_proposal_network.py` (anchors,
label assignment, host-`randperm` sampling, proposals through NMS), `Detection/roi/pooler.py` and `Detection/bbox.py`, with
the reference's state_dict layout — including its aliases: `_bn_modules.<i>.*` (model.py:27-28) and `detection.hidden.*`
(the same tensors as `features.layer4.*`).  Seeded construction reproduces the reference's tensors bit for bit
(tests/golden/det_frcnn_*: every module is created, default-initialised and re-initialised in the reference's order).

What differs is execution: every convolution runs on libafan_hip.so (tuned bf16 MFMA kernels or the general f32-MFMA ones),
a frozen BatchNorm (eval mode, no parameter gradients: model.py:31-35,46-47) is ONE fused affine(+residual)(+ReLU) launch
forward and one backward (`afan_bn_apply`, `afan_affine_relu_bwd`), pooling / ROIAlign / NMS are the library's kernels
(`afan_maxpool2d_*`, `afan_roi_align_*`, `afan_nms`), the linear heads run as 1x1 problems on the general kernel.  Label
assignment, sampling and the box arithmetic are small torch ops on the device, with the reference's host synchronisations
(`nonzero`, `randperm` on the CPU generator — so the draws can be matched)."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .deeplab import MaxPool2d, StemConv, _MaxPoolFn, _enter
from .det_ops import (box_assign, box_decode_clip, fg_bg_draw, fg_bg_sample, labels_limit_, nms, per_image_losses, proposal_rows,
                      roi_align, sample_lists)
from .resnet_s import (Conv2d, NormalizeByChannelMeanStd, _accumulates_in_place, _ConvFn, _dense, _Flags, _like_layout, _linear, dgrad_only,
                       _own_conv_ok, _to_compute, _WgradStream)

__all__ = ["Model", "ResNet101", "RegionProposalNetwork", "FrozenBatchNorm2d", "fasterrcnn_resnet101"]


def _int_tensor(v, device):
    """The reference's dicts carry sizes as [[v]] tensors on the device (model.py:104-108, :337): the same tensor, with the
    Python value riding along so that this package's own reader needs no device read (`_int_of`)."""
    t = torch.tensor([[v]], device=device)
    t._afan_int = int(v)
    return t


def _int_of(t):
    v = getattr(t, "_afan_int", None)
    return v if v is not None else t[0].item()


# ------------------------------------------------------------------------------------------------- box arithmetic (bbox.py)
def _centre(b):
    return torch.stack([(b[..., 0] + b[..., 2]) / 2, (b[..., 1] + b[..., 3]) / 2, b[..., 2] - b[..., 0], b[..., 3] - b[..., 1]], dim=-1)


def _corners(c):
    return torch.stack([c[..., 0] - c[..., 2] / 2, c[..., 1] - c[..., 3] / 2, c[..., 0] + c[..., 2] / 2, c[..., 1] + c[..., 3] / 2], dim=-1)


def box_apply(src, t):
    """bbox.py:54-64 `apply_transformer`."""
    s = _centre(src)
    return _corners(torch.stack([t[..., 0] * s[..., 2] + s[..., 0], t[..., 1] * s[..., 3] + s[..., 1],
                                 torch.exp(t[..., 2]) * s[..., 2], torch.exp(t[..., 3]) * s[..., 3]], dim=-1))


def box_clip_(b, right, bottom):
    """bbox.py:89-92 (left = top = 0), in place."""
    b[..., [0, 2]] = b[..., [0, 2]].clamp(min=0, max=right)
    b[..., [1, 3]] = b[..., [1, 3]].clamp(min=0, max=bottom)
    return b


# ------------------------------------------------------------------------------------------------------ fused affine layers
class _AffineFn(torch.autograd.Function):
    """y = [relu](x * alpha[c] + beta[c] [+ residual]) with coefficients that receive no gradient through this node (a frozen
    BatchNorm: alpha = w / sqrt(var + eps), beta = b - mean * alpha).  `coefs` [4, C] = mean | invstd | alpha | beta, computed
    once per set of buffers (FrozenBatchNorm2d._coefs): one launch forward (afan_affine_apply), one backward."""

    @staticmethod
    def forward(ctx, x, coefs, residual, relu, wb):
        x = _dense(x)
        if residual is not None:
            residual = _like_layout(residual, x)
        if ops.layout_of(x) == ops.AFAN_NHWC:
            y = ops.affine_apply(x, coefs, residual, relu)
        else:       # the reference's layout (fp32 parity mode): the NCHW kernel derives alpha / beta itself, per launch
            y = ops.bn_apply(x, coefs[0], coefs[1], wb[0], wb[1], residual, relu)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.alpha = coefs[2]
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _like_layout(g, y) if y is not None else _dense(g)
        if y is not None and g.dtype != y.dtype:
            g = g.to(y.dtype)
        dx, dres = ops.affine_relu_backward(g, y, ctx.alpha, ctx.relu, want_dx=ctx.needs_input_grad[0],
                                            want_dres=ctx.has_res and ctx.needs_input_grad[2])
        return dx, None, dres, None, None


class FrozenBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d's parameters and buffers (same state_dict keys), always normalising with the running statistics:
    Detection/model.py:46-47 puts every BatchNorm in eval mode at each forward and :31-35 switches their gradients off."""
    _coef = None

    def _coefs(self):
        ts = (self.running_var, self.running_mean, self.weight, self.bias)
        key = tuple(t._version for t in ts) + tuple(t.data_ptr() for t in ts)
        if self._coef is None or self._coef[0] != key:
            invstd = torch.rsqrt(self.running_var.float() + self.eps)
            self._coef = (key, ops.affine_coefs(self.running_mean.float().contiguous(), invstd, self.weight.detach().float().contiguous(),
                                                self.bias.detach().float().contiguous()))
        return self._coef[1]

    def fused(self, x, residual=None, relu=False, conv_stats=None):
        return _AffineFn.apply(x, self._coefs(), residual, relu, (self.weight.detach(), self.bias.detach()))

    def forward(self, x):
        return self.fused(x)


class _BiasFn(torch.autograd.Function):
    """y = [relu](x + bias[c]) after a bias-free convolution launch; d(bias) = sum of the (masked) gradient over n, h, w."""

    @staticmethod
    def forward(ctx, x, bias, relu, want_pgrad):
        x = _dense(x)
        c = x.shape[1]
        zero, one = _const(x.device, c)
        y = ops.bn_apply(x, zero, one, None, bias.detach().float(), None, relu)
        ctx.relu, ctx.pg = bool(relu), want_pgrad
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = _like_layout(g, y) if y is not None else _dense(g)
        if y is not None and g.dtype != y.dtype:
            g = g.to(y.dtype)
        dx, _ = ops.affine_relu_backward(g, y, None, ctx.relu, want_dx=True)
        db = dx.float().sum(dim=(0, 2, 3)) if (ctx.pg and ctx.needs_input_grad[1]) else None
        return (dx if ctx.needs_input_grad[0] else None), db, None, None


_consts = {}


def _const(dev, c):
    k = (dev.index, c)
    if k not in _consts:
        _consts[k] = (torch.zeros(c, device=dev), torch.ones(c, device=dev))
    return _consts[k]


class _GlobalMaxFn(torch.autograd.Function):
    """F.adaptive_max_pool2d(x, 1) (model.py:285): one window over the whole map."""

    @staticmethod
    def forward(ctx, x):
        x = _dense(x)
        h, w = x.shape[2], x.shape[3]
        if h != w:
            raise ops.AfanLibraryError("global max pooling of a non-square map is not on the Detection path")
        y, idx = ops.maxpool2d(x, h, h, 0, want_idx=ctx.needs_input_grad[0])
        ctx.shape, ctx.k, ctx.cl = tuple(x.shape), h, ops.layout_of(x) == ops.AFAN_NHWC
        ctx.save_for_backward(idx if idx is not None else x)
        ctx.have_idx = idx is not None
        return y

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        g = g.contiguous()              # [R, C, 1, 1]: the same memory in either layout
        return ops.maxpool2d_backward(g, t if ctx.have_idx else None, ctx.shape, ctx.k, ctx.k, 0, x=None if ctx.have_idx else t,
                                      channels_last=ctx.cl)


# ------------------------------------------------------------------------------------------------------------- backbone
class _FrozenBlockFn(torch.autograd.Function):
    """A frozen-BatchNorm bottleneck (backbone/resnet101_ori.py:78-127) as ONE autograd node on the bf16 channels-last path:
    the same launches as the layer-by-layer form — three (four) tuned convolutions, three (four) fused affine(+residual)
    (+ReLU) launches forward; affine backward, input gradient and weight gradient per layer backward, the identity shortcut's
    gradient added in the first convolution's dgrad epilogue — but one `Function.apply`, one backward node and ONE native call
    each way (afan_frozen_bottleneck_fwd / _bwd issue the launches from C++) instead of seven of each: the Detection
    iteration is bound by Python dispatch (2 600 applies per iteration before this node)."""

    @staticmethod
    def forward(ctx, x, blk, plan, *params):
        out, a1, a2 = ops.frozen_bottleneck_fwd_plan(x, plan)
        ctx.plan = plan
        ctx.save_for_backward(x, a1, a2, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, a1, a2, out = ctx.saved_tensors
        g = _like_layout(g, out)
        if g.dtype != out.dtype:
            g = g.to(out.dtype)
        dx = ops.frozen_bottleneck_bwd_plan(g, x, a1, a2, out, ctx.plan, ctx.needs_input_grad[0])
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


def _block_plan(blk, x):
    """The block's cached ops.FrozenBlockPlan for this input shape and autograd mode, or False where the one-node form does not
    apply (then the layer-by-layer modules run).  Valid for one weight epoch (an optimizer step or a shadow refresh bumps
    _Flags.weight_epoch: low-precision copies, transposed copies and coefficient rows are re-read then)."""
    if blk._plan_epoch != _Flags.weight_epoch:
        blk._plans, blk._plan_epoch = {}, _Flags.weight_epoch
    pg = bool(_Flags.param_grads and torch.is_grad_enabled())
    if blk._ver is None:      # tensors whose in-place edits (load_state_dict, manual surgery) must invalidate the cached pointers
        blk._ver = [t for m in blk.modules() if isinstance(m, (nn.Conv2d, nn.BatchNorm2d))
                    for t in ((m.weight,) if isinstance(m, nn.Conv2d) else (m.weight, m.bias, m.running_mean, m.running_var))]
    key = (tuple(x.shape), x.device.index, pg, _Flags.wgrad_stash, _WgradStream.ON, _FrozenBlockFn.ON,
           sum(t._version for t in blk._ver), blk._ver[0].data_ptr())
    plan = blk._plans.get(key)
    if plan is None:
        plan = False
        if _frozen_block_ok(blk, x):
            convs = [blk.conv1, blk.conv2, blk.conv3, blk.downsample[0] if blk.downsample is not None else None]
            ks = (blk.bn1._coefs(), blk.bn2._coefs(), blk.bn3._coefs(), blk.downsample[1]._coefs() if convs[3] is not None else None)
            gws = [c.weight.grad if (c is not None and pg and c.weight.requires_grad) else None for c in convs]
            plan = ops.frozen_bottleneck_plan(x, blk.conv1.out_channels, blk.conv2.stride[0],
                                              tuple(None if c is None else c.lp_weight() for c in convs), ks,
                                              tuple(None if c is None else c.lp_weight_t() for c in convs),
                                              tuple(None if k is None else k[2] for k in ks), gws)
        blk._plans[key] = plan
    return plan


def _stage_backward(plans, saved, g, want_dx):
    """The blocks of a stage backwards.  Block i + 1's input is block i's output, so the first step of block i's backward — the
    gradient masked by that output's ReLU, once as it is (the shortcut's share) and once times the last BatchNorm's alpha — rides in
    the epilogue of block i + 1's last input-gradient launch (ops.frozen_bottleneck_bwd_plan `prev`): no launch of its own."""
    pre = None
    for i in range(len(plans) - 1, -1, -1):
        x, a1, a2, out = saved[4 * i:4 * i + 4]
        res = ops.frozen_bottleneck_bwd_plan(g if pre is None else None, x, a1, a2, out, plans[i], i > 0 or want_dx, pre=pre,
                                             prev=plans[i - 1] if i > 0 else None)
        if i > 0:
            pre = res
        else:
            g = res
    return g


def _dgrad_plans(stage, out):
    """The input-gradient-only launch plans of `stage` for the activations a one-node forward left on `out`, or None."""
    saved = getattr(out, "_afan_stage_saved", None)
    if saved is None or len(saved) != 4 * len(stage) - 1:          # (every activation but `out` itself: no reference cycle)
        return None
    with dgrad_only():
        plans, probe = [], saved[0]
        for blk in stage:
            plan = _block_plan(blk, probe)
            if not plan:
                return None
            plans.append(plan)
            probe = _ShapeOnly((plan.n, plan.co, plan.ho, plan.wo), saved[0])
    return plans


def stage_input_gradient(stage, out, g=None):
    """d(loss)/d(stage input) from g = d(loss)/d(stage output) and the activations a forward of `stage` left on its output tensor
    `out` (`_FrozenStageFn`), with no parameter gradient touched (the dgrad-only launch plans) — what back-propagating g through
    a SECOND forward of the same stage on the same input under `resnet_s.dgrad_only()` would give, bit for bit, without that
    forward.  g None: only answers whether that is possible (True / False); None when it is not."""
    plans = _dgrad_plans(stage, out)
    if g is None:
        return plans is not None
    if plans is None:
        return None
    saved = out._afan_stage_saved + (out,)
    last = saved[-1]
    g = _like_layout(g, last)
    if g.dtype != last.dtype:
        g = g.to(last.dtype)
    with torch.no_grad():
        return _stage_backward(plans, saved, g, True)


class _FrozenStageFn(torch.autograd.Function):
    """A whole stage (layer1 .. layer4: 3 / 4 / 23 / 3 frozen-BatchNorm bottlenecks) as ONE autograd node: the same native calls
    as one `_FrozenBlockFn` per block, issued in a loop — the per-node cost of `Function.apply` and of the engine's scheduling
    (~50 us a block and direction on this host) is paid once per stage instead of 33 times per backbone pass, 27 passes per
    Detection iteration."""

    @staticmethod
    def forward(ctx, x, plans, *params):
        saved = []
        for plan in plans:
            out, a1, a2 = ops.frozen_bottleneck_fwd_plan(x, plan)
            saved += [x, a1, a2, out]
            x = out
        ctx.plans = plans
        ctx.save_for_backward(*saved)
        if any(ctx.needs_input_grad):             # (references only, and only where a graph keeps them alive anyway:
            # stage_input_gradient reuses a clean pass's activations.)  Everything BUT the output itself: a tuple holding
            # `x` on `x.__dict__` is a cycle that only the cyclic collector frees — the stage's activations would outlive
            # their graph by however long that takes (~4 GB resident in a CPU simulation)
            x._afan_stage_saved = tuple(saved[:-1])
        return x

    @staticmethod
    def backward(ctx, g):
        saved, plans = ctx.saved_tensors, ctx.plans
        last = saved[-1]
        g = _like_layout(g, last)
        if g.dtype != last.dtype:
            g = g.to(last.dtype)
        g = _stage_backward(plans, saved, g, ctx.needs_input_grad[0])
        return (g, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _StageInst:
    __slots__ = ("x", "out", "saved", "fwd", "bwd", "g", "dx", "busy", "plans", "want_dx")


class _StageGraphs:
    """Captured hipGraphs of whole backbone stages (forward, and backward on first use) per (stage, input shape, plan signature,
    gradient mode): the Detection iteration is shaped by its proposals everywhere EXCEPT in the backbone, whose 27 passes per
    iteration issue the same ~30 native calls each from Python — replayed here as one graph launch each way.  An instance owns
    its activations (static buffers of the graph's pool), so a forward whose backward is still pending keeps its instance
    busy and the next forward of the same kind takes (or captures) another; `begin_iteration()` frees them all (forwards whose
    graph autograd never walks back: the detached head passes, the ROI dict's pass).  More than MAX_INST busy instances of one
    kind (a caller that never calls begin_iteration): the eager stage node runs instead."""
    # Measured (MI355X, 600 x 904, 20 iterations, interleaved): 123.7 / 120.4 ms replayed against 119.9 / 111.9 ms eager, losses
    # identical — after the per-block launch plans the backbone's launch cost is no longer what the iteration waits for (its ~50
    # host reads of proposal / sample counts are): OFF by default, AFAN_DET_GRAPHS=1 switches it on.
    ON = os.environ.get("AFAN_DET_GRAPHS", "0") == "1"
    MAX_INST = 12
    cache = {}
    warm = {}                                  # key -> eager runs seen (workspaces must exist before a capture)

    @classmethod
    def begin_iteration(cls):
        for insts in cls.cache.values():
            for it in insts:
                it.busy = False

    @classmethod
    def get(cls, stage, x, plans, want_dx):
        key = (id(stage), want_dx, tuple(p.sig for p in plans))
        if cls.warm.get(key, 0) < 2:           # two eager passes first
            cls.warm[key] = cls.warm.get(key, 0) + 1
            return None
        insts = cls.cache.setdefault(key, [])
        for it in insts:
            if not it.busy:
                return it
        if len(insts) >= cls.MAX_INST or torch.cuda.is_current_stream_capturing():
            return None
        it = cls._capture_fwd(x, plans, want_dx)
        insts.append(it)
        return it

    @classmethod
    def _capture_fwd(cls, x, plans, want_dx):
        it = _StageInst()
        it.plans, it.want_dx, it.busy, it.bwd, it.g, it.dx = plans, want_dx, False, None, None, None
        it.x = torch.empty_like(x)
        torch.cuda.synchronize(x.device)
        g = torch.cuda.CUDAGraph()
        # A PRIVATE memory pool per graph: instances are replayed in any order and keep their activations from forward to backward,
        # so a temporary freed inside one capture must never be handed to another graph as something long-lived (a shared pool
        # assumes graphs replay in capture order).
        with ops.no_gc_during_capture(), torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
            cur, saved = it.x, []
            for plan in plans:
                out, a1, a2 = ops.frozen_bottleneck_fwd_plan(cur, plan)
                saved += [cur, a1, a2, out]
                cur = out
        it.saved, it.out, it.fwd = saved, cur, g
        return it

    @classmethod
    def capture_bwd(cls, it):
        it.g = torch.empty_like(it.out)
        torch.cuda.synchronize(it.out.device)
        g = torch.cuda.CUDAGraph()
        with ops.no_gc_during_capture(), torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):       # (runs on the engine's thread; private pool)
            cur = _stage_backward(it.plans, it.saved, it.g, it.want_dx)
        it.dx, it.bwd = cur, g


class _GraphedStageFn(torch.autograd.Function):
    """`_FrozenStageFn` as two graph replays (see _StageGraphs)."""

    @staticmethod
    def forward(ctx, x, inst, *params):
        inst.x.copy_(x, non_blocking=True)
        inst.fwd.replay()
        inst.busy = True
        ctx.inst = inst
        return inst.out.detach()

    @staticmethod
    def backward(ctx, g):
        it = ctx.inst
        if it.bwd is None:
            _StageGraphs.capture_bwd(it)
        it.g.copy_(g, non_blocking=True)          # (any dtype / layout of g: copy_ converts)
        it.bwd.replay()
        it.busy = False
        dx = it.dx.detach() if (it.want_dx and it.dx is not None) else None
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


def _run_stage(stage, x):
    """`stage(x)` for an nn.Sequential of Bottlenecks: one `_FrozenStageFn` node when every block takes the one-call form at the
    shape it will see, else block by block."""
    if not (_FrozenStageFn.ON and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last)):
        return stage(x)
    plans, shape_probe = [], x
    for blk in stage:
        plan = _block_plan(blk, shape_probe)
        if not plan:
            return stage(x)
        plans.append(plan)
        shape_probe = _ShapeOnly((plan.n, plan.co, plan.ho, plan.wo), x)
    params = getattr(stage, "_stage_params", None)
    if params is None:
        params = stage._stage_params = tuple(p for blk in stage for p in blk._block_params())
    if _StageGraphs.ON:
        grad_on = torch.is_grad_enabled()
        inst = _StageGraphs.get(stage, x, plans, bool(grad_on and x.requires_grad))
        if inst is not None:
            if not grad_on:                          # (no graph to build: replay and hand the buffer out; freed by begin_iteration)
                inst.x.copy_(x, non_blocking=True)
                inst.fwd.replay()
                inst.busy = True
                return inst.out.detach()
            return _GraphedStageFn.apply(x, inst, *params)
    return _FrozenStageFn.apply(x, tuple(plans), *params)


class _ShapeOnly:
    """What `_block_plan` / `_frozen_block_ok` look at of a block's input, for a tensor that does not exist yet (the previous
    block's output inside a stage node)."""
    __slots__ = ("shape", "device", "dtype", "is_cuda")

    def __init__(self, shape, like):
        self.shape, self.device, self.dtype, self.is_cuda = torch.Size(shape), like.device, like.dtype, like.is_cuda

    def dim(self):
        return len(self.shape)

    def is_contiguous(self, memory_format=None):
        return True


_FrozenStageFn.ON = os.environ.get("AFAN_DET_STAGE_NODE", "1") != "0"      # 0: one node per block (A/B)


def _frozen_block_ok(blk, x):
    """bf16 channels-last map, every convolution of the block on the tuned kernels, trainable weights owned by the arena."""
    if not (_FrozenBlockFn.ON and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and blk.conv1.compute_dtype == torch.bfloat16):
        return False
    convs = [blk.conv1, blk.conv2, blk.conv3] + ([blk.downsample[0]] if blk.downsample is not None else [])
    for c in convs:
        if c.bias is not None or not _own_conv_ok(x, c.lp_weight(), c.stride, c.padding, c.dilation):
            return False
        if not ops.conv_wgrad_supported(c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0]):
            return False
        if _Flags.param_grads and torch.is_grad_enabled() and c.weight.requires_grad and not (
                _accumulates_in_place(c.weight) and (c.weight.grad.is_contiguous(memory_format=torch.channels_last) or c.kernel_size[0] == 1)):
            return False
    return not _Flags.wgrad_stash and not _WgradStream.ON


_FrozenBlockFn.ON = os.environ.get("AFAN_DET_BLOCK_NODE", "1") != "0"      # 0: layer-by-layer autograd nodes (A/B)


class Bottleneck(nn.Module):
    """backbone/resnet101_ori.py:78-127 (torchvision v1.5: the 3x3 carries the stride)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, stride=1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, kernel_size=1, stride=1, bias=False)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    _plans, _plan_epoch, _params, _ver = None, -1, None, None

    def __getstate__(self):
        # launch-plan caches hold ctypes pointer objects (not picklable, and meaningless in a copy): a deepcopy / torch.save of
        # the module starts without them and rebuilds them on its first forward
        d = dict(self.__dict__)
        for k in ("_plans", "_plan_epoch", "_params", "_ver"):
            d.pop(k, None)
        return d

    def _block_params(self):
        if self._params is None:
            self._params = tuple(c.weight for c in (self.conv1, self.conv2, self.conv3)) + \
                ((self.downsample[0].weight,) if self.downsample is not None else ())
        return self._params

    def forward(self, x):
        x = _to_compute(x, self.conv1.compute_dtype)
        if x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last):
            plan = _block_plan(self, x)
            if plan:
                return _FrozenBlockFn.apply(x, self, plan, *self._block_params())
        out = self.bn1.fused(self.conv1(x), None, True)
        out = self.bn2.fused(self.conv2(out), None, True)
        res = x if self.downsample is None else self.downsample[1].fused(self.downsample[0](x))
        return self.bn3.fused(self.conv3(out), res, True)


class ResNet101(nn.Module):
    """backbone/resnet101_ori.py:130-262 with layers (3, 4, 23, 3): `forward(input_dict)` returns the feature map after
    layer `out_idx` (flag 'head'), runs the remaining layers up to layer3 from `adv` (flag 'tail'), or the whole stem ..
    layer3 (flag 'clean').  layer4 belongs to the module (and its state_dict) but is applied by the detection head."""

    def __init__(self, layers=(3, 4, 23, 3), num_classes=1000):
        super().__init__()
        self.normal = NormalizeByChannelMeanStd(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
        self.inplanes = 64
        self.conv1 = StemConv(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * 4, num_classes)
        for m in self.modules():                                            # resnet101_ori.py:166-171
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       FrozenBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _stem(self, x):
        x = self.normal(x)
        x = self.bn1.fused(self.conv1(x), None, True)
        return self.maxpool(x)

    def head_features(self, x, idxs=(1, 2, 3)):
        """The feature maps after layers `idxs` from ONE pass, detached: what the three `flag: 'head'` calls of
        train_aug_sat_muti_advt.py:78-80 return (same images, frozen BatchNorm, no dropout: each is a prefix of the next)."""
        out = {}
        with torch.no_grad():
            x = self._stem(x)
            for i, st in enumerate([self.layer1, self.layer2, self.layer3], start=1):
                if i > max(idxs):
                    break
                x = _run_stage(st, x)
                if i in idxs:
                    out[i] = x.clone() if _StageGraphs.ON else x      # (a replayed stage hands out its own static buffer)
        return [out[i] for i in idxs]

    def forward_many(self, dicts):
        """The conv4 feature maps of several INDEPENDENT passes ('clean' from the image, 'tail' from a feature map entering behind layer
        1 or 2) with layer3 — 23 of the backbone's 30 bottlenecks — run ONCE on their concatenated batch: the BatchNorms are frozen, so a
        row of a convolution does not know its batch (every tiled variant adds a row's products in the same order: the same activations
        and input gradients, bit for bit); the layer's launches see three images' rows instead of one's (2 166 rows per image leave the
        chip a quarter full), and its weight gradients sum over the passes inside one reduction instead of three accumulations (fp32
        order: the only difference).  Returns one feature map per dict, each carrying its share of the autograd graph; None with the
        stage graphs on (AFAN_DET_GRAPHS=1: their instances are captured per pass)."""
        if _StageGraphs.ON:
            return None
        stages = [self.layer1, self.layer2, self.layer3]
        xs = []
        for d in dicts:
            if d["flag"] == "clean":
                x, first = self._stem(d["x"]), 0
            else:
                assert d["flag"] == "tail" and d["out_idx"] in (1, 2)
                x, first = _enter(d["adv"], self.conv1.compute_dtype, self.normal.channels_last), d["out_idx"]
            for st in stages[first:2]:
                x = _run_stage(st, x)
            xs.append(x)
        y = _run_stage(self.layer3, torch.cat(xs, dim=0) if len(xs) > 1 else xs[0])
        return list(torch.split(y, [x.shape[0] for x in xs], dim=0))

    def forward(self, input_dict):
        flag = input_dict["flag"]
        stages = [self.layer1, self.layer2, self.layer3]
        if flag in ("head", "clean"):
            last = 3 if flag == "clean" else input_dict["out_idx"]
            assert last in (1, 2, 3)
            x = self._stem(input_dict["x"])
            col = input_dict.get("collect")           # {}: filled with {stage index: that stage's output, detached} on the way
            for i, st in enumerate(stages[:last], start=1):
                x = _run_stage(st, x)
                if col is not None:
                    col[i] = x.detach().clone() if _StageGraphs.ON else x.detach()
                    col["out", i] = x                 # (the graph tensor: carries the stage's activations, see stage_input_gradient)
            return x
        assert flag == "tail" and input_dict["out_idx"] in (1, 2, 3)
        x = _enter(input_dict["adv"], self.conv1.compute_dtype, self.normal.channels_last)
        for st in stages[input_dict["out_idx"]:]:
            x = _run_stage(st, x)
        return x


# ------------------------------------------------------------------------------------------------------------------ RPN
EARLY_READ = os.environ.get("AFAN_DET_EARLY_READ", "1") != "0"      # 0: the anchor sampling's counts travel with the iteration's one host read (A/B, tests)
_early_ring = {}


def _early_slot(counts_dev):
    """Queue counts_dev ([2] int64 on the device) -> pinned host memory behind what has been launched so far; an event marks the copy."""
    dev = counts_dev.device.index
    ring = _early_ring.get(dev)
    if ring is None:
        ring = _early_ring[dev] = {"i": 0, "slots": [(torch.empty(2, dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(4)]}
    buf, ev = ring["slots"][ring["i"]]
    ring["i"] = (ring["i"] + 1) % len(ring["slots"])
    buf.copy_(counts_dev, non_blocking=True)
    ev.record()
    return buf, ev


def _early_wait(slot):
    buf, ev = slot
    ev.synchronize()
    return int(buf[0]), int(buf[1])


BATCH_RPN_TRUNK = os.environ.get("AFAN_DET_BATCH_RPN", "1") != "0"   # 0: forward_heads_many runs the RPN trunk and heads pass by pass (A/B, tests)
ANCHOR_LABEL_CACHE = os.environ.get("AFAN_DET_LABEL_CACHE", "1") != "0"   # 0: every pass labels the anchors and reads the list lengths itself (A/B, tests)
LINEAR_PAIR = os.environ.get("AFAN_DET_LINEAR_PAIR", "1") != "0"      # 0: each head layer on the general fp32 kernels (A/B, tests)


class _LinearPairFn(torch.autograd.Function):
    """(x w1^T + b1, x w2^T + b2) for the two heads that share an input — the ROI head's class / box Linear layers (model.py:255-256)
    and the RPN's objectness / box 1x1 convolutions (region_proposal_network.py:53-54, weights [N, K, 1, 1]) — on afan_linear.hip: one
    launch forward (two for the ROI head's 128 rows: split reduction), ONE input gradient for both layers (their sum is its reduction),
    both layers' parameter gradients in one launch pair; fp32."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, want_pgrad):
        ctx.save_for_backward(x, w1, w2)
        ctx.pg, ctx.params = want_pgrad, (w1, b1, w2, b2)
        return ops.linear_pair_fwd(x, w1.detach().reshape(w1.shape[0], -1), None if b1 is None else b1.detach(),
                                   w2.detach().reshape(w2.shape[0], -1), None if b2 is None else b2.detach())

    @staticmethod
    def backward(ctx, g1, g2):
        x, w1, w2 = ctx.saved_tensors
        p1, b1, p2, b2 = ctx.params
        n1, n2 = w1.shape[0], w2.shape[0]
        g1 = torch.zeros((x.shape[0], n1), dtype=torch.float32, device=x.device) if g1 is None else g1.contiguous().float()
        g2 = torch.zeros((x.shape[0], n2), dtype=torch.float32, device=x.device) if g2 is None else g2.contiguous().float()
        gx = ops.linear_pair_dgrad(g1, g2, w1.detach().reshape(n1, -1), w2.detach().reshape(n2, -1)) if ctx.needs_input_grad[0] else None
        gw1 = gb1 = gw2 = gb2 = None
        if ctx.pg and (ctx.needs_input_grad[1] or ctx.needs_input_grad[3]):
            if all(q is None or _accumulates_in_place(q) for q in (p1, b1, p2, b2)):
                ops.linear_pair_wgrad(g1, g2, x, p1.grad, None if b1 is None else b1.grad, p2.grad, None if b2 is None else b2.grad, True)
            else:
                gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
                gb1 = None if b1 is None else torch.empty_like(b1)
                gb2 = None if b2 is None else torch.empty_like(b2)
                ops.linear_pair_wgrad(g1, g2, x, gw1, gb1, gw2, gb2, False)
        return gx, gw1, gb1, gw2, gb2, None


def _linear_pair(x, lin1, lin2):
    """The two layers' outputs on afan_linear.hip, or None (the caller takes the general kernels)."""
    w1, w2 = lin1.weight, lin2.weight
    if not (x.is_cuda and x.dim() == 2 and (w1.dim() == 2 or w1.shape[2] * w1.shape[3] == 1) and (w2.dim() == 2 or w2.shape[2] * w2.shape[3] == 1)):
        return None
    if not (_dense_rows(w1) and _dense_rows(w2) and ops.linear_pair_ok(x, w1.detach().reshape(w1.shape[0], -1), w2.detach().reshape(w2.shape[0], -1))):
        return None
    for b in (lin1.bias, lin2.bias):
        if b is not None and (b.dtype != torch.float32 or not b.is_contiguous()):
            return None
    return _LinearPairFn.apply(x, w1, lin1.bias, w2, lin2.bias, _Flags.param_grads)


def _dense_rows(w):
    """[N, K] or [N, K, 1, 1] whose rows are K consecutive elements (either memory format of a 1x1 kernel)."""
    return w.stride(0) == w.shape[1] and w.stride(1) == 1 and w.reshape(w.shape[0], -1).data_ptr() == w.data_ptr()


class RegionProposalNetwork(nn.Module):
    """rpn/region_proposal_network.py:13-271."""

    def __init__(self, num_features_out, anchor_ratios, anchor_sizes, pre_nms_top_n, post_nms_top_n, anchor_smooth_l1_loss_beta):
        super().__init__()
        self._features = nn.Sequential(Conv2d(num_features_out, 512, kernel_size=3, padding=1), nn.ReLU())
        self._anchor_ratios, self._anchor_sizes = anchor_ratios, anchor_sizes
        num_anchors = len(anchor_ratios) * len(anchor_sizes)
        self._pre_nms_top_n, self._post_nms_top_n = pre_nms_top_n, post_nms_top_n
        self._anchor_smooth_l1_loss_beta = anchor_smooth_l1_loss_beta
        self._anchor_objectness = Conv2d(512, num_anchors * 2, kernel_size=1)
        self._anchor_transformer = Conv2d(512, num_anchors * 4, kernel_size=1)
        self._inside_cache = {}

    # -- layers
    def _trunk(self, features):
        c = self._features[0]
        x = _to_compute(features, c.compute_dtype)
        y = _ConvFn.apply(x, c.weight, c.lp_weight().detach(), c.lp_weight_t, c.stride, c.padding, _Flags.param_grads, None, c.dilation)
        return _BiasFn.apply(y, c.bias, True, _Flags.param_grads)

    def _heads(self, trunk):
        """The two 1x1 heads in fp32 (18 / 36 output channels: the general kernel, bias in its epilogue)."""
        b = trunk.shape[0]
        x = trunk if trunk.dtype == torch.float32 else trunk.float()
        if LINEAR_PAIR and (ops.layout_of(x) == ops.AFAN_NHWC or x.shape[2] * x.shape[3] == 1):      # channels-last: the map IS [pixels, 512]
            ys = _linear_pair(x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]), self._anchor_objectness, self._anchor_transformer)
            if ys is not None:
                return [ys[0].view(b, -1, 2), ys[1].view(b, -1, 4)]
        outs = []
        for conv, k in ((self._anchor_objectness, 2), (self._anchor_transformer, 4)):
            y = _ConvFn.apply(x, conv.weight, conv.weight.detach(), None, conv.stride, conv.padding, _Flags.param_grads, None,
                              conv.dilation, conv.bias)
            outs.append(y.permute(0, 2, 3, 1).contiguous().view(b, -1, k))
        return outs

    # -- training targets (:58-105; the reference repeats this block in its 'clean' and 'tail' branches)
    def _losses(self, objectnesses, transformers, anchor_bboxes, gt_bboxes_batch, image_width, image_height, pending=False):
        """Labels (:66-82), sampling (:84-90), regression targets (:92-100) and the two per-image losses (:163-185) on the anchors
        inside the image: four launches and one host read (det_ops.box_assign / fg_bg_sample / per_image_losses) where the
        reference's tensor operations are about a hundred."""
        b = anchor_bboxes.shape[0]
        # (the anchors inside the image: a function of the cached anchor grid, computed — and synchronised on — once per grid)
        ikey = (anchor_bboxes.data_ptr(), tuple(anchor_bboxes.shape), int(image_width), int(image_height))
        hit = self._inside_cache.get(ikey)
        if hit is None:
            inside = ((anchor_bboxes[..., 0] >= 0) * (anchor_bboxes[..., 1] >= 0) * (anchor_bboxes[..., 2] <= image_width) *
                      (anchor_bboxes[..., 3] <= image_height)).nonzero().unbind(dim=1)
            in_boxes = anchor_bboxes[inside].view(b, -1, 4).contiguous()
            flat = (inside[0] * anchor_bboxes.shape[1] + inside[1]).contiguous()
            if len(self._inside_cache) > 64:
                self._inside_cache.clear()
            hit = self._inside_cache[ikey] = (in_boxes, flat, anchor_bboxes)      # (holding the grid keeps its address from being reused)
        in_boxes, flat = hit[0], hit[1]
        if pending and ANCHOR_LABEL_CACHE:
            # (round 6) the anchors' labels and the two candidate lists are a function of the anchor grid and the ground truth alone: every
            # pass of an iteration (16 of them) labelled the same anchors against the same boxes.  Remembered for the tensor the caller
            # passes (address, version, shape: an in-place edit or another batch is another key); once the first pass's counts have
            # reached the host they are kept too, and later passes neither launch nor read anything for them
            gkey = (ikey, gt_bboxes_batch.data_ptr(), gt_bboxes_batch._version, tuple(gt_bboxes_batch.shape))
            lab = self._label_cache if getattr(self, "_label_cache", None) is not None and self._label_cache[0] == gkey else None
            if lab is None:
                labels, assign = box_assign(in_boxes, gt_bboxes_batch, "anchor", 0.3, 0.7)
                lab = self._label_cache = [gkey, sample_lists(labels), labels, assign, gt_bboxes_batch, None]   # [.., host counts]
            return (lab[1], lab[2], lab[3], in_boxes, flat, gt_bboxes_batch, objectnesses, transformers, b)
        labels, assign = box_assign(in_boxes, gt_bboxes_batch, "anchor", 0.3, 0.7)
        if pending:                     # launches only: the caller reads the two list lengths together with its other counts
            return (sample_lists(labels), labels, assign, in_boxes, flat, gt_bboxes_batch, objectnesses, transformers, b)
        sel, _, lab, gt_deltas, bi = fg_bg_sample(labels, assign, in_boxes, gt_bboxes_batch, 128 * b, 256 * b)
        return per_image_losses(objectnesses, transformers, flat[sel], lab, gt_deltas, bi, b, self._anchor_smooth_l1_loss_beta)

    def _losses_finish(self, pend, nf, nb):
        lists, labels, assign, in_boxes, flat, gt, objectnesses, transformers, b = pend
        sel, _, lab, gt_deltas, bi = fg_bg_draw(lists, nf, nb, labels, assign, in_boxes, gt, 128 * b, 256 * b)
        return per_image_losses(objectnesses, transformers, flat[sel], lab, gt_deltas, bi, b, self._anchor_smooth_l1_loss_beta)

    def forward_and_propose(self, features, anchor_bboxes, gt_bboxes_batch, image_width, image_height, return_type="clean", roi_targets=None,
                            heads=None, share=None):
        """`forward(...)` (training) followed by `generate_proposals(...)` — the order model.py:95-100 calls them in — with ONE host
        read for both: the anchor sampling's two list lengths and every image's NMS survivor count travel together (the label /
        list launches and the decode / sort / NMS launches are all queued before it).  The host's random draws keep their order
        (the anchor sampling's three; the proposal layer draws nothing).  Returns (objectnesses, transformers, ce, sl1, proposals, roi_t)
        with roi_t = None unless roi_targets is given.
        roi_targets (round 5): the ROI head's own sampling (model.py:256-282) needs the proposals' labels' list lengths — a second host
        read right behind this one.  Given `roi_targets(padded [B, top_n, 4], kept [B] on the device)` -> a pending object with
        `.counts` (device, [2]) and `.finish(nf, nb)`, the label / list launches of THAT sampling are queued on the padded proposals (rows at and beyond the
        longest image's survivor count labelled -1, shorter images zero-padded like the reference's stack) before the one read, and its
        three draws follow the anchor sampling's on the host generator, as in the reference: a sixth return value, 16 host reads per
        iteration fewer."""
        if heads is not None:           # (the two head outputs computed by the caller: Model.forward_heads_many runs the trunk once for several passes)
            objectnesses, transformers = heads
        else:
            trunk = features["rpn_feature"] if return_type == "tail" else self._trunk(features)
            objectnesses, transformers = self._heads(trunk)
        if share is not None and "state" in share:
            # (round 6) a pass whose RPN input holds the SAME values as an earlier pass's (the one-step feature PGDs' tails and the clean
            # ROI-head pass all start from the clean conv4 map: :81-88): anchor labels, proposals (decode, sort, NMS) and the ROI
            # sampling's candidate lists are that pass's — only the host generator's draws are this pass's own, in the same order
            # (anchor sampling's three, then the ROI sampling's three), and the losses hang on THIS pass's head outputs
            pend0, counts, proposals, roi_pend, nb_img = share["state"]
            ce, sl1 = self._losses_finish(pend0[:6] + (objectnesses, transformers, pend0[8]), counts[0], counts[1])
            roi_t = roi_pend.finish(counts[2 + nb_img], counts[3 + nb_img]) if roi_pend is not None else None
            return objectnesses, transformers, ce, sl1, proposals, roi_t
        pend = self._losses(objectnesses, transformers, anchor_bboxes, gt_bboxes_batch, image_width, image_height, pending=True)
        early, known = None, None
        lc = getattr(self, "_label_cache", None)
        if ANCHOR_LABEL_CACHE and lc is not None and lc[1] is pend[0] and lc[5] is not None:
            known = lc[5]               # this ground truth's list lengths are on the host already (an earlier pass read them)
        elif EARLY_READ and not torch.cuda.is_current_stream_capturing():
            # (round 6) the anchor sampling's two list lengths leave for the host NOW, behind the label / list launches only: the host
            # waits for THAT copy (an event, not a drain) while the proposal layer's decode / sort / NMS launches keep the GPU busy, and
            # composes the anchor draws — torch.randperm over ~17 000 background anchors: ~0.16 ms of host time per pass that used to
            # sit between the iteration's host read and the next launch.  The host generator's draws keep their order (anchor
            # sampling's three, then the ROI sampling's three behind the second read).
            early = _early_slot(pend[0][2])
        boxes = box_decode_clip(anchor_bboxes, transformers.float(), image_width, image_height)
        probs = F.softmax(objectnesses.float()[:, :, 1], dim=-1)
        _, order = torch.sort(probs, dim=-1, descending=True)
        cand, keeps = [], []
        for b in range(anchor_bboxes.shape[0]):
            sb = boxes[b][order[b][:self._pre_nms_top_n]]
            keeps.append(nms(sb, None, 0.7, max_keep=self._post_nms_top_n, presorted=True, padded=True))
            cand.append(sb)
        roi_pend, nb_img = None, len(keeps)
        if roi_targets is not None:
            padded, kept_n = proposal_rows(cand, keeps, self._post_nms_top_n)     # (the scan may report up to 63 survivors more: clamped there)
            roi_pend = roi_targets(padded, kept_n)
        if early is not None or known is not None:
            nf, nb_ = known if known is not None else _early_wait(early)
            ce, sl1 = self._losses_finish(pend, nf, nb_)                      # host draws + launches while the NMS is still running
            counts = [nf, nb_] + torch.cat([c for _, c in keeps] + ([roi_pend.counts] if roi_pend is not None else [])).tolist()
        else:
            counts = torch.cat([pend[0][2]] + [c for _, c in keeps] + ([roi_pend.counts] if roi_pend is not None else [])).tolist()   # the one read
            ce, sl1 = self._losses_finish(pend, counts[0], counts[1])
        if ANCHOR_LABEL_CACHE and lc is not None and lc[1] is pend[0] and lc[5] is None:
            lc[5] = (counts[0], counts[1])
        kept = [sb[k[:n]][:self._post_nms_top_n] for sb, (k, _), n in zip(cand, keeps, counts[2:2 + nb_img])]
        if len(kept) == 1:
            proposals = kept[0].unsqueeze(0)
        else:
            longest = max(len(k) for k in kept)
            proposals = torch.stack([torch.cat([k, torch.zeros(longest - len(k), 4).to(k)]) for k in kept], dim=0)
        roi_t = roi_pend.finish(counts[2 + nb_img], counts[3 + nb_img]) if roi_pend is not None else None
        if share is not None and (roi_pend is not None) == (roi_targets is not None):
            share["state"] = (pend[:6] + (None, None, pend[8]), counts, proposals.detach(), roi_pend, nb_img)
        return objectnesses, transformers, ce, sl1, proposals.detach(), roi_t

    def forward(self, features, anchor_bboxes=None, gt_bboxes_batch=None, image_width=None, image_height=None, return_type="clean"):
        if return_type == "head":
            trunk = self._trunk(features)
            return {"batch_size": _int_tensor(trunk.shape[0], trunk.device), "rpn_feature": trunk}
        if return_type == "tail":
            trunk = features["rpn_feature"]
        else:
            assert return_type == "clean"
            trunk = self._trunk(features)
        objectnesses, transformers = self._heads(trunk)
        if not self.training:
            return objectnesses, transformers
        ce, sl1 = self._losses(objectnesses, transformers, anchor_bboxes, gt_bboxes_batch, image_width, image_height)
        return objectnesses, transformers, ce, sl1

    def generate_anchors(self, image_width, image_height, num_x_anchors, num_y_anchors):
        """:187-221: centres on a linspace grid without its end points, ratio-major / size-minor, y-major order."""
        ys = np.linspace(start=0, stop=image_height, num=num_y_anchors + 2)[1:-1]
        xs = np.linspace(start=0, stop=image_width, num=num_x_anchors + 2)[1:-1]
        ratios = np.array(self._anchor_ratios)
        ratios = ratios[:, 0] / ratios[:, 1]
        sizes = np.array(self._anchor_sizes)
        ys, xs, ratios, sizes = (a.reshape(-1) for a in np.meshgrid(ys, xs, ratios, sizes, indexing="ij"))
        centre = np.stack((xs, ys, sizes * np.sqrt(1 / ratios), sizes * np.sqrt(ratios)), axis=1)
        return _corners(torch.from_numpy(centre).float())

    def generate_proposals(self, anchor_bboxes, objectnesses, transformers, image_width, image_height):
        """:223-271: decode, clip, sort by the softmax over ALL anchors of the foreground logit, NMS at 0.7 per image (the
        library's kernel), top-N, zero-pad to the longest image."""
        boxes = box_decode_clip(anchor_bboxes, transformers, image_width, image_height)
        probs = F.softmax(objectnesses[:, :, 1], dim=-1)
        _, order = torch.sort(probs, dim=-1, descending=True)
        kept = []
        for b in range(anchor_bboxes.shape[0]):
            sb = boxes[b][order[b][:self._pre_nms_top_n]]
            # (sb is in descending score order: no second sort, and the scan stops at the top-N survivors)
            keep = nms(sb, None, 0.7, max_keep=self._post_nms_top_n, presorted=True)
            kept.append(sb[keep.to(sb.device)][:self._post_nms_top_n])
        if len(kept) == 1:
            return kept[0].unsqueeze(0)
        longest = max(len(k) for k in kept)
        return torch.stack([torch.cat([k, torch.zeros(longest - len(k), 4).to(k)]) for k in kept], dim=0)


class _RoiPending:
    """The ROI head's sampling between its label / list launches and its draws (Detection._targets_pending)."""

    def __init__(self, lists, labels, assign, padded, gt):
        self.lists, self.labels, self.assign, self.padded, self.gt = lists, labels, assign, padded, gt
        self.counts = lists[2]

    def finish(self, nf, nb):
        b = self.padded.shape[0]
        _, boxes, lab, deltas, bi = fg_bg_draw(self.lists, nf, nb, self.labels, self.assign, self.padded, self.gt, 32 * b, 128 * b)
        return boxes, lab, deltas, bi


# --------------------------------------------------------------------------------------------------------------- pooler
def pool_rois(features, proposal_bboxes, proposal_batch_indices, mode):
    """roi/pooler.py:21-44: 14 x 14 per region — 'align': ROIAlign at scale 1/16 with adaptive sampling (afan_roi_align_*);
    'pooling': adaptive max pooling of the box rounded to feature cells, region by region on the host like the reference
    (its CPU-runnable mode: the one the reference goldens can be generated in) — then a 2 x 2 / stride 2 max pool."""
    mode = getattr(mode, "value", mode)
    _, _, fh, fw = features.shape
    scale = 1 / 16
    if mode == "pooling":
        pool = []
        for box, bi in zip(proposal_bboxes, proposal_batch_indices):
            x0 = max(min(round(box[0].item() * scale), fw - 1), 0)
            y0 = max(min(round(box[1].item() * scale), fh - 1), 0)
            x1 = max(min(round(box[2].item() * scale) + 1, fw), 1)
            y1 = max(min(round(box[3].item() * scale) + 1, fh), 1)
            pool.append(F.adaptive_max_pool2d(input=features[bi, :, y0:y1, x0:x1], output_size=(14, 14)))
        pool = torch.stack(pool, dim=0)
    elif mode == "align":
        rois = torch.cat([proposal_batch_indices.view(-1, 1).float(), proposal_bboxes], dim=1)
        pool = roi_align(features, rois, (14, 14), scale, 0)
    else:
        raise ValueError(mode)
    cl = ops.layout_of(features) == ops.AFAN_NHWC
    pool = pool.contiguous(memory_format=torch.channels_last) if cl else pool.contiguous()
    return _MaxPoolFn.apply(pool, 2, 2, 0)


# ---------------------------------------------------------------------------------------------------------------- model
class Model(nn.Module):
    """Detection/model.py:18-185."""

    def __init__(self, backbone=None, num_classes=21, pooler_mode="align", anchor_ratios=((1, 2), (1, 1), (2, 1)),
                 anchor_sizes=(128, 256, 512), rpn_pre_nms_top_n=12000, rpn_post_nms_top_n=2000,
                 anchor_smooth_l1_loss_beta=1.0, proposal_smooth_l1_loss_beta=1.0):
        super().__init__()
        self.features = backbone if backbone is not None else ResNet101()
        hidden, num_features_out, num_hidden_out = self.features.layer4, 1024, 2048
        for part in (self.features.conv1, self.features.bn1, self.features.layer1):        # backbone/resnet101.py:30-32
            for p in part.parameters():
                p.requires_grad = False
        self._bn_modules = nn.ModuleList([m for m in self.features.modules() if isinstance(m, nn.BatchNorm2d)] +
                                         [m for m in hidden.modules() if isinstance(m, nn.BatchNorm2d)])
        for bn in self._bn_modules:
            for p in bn.parameters():
                p.requires_grad = False
        self.rpn = RegionProposalNetwork(num_features_out, list(anchor_ratios), list(anchor_sizes), rpn_pre_nms_top_n,
                                         rpn_post_nms_top_n, anchor_smooth_l1_loss_beta)
        self.detection = Model.Detection(pooler_mode, hidden, num_hidden_out, num_classes, proposal_smooth_l1_loss_beta)
        self.compute_dtype, self.channels_last = torch.float32, False
        self._anchor_cache = {}

    def train(self, mode=True):
        """`model.train().forward(...)` is the reference's calling idiom — once per forward, 21 times per A-FAN iteration
        (attack_algo.py:24,31, train_aug_sat_muti_advt.py:106-150); nn.Module.train walks all ~430 modules each time
        (54 ms of host time per iteration).  Walk only when the mode changes (sub-modules switched by hand in between are
        re-synchronised by any real mode change)."""
        mode = bool(mode)
        if self.training == mode and getattr(self, "_mode_walked", None) == mode:
            return self
        super().train(mode)
        self._mode_walked = mode
        return self

    _cuts = None

    def cut_features(self, cuts):
        """Context: every training forward inside it detaches the backbone's output `features` (where it carries a graph)
        into a fresh leaf and appends (features, leaf) to `cuts`; the RPN and the ROI head consume the leaf.  A backward from
        the losses then stops at the leaves; `torch.autograd.backward([f...], [leaf.grad...])` finishes it
        (det_attack_algo.det_train_phases: the two-part backward of the data-parallel iteration)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._cuts = self._cuts, cuts
            try:
                yield cuts
            finally:
                self._cuts = old
        return ctx()

    def _cut(self, features):
        if self._cuts is None or features.grad_fn is None:
            return features
        leaf = features.detach().requires_grad_(True)
        self._cuts.append((features, leaf))
        return leaf

    def begin_iteration(self):
        """Start of a training iteration: the captured backbone-stage instances of the previous one are free again."""
        _StageGraphs.begin_iteration()

    def head_features(self, x, idxs=(1, 2, 3)):
        """[forward({'x': x, 'flag': 'head', 'out_idx': i}).detach() for i in idxs] from one backbone pass."""
        return self.features.head_features(x, idxs)

    # A 'clean' forward whose dict carries "collect": {} also leaves the three head passes' values there (the backbone's stage
    # outputs, detached): the same numbers as head_features(x) — same images, frozen BatchNorm, no dropout — without the pass.
    collects_head_features = True

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, Conv2d) and m not in (self.rpn._anchor_objectness, self.rpn._anchor_transformer):
                m.compute_dtype = dtype
            elif isinstance(m, NormalizeByChannelMeanStd):
                m.out_dtype = dtype
        return self

    def set_channels_last(self, on=True):
        self.channels_last = bool(on)
        for m in self.modules():
            if isinstance(m, NormalizeByChannelMeanStd):
                m.channels_last = self.channels_last
        return self

    def _anchors(self, features, image_shape):
        """The anchor grid of this (image, feature map) geometry on the device, computed once: the reference rebuilds it in
        numpy at every forward (model.py:75-77); on a 256-core host the small CPU tensor ops behind it cost 5 ms per forward."""
        b, _, ih, iw = image_shape
        _, _, fh, fw = features.shape
        key = (int(iw), int(ih), int(fw), int(fh), int(b), features.device)
        a = self._anchor_cache.get(key)
        if a is None:
            a = self.rpn.generate_anchors(iw, ih, num_x_anchors=fw, num_y_anchors=fh).to(features.device).repeat(b, 1, 1)
            if len(self._anchor_cache) > 64:
                self._anchor_cache.clear()
            self._anchor_cache[key] = a
        return a, iw, ih

    def forward(self, input_dict, gt_bboxes_batch=None, gt_classes_batch=None):
        flag = input_dict["flag"]
        if flag == "head":
            return self.features(input_dict)
        assert flag in ("tail", "clean")
        if not self.training:
            features = self.features(input_dict)
            anchors, iw, ih = self._anchors(features, input_dict["x"].shape)
            obj, tr = self.rpn.forward(features)
            proposals = self.rpn.generate_proposals(anchors, obj, tr, iw, ih)
            classes, transformers = self.detection.forward(features, proposals)
            return self.detection.generate_detections(proposals, classes, transformers, iw, ih)
        idx = input_dict["out_idx"]
        if idx == "roi_tail":
            d = input_dict["adv"]
            _, _, pc, pt = self.detection.forward(d["roi_output_dict"], return_type="tail")
            return d["anchor_objectness_losses"], d["anchor_transformer_losses"], pc, pt
        if idx == "rpn_tail":
            d = input_dict["adv"]
            features, anchors = d["features"], d["anchor_bboxes"]
            iw, ih = _int_of(d["image_width"]), _int_of(d["image_height"])
            obj, tr, ao, at, proposals, roi_t = self.rpn.forward_and_propose(d["rpn_feature_map_dict"], anchors, gt_bboxes_batch, iw, ih, return_type="tail",
                                                                             roi_targets=self._roi_targets(gt_classes_batch, gt_bboxes_batch))
        else:
            features = self._cut(self.features(input_dict))
            anchors, iw, ih = self._anchors(features, input_dict["x"].shape)
            if idx == "rpn_head":
                return {"features": features, "image_height": _int_tensor(ih, features.device),
                        "image_width": _int_tensor(iw, features.device), "anchor_bboxes": anchors,
                        "rpn_feature_map_dict": self.rpn.forward(features, anchors, gt_bboxes_batch, iw, ih, return_type="head")}
            assert type(idx) == int or idx == "roi_head"
            obj, tr, ao, at, proposals, roi_t = self.rpn.forward_and_propose(features, anchors, gt_bboxes_batch, iw, ih,
                                                                             roi_targets=self._roi_targets(gt_classes_batch, gt_bboxes_batch),
                                                                             share=input_dict.get("rpn_share"))
        if idx == "roi_head":
            return {"anchor_objectness_losses": ao, "anchor_transformer_losses": at,
                    "roi_output_dict": self.detection.forward(features, proposals, gt_classes_batch, gt_bboxes_batch, return_type="head", targets=roi_t)}
        _, _, pc, pt = self.detection.forward(features, proposals, gt_classes_batch, gt_bboxes_batch, targets=roi_t)
        return ao, at, pc, pt

    def forward_heads_many(self, dicts, gt_bboxes_batch=None, gt_classes_batch=None, share=None):
        """[forward(d, gt_bboxes_batch, gt_classes_batch) for d in dicts] for training passes that are independent of one another (the
        final passes of train_aug_sat_muti_advt.py:141-153), with the ROI head — ROIAlign, layer4, the two Linear layers — run ONCE on
        all passes' sampled regions: pass by pass, in order, everything up to the sampling (backbone, RPN, proposals, the host
        generator's draws in the reference's sequence); then one ROIAlign over the concatenated feature maps (a region's batch index
        names its pass's map), one layer4 on 7 x 128 regions instead of seven on 128 (2 048 rows leave the chip half full), and each
        pass's two losses from its rows.  Frozen BatchNorm: a region's features do not depend on its batch; the small fp32 Linear
        layers split their reduction per chunk whatever the row count; the parameter gradients sum in one reduction: fp32 order is the
        only difference.  share: a dict — the passes' RPN inputs hold the SAME values (RegionProposalNetwork.forward_and_propose): labels,
        proposals and candidate lists are computed once (by the first pass, or by an earlier pass that left its state in the dict)."""
        pend, feats = [], []
        for d in dicts:
            assert d["flag"] in ("tail", "clean") and type(d["out_idx"]) == int and self.training
            feats.append(self._cut(self.features(d)))
        # the RPN's trunk convolution and its two 1x1 heads once for all passes (the same argument: no BatchNorm at all here)
        heads = [None] * len(dicts)
        if BATCH_RPN_TRUNK and len(feats) > 1 and all(f.shape == feats[0].shape for f in feats):
            b = feats[0].shape[0]
            obj_all, tr_all = self.rpn._heads(self.rpn._trunk(torch.cat(feats, dim=0)))
            heads = [(obj_all[k * b:(k + 1) * b], tr_all[k * b:(k + 1) * b]) for k in range(len(feats))]
        for d, features, hd in zip(dicts, feats, heads):
            anchors, iw, ih = self._anchors(features, d["x"].shape)
            _, _, ao, at, proposals, roi_t = self.rpn.forward_and_propose(features, anchors, gt_bboxes_batch, iw, ih,
                                                                          roi_targets=self._roi_targets(gt_classes_batch, gt_bboxes_batch), heads=hd,
                                                                          share=share)
            if roi_t is None:
                roi_t = self.detection._targets(proposals, gt_classes_batch, gt_bboxes_batch)
            pend.append((features, ao, at, roi_t))
        return self.detection.forward_many(pend)

    MERGE_READS = os.environ.get("AFAN_DET_MERGE_READS", "1") != "0"       # 0: the ROI head's sampling with its own host read (A/B, tests)

    def _roi_targets(self, gt_classes_batch, gt_bboxes_batch):
        if not Model.MERGE_READS:
            return None
        det = self.detection
        return lambda padded, limit: det._targets_pending(padded, limit, gt_classes_batch, gt_bboxes_batch)

    class Detection(nn.Module):
        """model.py:231-367."""

        def __init__(self, pooler_mode, hidden, num_hidden_out, num_classes, proposal_smooth_l1_loss_beta):
            super().__init__()
            self._pooler_mode = pooler_mode
            self.hidden = hidden
            self.num_classes = num_classes
            self._proposal_class = nn.Linear(num_hidden_out, num_classes)
            self._proposal_transformer = nn.Linear(num_hidden_out, num_classes * 4)
            self._proposal_smooth_l1_loss_beta = proposal_smooth_l1_loss_beta
            self._transformer_normalize_mean = torch.tensor([0., 0., 0., 0.], dtype=torch.float)
            self._transformer_normalize_std = torch.tensor([.1, .1, .2, .2], dtype=torch.float)

        def _roi_features(self, features, boxes, batch_indices):
            """Pooler -> layer4 -> global max: [R, 2048, 1, 1]."""
            h = _run_stage(self.hidden, _to_compute(pool_rois(features, boxes, batch_indices, self._pooler_mode), self.hidden[0].conv1.compute_dtype))
            return _GlobalMaxFn.apply(h)

        def _linears(self, hidden):
            x = hidden.view(hidden.shape[0], -1).float()
            ys = _linear_pair(x.contiguous(), self._proposal_class, self._proposal_transformer) if LINEAR_PAIR else None
            if ys is not None:
                return ys
            return _linear(x, self._proposal_class), _linear(x, self._proposal_transformer)

        def _targets(self, proposal_bboxes, gt_classes_batch, gt_bboxes_batch):
            """:256-282 (repeated at :300-326): IoU >= 0.5 takes its ground truth's class, 32 / 128 per image sampled."""
            b = proposal_bboxes.shape[0]
            labels, assign = box_assign(proposal_bboxes, gt_bboxes_batch, "proposal", 0.5, gt_classes=gt_classes_batch)
            _, boxes, lab, deltas, bi = fg_bg_sample(labels, assign, proposal_bboxes, gt_bboxes_batch, 32 * b, 128 * b)
            return boxes, lab, deltas, bi

        def _targets_pending(self, padded, limit, gt_classes_batch, gt_bboxes_batch):
            """The label / list launches of `_targets` on the PADDED proposals [B, top_n, 4] (rows >= max(limit) = the longest image's
            survivor count — limit [B] int64 on the device — are no candidates: label -1), nothing read: the caller's one host read
            brings `.counts`, `.finish(nf, nb)` draws."""
            labels, assign = box_assign(padded, gt_bboxes_batch, "proposal", 0.5, gt_classes=gt_classes_batch)
            labels_limit_(labels, limit)
            return _RoiPending(sample_lists(labels), labels, assign, padded, gt_bboxes_batch)

        def forward(self, features, proposal_bboxes=None, gt_classes_batch=None, gt_bboxes_batch=None, return_type="clean", targets=None):
            if return_type == "tail":
                d = features
                classes, transformers = self._linears(d["roi_feature_map"])
                ce, sl1 = self.loss(classes, transformers, d["gt_proposal_classes"], d["gt_proposal_transformers"],
                                    _int_of(d["batch_size"]), d["batch_indices"])
                return classes, transformers, ce, sl1
            b = features.shape[0]
            if return_type == "clean" and not self.training:
                bi = torch.arange(end=b, dtype=torch.long, device=proposal_bboxes.device).view(-1, 1).repeat(1, proposal_bboxes.shape[1])
                classes, transformers = self._linears(self._roi_features(features, proposal_bboxes.view(-1, 4), bi.view(-1)))
                return classes.view(b, -1, classes.shape[-1]), transformers.view(b, -1, transformers.shape[-1])
            boxes, gt_classes, gt_deltas, bi = targets if targets is not None else self._targets(proposal_bboxes, gt_classes_batch, gt_bboxes_batch)
            hidden = self._roi_features(features, boxes, bi)
            if return_type == "head":
                return {"roi_feature_map": hidden, "gt_proposal_classes": gt_classes, "gt_proposal_transformers": gt_deltas,
                        "batch_size": _int_tensor(b, hidden.device), "batch_indices": bi}
            assert return_type == "clean"
            classes, transformers = self._linears(hidden)
            ce, sl1 = self.loss(classes, transformers, gt_classes, gt_deltas, b, bi)
            return classes, transformers, ce, sl1

        def forward_many(self, pend):
            """pend: [(features [B, C, H, W], anchor ce, anchor sl1, (boxes, gt_classes, gt_deltas, batch_indices))] of independent training
            passes -> [(anchor ce, anchor sl1, proposal ce, proposal sl1)]: `forward(..., targets=...)` per pass with ONE `_roi_features` /
            `_linears` over all passes' regions (Model.forward_heads_many)."""
            b = pend[0][0].shape[0]
            feats = torch.cat([p[0] for p in pend], dim=0) if len(pend) > 1 else pend[0][0]
            boxes = torch.cat([p[3][0] for p in pend], dim=0)
            bi_all = torch.cat([p[3][3] + k * b for k, p in enumerate(pend)], dim=0)
            classes, transformers = self._linears(self._roi_features(feats, boxes, bi_all))
            out, o = [], 0
            for _, ao, at, (bx, gt_classes, gt_deltas, bi) in pend:
                n = bx.shape[0]
                ce, sl1 = self.loss(classes[o:o + n], transformers[o:o + n], gt_classes, gt_deltas, b, bi)
                out.append((ao, at, ce, sl1))
                o += n
            return out

        def loss(self, proposal_classes, proposal_transformers, gt_proposal_classes, gt_proposal_transformers, batch_size, batch_indices):
            """:343-367: the regression output of each sample's OWN class, targets normalised by (0, 0, 0, 0) / (.1, .1, .2, .2)."""
            norm = tuple(self._transformer_normalize_mean.tolist()) + tuple(self._transformer_normalize_std.tolist())
            return per_image_losses(proposal_classes, proposal_transformers, None, gt_proposal_classes, gt_proposal_transformers,
                                    batch_indices, batch_size, self._proposal_smooth_l1_loss_beta, norm=norm)

        def generate_detections(self, proposal_bboxes, proposal_classes, proposal_transformers, image_width, image_height):
            """:369-407 (inference): per-class decode, clip, softmax, NMS at 0.3."""
            b = proposal_bboxes.shape[0]
            tr = proposal_transformers.view(b, -1, self.num_classes, 4)
            tr = tr * self._transformer_normalize_std.to(tr.device) + self._transformer_normalize_mean.to(tr.device)
            boxes = box_clip_(box_apply(proposal_bboxes.unsqueeze(dim=2).repeat(1, 1, self.num_classes, 1), tr), image_width, image_height)
            probs = F.softmax(proposal_classes, dim=-1)
            out_b, out_c, out_p, out_i = [], [], [], []
            for bi in range(b):
                for c in range(1, self.num_classes):
                    cb, cp = boxes[bi, :, c, :], probs[bi, :, c]
                    keep = nms(cb, cp, 0.3).to(cb.device)
                    out_b.append(cb[keep])
                    out_c.append(torch.full((len(keep),), c, dtype=torch.int))
                    out_p.append(cp[keep])
                    out_i.append(torch.full((len(keep),), bi, dtype=torch.long))
            return torch.cat(out_b, dim=0), torch.cat(out_c, dim=0), torch.cat(out_p, dim=0), torch.cat(out_i, dim=0)


def fasterrcnn_resnet101(num_classes=21, pooler_mode="align", **kw):
    """train_aug_sat_muti_advt.py:36-44 with the VOC configuration (config/train_config.py): 21 classes, 9 anchors."""
    return Model(ResNet101(), num_classes, pooler_mode, **kw)
