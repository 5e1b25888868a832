"""Split-forward ResNets for the A-FAN hot path, MI355X-native.

Host-side mirror of the reference's model interface (Classification/resnet_s.py:79-124):
  * one flat `sequential_model`, `model(x, end_point=E, start_point=S)` == `sequential_model[S:E](x)`
    (resnet_s.py:119-121) — the protocol `PGD` calls back into;
  * state_dict keys identical to the reference's (`w`, `sequential_model.0.mean/std`,
    `sequential_model.<i>.{conv1,bn1,conv2,bn2}.*`, ...), so checkpoints interchange.
What differs is how it executes: BatchNorm (+residual, +ReLU) runs as fused HIP kernels from
libafan_hip.so (two launches forward, two backward, instead of ~6 eager ones), convolutions go to
MIOpen's MFMA kernels in the model's compute dtype (bf16 by default) from cached low-precision weights,
and the slice executor fuses BN->ReLU pairs that sit at adjacent indices of the Sequential.
ResNet-18 (CIFAR stem) is build-defined: the reference ships only resnet56 (SURVEY.md warning 3).
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

__all__ = ["ResNet", "ResNet50", "BasicBlock", "Bottleneck", "resnet20", "resnet56", "resnet18", "resnet50", "ARCHS",
           "dgrad_only"]


class _Flags:
    param_grads = True  # False inside PGD: only d(loss)/d(feature) is needed (attack_algo.py:52 only_inputs=True)
    weight_epoch = 0    # bumped by the arena's fused SGD step (it updates weights without touching tensor versions)
    block_fusion = True  # BasicBlock as one autograd node on the bf16 channels-last fast path (_BlockFn)
    wgrad_stash = False  # True: block weight gradients are not launched but their operands kept on the conv module (stash_wgrad)
    bn_groups = 1        # 2: the batch is [adv half | clean half]; BatchNorm statistics / running updates per half, in order
    bn_branch = "main"   # "adv" inside bn_branch(model, "adv"): BatchNorms with an auxiliary set use it (dual-BN option)


@contextlib.contextmanager
def dgrad_only():
    """Inside this context the HIP-backed layers skip weight/affine gradients (the PGD inner loop)."""
    old = _Flags.param_grads
    _Flags.param_grads = False
    try:
        yield
    finally:
        _Flags.param_grads = old


@contextlib.contextmanager
def bn_groups(g):
    """Inside this context a batch is g concatenated half-batches that BatchNorm treats as separate forward passes
    (statistics, running-stat updates in order, backward sums) while convolutions run once over the whole batch:
    main_perturb.py:195-196's adversarial and clean passes as one pass over the tail.  Only the one-node residual
    blocks implement it; any other BatchNorm raises."""
    old = _Flags.bn_groups
    _Flags.bn_groups = int(g)
    try:
        yield
    finally:
        _Flags.bn_groups = old


def enable_dual_bn(model):
    """Dual-BN option (north-star wording; the reference has ONE BatchNorm set, SURVEY section 1 — default off): give every
    BatchNorm2d of `model` an auxiliary parameter / running-statistics set for adversarial features, initialised as a copy
    of the main one.  state_dict gains `<bn>.adv.{weight,bias,running_mean,running_var,num_batches_tracked}`; eval mode and
    the clean passes use the main set.  Call before the parameter arena is built."""
    bns = [m for m in model.modules() if isinstance(m, BatchNorm2d)]
    for b in bns:
        b.enable_dual()
    model._dual_bns = bns
    return model


@contextlib.contextmanager
def bn_branch(model, branch):
    """Inside: the BatchNorms of a dual-BN model normalise with their `branch` set ("adv" or "main") — affine parameters,
    their gradients and the running statistics.  No-op for a model without auxiliary sets."""
    bns = getattr(model, "_dual_bns", None)
    if not bns:
        yield
        return
    old = _Flags.bn_branch
    for b in bns:
        b.select(branch)
    _Flags.bn_branch = branch
    try:
        yield
    finally:
        for b in bns:
            b.select(old)
        _Flags.bn_branch = old


def _bn_fwd_g(raw, b, res, relu, st, G, mom):
    """BatchNorm forward of one layer over G half-batches (G = 1: the plain call)."""
    if G == 1 or st.acc is not None:     # one launch: the kernel walks the groups (statistics per half, updates in order)
        return ops.bn_train_forward(raw, b.weight, b.bias, res, relu, b.eps, mom, b.running_mean, b.running_var,
                                    b.num_batches_tracked, st, groups=G)
    c, n = raw.shape[1], raw.shape[0] // G
    y = torch.empty_like(raw)
    stats = torch.empty(G, 4, c, dtype=torch.float32, device=raw.device)
    for g in range(G):                         # in order: the adversarial half updates the running statistics first
        sl = slice(g * n, (g + 1) * n)
        ops.bn_train_forward(raw[sl], b.weight, b.bias, None if res is None else res[sl], relu, b.eps, mom,
                             b.running_mean, b.running_var, b.num_batches_tracked, st.group(g, c), out=y[sl],
                             stats_out=stats[g])
    return y, stats


def _bn_bwd_g(dy, raw, y, stats, b, relu, want_dres, pg, part, G, dx_out=None):
    gw, gb = (b.weight.grad, b.bias.grad) if pg else (None, None)
    if G == 1 or (part is not None and part.acc is not None):
        return ops.bn_backward(dy, raw, y, stats, b.weight, b.bias, relu, want_dres, gw, gb, pg, partials=part, groups=G,
                               dx_out=dx_out)
    c, n = raw.shape[1], raw.shape[0] // G
    dx = torch.empty_like(raw)
    dres = torch.empty_like(raw) if want_dres else None
    for g in range(G):
        sl = slice(g * n, (g + 1) * n)
        ops.bn_backward(dy[sl], raw[sl], None if y is None else y[sl], stats[g], b.weight, b.bias, relu, want_dres, gw, gb,
                        pg, partials=None if part is None else part.group(g, c), dx_out=dx[sl],
                        dres_out=dres[sl] if want_dres else None)
    return dx, dres


# ------------------------------------------------------------------------------------------------ ops
class _CastFn(torch.autograd.Function):
    """fp32 -> bf16 through afan_cast_bf16; the gradient flows back in fp32."""

    @staticmethod
    def forward(ctx, x):
        return ops.cast_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g.float()


def _to_compute(x, dtype):
    if x.dtype == dtype:
        return x
    if x.dtype == torch.float32 and dtype == torch.bfloat16:
        return _CastFn.apply(_dense(x))
    return x.to(dtype)


def _dense(t):
    """Dense NCHW or dense channels_last, whichever `t` already is (copy only if it is neither)."""
    if t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
        return t
    return t.contiguous()


def _like_layout(t, ref):
    """`t` with `ref`'s memory layout (BN operands and incoming gradients must share strides)."""
    if t.stride() == ref.stride():
        return t
    if ref.dim() == 4 and ops.layout_of(ref) == ops.AFAN_NHWC:
        return t.contiguous(memory_format=torch.channels_last)
    return t.contiguous()


def _sq(v, what):
    """The convolutions of the path are square in stride / padding / dilation (every layer of the reference's networks)."""
    if v[0] != v[1]:
        raise ops.AfanLibraryError(f"convolution with a non-square {what} {tuple(v)} is not implemented by libafan_hip")
    return int(v[0])


class _ConvFn(torch.autograd.Function):
    """Convolution on the compute-dtype weight copy, always on the library's own kernels: bf16 channels-last tensors of
    the shapes the tuned implicit-GEMM MFMA kernels take run those (afan_conv_*_nhwc_bf16: forward, input gradient,
    weight gradient); everything else — fp32 parity mode in either layout, the 3-channel stems at odd widths, NCHW
    bf16 — runs the general fp32-arithmetic kernels (afan_conv_fwd / _dgrad / _wgrad, v_mfma_f32_32x32x2_f32).  There is
    no vendor-library path: a shape neither family takes raises AfanLibraryError.  wgrad is returned for (or added
    straight into) the fp32 master weight."""

    @staticmethod
    def forward(ctx, x, w_master, w_lp, wt_fn, stride, padding, want_wgrad, stats_req=None, dilation=(1, 1), bias=None):
        ctx.stride, ctx.padding, ctx.want_wgrad, ctx.wt_fn = stride, padding, want_wgrad, wt_fn
        ctx.dilation = dilation = tuple(dilation)
        ctx.w_master = w_master
        ctx.own = _own_conv_ok(x, w_lp, stride, padding, dilation) and bias is None
        ctx.has_bias = bias is not None
        if ctx.own:
            ctx.save_for_backward(x, w_lp)
            if stats_req is not None:   # [shift tensor | None, reusable partials buffer | None] -> filled with ConvStats
                G = _Flags.bn_groups
                if G != 1 and not ops._conv_acc_ok(w_lp.shape[0]):
                    stats_req.append(None)                   # (no accumulator form for this channel count: the BatchNorm sums per half itself)
                    return ops.conv_fwd(x, w_lp, stride[0], dilation=dilation[0])
                y, st = ops.conv_fwd(x, w_lp, stride[0], stats_shift=stats_req[0], want_stats=True,
                                     stats_buf=stats_req[1], dilation=dilation[0], groups=G)
                stats_req.append(st)
                return y
            return ops.conv_fwd(x, w_lp, stride[0], dilation=dilation[0])
        x = _dense(x)
        ctx.save_for_backward(x, w_lp)
        return ops.conv_general_fwd(x, w_lp, None if bias is None else bias.detach().float(), _sq(stride, "stride"),
                                    _sq(padding, "padding"), _sq(dilation, "dilation"))

    @staticmethod
    def backward(ctx, gy):
        x, w_lp = ctx.saved_tensors
        need_gx = ctx.needs_input_grad[0]
        need_gw = ctx.want_wgrad and ctx.needs_input_grad[1]
        need_gb = ctx.has_bias and ctx.want_wgrad and ctx.needs_input_grad[9]
        gy = _like_layout(gy, x) if x.dim() == 4 and gy.shape[2:] == x.shape[2:] else _dense(gy)
        if ops.layout_of(gy) != ops.layout_of(x) and gy.shape[1] > 1 and gy.shape[2] * gy.shape[3] > 1:
            gy = gy.contiguous(memory_format=torch.channels_last if ops.layout_of(x) == ops.AFAN_NHWC else torch.contiguous_format)
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gx = gw = gb = None
        k = w_lp.shape[2]
        st, pad, dil = ctx.stride[0], ctx.padding[0], ctx.dilation[0]
        own = ctx.own and gy.is_contiguous(memory_format=torch.channels_last)
        if need_gx:
            if own:
                gx = ops.conv_dgrad(gy, ctx.wt_fn(), x.shape[2:], st, dilation=dil)
            else:
                gx = ops.conv_general_dgrad(gy, w_lp, x.shape[2:], st, pad, dil)
        if need_gw:
            w_master = ctx.w_master
            direct = _accumulates_in_place(w_master)
            if own and ops.conv_wgrad_supported(x.shape[1], gy.shape[1], k, st, (x.shape[0], x.shape[2], x.shape[3])):
                if direct and (w_master.grad.is_contiguous(memory_format=torch.channels_last) or k == 1):
                    # summed by the kernel straight into the fp32 gradient arena (KRSC): nothing goes back through autograd
                    g_ = w_master.grad
                    _WgradStream.run(lambda: ops.conv_wgrad(x, gy, k, st, g_, accumulate=True, dilation=dil), x, gy)
                else:
                    gw = ops.conv_wgrad(x, gy, k, st, dilation=dil)
            elif direct and _dense_weight(w_master.grad):
                g_ = w_master.grad
                _WgradStream.run(lambda: ops.conv_general_wgrad(x, gy, k, st, pad, dil, grad=g_, accumulate=True), x, gy)
            else:
                gw = ops.conv_general_wgrad(x, gy, k, st, pad, dil)
        if need_gb:
            gb = gy.float().sum(dim=(0, 2, 3))
        return gx, gw, None, None, None, None, None, None, None, gb


def _dense_weight(g):
    return g.is_contiguous() or g.is_contiguous(memory_format=torch.channels_last)


class _WgradStream:
    """Weight-gradient launches off the critical path: nothing in the backward waits for a weight gradient, so while the
    autograd engine runs (and the switch is on), `_wgrad_accumulate` issues its launches on ONE side stream per device —
    ordered after the producer of dy by an event, in program order among themselves, so the sums into the fp32 gradient
    arena happen in the same order as on the main stream: bit-identical results (tests/test_graph_safety_gpu.py) — and the
    engine's end-of-backward callback makes the main stream wait for it.  Tensors the launches read are held until that
    join.  In a captured iteration the side stream is a parallel branch of the hipGraph.  Measured (MI355X, graph replay):
    DeepLabv3+ R101 513^2 batch 8: 55.2 -> 52.6 ms; batch 2: 25.3 -> 25.5; ResNet-18 batch 256: 9.82 -> 9.95 — the replayed
    graph overlaps the TAILS of neighbouring independent kernels, it does not run branches side by side (batching the side
    launches in groups of 8-64 loses the gain: 55.3 ms), so it pays where kernels are long.  Hence off by default; the
    trainers switch it on by workload size (seg_trainer.SegTrainer) and AFAN_WGRAD_STREAM=1/0 forces it."""
    FORCE = os.environ.get("AFAN_WGRAD_STREAM")      # "1" / "0": override every trainer's choice (A/B)
    ON = FORCE == "1"
    streams = {}
    held = []
    mains = []
    task = -1            # the graph task whose end-of-backward callback is queued

    @classmethod
    def run(cls, fn, *tensors):
        t0 = tensors[0]
        if not cls.ON or not t0.is_cuda or torch._C._current_graph_task_id() < 0:
            return fn()                                 # outside a backward (flush_wgrad, direct calls): nothing to overlap with
        dev = t0.device
        side = cls.streams.get(dev.index)
        if side is None:
            side = cls.streams[dev.index] = torch.cuda.Stream(device=dev)
            if cls.FORCE == "1" and ops.GRID_BN:        # (forced on outside a wgrad_stream() context: tell the library once)
                ops.grid_shared(True)
        main = torch.cuda.current_stream(dev)
        # one join per backward (graph task): keyed on the task id, not on `held` being empty — a backward that raised after
        # its first side launch never ran its callback, and its leftovers must not keep later backwards from queueing theirs
        task = torch._C._current_graph_task_id()
        if task != cls.task:
            if cls.held:                                # leftovers of an aborted backward: wait for them here
                cls.join()
            cls.task = task
            torch.autograd.Variable._execution_engine.queue_callback(cls.join)
        if main not in cls.mains:
            cls.mains.append(main)
        cls.held.append(tensors)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn()

    @classmethod
    def join(cls):
        for main in cls.mains:
            side = cls.streams.get(main.device.index)
            if side is not None:
                main.wait_stream(side)
        cls.mains.clear()
        cls.held.clear()
        cls.task = -1


@contextlib.contextmanager
def wgrad_stream(on):
    """Weight gradients of the backward passes inside this context on the side stream (see _WgradStream)."""
    old = _WgradStream.ON
    _WgradStream.ON = bool(on) if _WgradStream.FORCE is None else _WgradStream.FORCE == "1"
    # (side-stream kernels run beside the main stream's: the in-launch BatchNorm then only takes launches of one workgroup per CU)
    old_shared = ops.grid_shared(_WgradStream.ON) if (ops.GRID_BN and torch.cuda.is_available()) else None
    try:
        yield
    finally:
        _WgradStream.ON = old
        if old_shared is not None:
            ops.grid_shared(old_shared)


class _WgradBatch:
    """Inside a block's backward: the block's weight-gradient problems are collected and issued together at the end — those
    that the tuned kernel takes with the SAME tile configuration as ONE launch + one reduction launch
    (ops.conv_wgrad_multi: bit-identical to separate launches), the rest one by one.  At DeepLab's per-GPU share a
    bottleneck's three weight gradients are 35-step reductions of 13-25 us each; together they cost the time of one."""
    ON = os.environ.get("AFAN_WGRAD_MULTI", "1") != "0"
    pending = None          # list while a block backward collects

    @classmethod
    def flush(cls):
        items, cls.pending = cls.pending, None
        if not items:
            return
        groups = {}
        for it in items:
            x, dy, c = it
            ok = (not _Flags.wgrad_stash and getattr(c, "_pending_wgrad", None) is None and _accumulates_in_place(c.weight)
                  and (c.weight.grad.is_contiguous(memory_format=torch.channels_last) or c.kernel_size[0] == 1))
            groups.setdefault(ops.conv_wgrad_plan(x, dy, c.kernel_size[0], c.stride[0]) if ok else 0, []).append(it)
        for code, its in groups.items():
            while code and len(its) >= 2:
                chunk, its = its[:4], its[4:]
                args = [(x, dy, c.kernel_size[0], c.stride[0], c.dilation[0], c.weight.grad) for x, dy, c in chunk]
                _WgradStream.run(lambda args=args: ops.conv_wgrad_multi(args), *[t for a in args for t in a[:2]])
            for x, dy, c in its:
                _wgrad_accumulate(x, dy, c, batch=False)


def _wgrad_accumulate(x, dy, c, batch=True):
    """Weight gradient of conv module c added into its arena gradient view (queued on the weight-gradient stream while a
    backward runs, see _WgradStream; collected per block, see _WgradBatch)."""
    if batch and _WgradBatch.pending is not None:
        _WgradBatch.pending.append((x, dy, c))
        return
    if _Flags.wgrad_stash and ops.conv_wgrad_supported(x.shape[1], dy.shape[1], c.kernel_size[0], c.stride[0],
                                                       (x.shape[0], x.shape[2], x.shape[3])) \
            and ops.wgrad_pairable(x, dy, c.kernel_size[0], c.stride[0]):
        c._pending_wgrad = (x, dy)          # summed into the launch of the next pass over this layer (stash_wgrad)
        return
    pend = getattr(c, "_pending_wgrad", None)
    _WgradStream.run(lambda: _wgrad_launch(x, dy, c), x, dy, pend)


def _wgrad_launch(x, dy, c):
    """The tuned bf16 kernels where they tile, else the general fp32-arithmetic kernel on the same bf16 tensors."""
    k, st, dil = c.kernel_size[0], c.stride[0], c.dilation[0]
    if ops.conv_wgrad_supported(x.shape[1], dy.shape[1], k, st, (x.shape[0], x.shape[2], x.shape[3])):
        pairable = ops.wgrad_pairable(x, dy, k, st)
        pend = getattr(c, "_pending_wgrad", None)
        if pend is not None:
            c._pending_wgrad = None
            if pairable and pend[0].shape[1:] == x.shape[1:] and pend[1].shape[1:] == dy.shape[1:]:
                ops.conv_wgrad(pend[0], pend[1], k, st, c.weight.grad, accumulate=True, second=(x, dy), dilation=dil)
                return
            ops.conv_wgrad(pend[0], pend[1], k, st, c.weight.grad, accumulate=True, dilation=dil)
        ops.conv_wgrad(x, dy, k, st, c.weight.grad, accumulate=True, dilation=dil)
        return
    ops.conv_general_wgrad(x, dy, k, st, c.padding[0], dil, grad=c.weight.grad, accumulate=True)


class stash_wgrad:
    """Context for the backward of the FIRST of two passes over the same layers within one iteration (the clean and the
    adversarial tail pass): the residual blocks' weight-gradient launches are held back — each conv module keeps its
    (input, output-gradient) pair — and the next backward over the layer sums both pairs in ONE launch and one slab
    reduction (afan_conv_wgrad2_nhwc_bf16).  flush_wgrad() afterwards launches whatever was not picked up."""

    ON = os.environ.get("AFAN_WGRAD_STASH", "1") != "0"     # 0: every pass launches its own weight gradients (A/B)

    def __enter__(self):
        self.old = _Flags.wgrad_stash
        _Flags.wgrad_stash = self.ON
        return self

    def __exit__(self, *exc):
        _Flags.wgrad_stash = self.old
        return False


def flush_wgrad(model):
    for c in model.modules():
        pend = getattr(c, "_pending_wgrad", None)
        if pend is not None:
            c._pending_wgrad = None
            ops.conv_wgrad(pend[0], pend[1], c.kernel_size[0], c.stride[0], c.weight.grad, accumulate=True,
                           dilation=c.dilation[0])


def _accumulates_in_place(p):
    """True for a leaf parameter whose .grad buffer is pre-attached and owned by a ParamArena (train_step zeroes it
    at the start of every step), so a kernel may add into it directly instead of going through AccumulateGrad."""
    return isinstance(p, torch.nn.Parameter) and p.grad is not None and getattr(p, "_afan_arena_grad", False)


def _own_conv_ok_shape(w, stride, padding, dilation=(1, 1)):
    if w.dtype != torch.bfloat16 or w.shape[2] != w.shape[3]:
        return False
    k = w.shape[2]
    if not (w.is_contiguous(memory_format=torch.channels_last) or k == 1):
        return False
    d = dilation[0]
    if stride[0] != stride[1] or dilation[0] != dilation[1] or padding[0] != d * (k // 2) or padding[1] != d * (k // 2):
        return False
    return ops.conv_supported(w.shape[1], w.shape[0], k, stride[0], d if k > 1 else 1)


def general_convs(model):
    """Names of the convolutions of `model` that, in its present configuration, run on the general fp32-arithmetic
    kernels (afan_conv_f32.hip: fp32 parity mode, NCHW weights, shapes the tuned bf16 kernels decline) instead of the
    tuned bf16 MFMA kernels; the image stem (3 input channels) is not listed (it decides by image width at run time).
    Informational: both families are the library's own and both are hipGraph-capturable."""
    out = []
    nchw = not getattr(model, "channels_last", True)          # NCHW activations: nothing reaches the tuned (channels-last) kernels
    for name, m in model.named_modules():
        if isinstance(m, Conv2d) and m.in_channels > 4 and not getattr(m, "own_kernel", False):
            if nchw or m.compute_dtype != torch.bfloat16 or not _own_conv_ok_shape(m.lp_weight(), m.stride, m.padding, m.dilation):
                out.append(name)
    return out


def vendor_convs(model):
    """Convolutions of `model` that would leave the library for the vendor's: none, by construction — rounds 1-2 sent fp32
    parity mode and odd shapes to MIOpen through aten; those branches are gone (see _ConvFn).  Kept as the invariant the
    entry points log and the tests assert."""
    return []


def _own_conv_ok(x, w, stride, padding, dilation=(1, 1)):
    if x.dtype != torch.bfloat16 or x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    if x.shape[1] == 3 and x.shape[3] % 32:      # the stem kernel walks 32-pixel row segments
        return False
    # the tuned kernels address activations with 32-bit byte offsets (afan_conv.hip check_dims: the larger of the two
    # tensors + the DMA descriptor's bias); a larger problem goes to the general kernels (64-bit strides) instead of raising
    c = max(int(w.shape[0]), int(w.shape[1]))
    if x.shape[0] * x.shape[2] * x.shape[3] * c * 2 + 2 * int(dilation[0]) * (x.shape[3] + 1) * c > 0x7FFFFFFF:
        return False
    return _own_conv_ok_shape(w, stride, padding, dilation)


class _BNTrainFn(torch.autograd.Function):
    """y = [relu](bn_train(x) [+ residual]) via afan_bn_train_forward / afan_bn_backward.  groups = 2 (resnet_s.bn_groups): x
    is two concatenated half-batches normalised separately, running statistics updated half by half in order — one launch
    where the producing convolution summed per half (conv_stats with accumulators), else one launch per half."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, relu, eps, momentum, rmean, rvar, nbt, want_pgrad, conv_stats=None, groups=1):
        x = _dense(x)
        if residual is not None:
            residual = _like_layout(residual, x)
        if groups == 1:
            y, stats = ops.bn_train_forward(x, weight, bias, residual, relu, eps, momentum, rmean, rvar, nbt, conv_stats)
        elif conv_stats is not None and conv_stats.acc is not None:
            y, stats = ops.bn_train_forward(x, weight, bias, residual, relu, eps, momentum, rmean, rvar, nbt, conv_stats, groups=groups)
        else:
            n = x.shape[0] // groups
            y = torch.empty_like(x)
            stats = torch.empty(groups, 4, x.shape[1], dtype=torch.float32, device=x.device)
            for g in range(groups):
                sl = slice(g * n, (g + 1) * n)
                ops.bn_train_forward(x[sl], weight, bias, None if residual is None else residual[sl], relu, eps, momentum, rmean, rvar,
                                     nbt, None, out=y[sl], stats_out=stats[g])
        ctx.relu, ctx.has_res, ctx.want_pgrad, ctx.groups = relu, residual is not None, want_pgrad, groups
        # the ReLU mask is recomputed from x when there is no residual; otherwise y carries it
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, stats, weight, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, stats, weight, bias = ctx.saved_tensors
        want_p = ctx.want_pgrad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        dw = db = None
        direct = False
        if want_p:
            # parameters whose .grad is pre-attached (ParamArena views) are accumulated by the kernel itself:
            # no temporary, no AccumulateGrad add launch.  Otherwise hand the gradients to autograd as usual.
            direct = _accumulates_in_place(weight) and _accumulates_in_place(bias)
            if direct:
                dw, db = weight.grad, bias.grad
            else:
                dwb = torch.empty(2, x.shape[1], dtype=torch.float32, device=x.device)
                dw, db = dwb[0], dwb[1]
        want_dres = ctx.has_res and ctx.needs_input_grad[3]
        gy = _like_layout(gy, x)
        if ctx.groups == 1:
            dx, dres = ops.bn_backward(gy, x, y, stats, weight, bias, ctx.relu, want_dres, dw, db, accumulate=direct)
        else:
            G, n = ctx.groups, x.shape[0] // ctx.groups
            dx = torch.empty_like(x)
            dres = torch.empty_like(x) if want_dres else None
            for g in range(G):                     # (both halves' parameter gradients: the second launch adds to the first's)
                sl = slice(g * n, (g + 1) * n)
                ops.bn_backward(gy[sl], x[sl], None if y is None else y[sl], stats[g], weight, bias, ctx.relu, want_dres, dw, db,
                                accumulate=direct or g > 0, dx_out=dx[sl], dres_out=dres[sl] if want_dres else None)
        if direct:
            dw = db = None
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None


class _BlockFn(torch.autograd.Function):
    """A whole residual block as ONE autograd node on the bf16 channels-last fast path.  The main branch is a chain of
    conv -> BN stages (2 for a BasicBlock, 3 for a Bottleneck), ReLU after each, the last one after adding the shortcut:
         raw_i = conv_i(a_{i-1}) [+moments]   a_i = relu(bn_i(raw_i))   out = relu(bn_n(raw_n) + shortcut(x))
    The hand-ordered backward fuses what autograd's per-op graph cannot: the reduction pass of bn_{i-1}'s backward rides
    in conv_i's dgrad epilogue, the sum of the two gradient branches arriving at the block input rides in conv_1's dgrad
    epilogue (no separate add launches), parameter gradients are accumulated by the kernels into the arena views, and
    the dgrad producing the gradient for the PREVIOUS block's output also takes that block's last-BN backward sums.
    `params` only tell autograd which leaves the node depends on; values are read from the modules."""
    MULTI_SC = os.environ.get("AFAN_BLOCK_MULTI_SC", "1") != "0"     # 0: the projection shortcut as its own launch (A/B)

    @staticmethod
    def forward(ctx, x, blk, want_pgrad, *params):
        chain = blk._chain()
        n = len(chain)
        G = _Flags.bn_groups
        mom = lambda bn: bn.momentum if bn.momentum is not None else 0.1
        rawsc = ssc = first = first_done = None
        dual_sc = False
        if blk._sc_kind == "conv":
            csc, bsc = blk.shortcut[0], blk.shortcut[1]
            c1, b1 = chain[0]
            if (_BlockFn.MULTI_SC and G == 1 and c1.stride == csc.stride and c1.dilation[0] == 1
                    and c1.out_channels == csc.out_channels
                    and ops.conv_fwd_multi_ok(x, [c1.lp_weight(), csc.lp_weight()], c1.stride[0])):
                # a BasicBlock's first 3x3 and its 1x1 projection read the same x at the same stride and write the same
                # shape: one launch, two problems (afan_conv_fwd_multi_nhwc_bf16)
                # ... with the first convolution's BatchNorm + ReLU inside that launch where its workgroups fit the chip
                fm = ops.conv_fwd_multi_bn(x, [c1.lp_weight(), csc.lp_weight()], c1.stride[0], [1, 1],
                                           [b1.running_mean, bsc.running_mean], b1, mom(b1)) if ops.GRID_BN else None
                if fm is not None:
                    (raw1, rawsc), (st1, stc), act1, s1 = fm
                    first_done = (act1, s1)
                else:
                    (raw1, rawsc), (st1, stc) = ops.conv_fwd_multi(x, [c1.lp_weight(), csc.lp_weight()], c1.stride[0], [1, 1],
                                                                   [b1.running_mean, bsc.running_mean])
                first = (raw1, st1)
            else:
                rawsc, stc = ops.conv_fwd(x, csc.lp_weight(), csc.stride[0], stats_shift=bsc.running_mean, want_stats=True,
                                          stats_buf=csc._stats_buf, groups=G)      # (1x1: no dilation)
                csc._stats_buf = stc.partials
            # with both convolutions' moments in accumulator blocks the projection's BatchNorm is applied by the block's
            # LAST launch (ops.bn_train_forward_dual: relu(bn_n(raw_n) + bn_sc(raw_sc))), its output tensor never written
            dual_sc = first is not None and stc.acc is not None
            if not dual_sc:
                res, ssc = _bn_fwd_g(rawsc, bsc, None, False, stc, G, mom(bsc))
        elif blk._sc_kind == "pad":
            # option-A shortcut (resnet_s.py:64-65): every second pixel, zero channels either side — data movement only
            res = blk.shortcut(x).contiguous(memory_format=torch.channels_last)
        else:
            res = x
        a, saved = x, []
        for i, (c, b) in enumerate(chain):
            last = i == n - 1
            if i == 0 and first_done is not None and not last:
                saved += [first[0], first_done[0], first_done[1]]
                a = first_done[0]
                continue
            if G == 1 and ops.GRID_BN and not (i == 0 and first is not None) and c.kernel_size[0] in (1, 3) and c.stride[0] == 1:
                # convolution + BatchNorm (+ shortcut) + ReLU as ONE launch (grid barrier between the sums and the second pass:
                # ops.conv_fwd_bn); None = this launch does not take that form, the two launches below run instead (same bits)
                if last and dual_sc:
                    fused = ops.conv_fwd_bn(a, c.lp_weight(), b, mom(b), relu=True, sc=(rawsc, bsc, stc, mom(bsc)), dilation=c.dilation[0])
                else:
                    fused = ops.conv_fwd_bn(a, c.lp_weight(), b, mom(b), residual=res if last else None, relu=True, dilation=c.dilation[0])
                if fused is not None:
                    if len(fused) == 4:
                        raw, a, s_i, ssc = fused
                    else:
                        raw, a, s_i = fused
                    saved += [raw, a, s_i]
                    continue
            if i == 0 and first is not None:
                raw, st = first
            else:
                raw, st = ops.conv_fwd(a, c.lp_weight(), c.stride[0], stats_shift=b.running_mean, want_stats=True,
                                       stats_buf=c._stats_buf, groups=G, dilation=c.dilation[0])
                c._stats_buf = st.partials
            if i == n - 1 and dual_sc and st.acc is not None:
                a, s_i, ssc = ops.bn_train_forward_dual(raw, b, st, mom(b), rawsc, bsc, stc, mom(bsc))
            else:
                if i == n - 1 and dual_sc:          # (the last convolution's moments did not go to accumulators after all)
                    res, ssc = _bn_fwd_g(rawsc, bsc, None, False, stc, G, mom(bsc))
                a, s_i = _bn_fwd_g(raw, b, res if i == n - 1 else None, True, st, G, mom(b))
            saved += [raw, a, s_i]
        ctx.blk, ctx.want_pgrad, ctx.n, ctx.G = blk, want_pgrad, n, G
        ctx.branch = _Flags.bn_branch      # dual-BN: the backward reads the same parameter set from the modules
        # cross-block fusion: when x is the output of another _BlockFn, this block's input-gradient dgrad also takes the
        # reduction sums of THAT block's last BN backward (its ReLU mask is x > 0) — see backward
        prev = getattr(x, "_afan_bn2", None) if _Flags.block_fusion else None
        ctx.prev_bn = prev if (prev is not None and prev[2] == G) else None
        if prev is not None:
            prev[6]["consumers"] += 1      # (block nodes reading the producer's output: the in-launch hand-off below is for exactly one)
        ctx.handoff = {"consumers": 0, "done": 0}
        ctx.save_for_backward(x, rawsc, ssc, *saved)
        # (raw_n, stats_n, G, bn_n, want_pgrad, the projection shortcut's (raw, stats, bn) or None): what a consumer block's first
        # input-gradient launch needs to run this block's last-BatchNorm backward(s) inside itself
        # [6]: the hand-off's bookkeeping, shared with this node's backward (see _handoff_check)
        a._afan_bn2 = (saved[-3], saved[-1], G, chain[-1][1], want_pgrad,
                       (rawsc, ssc, blk.shortcut[1]) if (blk._sc_kind == "conv" and ssc is not None) else None, ctx.handoff)
        return a

    @staticmethod
    def _handoff_check(ctx, gout):
        """The consumer block's first input-gradient launch may have run THIS block's last-BatchNorm backward inside itself and
        handed the results over as attributes of the gradient tensor (`_afan_bn_done`: the masked gradient IS the tensor, d_raw rides
        on it, the BatchNorm's parameter gradients are already accumulated).  That is only right if autograd delivers that very
        tensor, unmodified: a second consumer of the block's output makes the engine add another contribution — in place (the
        attribute survives, the data does not match it) or into a new tensor (the attribute is lost and a recomputation would count
        the parameter gradients twice).  Both are refused loudly here; a consumer only takes the form when it is the output's one
        block consumer (`consumers == 1`)."""
        st = ctx.handoff
        done = getattr(gout, "_afan_bn_done", None)
        mark = getattr(gout, "_afan_handoff", None)
        was_done, st["done"] = st["done"], 0
        if done is not None:
            if mark != (gout.data_ptr(), gout._version):
                raise ops.AfanLibraryError("block backward: the gradient tensor carrying an in-launch BatchNorm backward's results was "
                                           "modified on its way (the block's output has another consumer, a hook or retain_grad): "
                                           "run with resnet_s._Flags.block_fusion = False or AFAN_GRID_BN=0")
        elif was_done:
            raise ops.AfanLibraryError("block backward: a consumer block ran this block's last-BatchNorm backward inside its input-"
                                       "gradient launch, but autograd delivered a different gradient tensor (the block's output has "
                                       "another consumer): run with resnet_s._Flags.block_fusion = False or AFAN_GRID_BN=0")
        pre = getattr(gout, "_afan_bn_sums", None)
        if pre is not None and getattr(gout, "_afan_sums_mark", None) != (gout.data_ptr(), gout._version):
            pre = None                                 # (sums of a tensor that has changed since: reduce again)
        return pre, done

    @staticmethod
    def backward(ctx, gout):
        blk = ctx.blk
        bns = [b for _, b in blk._chain()] + ([blk.shortcut[1]] if blk._sc_kind == "conv" else [])
        if all(b._branch == ctx.branch for b in bns):
            return _BlockFn._backward(ctx, gout)
        was = [b._branch for b in bns]
        for b in bns:
            b.select(ctx.branch)
        try:
            return _BlockFn._backward(ctx, gout)
        finally:
            for b, w in zip(bns, was):
                b.select(w)

    @staticmethod
    def _backward(ctx, gout):
        if not (_WgradBatch.ON and ctx.want_pgrad and _WgradBatch.pending is None):
            return _BlockFn._backward_impl(ctx, gout)
        _WgradBatch.pending = []
        try:
            out = _BlockFn._backward_impl(ctx, gout)
        except BaseException:
            _WgradBatch.pending = None
            raise
        _WgradBatch.flush()
        return out

    @staticmethod
    def _backward_impl(ctx, gout):
        x, rawsc, ssc, *saved = ctx.saved_tensors
        blk, pg, n, G = ctx.blk, ctx.want_pgrad, ctx.n, ctx.G
        chain = blk._chain()
        raws, acts, stats = saved[0::3], saved[1::3], saved[2::3]
        need_dx = ctx.needs_input_grad[0]
        g = lambda p: p.grad if pg else None
        out = acts[-1]
        # pre: this block's last-BatchNorm backward SUMS, taken by the consumer block's dgrad epilogue (same tensor object);
        # done: ... or that whole backward, done inside that launch:
        pre, done = _BlockFn._handoff_check(ctx, gout)
        # last BN (+residual, ReLU mask from `out`): gradient to its conv output and to the shortcut branch
        bl = chain[-1][1]
        if done is not None:
            d_raw, dres = done, gout                   # the tensor handed back IS the shortcut's share; the other result rides on it
        else:
            gout = _like_layout(gout, out)
            d_raw, dres = _bn_bwd_g(gout, raws[-1], out, stats[-1], bl, True, True, pg, pre, G)
        # the first convolution's and the projection's output gradients side by side in one buffer: their input gradients
        # are ONE launch (afan_conv_dgrad_sc_nhwc_bf16) when the arena holds the block's combined [Ci][10][Co] operand
        c1 = chain[0][0]
        wt10 = getattr(c1, "_arena_wt10", None) if (_BlockFn.MULTI_SC and blk._sc_kind == "conv" and G == 1 and need_dx
                                                     and n >= 2) else None
        pair = None
        # ... and the projection BatchNorm's backward too: its result sits in the second half of a pair buffer the consumer made
        sc_done = getattr(gout, "_afan_sc_done", None) if done is not None else None
        r0 = raws[0]
        if wt10 is not None:
            pair = sc_done if sc_done is not None else torch.empty((2 * r0.shape[0],) + tuple(r0.shape[1:]), dtype=r0.dtype,
                                                                   device=r0.device, memory_format=torch.channels_last)
        for i in range(n - 1, 0, -1):
            c, bp = chain[i][0], chain[i - 1][1]
            # conv_i: dgrad carries bn_{i-1}'s backward reduction in its epilogue; wgrad straight into the arena
            fused = None
            if G == 1 and ops.GRID_BN and c.kernel_size[0] in (1, 3) and c.stride[0] == 1:
                # ... or bn_{i-1}'s whole backward (sums -> grid barrier -> the gradient entering its input): ONE launch
                fused = ops.conv_dgrad_bn(d_raw, c.lp_weight_t(), acts[i - 1].shape[2:], raws[i - 1], stats[i - 1], True,
                                          dweight=g(bp.weight), dbias=g(bp.bias), accumulate=pg,
                                          dx_out=pair[:r0.shape[0]] if (pair is not None and i == 1) else None, dilation=c.dilation[0])
            if fused is None:
                d_a, part = ops.conv_dgrad(d_raw, c.lp_weight_t(), acts[i - 1].shape[2:], c.stride[0],
                                           bn_bwd=(raws[i - 1], stats[i - 1], True), partials_buf=c._bwd_buf, groups=G,
                                           dilation=c.dilation[0])
                c._bwd_buf = part.partials
            if pg:
                _wgrad_accumulate(acts[i - 1], d_raw, c)
            if fused is not None:
                d_raw = fused[0]
                continue
            d_raw, _ = _bn_bwd_g(d_a, raws[i - 1], None, stats[i - 1], bp, True, False, pg, part, G,
                                 dx_out=pair[:r0.shape[0]] if (pair is not None and i == 1) else None)
        if pg:
            _wgrad_accumulate(x, d_raw, c1)
        dx = None
        prev = ctx.prev_bn
        fuse = dict(bn_bwd=(prev[0], prev[1], True), bn_y=x, groups=G) if (prev is not None and need_dx) else {}
        fuse["dilation"] = c1.dilation[0]
        if blk._sc_kind == "conv":
            csc, bsc = blk.shortcut[0], blk.shortcut[1]
            if pg or need_dx:
                if sc_done is not None:
                    d_rawsc = sc_done[r0.shape[0]:]
                else:
                    d_rawsc, _ = _bn_bwd_g(dres, rawsc, None, ssc, bsc, False, False, pg, None, G,
                                           dx_out=pair[r0.shape[0]:] if pair is not None else None)
                if pg:
                    _wgrad_accumulate(x, d_rawsc, csc)
                if need_dx and pair is not None:
                    if (prev is not None and G == 1 and ops.GRID_BN and len(prev) >= 5 and c1.stride[0] == 2
                            and prev[6]["consumers"] == 1 and getattr(prev[3], "_branch", "main") == ctx.branch):
                        # the producing block's last-BatchNorm backward inside this launch too (the stride-2 pair form: one set of
                        # sums over its four output-parity classes); that block's node finds its results on the tensor handed back
                        pbn, ppg = prev[3], prev[4]
                        fused = ops.conv_dgrad_bn(d_raw, None, x.shape[2:], prev[0], prev[1], True, bn_y=x, want_dres=True,
                                                  dweight=pbn.weight.grad if ppg else None, dbias=pbn.bias.grad if ppg else None,
                                                  accumulate=ppg, pair=(d_rawsc, wt10))
                        if fused is not None:
                            dx = fused[1]
                            dx._afan_bn_done = fused[0]
                            dx._afan_handoff = (dx.data_ptr(), dx._version)
                            prev[6]["done"] = 1
                            return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)
                    dx = ops.conv_dgrad(d_raw, c1.lp_weight_t(), x.shape[2:], c1.stride[0], sc=(d_rawsc, wt10), **fuse)
                elif need_dx:
                    dx_sc = ops.conv_dgrad(d_rawsc, csc.lp_weight_t(), x.shape[2:], csc.stride[0])
                    dx = ops.conv_dgrad(d_raw, c1.lp_weight_t(), x.shape[2:], c1.stride[0], addend=dx_sc, **fuse)
        elif blk._sc_kind == "pad":
            if need_dx:
                # gradient of the subsample + channel padding: the middle channels of dres land on the even pixels
                pad = blk.shortcut.pad
                dx_sc = torch.zeros_like(x)
                dx_sc[:, :, ::2, ::2] = dres[:, pad:pad + x.shape[1]]
                dx = ops.conv_dgrad(d_raw, c1.lp_weight_t(), x.shape[2:], c1.stride[0], addend=dx_sc, **fuse)
        elif need_dx:
            fused = pair_p = None
            if (prev is not None and G == 1 and ops.GRID_BN and len(prev) >= 5 and c1.kernel_size[0] in (1, 3) and c1.stride[0] == 1
                    and prev[6]["consumers"] == 1 and getattr(prev[3], "_branch", "main") == ctx.branch):
                # the gradient leaving this block is only ever read by the producing block's last-BatchNorm backward: that
                # backward runs inside this launch and the producer's node finds its two results on the tensor handed back
                pbn, ppg = prev[3], prev[4]
                kw = dict(bn_y=x, addend=dres, want_dres=True, dweight=pbn.weight.grad if ppg else None,
                          dbias=pbn.bias.grad if ppg else None, accumulate=ppg, dilation=c1.dilation[0])
                psc = prev[5] if (len(prev) > 5 and ops.GRID_BN_SC and _BlockFn.MULTI_SC) else None
                if psc is not None and getattr(psc[2], "_branch", "main") == ctx.branch:
                    # the producing block has a projection shortcut: its BatchNorm's backward here as well, the result in the second
                    # half of the buffer that block's fused stride-2 input gradient reads (first half: its own first convolution's)
                    pair_p = torch.empty((2 * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device,
                                         memory_format=torch.channels_last)
                    fused = ops.conv_dgrad_bn(d_raw, c1.lp_weight_t(), x.shape[2:], prev[0], prev[1], True,
                                              sc=(psc[0], psc[1], pair_p[x.shape[0]:], psc[2].weight.grad if ppg else None,
                                                  psc[2].bias.grad if ppg else None), **kw)
                    if fused is None:
                        pair_p = None
                if fused is None:
                    fused = ops.conv_dgrad_bn(d_raw, c1.lp_weight_t(), x.shape[2:], prev[0], prev[1], True, **kw)
            if fused is not None:
                dx = fused[1]
                dx._afan_bn_done = fused[0]            # (not the pair: a tuple holding dx on dx would be a reference cycle)
                dx._afan_handoff = (dx.data_ptr(), dx._version)
                prev[6]["done"] = 1
                if pair_p is not None:
                    dx._afan_sc_done = pair_p
                return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)
            dx = ops.conv_dgrad(d_raw, c1.lp_weight_t(), x.shape[2:], c1.stride[0], addend=dres, **fuse)
        if "bn_bwd" in fuse and dx is not None:
            dx, sums = dx
            dx._afan_bn_sums = sums
            dx._afan_sums_mark = (dx.data_ptr(), dx._version)
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


def _block_fast_path_ok(blk, x):
    """bf16 channels-last training step with every convolution of the block on the library's MFMA kernels and every
    parameter's gradient buffer owned by the arena (or parameter gradients switched off, as inside PGD)."""
    chain = blk._chain()
    c1 = chain[0][0]
    if c1.compute_dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or x.dim() != 4:
        return False
    if not x.is_contiguous(memory_format=torch.channels_last) or not x.is_cuda:
        return False
    convs, mods = [c for c, _ in chain], [b for _, b in chain]
    if blk._sc_kind == "conv":
        convs.append(blk.shortcut[0])
        mods.append(blk.shortcut[1])
    elif blk._sc_kind == "pad":
        if x.shape[2] % 2 or x.shape[3] % 2:
            return False
    elif blk._sc_kind != "identity":
        return False
    if not all(m.training and m.track_running_stats for m in mods):
        return False
    for c in convs:
        if not _own_conv_ok_shape(c.lp_weight(), c.stride, c.padding, c.dilation):
            return False
        cm = max(c.in_channels, c.out_channels)         # 32-bit byte offsets in the tuned kernels (see _own_conv_ok)
        if x.shape[0] * x.shape[2] * x.shape[3] * cm * 2 + 2 * c.dilation[0] * (x.shape[3] + 1) * cm > 0x7FFFFFFF:
            return False
    if _Flags.param_grads and torch.is_grad_enabled():
        ps = [c.weight for c in convs] + [m.weight for m in mods] + [m.bias for m in mods]
        if not all(_accumulates_in_place(p) for p in ps):
            return False
    return True


def _block_params(blk):
    ps = []
    for c, b in blk._chain():
        ps += [c.weight, b.weight, b.bias]
    if blk._sc_kind == "conv":
        ps += [blk.shortcut[0].weight, blk.shortcut[1].weight, blk.shortcut[1].bias]
    return ps


# ---------------------------------------------------------------------------------------------- layers
class NormalizeByChannelMeanStd(nn.Module):
    """(x - mean[c]) / std[c] with `mean`/`std` buffers (advertorch's module, used at resnet_s.py:87).
    Input images are constants of the graph (no gradient), output is in the model's compute dtype."""

    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean, dtype=torch.float32))
        self.register_buffer("std", torch.tensor(std, dtype=torch.float32))
        self.out_dtype = torch.float32
        self.channels_last = False

    def forward(self, x):
        if x.requires_grad:        # image-space PGD (Detection / Segmentation `adv_input`): d/dx = g / std[c]
            return _NormalizeFn.apply(x, self.mean, self.std, self.out_dtype, self.channels_last)
        return ops.normalize_nchw(x.contiguous().float(), self.mean, self.std, self.out_dtype, self.channels_last)


class _NormalizeFn(torch.autograd.Function):
    """(x - mean[c]) / std[c] with the gradient to the image: g / std[c] in the image's own fp32 NCHW form
    (afan_affine_relu_bwd with alpha = 1 / std)."""

    @staticmethod
    def forward(ctx, x, mean, std, out_dtype, channels_last):
        ctx.inv_std = (1.0 / std.float()).contiguous()
        return ops.normalize_nchw(x.detach().contiguous().float(), mean, std, out_dtype, channels_last)

    @staticmethod
    def backward(ctx, g):
        g = g.float().contiguous()
        dx, _ = ops.affine_relu_backward(g, None, ctx.inv_std, False, want_dx=True)
        return dx, None, None, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters (fp32 master `weight`); runs in `compute_dtype` from a cached low-precision copy."""

    compute_dtype = torch.float32
    _lp = None
    _lp_version = -1

    def lp_weight(self):
        if self.compute_dtype == torch.float32:
            return self.weight
        shadow = getattr(self, "_arena_shadow", None)
        if shadow is not None:  # kept fresh by afan_sgd_step (ParamArena)
            return shadow
        if self._lp is None or self._lp_version != self.weight._version or self._lp.device != self.weight.device:
            # KRSC (channels-last) memory: what the tuned kernels read — also for weights outside a parameter arena
            # (frozen layers: Detection/backbone/resnet101.py:30-32)
            self._lp = ops.cast_bf16(self.weight.detach().contiguous(memory_format=torch.channels_last))
            self._lp_version = self.weight._version
        return self._lp

    def lp_weight_t(self):
        """[Ci, Co, k, k] channels-last (CRSK memory) copy of the low-precision weight for the dgrad kernel; rebuilt
        when the weights change (once per SGD step: the K+2 input-gradient passes of an iteration share it)."""
        wt = getattr(self, "_arena_wt", None)
        if wt is not None:   # kept fresh by ParamArena.refresh_transposed (one launch per SGD step)
            return wt
        w = self.lp_weight()
        key = (w.data_ptr(), self.weight._version, _Flags.weight_epoch)
        if self._wt is None or self._wt_key != key:
            self._wt = w.detach().permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
            self._wt_key = key
        return self._wt

    _wt = None
    _wt_key = None

    def forward(self, x):
        x = _to_compute(x, self.compute_dtype)
        return _ConvFn.apply(x, self.weight, self.lp_weight().detach(), self.lp_weight_t, self.stride, self.padding,
                             _Flags.param_grads, None, self.dilation)

    _stats_buf = None
    _bwd_buf = None

    def forward_with_stats(self, x, bn):
        """Convolution whose epilogue also sums the moments the train-mode BatchNorm `bn` is about to need.
        Returns (y, ConvStats | None): None whenever the fused path does not apply (eval mode, fp32, MIOpen conv)."""
        if not (bn.training and self.compute_dtype == torch.bfloat16):
            return self.forward(x), None
        x = _to_compute(x, self.compute_dtype)
        req = [bn.running_mean if bn.track_running_stats else None, self._stats_buf]
        y = _ConvFn.apply(x, self.weight, self.lp_weight().detach(), self.lp_weight_t, self.stride, self.padding,
                          _Flags.param_grads, req, self.dilation)
        st = req[2] if len(req) > 2 else None
        if st is not None:
            self._stats_buf = st.partials
        return y, st


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d parameters/buffers; fused HIP execution."""
    _branch = "main"       # which set currently sits in weight / bias / running_* (see enable_dual_bn)

    def enable_dual(self):
        if getattr(self, "adv", None) is None:
            aux = nn.BatchNorm2d(self.num_features, eps=self.eps, momentum=self.momentum).to(self.weight.device)
            aux.load_state_dict({k: v.clone() for k, v in nn.BatchNorm2d.state_dict(self).items() if not k.startswith("adv.")})
            self.adv = aux         # a holder: never called, its tensors are exchanged with this module's by select()

    def select(self, branch):
        """Exchange the module's parameter / buffer ENTRIES with the auxiliary set's (no copy): afterwards `self.weight`
        etc. are the tensors of `branch`."""
        if getattr(self, "adv", None) is None or branch == self._branch:
            return
        a = self.adv
        for k in ("weight", "bias"):
            self._parameters[k], a._parameters[k] = a._parameters[k], self._parameters[k]
        for k in ("running_mean", "running_var", "num_batches_tracked"):
            self._buffers[k], a._buffers[k] = a._buffers[k], self._buffers[k]
        self._branch = branch

    def fused(self, x, residual=None, relu=False, conv_stats=None):
        if self.training:
            mom = self.momentum if self.momentum is not None else 0.1
            G = _Flags.bn_groups
            if G != 1 and (x.shape[0] % G or not x.is_cuda):
                raise ValueError("grouped BatchNorm statistics: the batch must split into equal half-batches on the GPU")
            return _BNTrainFn.apply(x, self.weight, self.bias, residual, relu, self.eps, mom, self.running_mean,
                                    self.running_var, self.num_batches_tracked, _Flags.param_grads, conv_stats, G)
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad and _Flags.param_grads):
            if x.requires_grad:
                raise NotImplementedError("eval-mode BatchNorm backward is not on the A-FAN path (the reference "
                                          "runs PGD and the joint step in train mode, main_perturb.py:159)")
        with torch.no_grad():
            invstd = torch.rsqrt(self.running_var + self.eps)
            x = _dense(x)
            return ops.bn_apply(x, self.running_mean, invstd, self.weight, self.bias,
                                None if residual is None else _like_layout(residual, x), relu)

    def forward(self, x):
        return self.fused(x)


class _PadShortcut(nn.Module):
    """Option-A identity (resnet_s.py:64-65): stride-2 subsample, zero-pad planes//4 channels per side."""

    def __init__(self, planes):
        super().__init__()
        self.pad = planes // 4

    def forward(self, x):
        return F.pad(x[:, :, ::2, ::2], (0, 0, 0, 0, self.pad, self.pad), "constant", 0)


class BasicBlock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN-(+shortcut)-ReLU (resnet_s.py:48-77); BN+ReLU and BN+add+ReLU fused."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, option="A"):
        super().__init__()
        self.conv1 = Conv2d(in_planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        self._sc_kind = "identity"
        if stride != 1 or in_planes != planes:
            if option == "A":
                self.shortcut = _PadShortcut(planes)
                self._sc_kind = "pad"
            else:
                self.shortcut = nn.Sequential(
                    Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False),
                    BatchNorm2d(self.expansion * planes))
                self._sc_kind = "conv"

    def _chain(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]

    def forward(self, x):
        x = _to_compute(x, self.conv1.compute_dtype)
        if _Flags.block_fusion and _block_fast_path_ok(self, x):
            return _BlockFn.apply(x, self, _Flags.param_grads, *_block_params(self))
        out, st = self.conv1.forward_with_stats(x, self.bn1)
        out = self.bn1.fused(out, None, True, st)
        out, st = self.conv2.forward_with_stats(out, self.bn2)
        sc = self.shortcut
        if isinstance(sc, nn.Sequential) and len(sc) == 2:     # option B: 1x1 conv + BN
            r, st_sc = sc[0].forward_with_stats(x, sc[1])
            res = sc[1].fused(r, None, False, st_sc)
        else:
            res = sc(x)
        return self.bn2.fused(out, res, True, st)


class Bottleneck(nn.Module):
    """ResNet-v1.5 bottleneck (1x1 -> 3x3 (stride) -> 1x1 x4; projection shortcut when the shape changes) for the
    build-defined ImageNet-shape ResNet-50 (BASELINE config 3).  Every conv -> BN pair uses the fused moments epilogue;
    BN+ReLU and BN+add+ReLU are single fused launches."""
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, option="B"):
        super().__init__()
        out_planes = planes * 4
        self.conv1 = Conv2d(in_planes, planes, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, out_planes, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn3 = BatchNorm2d(out_planes)
        self.shortcut = nn.Sequential()
        self._sc_kind = "identity"
        if stride != 1 or in_planes != out_planes:
            self.shortcut = nn.Sequential(Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False),
                                          BatchNorm2d(out_planes))
            self._sc_kind = "conv"

    def _chain(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]

    def forward(self, x):
        x = _to_compute(x, self.conv1.compute_dtype)
        if _Flags.block_fusion and _block_fast_path_ok(self, x):
            return _BlockFn.apply(x, self, _Flags.param_grads, *_block_params(self))
        out, st = self.conv1.forward_with_stats(x, self.bn1)
        out = self.bn1.fused(out, None, True, st)
        out, st = self.conv2.forward_with_stats(out, self.bn2)
        out = self.bn2.fused(out, None, True, st)
        out, st = self.conv3.forward_with_stats(out, self.bn3)
        if len(self.shortcut) == 2:
            r, st_sc = self.shortcut[0].forward_with_stats(x, self.shortcut[1])
            res = self.shortcut[1].fused(r, None, False, st_sc)
        else:
            res = x
        return self.bn3.fused(out, res, True, st)


class _HeadFn(torch.autograd.Function):
    """AdaptiveAvgPool2d((1,1)) -> Flatten -> Linear (resnet_s.py:108-110) as one forward and two backward launches
    (afan_head_*): the classifier head runs at the end of every tail pass, K + 2 times per iteration."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_pgrad):
        logits, pooled = ops.head_forward(x, weight, bias)
        ctx.save_for_backward(x, weight, pooled)
        ctx.bias, ctx.pg = bias, want_pgrad
        return logits

    @staticmethod
    def backward(ctx, g):
        x, weight, pooled = ctx.saved_tensors
        bias = ctx.bias
        want_p = ctx.pg and ctx.needs_input_grad[1]
        dw = db = None
        direct = False
        if want_p:
            direct = _accumulates_in_place(weight) and (bias is None or _accumulates_in_place(bias))
            if direct:
                dw, db = weight.grad, (bias.grad if bias is not None else None)
            else:
                dw = torch.empty_like(weight)
                db = torch.empty_like(bias) if bias is not None else None
        dx = ops.head_backward(g.float().contiguous(), weight, pooled, x, ctx.needs_input_grad[0], dw, db, accumulate=direct)
        if direct:
            dw = db = None
        return dx, dw, db, None


class _CEFn(torch.autograd.Function):
    """nn.CrossEntropyLoss()(logits, target) with its defaults as one forward launch that also leaves d(loss)/d(logits)
    (afan_cross_entropy); backward = that gradient times the incoming scalar."""

    @staticmethod
    def forward(ctx, logits, target):
        loss, dlogits = ops.cross_entropy(logits, target)
        ctx.save_for_backward(dlogits)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        if g.data_ptr() == ops.one(g.device).data_ptr():      # the root gradient ops.one(): d(loss)/d(logits) as it is
            return dlogits, None
        return dlogits * g, None


_FUSED_CE = os.environ.get("AFAN_FUSED_CE", "1") != "0"


def fused_criterion(criterion, model):
    """`criterion` as the step applies it to the classifier's logits.  A plain nn.CrossEntropyLoss() on the bf16 product
    path becomes the one-launch fused form when its arguments allow; anything else (a caller's own loss_fn, class weights,
    label smoothing, fp32 parity mode — which stays on torch's own loss kernels) is returned unchanged."""
    if not (_FUSED_CE and type(criterion) is nn.CrossEntropyLoss and criterion.weight is None and criterion.ignore_index == -100
            and criterion.reduction == "mean" and getattr(criterion, "label_smoothing", 0.0) == 0.0
            and getattr(model, "compute_dtype", torch.float32) == torch.bfloat16):
        return criterion

    def ce(out, y):
        if (out.is_cuda and out.dtype == torch.float32 and out.dim() == 2 and out.is_contiguous() and y.dtype == torch.int64
                and y.dim() == 1 and 0 < out.numel() <= ops.CE_MAX_ELEMS):
            return _CEFn.apply(out, y)
        return criterion(out, y)
    return ce


_FUSED_HEAD = os.environ.get("AFAN_FUSED_HEAD", "1") != "0"   # 0: pool / flatten / linear as separate torch ops (A/B)


def _head_ok(x, lin):
    return (_FUSED_HEAD and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and lin.weight.dtype == torch.float32
            and (x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous() or x.shape[2] * x.shape[3] == 1)
            and lin.out_features <= 16)


def _head_in(x):
    """The classifier-head kernels walk channels-last rows: an NCHW map (fp32 parity mode in the reference's layout) is
    re-laid — data movement of the last, smallest feature map only."""
    if x.is_cuda and x.dim() == 4 and x.is_contiguous() and x.shape[2] * x.shape[3] > 1 and x.shape[1] > 1:
        return x.contiguous(memory_format=torch.channels_last)
    return x


class _LinearFn(torch.autograd.Function):
    """nn.Linear on the GPU as a 1x1 convolution of an [N, Cin, 1, 1] map on the general fp32-arithmetic kernels (no
    vendor GEMM): y = x w^T + b."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_pgrad):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.pg, ctx.bias = want_pgrad, bias
        n, ci = x.shape
        co = weight.shape[0]
        if ci >= 1024 and n <= 1024 and x.dtype == torch.float32 and weight.dtype == torch.float32:
            # few rows, long reduction (the detection heads: 128 ROIs x 2048 -> 21 / 84): as a 1x1 convolution this is ONE
            # workgroup walking the whole K chain (290-390 us); read as a weight-gradient problem — y^T[co, n] = sum_k
            # w[co, k] x[n, k], k in the role of the pixels — the same kernels split the reduction over workgroups (~25 us)
            yt = ops.conv_general_wgrad(x.view(1, n, ci, 1), weight.detach().contiguous().view(1, co, ci, 1), 1)
            y = yt.view(co, n).t()
            return (y + bias.detach()) if bias is not None else y.contiguous()
        y = ops.conv_general_fwd(x.view(n, ci, 1, 1), weight.detach().view(co, ci, 1, 1),
                                 None if bias is None else bias.detach())
        return y.view(n, -1)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        n, ci = x.shape
        co = weight.shape[0]
        g = g.contiguous().float()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if co >= 512 and n <= 1024 and weight.dtype == torch.float32 and weight.is_contiguous():
                # few rows, long reduction over the OUTPUT features (ResNet-50's 2048 -> 1000 classifier at batch 64: as a 1x1
                # input gradient 16 workgroups walk 1000 channels each, 238 us): the forward's trick again — dx[n, ci] =
                # sum_k g[n, k] w[k, ci] is a weight-gradient problem with k in the role of the pixels, channels-last:
                # "x" = w read as [1, ci, co, 1] (memory [k][ci]), "dy" = g^T as [1, n, co, 1] (memory [k][n])
                w_cl = weight.detach().view(1, co, 1, ci).permute(0, 3, 1, 2)
                g_cl = g.t().contiguous().view(1, co, 1, n).permute(0, 3, 1, 2)
                gx = ops.conv_general_wgrad(w_cl, g_cl, 1).reshape(n, ci)
            else:
                gx = ops.conv_general_dgrad(g.view(n, co, 1, 1), weight.detach().view(co, ci, 1, 1), (1, 1)).view(n, ci)
        if ctx.pg and ctx.needs_input_grad[1]:
            direct = _accumulates_in_place(weight)
            gwt = ops.conv_general_wgrad(x.view(n, ci, 1, 1), g.view(n, co, 1, 1), 1,
                                         grad=weight.grad.view(co, ci, 1, 1) if direct else None, accumulate=direct)
            gw = None if direct else gwt.view(co, ci)
            bias = ctx.bias
            if bias is not None:
                if _accumulates_in_place(bias):
                    bias.grad.add_(g.sum(dim=0))
                else:
                    gb = g.sum(dim=0)
        return gx, gw, gb, None


def _linear(x, lin):
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and lin.weight.dtype == torch.float32:
        return _LinearFn.apply(x, lin.weight, lin.bias, _Flags.param_grads)
    return lin(x)


class _HeadPool(nn.AdaptiveAvgPool2d):
    """Global average pool; accumulates and returns fp32 (the classifier head runs in fp32)."""

    def forward(self, x):
        return x.float().mean(dim=(2, 3), keepdim=True)


class ResNet(nn.Module):
    """Flat-Sequential ResNet with the slice protocol.  `widths`/`option` generalise the reference class
    (16-32-64, option A) to the build-defined ResNet-18-CIFAR (64-128-256-512, option B)."""

    def __init__(self, block, num_blocks, num_classes=10, init_weight=1, widths=(16, 32, 64), option="A"):
        super().__init__()
        self.all_layers = 9
        layers = [NormalizeByChannelMeanStd(mean=[0.4914, 0.4822, 0.4465], std=[0.2470, 0.2435, 0.2616]),
                  Conv2d(3, widths[0], kernel_size=3, stride=1, padding=1, bias=False),
                  BatchNorm2d(widths[0]), nn.ReLU()]
        in_planes = widths[0]
        for stage, (planes, nb) in enumerate(zip(widths, num_blocks)):
            for b in range(nb):
                layers.append(block(in_planes, planes, 2 if (stage > 0 and b == 0) else 1, option))
                in_planes = planes * block.expansion
        layers += [_HeadPool((1, 1)), nn.Flatten(), nn.Linear(in_planes, num_classes)]
        self.sequential_model = nn.Sequential(*layers)
        self.w = nn.Parameter(torch.full((self.all_layers,), float(init_weight)), requires_grad=True)
        for m in self.modules():  # same leaf order as resnet_s.py:116 `self.apply(_weights_init)`
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)
        self.compute_dtype = torch.float32
        self.channels_last = False

    def set_channels_last(self, on=True):
        """Keep activations (and, with a ParamArena, conv weights) channels-last: the layout the MFMA convolutions
        consume without transposes.  Logical shapes stay NCHW, so the slice protocol and PGD are unchanged."""
        self.channels_last = bool(on)
        for m in self.modules():
            if isinstance(m, NormalizeByChannelMeanStd):
                m.channels_last = self.channels_last
        return self

    @property
    def layer_number(self):
        return len(self.sequential_model)

    def set_compute_dtype(self, dtype):
        """fp32 (parity mode) or bf16 (MFMA convs, bf16 activations, fp32 statistics / master weights)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, Conv2d):
                m.compute_dtype = dtype
            elif isinstance(m, NormalizeByChannelMeanStd):
                m.out_dtype = dtype
        return self

    def forward(self, x, end_point=34, start_point=0):
        layers = list(self.sequential_model[start_point:end_point])
        if self.channels_last and x.dim() == 4 and x.is_floating_point():
            x = x if x.is_contiguous(memory_format=torch.channels_last) else \
                x.contiguous(memory_format=torch.channels_last)   # no-op inside the step: tensors already are
        i, n = 0, len(layers)
        while i < n:
            L = layers[i]
            if isinstance(L, BatchNorm2d):
                x = _to_compute(x, self.compute_dtype)
                if i + 1 < n and isinstance(layers[i + 1], nn.ReLU):
                    x = L.fused(x, None, True)
                    i += 2
                    continue
                x = L.fused(x)
            elif (isinstance(L, _HeadPool) and i + 2 < n and isinstance(layers[i + 1], nn.Flatten)
                  and isinstance(layers[i + 2], nn.Linear) and _head_ok(_head_in(x), layers[i + 2])):
                lin = layers[i + 2]
                x = _HeadFn.apply(_head_in(x), lin.weight, lin.bias, _Flags.param_grads)
                i += 3
                continue
            elif isinstance(L, nn.Linear):
                x = _linear(x.float(), L)
            else:
                x = L(x)
            i += 1
        return x


class ResNet50(ResNet):
    """ImageNet-shape ResNet-50, same flat-Sequential slice protocol: 0 normalise, 1 conv7x7/2, 2 BN, 3 ReLU, 4 maxpool,
    5-7 layer1, 8-11 layer2, 12-17 layer3, 18-20 layer4, 21 avgpool, 22 flatten, 23 fc (perturb_idx 8 = after layer1:
    256 x 56 x 56 at 224^2)."""

    def __init__(self, num_classes=1000, blocks=(3, 4, 6, 3), init_weight=1):
        nn.Module.__init__(self)
        from .deeplab import MaxPool2d, StemConv      # the 7x7 / 2 stem (im2col + MFMA 1x1) and the 3x3 / 2 max-pool kernels
        self.all_layers = 9
        layers = [NormalizeByChannelMeanStd(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]),
                  StemConv(3, 64, kernel_size=7, stride=2, padding=3, bias=False), BatchNorm2d(64), nn.ReLU(),
                  MaxPool2d(3, 2, 1)]
        in_planes = 64
        for stage, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
            for b in range(nb):
                layers.append(Bottleneck(in_planes, planes, 2 if (stage > 0 and b == 0) else 1))
                in_planes = planes * 4
        layers += [_HeadPool((1, 1)), nn.Flatten(), nn.Linear(in_planes, num_classes)]
        self.sequential_model = nn.Sequential(*layers)
        self.w = nn.Parameter(torch.full((self.all_layers,), float(init_weight)), requires_grad=True)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)
        self.compute_dtype = torch.float32
        self.channels_last = False


def resnet50(init_weight_eta=1, num_classes=1000):
    return ResNet50(num_classes=num_classes, init_weight=init_weight_eta)


def resnet20(init_weight_eta=1):
    return ResNet(BasicBlock, [3, 3, 3], init_weight=init_weight_eta)


def resnet56(init_weight_eta=1):  # resnet_s.py:123-124
    return ResNet(BasicBlock, [9, 9, 9], init_weight=init_weight_eta)


def resnet18(init_weight_eta=1, num_classes=10):
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes=num_classes, init_weight=init_weight_eta,
                  widths=(64, 128, 256, 512), option="B")


# name -> (constructor, default --perturb_idx): end of stage 1 in each flat index map
ARCHS = {"resnet20s": (resnet20, 7), "resnet56s": (resnet56, 13), "resnet18": (resnet18, 6), "resnet50": (resnet50, 8)}
