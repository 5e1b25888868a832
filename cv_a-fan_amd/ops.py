"""Tensor-level wrappers over the C-ABI (include/afan_hip.h).  torch is plumbing here: it owns the
device memory and the stream; every arithmetic op below runs in libafan_hip.so.  No fallbacks: a CPU
tensor, a non-contiguous tensor or a missing library raises.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import AFAN_BF16, AFAN_F32, AFAN_NCHW, AFAN_NHWC, AfanLibraryError, check

_DT = {torch.float32: AFAN_F32, torch.bfloat16: AFAN_BF16}
_ws_cache = {}
_tls = threading.local()
# how many convolution passes ran on the library's kernels / went to the vendor library since import (tests and the
# entry points' logs use it to show which path a configuration really takes)
# ("vendor_conv" stays in the table as the invariant the tests assert: the package holds no vendor convolution any more;
# "conv_general" counts passes of the fp32-arithmetic general kernels, afan_conv_f32.hip)
CALLS = {"conv_fwd": 0, "conv_dgrad": 0, "conv_wgrad": 0, "conv_general": 0, "vendor_conv": 0, "conv_bn_fused": 0}


def _need(t, name, dtype=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if t.device.type != "cuda":
        raise AfanLibraryError(f"{name}: tensor is on '{t.device}'. The A-FAN kernels are MI355X-only; "
                               "there is no CPU path (the CPU restatement under oracle/ is test infrastructure).")
    if not _dense(t):
        raise ValueError(f"{name}: tensor must be dense (contiguous NCHW or channels_last)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def layout_of(t):
    """AFAN_NHWC for a channels_last 4-D tensor, else AFAN_NCHW (degenerate shapes count as NCHW)."""
    if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
        return AFAN_NHWC
    return AFAN_NCHW


def _strides_eq(a, b):
    """Same memory order: strides of size-1 dimensions carry no information (a [N,C,1,1] tensor is both layouts)."""
    return all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


def _same_layout(ref, *others):
    for o in others:
        if o is not None and ref.numel() > 0 and (o.shape != ref.shape or not _strides_eq(o, ref)):
            raise ValueError("operands must share shape and memory layout (strides): "
                             f"{tuple(ref.shape)}/{ref.stride()} vs {tuple(o.shape)}/{o.stride()}")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = torch._C._cuda_getCurrentRawStream      # (torch.cuda.current_stream() builds a Stream object: 15 us per launch)


def _stream(t):
    return C.c_void_p(_raw_stream(t.device.index))


def _workspace(ref, nfloats, tag):
    """Per (device, stream, tag) fp32 scratch, grown on demand, never shrunk (graph-capture safe once warm)."""
    key = (ref.device.index, _raw_stream(ref.device.index), tag)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nfloats:
        ws = torch.empty(max(int(nfloats), 1024), dtype=torch.float32, device=ref.device)
        _ws_cache[key] = ws
    return ws


# ---- two small linear layers on one input (afan_linear.hip): the Faster-RCNN heads
def linear_pair_ok(x, w1, w2):
    """x [M, K] fp32 row-major on the GPU, weights [N1, K] / [N2, K] fp32 contiguous, K % 4 == 0, N1 + N2 <= 128."""
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous() and w1.dtype == torch.float32 and w2.dtype == torch.float32
            and w1.dim() == 2 and w2.dim() == 2 and w1.is_contiguous() and w2.is_contiguous() and w1.shape[1] == x.shape[1] == w2.shape[1]
            and x.shape[1] % 4 == 0 and x.shape[1] >= 4 and w1.shape[0] + w2.shape[0] <= 128 and x.data_ptr() % 16 == 0
            and w1.data_ptr() % 16 == 0 and w2.data_ptr() % 16 == 0)


def linear_pair_fwd(x, w1, b1, w2, b2):
    """(x w1^T + b1, x w2^T + b2), fp32, one launch (two when the reduction is split: few rows)."""
    lib = _lib.load()
    M, K = x.shape
    n1, n2 = w1.shape[0], w2.shape[0]
    y1 = torch.empty((M, n1), dtype=torch.float32, device=x.device)
    y2 = torch.empty((M, n2), dtype=torch.float32, device=x.device)
    nws = lib.afan_linear_pair_workspace_floats(0, M, n1, n2, K)
    if nws < 0:
        raise ValueError("linear_pair_fwd: shape outside the kernel's range")
    ws = _workspace(x, nws, "linear_pair") if nws else None
    check(lib.afan_linear_pair_fwd_f32(_ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(y1), _ptr(y2), M, n1, n2, K, _ptr(ws), _stream(x)),
          "afan_linear_pair_fwd_f32")
    return y1, y2


def linear_pair_dgrad(g1, g2, w1, w2):
    """g1 w1 + g2 w2 [M, K], fp32, one launch."""
    M, K = g1.shape[0], w1.shape[1]
    gx = torch.empty((M, K), dtype=torch.float32, device=g1.device)
    check(_lib.load().afan_linear_pair_dgrad_f32(_ptr(g1), _ptr(g2), _ptr(w1), _ptr(w2), _ptr(gx), M, w1.shape[0], w2.shape[0], K, _stream(g1)),
          "afan_linear_pair_dgrad_f32")
    return gx


def linear_pair_wgrad(g1, g2, x, gw1, gb1, gw2, gb2, accumulate):
    """gw1 (+)= g1^T x, gb1 (+)= g1's column sums, the same for layer 2 (biases optional), fp32, two launches."""
    lib = _lib.load()
    M, K = x.shape
    n1, n2 = g1.shape[1], g2.shape[1]
    ws = _workspace(x, lib.afan_linear_pair_workspace_floats(1, M, n1, n2, K), "linear_pair")
    check(lib.afan_linear_pair_wgrad_f32(_ptr(g1), _ptr(g2), _ptr(x), _ptr(gw1), _ptr(gb1), _ptr(gw2), _ptr(gb2), int(bool(accumulate)), M, n1, n2, K,
                                         _ptr(ws), _stream(x)), "afan_linear_pair_wgrad_f32")


# BatchNorm sums in f64 accumulators (no partial slabs, no finalize launches) where the channel count allows it;
# AFAN_BN_ACC=0 selects the partial-slab path everywhere (bitwise run-to-run reproducible, ~0.45 more launches per BN).
BN_ACC = os.environ.get("AFAN_BN_ACC", "1") != "0"
_ACC_DOUBLES = 1 << 20
_acc_arenas = {}


class _AccArena:
    __slots__ = ("buf", "off")

    def __init__(self, device):
        self.buf = torch.zeros(_ACC_DOUBLES, dtype=torch.float64, device=device)
        self.off = 0


def _acc_arena(device):
    key = (device.index, _raw_stream(device.index))
    a = _acc_arenas.get(key)
    if a is None:
        a = _acc_arenas[key] = _AccArena(device)
    return a


def acc_reset(device):
    """Zero the accumulator arena of the current stream and rewind it (once per training step; one memset)."""
    a = _acc_arena(torch.device(device))
    a.buf.zero_()
    a.off = 0


def acc_take(device, c, groups=1):
    """A zeroed accumulator block for one BatchNorm pass over c channels (see afan_bn_acc_doubles); groups = 2: two
    consecutive blocks, one per half-batch (stride = acc_block_doubles(c)).  Blocks are cut
    from a per-stream arena; when it runs out the arena is zeroed in stream order and reused — every block is consumed
    by the launch right after its producer, so nothing live is lost."""
    a = _acc_arena(device)
    n = acc_block_doubles(c) * int(groups)
    if n > _ACC_DOUBLES:
        raise ValueError("too many channels for the accumulator arena")
    if a.off + n > _ACC_DOUBLES:
        a.buf.zero_()
        a.off = 0
    blk = a.buf[a.off:a.off + n]
    a.off += n
    return blk


def acc_block_doubles(c):
    return (int(_lib.load().afan_bn_acc_doubles(int(c))) + 1) & ~1


def bn_acc_ok(x):
    """Accumulator path usable for BatchNorm over x: channels-last memory (or 1x1 spatial) and a channel count whose
    16-byte vectors tile a 256-thread block."""
    if not BN_ACC or x.dim() != 4 or x.dtype not in _DT:
        return False
    if layout_of(x) != AFAN_NHWC and not (x.shape[2] == 1 and x.shape[3] == 1):
        return False
    return bool(_lib.load().afan_bn_acc_supported(_DT[x.dtype], int(x.shape[1])))


def _nchw(t):
    if t.dim() < 2:
        raise ValueError("expected a tensor of shape [N, C, ...]")
    n, c = t.shape[0], t.shape[1]
    hw = 1
    for s in t.shape[2:]:
        hw *= s
    return n, c, hw


def cross_entropy(logits, target):
    """Mean cross-entropy (nn.CrossEntropyLoss defaults) and d(loss)/d(logits) in one launch: returns (loss [1], dlogits)."""
    lib = _lib.load()
    _need(logits, "logits", torch.float32)
    if logits.dim() != 2 or not logits.is_contiguous() or target.dtype != torch.int64 or target.dim() != 1 \
            or target.shape[0] != logits.shape[0] or not target.is_cuda:
        raise TypeError("cross_entropy: logits [N,K] fp32 contiguous and target [N] int64 on the GPU")
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    check(lib.afan_cross_entropy(_ptr(logits), _ptr(target.contiguous()), logits.shape[0], logits.shape[1], _ptr(loss),
                                 _ptr(dlogits), _stream(logits)), "afan_cross_entropy")
    return loss, dlogits


CE_MAX_ELEMS = 1 << 16
_ones = {}


def half(device):
    """A cached fp32 scalar 0.5: root gradient of each half of the joint loss (CE_adv + CE_clean) / 2."""
    t = _ones.get(("half", device.index))
    if t is None:
        t = _ones[("half", device.index)] = torch.full((), 0.5, dtype=torch.float32, device=device)
    return t


def one(device):
    """A cached fp32 scalar 1.0 on `device`: the explicit root gradient of the step's backward passes (autograd would
    otherwise allocate and fill a ones_like(loss) per pass); never written."""
    t = _ones.get(device.index)
    if t is None:
        t = _ones[device.index] = torch.ones((), dtype=torch.float32, device=device)
    return t


class bn_running_updates:
    """Context: the channels-last train-mode BatchNorm forwards issued inside stand for `n` identical passes — their
    running statistics are updated n times in sequence (afan_bn_set_running_updates; main_perturb.py:173 + :196 run the
    head twice on the same images with the same weights)."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.old = _lib.load().afan_bn_set_running_updates(self.n)
        return self

    def __exit__(self, *exc):
        _lib.load().afan_bn_set_running_updates(self.old)
        return False


# ---------------------------------------------------------------------------------------------- PGD
def pgd_step_(x_adv, grad, gamma, x_clean=None, eps=0.0, clip=False, shadow=None):
    """In place: x_adv += gamma*sign(grad) [then project onto the eps-ball around x_clean]."""
    lib = _lib.load()
    _need(x_adv, "x_adv", torch.float32)
    _need(grad, "grad")
    if grad.dtype not in _DT or grad.numel() != x_adv.numel():
        raise TypeError("grad must be fp32/bf16 with x_adv's number of elements")
    if clip:
        _need(x_clean, "x_clean", torch.float32)
        if x_clean.numel() != x_adv.numel():
            raise ValueError("x_clean must match x_adv")
    if shadow is not None:
        _need(shadow, "shadow", torch.bfloat16)
    _same_layout(x_adv, grad, x_clean if clip else None, shadow)
    check(lib.afan_pgd_step(_ptr(x_adv), _ptr(grad), _DT[grad.dtype], _ptr(x_clean) if clip else None,
                            _ptr(shadow), x_adv.numel(), float(gamma), float(eps), int(bool(clip)),
                            _stream(x_adv)), "afan_pgd_step")
    return x_adv


def tensor_clamp_(t, lo, hi):
    lib = _lib.load()
    _need(t, "t", torch.float32)
    _need(lo, "min", torch.float32)
    _need(hi, "max", torch.float32)
    if lo.numel() != t.numel() or hi.numel() != t.numel():
        raise ValueError("min/max must match t")
    check(lib.afan_tensor_clamp(_ptr(t), _ptr(lo), _ptr(hi), t.numel(), _stream(t)), "afan_tensor_clamp")
    return t


def pgd_step_norms_(x_adv, grad, gamma, x_clean, eps=0.0, clip=False, shadow=None):
    """Last PGD step fused with per-sample L2/Linf norms of (x_adv - x_clean). Returns (l2, linf) [N]."""
    lib = _lib.load()
    _need(x_adv, "x_adv", torch.float32)
    _need(grad, "grad")
    _need(x_clean, "x_clean", torch.float32)
    if grad.dtype not in _DT or grad.numel() != x_adv.numel() or x_clean.numel() != x_adv.numel():
        raise TypeError("grad/x_clean must match x_adv")
    if shadow is not None:
        _need(shadow, "shadow", torch.bfloat16)
    _same_layout(x_adv, grad, x_clean, shadow)
    batch = x_adv.shape[0]
    per = x_adv.numel() // max(batch, 1)
    out = torch.empty(2, batch, dtype=torch.float32, device=x_adv.device)
    if batch == 0:
        return out[0], out[1]
    ws = _workspace(x_adv, lib.afan_norms_workspace_floats(batch, per), "norms")
    check(lib.afan_pgd_step_norms(_ptr(x_adv), _ptr(grad), _DT[grad.dtype], _ptr(x_clean), _ptr(shadow),
                                  batch, per, float(gamma), float(eps), int(bool(clip)), _ptr(ws),
                                  _ptr(out[0]), _ptr(out[1]), _stream(x_adv)), "afan_pgd_step_norms")
    return out[0], out[1]


def perturb_norms(x_adv, x_clean):
    lib = _lib.load()
    _need(x_adv, "x_adv", torch.float32)
    _need(x_clean, "x_clean", torch.float32)
    if x_clean.numel() != x_adv.numel():
        raise ValueError("x_clean must match x_adv")
    _same_layout(x_adv, x_clean)
    batch = x_adv.shape[0]
    per = x_adv.numel() // max(batch, 1)
    out = torch.empty(2, batch, dtype=torch.float32, device=x_adv.device)
    if batch == 0:
        return out[0], out[1]
    ws = _workspace(x_adv, lib.afan_norms_workspace_floats(batch, per), "norms")
    check(lib.afan_perturb_norms(_ptr(x_adv), _ptr(x_clean), batch, per, _ptr(ws), _ptr(out[0]),
                                 _ptr(out[1]), _stream(x_adv)), "afan_perturb_norms")
    return out[0], out[1]


def axpy_noise_(x_adv, u, eps, shadow=None):
    """In place: x_adv += (2u-1)*eps with host-drawn u already uploaded."""
    lib = _lib.load()
    _need(x_adv, "x_adv", torch.float32)
    _need(u, "u", torch.float32)
    if u.numel() != x_adv.numel():
        raise ValueError("u must match x_adv")
    if shadow is not None:
        _need(shadow, "shadow", torch.bfloat16)
    _same_layout(x_adv, shadow)
    if u.shape == x_adv.shape and u.stride() != x_adv.stride():
        u = u.contiguous(memory_format=torch.channels_last) if layout_of(x_adv) == AFAN_NHWC else u.contiguous()
    check(lib.afan_axpy_noise(_ptr(x_adv), _ptr(u), x_adv.numel(), float(eps), _ptr(shadow),
                              _stream(x_adv)), "afan_axpy_noise")
    return x_adv


def cast_bf16(src, out=None):
    lib = _lib.load()
    _need(src, "src", torch.float32)
    if out is None:
        out = torch.empty_like(src, dtype=torch.bfloat16)
    _need(out, "out", torch.bfloat16)
    _same_layout(src, out)
    check(lib.afan_cast_bf16(_ptr(src), _ptr(out), src.numel(), _stream(src)), "afan_cast_bf16")
    return out


def pgd_init(x, want_shadow=False):
    """PGD's start in one launch: (x as fp32, x_adv = its clone[, bf16 shadow of x_adv]) from a dense fp32 / bf16 feature
    map, all in x's memory layout."""
    lib = _lib.load()
    _need(x, "x")
    if x.dtype not in _DT:
        raise TypeError("x must be fp32 or bf16")
    x32 = x if x.dtype == torch.float32 else torch.empty_like(x, dtype=torch.float32)
    x_adv = torch.empty_like(x, dtype=torch.float32)
    shadow = torch.empty_like(x, dtype=torch.bfloat16) if want_shadow else None
    check(lib.afan_pgd_init(_ptr(x), _DT[x.dtype], None if x32 is x else _ptr(x32), _ptr(x_adv),
                            None if shadow is None else _ptr(shadow), x.numel(), _stream(x)), "afan_pgd_init")
    return x32, x_adv, shadow


# ------------------------------------------------------------------------------ mix_feature / lerp
def mix_feature(clean, adv, eps=1e-5):
    lib = _lib.load()
    _need(clean, "clean")
    _need(adv, "adv", clean.dtype)
    if clean.dtype not in _DT or clean.shape != adv.shape:
        raise TypeError("clean/adv must be fp32 or bf16 tensors of the same shape")
    if layout_of(clean) != layout_of(adv):
        adv = adv.contiguous(memory_format=torch.channels_last if layout_of(clean) == AFAN_NHWC else torch.contiguous_format)
    n, c, hw = _nchw(clean)
    out = torch.empty_like(clean)
    fn = lib.afan_mix_feature_nhwc if layout_of(clean) == AFAN_NHWC else lib.afan_mix_feature
    check(fn(_ptr(clean), _ptr(adv), _ptr(out), n, c, hw, float(eps), _DT[clean.dtype], _stream(clean)),
          "afan_mix_feature")
    return out


def lerp_points(x, y, number):
    """Interior points of get_sample_points(x, y, number): list of number-2 tensors."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    _need(y, "y", torch.float32)
    if x.shape != y.shape:
        raise ValueError("x/y shape mismatch")
    _same_layout(x, y)
    k = number - 2
    if k <= 0:
        return []
    if k > 8:
        raise ValueError("at most 10 sample points")
    percent = 1.0 / (number - 1)  # python double, as the reference
    w = (C.c_float * k)(*[i * percent for i in range(1, number - 1)])
    # k dense blocks of x.numel() elements, each carrying x's own strides
    flat = torch.empty(k * x.numel(), dtype=torch.float32, device=x.device)
    out = [flat[i * x.numel():(i + 1) * x.numel()].as_strided(x.shape, x.stride()) for i in range(k)]
    check(lib.afan_lerp_points(_ptr(x), _ptr(y), _ptr(flat), x.numel(), w, k, _stream(x)), "afan_lerp_points")
    return out


def lerp_mix(x, y, number, mix, eps=1e-5):
    """Points 1 .. number-1 of get_sample_points(x, y, number), point j re-normalised by mix_feature(x, point_j) where
    mix[j-1] is true — one launch (afan_lerp_mix).  Returns number-1 tensors; an end point without its flag is `y` itself."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    _need(y, "y", torch.float32)
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("x/y must be 4-d tensors of the same shape")
    _same_layout(x, y)
    npts = number - 1
    if npts < 1 or npts > 4 or len(mix) != npts:
        raise ValueError("2..5 sample points, one mix flag per point after the first")
    n, c, hw = _nchw(x)
    percent = 1.0 / (number - 1)
    w = (C.c_float * max(npts - 1, 1))(*[i * percent for i in range(1, number - 1)])
    mask = sum(1 << j for j, f in enumerate(mix) if f)
    flat = torch.empty(npts * x.numel(), dtype=torch.float32, device=x.device)
    out = [flat[i * x.numel():(i + 1) * x.numel()].as_strided(x.shape, x.stride()) for i in range(npts)]
    check(lib.afan_lerp_mix(_ptr(x), _ptr(y), _ptr(flat), n, c, hw, w, number, mask, float(eps), layout_of(x), _stream(x)),
          "afan_lerp_mix")
    if not mix[-1]:
        out[-1] = y
    return out


def mix_w(clean, adv, w_elem, out_dtype=torch.float32):
    """clean + w * (adv - clean) (main_learnable.py:226): clean/adv fp32 with identical strides, w_elem a 1-element
    fp32 DEVICE tensor (a view of the model's `w`), output in `out_dtype` with the inputs' memory layout."""
    lib = _lib.load()
    _need(clean, "clean", torch.float32), _need(adv, "adv", torch.float32), _need(w_elem, "w", torch.float32)
    _same_layout(clean, adv)
    if w_elem.numel() != 1 or out_dtype not in _DT:
        raise ValueError("w must be a single fp32 element; out_dtype fp32 or bf16")
    out = torch.empty_like(clean, dtype=out_dtype)
    check(lib.afan_mix_w(_ptr(clean), _ptr(adv), _ptr(w_elem), _ptr(out), _DT[out_dtype], clean.numel(),
                         _stream(clean)), "afan_mix_w")
    return out


def mix_w_backward(grad_out, clean, adv, dw_elem, accumulate=False):
    """d(loss)/dw = sum grad_out * (adv - clean) into the 1-element fp32 tensor dw_elem."""
    lib = _lib.load()
    _need(grad_out, "grad_out"), _need(clean, "clean", torch.float32), _need(adv, "adv", torch.float32)
    _need(dw_elem, "dw", torch.float32)
    _same_layout(clean, adv, grad_out)
    if grad_out.dtype not in _DT or dw_elem.numel() != 1:
        raise TypeError("grad_out must be fp32 or bf16, dw a single fp32 element")
    ws = _workspace(clean, lib.afan_mix_w_workspace_floats(), "mixw")
    check(lib.afan_mix_w_backward(_ptr(grad_out), _DT[grad_out.dtype], _ptr(clean), _ptr(adv), clean.numel(), _ptr(ws),
                                  _ptr(dw_elem), int(bool(accumulate)), _stream(clean)), "afan_mix_w_backward")
    return dw_elem


def head_forward(x, weight, bias):
    """Global average pool + flatten + linear on a channels-last [N,C,H,W] tensor: (logits fp32 [N,K], pooled fp32 [N,C])."""
    lib = _lib.load()
    _need(x, "x"), _need(weight, "weight", torch.float32)
    if x.dim() != 4 or not (layout_of(x) == AFAN_NHWC or x.shape[2] * x.shape[3] == 1) or x.dtype not in _DT:
        raise ValueError("head_forward: x must be a channels-last fp32/bf16 feature map")
    n, c, hw = _nchw(x)
    k = weight.shape[0]
    pooled = torch.empty(n, c, dtype=torch.float32, device=x.device)
    logits = torch.empty(n, k, dtype=torch.float32, device=x.device)
    check(lib.afan_head_forward(_ptr(x), _DT[x.dtype], n, c, hw, _ptr(weight), _ptr(bias), k, _ptr(pooled), _ptr(logits),
                                _stream(x)), "afan_head_forward")
    return logits, pooled


def head_backward(dlogits, weight, pooled, x_like, want_dx, dweight=None, dbias=None, accumulate=False):
    """Backward of head_forward: dx (x_like's shape / dtype / layout) or None; dweight / dbias written or added into."""
    lib = _lib.load()
    _need(dlogits, "dlogits", torch.float32), _need(pooled, "pooled", torch.float32)
    n, c, hw = _nchw(x_like)
    dx = torch.empty_like(x_like) if want_dx else None
    check(lib.afan_head_backward(_ptr(dlogits.contiguous()), _ptr(weight), _ptr(pooled), n, c, hw, weight.shape[0], _ptr(dx),
                                 _DT[x_like.dtype], _ptr(dweight), _ptr(dbias), int(bool(accumulate)), _stream(x_like)),
          "afan_head_backward")
    return dx


# ------------------------------------------------------------------------------------- BatchNorm
def bn_stats(x, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, num_batches=None):
    """Per-channel batch mean and 1/sqrt(var_biased + eps) (optionally updating running statistics)."""
    lib = _lib.load()
    _need(x, "x")
    n, c, hw = _nchw(x)
    stats = torch.empty(4, c, dtype=torch.float32, device=x.device)
    ws = _workspace(x, lib.afan_bn_workspace_floats(c), "bn")
    check(lib.afan_bn_stats(_ptr(x), _DT[x.dtype], layout_of(x), n, c, hw, float(eps), float(momentum), _ptr(ws),
                            _ptr(stats), _ptr(running_mean), _ptr(running_var), _ptr(num_batches), _stream(x)),
          "afan_bn_stats")
    return stats[0], stats[1]


class record_bn_updates:
    """Context: remember (running buffers, saved statistics, count, eps, momentum) of every train-mode BatchNorm forward
    issued inside by this thread, so that `replay()` can apply each one's running-statistics update once more later —
    see afan_bn_running_update (the clean tail pass that stands for two of the reference's)."""

    def __enter__(self):
        self.old = getattr(_tls, "bn_rec", None)
        self.items = _tls.bn_rec = []
        return self

    def __exit__(self, *exc):
        _tls.bn_rec = self.old
        return False

    def replay(self):
        lib = _lib.load()
        it = self.items
        n = len(it)
        if n == 0:
            return
        ptrs = lambda k: (C.c_void_p * n)(*[(t[k].data_ptr() if t[k] is not None else None) for t in it])
        check(lib.afan_bn_running_update_batched(
            ptrs(3), ptrs(0), ptrs(1), ptrs(2), (C.c_int64 * n)(*[t[0].numel() for t in it]),
            (C.c_double * n)(*[float(t[4]) for t in it]), (C.c_float * n)(*[float(t[5]) for t in it]),
            (C.c_float * n)(*[float(t[6]) for t in it]), n, _stream(it[0][0])), "afan_bn_running_update_batched")


def _bn_record(running_mean, running_var, num_batches, stats, m, eps, momentum, groups):
    rec = getattr(_tls, "bn_rec", None)
    if rec is not None and running_mean is not None:
        if groups != 1:
            raise ValueError("record_bn_updates: grouped BatchNorm launches cannot be replayed")
        rec.append((running_mean, running_var, num_batches, stats, m, eps, momentum))


def bn_train_forward(x, weight, bias, residual, relu, eps, momentum, running_mean, running_var, num_batches,
                     conv_stats=None, out=None, stats_out=None, groups=1):
    """Returns (y, stats) with stats = [4, C] fp32: mean, invstd, alpha, beta (kept for bn_backward).
    conv_stats: ConvStats from the producing convolution -> the moments pass over x is skipped.
    out / stats_out: write into these tensors (views of a batched buffer) instead of allocating.
    groups = 2: x is two concatenated half-batches normalised separately in one launch (conv_stats from a grouped
    convolution); stats is then [2, 4, C]."""
    lib = _lib.load()
    _need(x, "x")
    if x.dtype not in _DT:
        raise TypeError("x must be fp32 or bf16")
    if residual is not None:
        _need(residual, "residual", x.dtype)
        _same_layout(x, residual)
    n, c, hw = _nchw(x)
    if out is not None:
        _need(out, "out", x.dtype)
        _same_layout(x, out)
    y = out if out is not None else torch.empty_like(x)
    if groups != 1 and (conv_stats is None or conv_stats.acc is None):
        raise ValueError("grouped BatchNorm needs the accumulators of a grouped convolution")
    stats = stats_out if stats_out is not None else torch.empty((4, c) if groups == 1 else (groups, 4, c),
                                                                dtype=torch.float32, device=x.device)
    _bn_record(running_mean, running_var, num_batches, stats, n * hw, eps, momentum, groups)
    acc = conv_stats.acc if conv_stats is not None else None
    ready = acc is not None
    if acc is None and conv_stats is None and bn_acc_ok(x):
        acc = acc_take(x.device, c)
    if acc is not None:
        if layout_of(x) != AFAN_NHWC and not (x.shape[2] == 1 and x.shape[3] == 1):
            raise ValueError("accumulator statistics need a channels_last x")
        check(lib.afan_bn_train_forward_acc(_ptr(x), _ptr(residual), _ptr(y), _DT[x.dtype], n, c, hw, float(eps),
                                            float(momentum), _ptr(weight), _ptr(bias), int(bool(relu)), _ptr(acc),
                                            int(ready), _ptr(stats), _ptr(running_mean), _ptr(running_var),
                                            _ptr(num_batches), int(groups), _stream(x)), "afan_bn_train_forward_acc")
        return y, stats
    if conv_stats is not None:
        if layout_of(x) != AFAN_NHWC and not (x.shape[2] == 1 and x.shape[3] == 1):
            raise ValueError("conv_stats need a channels_last x")
        check(lib.afan_bn_train_forward_partials(_ptr(x), _ptr(residual), _ptr(y), _DT[x.dtype], n, c, hw, float(eps),
                                                 float(momentum), _ptr(weight), _ptr(bias), int(bool(relu)),
                                                 _ptr(conv_stats.partials), int(conv_stats.g), _ptr(conv_stats.shift),
                                                 _ptr(stats), _ptr(running_mean), _ptr(running_var), _ptr(num_batches),
                                                 _stream(x)), "afan_bn_train_forward_partials")
        return y, stats
    ws = _workspace(x, lib.afan_bn_workspace_floats(c), "bn")
    check(lib.afan_bn_train_forward(_ptr(x), _ptr(residual), _ptr(y), _DT[x.dtype], layout_of(x), n, c, hw, float(eps),
                                    float(momentum), _ptr(weight), _ptr(bias), int(bool(relu)), _ptr(ws), _ptr(stats),
                                    _ptr(running_mean), _ptr(running_var), _ptr(num_batches), _stream(x)),
          "afan_bn_train_forward")
    return y, stats


def bn_train_forward_dual(xa, bna, sta, mom_a, xb, bnb, stb, mom_b):
    """y = relu(bn_a(xa) + bn_b(xb)) in ONE launch (afan_bn_train_forward_acc_dual): a residual block's last BatchNorm and its
    projection shortcut's, both fed by convolution-epilogue accumulators (ConvStats sta / stb).  bna / bnb: the modules
    (weight, bias, eps, running buffers).  Returns (y, stats_a [4, C], stats_b [4, C])."""
    lib = _lib.load()
    _need(xa, "xa"), _need(xb, "xb", xa.dtype)
    _same_layout(xa, xb)
    n, c, hw = _nchw(xa)
    y = torch.empty_like(xa)
    stats = torch.empty((2, 4, c), dtype=torch.float32, device=xa.device)
    _bn_record(bnb.running_mean, bnb.running_var, bnb.num_batches_tracked, stats[1], n * hw, bnb.eps, mom_b, 1)
    _bn_record(bna.running_mean, bna.running_var, bna.num_batches_tracked, stats[0], n * hw, bna.eps, mom_a, 1)
    check(lib.afan_bn_train_forward_acc_dual(
        _ptr(xa), _ptr(xb), _ptr(y), _DT[xa.dtype], n, c, hw,
        float(bna.eps), float(mom_a), _ptr(bna.weight), _ptr(bna.bias), _ptr(sta.acc), _ptr(stats[0]), _ptr(bna.running_mean),
        _ptr(bna.running_var), _ptr(bna.num_batches_tracked),
        float(bnb.eps), float(mom_b), _ptr(bnb.weight), _ptr(bnb.bias), _ptr(stb.acc), _ptr(stats[1]), _ptr(bnb.running_mean),
        _ptr(bnb.running_var), _ptr(bnb.num_batches_tracked), _stream(xa)), "afan_bn_train_forward_acc_dual")
    return y, stats[0], stats[1]


def frozen_bottleneck_fwd(x, planes, stride, ws, ks):
    """One native call for a frozen-BatchNorm bottleneck's forward (afan_frozen_bottleneck_fwd): ws = (w1, w2, w3, wd | None)
    KRSC bf16, ks = (k1, k2, k3, kd | None) coefficient blocks.  Returns (out, a1, a2)."""
    lib = _lib.load()
    _cl4(x, "x")
    n, cin, h, w = x.shape
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    co = 4 * planes
    cl = torch.channels_last
    a1 = torch.empty((n, planes, h, w), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    a2 = torch.empty((n, planes, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    out = torch.empty((n, co, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    scratch = _workspace(x, (n * (planes * h * w + 2 * co * ho * wo) + 1) // 2, "fbk_fwd")
    CALLS["conv_fwd"] += 3 + (ws[3] is not None)
    check(lib.afan_frozen_bottleneck_fwd(_ptr(x), n, h, w, cin, planes, int(stride), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(ws[3]),
                                         _ptr(ks[0]), _ptr(ks[1]), _ptr(ks[2]), _ptr(ks[3]), _ptr(scratch), _ptr(a1), _ptr(a2),
                                         _ptr(out), _stream(x)), "afan_frozen_bottleneck_fwd")
    return out, a1, a2


class FrozenBlockPlan:
    """Everything about one frozen bottleneck at one input shape that does not change between optimizer steps — sizes, scratch
    requirements, the ctypes pointer objects of its weights / transposed weights / coefficient rows / gradient views — gathered
    once (det_model._block_plan) so that a block's forward and backward are a few allocations and ONE ctypes call each: the
    Detection iteration runs 388 block forwards and ~250 block backwards, and at ~100 us of Python per call (eligibility
    checks, 20 pointer wrappers, four workspace-size queries) the host, not the GPU, set its pace."""
    __slots__ = ("n", "cin", "h", "w", "ho", "wo", "planes", "co", "stride", "fwd_scratch", "bwd_scratch", "wgrad_ws", "w_ptrs", "k_ptrs",
                 "wt_ptrs", "al_ptrs", "gw_ptrs", "n_fwd", "n_wgrad", "has_ds", "keep", "sig")


def frozen_bottleneck_plan(x, planes, stride, ws, ks, wts, als, gws):
    lib = _lib.load()
    p = FrozenBlockPlan()
    p.n, p.cin, p.h, p.w = (int(v) for v in x.shape)
    p.ho, p.wo = (p.h - 1) // stride + 1, (p.w - 1) // stride + 1
    p.planes, p.co, p.stride = int(planes), 4 * int(planes), int(stride)
    p.fwd_scratch = (p.n * (p.planes * p.h * p.w + 2 * p.co * p.ho * p.wo) + 1) // 2
    p.bwd_scratch = (lib.afan_frozen_bottleneck_bwd_scratch(p.n, p.h, p.w, p.cin, p.planes, p.stride) + 1) // 2
    shapes = ((p.n, p.ho, p.wo, p.planes, p.co, 1, 1), (p.n, p.h, p.w, p.planes, p.planes, 3, p.stride), (p.n, p.h, p.w, p.cin, p.planes, 1, 1),
              (p.n, p.h, p.w, p.cin, p.co, 1, p.stride))
    p.wgrad_ws = sum(lib.afan_conv_wgrad_workspace_floats(*sh) for sh, gw in zip(shapes, (gws[2], gws[1], gws[0], gws[3])) if gw is not None)
    p.w_ptrs, p.k_ptrs = tuple(_ptr(t) for t in ws), tuple(_ptr(t) for t in ks)
    p.wt_ptrs, p.al_ptrs, p.gw_ptrs = tuple(_ptr(t) for t in wts), tuple(_ptr(t) for t in als), tuple(_ptr(t) for t in gws)
    p.has_ds = ws[3] is not None
    p.n_fwd, p.n_wgrad = 3 + int(p.has_ds), sum(gw is not None for gw in gws)
    p.keep = (ws, ks, wts, als, gws)              # the tensors behind the pointers
    # every address a launch of this plan bakes in: two plans with equal signatures issue identical launches (hipGraph reuse)
    p.sig = (p.n, p.cin, p.h, p.w, p.planes, p.stride) + tuple(None if t is None else t.data_ptr() for grp in p.keep for t in grp)
    return p


def frozen_bottleneck_fwd_plan(x, p):
    lib = _lib.load()
    cl, dev = torch.channels_last, x.device
    a1 = torch.empty((p.n, p.planes, p.h, p.w), dtype=torch.bfloat16, device=dev, memory_format=cl)
    a2 = torch.empty((p.n, p.planes, p.ho, p.wo), dtype=torch.bfloat16, device=dev, memory_format=cl)
    out = torch.empty((p.n, p.co, p.ho, p.wo), dtype=torch.bfloat16, device=dev, memory_format=cl)
    scratch = _workspace(x, p.fwd_scratch, "fbk_fwd")
    CALLS["conv_fwd"] += p.n_fwd
    w, k = p.w_ptrs, p.k_ptrs
    check(lib.afan_frozen_bottleneck_fwd(C.c_void_p(x.data_ptr()), p.n, p.h, p.w, p.cin, p.planes, p.stride, w[0], w[1], w[2], w[3], k[0], k[1], k[2], k[3],
                                         C.c_void_p(scratch.data_ptr()), C.c_void_p(a1.data_ptr()), C.c_void_p(a2.data_ptr()),
                                         C.c_void_p(out.data_ptr()), C.c_void_p(_raw_stream(dev.index))), "afan_frozen_bottleneck_fwd")
    return out, a1, a2


def frozen_bottleneck_bwd_plan(g, x, a1, a2, out, p, want_dx, pre=None, prev=None):
    """One block's backward.  Inside a stage's chain (det_model._stage_backward): `pre` = (d3, dres), this block's first backward
    step as the block behind it left it (g is then None); `prev` = the plan of the block in front, whose first step this call's
    last input-gradient launch performs on the way out — returns that block's (d3, dres) instead of dx
    (afan_frozen_bottleneck_bwd_chain)."""
    lib = _lib.load()
    scratch = _workspace(x, p.bwd_scratch, "fbk_bwd")
    wws = _workspace(x, p.wgrad_ws, "wgrad") if p.wgrad_ws else None
    CALLS["conv_dgrad"] += 2 + (1 if want_dx else 0) + (1 if (want_dx and p.has_ds) else 0)
    CALLS["conv_wgrad"] += p.n_wgrad
    wt, al, gw = p.wt_ptrs, p.al_ptrs, p.gw_ptrs
    dx = nd3 = ndres = None
    if prev is not None:
        nd3, ndres = torch.empty_like(x), torch.empty_like(x)
    elif want_dx:
        dx = torch.empty_like(x)
    check(lib.afan_frozen_bottleneck_bwd_chain(_ptr(g), _ptr(pre[0]) if pre else None, _ptr(pre[1]) if pre else None, C.c_void_p(x.data_ptr()),
                                               C.c_void_p(a1.data_ptr()), C.c_void_p(a2.data_ptr()), C.c_void_p(out.data_ptr()), p.n, p.h, p.w, p.cin,
                                               p.planes, p.stride, wt[0], wt[1], wt[2], wt[3], al[0], al[1], al[2], al[3], gw[0], gw[1], gw[2], gw[3],
                                               _ptr(wws), C.c_void_p(scratch.data_ptr()), _ptr(dx), prev.al_ptrs[2] if prev is not None else None,
                                               _ptr(nd3), _ptr(ndres), C.c_void_p(_raw_stream(x.device.index))), "afan_frozen_bottleneck_bwd_chain")
    return (nd3, ndres) if prev is not None else dx


def frozen_bottleneck_bwd(g, x, a1, a2, out, planes, stride, wts, als, gws, want_dx):
    """The backward of the same block in one native call: wts = transposed weights, als = alpha rows, gws = fp32 arena gradient
    views to add into (None entries: not wanted).  Returns dx | None."""
    lib = _lib.load()
    n, cin, h, w = x.shape
    co = 4 * planes
    scratch = _workspace(x, (lib.afan_frozen_bottleneck_bwd_scratch(n, h, w, cin, planes, int(stride)) + 1) // 2, "fbk_bwd")
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    shapes = ((n, ho, wo, planes, co, 1, 1), (n, h, w, planes, planes, 3, stride), (n, h, w, cin, planes, 1, 1), (n, h, w, cin, co, 1, stride))
    total = sum(lib.afan_conv_wgrad_workspace_floats(*sh) for sh, gw in zip(shapes, (gws[2], gws[1], gws[0], gws[3])) if gw is not None)
    wws = _workspace(x, total, "wgrad") if total else None
    dx = torch.empty_like(x) if want_dx else None
    CALLS["conv_dgrad"] += 2 + (1 if want_dx else 0) + (1 if (want_dx and wts[3] is not None) else 0)
    CALLS["conv_wgrad"] += sum(gw is not None for gw in gws)
    check(lib.afan_frozen_bottleneck_bwd(_ptr(g), _ptr(x), _ptr(a1), _ptr(a2), _ptr(out), n, h, w, cin, planes, int(stride),
                                         _ptr(wts[0]), _ptr(wts[1]), _ptr(wts[2]), _ptr(wts[3]), _ptr(als[0]), _ptr(als[1]),
                                         _ptr(als[2]), _ptr(als[3]), _ptr(gws[0]), _ptr(gws[1]), _ptr(gws[2]), _ptr(gws[3]),
                                         _ptr(wws), _ptr(scratch), _ptr(dx), _stream(x)), "afan_frozen_bottleneck_bwd")
    return dx


def bn_apply(x, mean, invstd, weight, bias, residual=None, relu=False):
    lib = _lib.load()
    _need(x, "x")
    if residual is not None:
        _need(residual, "residual", x.dtype)
        _same_layout(x, residual)
    n, c, hw = _nchw(x)
    y = torch.empty_like(x)
    ws = _workspace(x, lib.afan_bn_workspace_floats(c), "bn")
    check(lib.afan_bn_apply(_ptr(x), _ptr(residual), _ptr(y), _DT[x.dtype], layout_of(x), n, c, hw, _ptr(mean),
                            _ptr(invstd), _ptr(weight), _ptr(bias), int(bool(relu)), _ptr(ws), _stream(x)),
          "afan_bn_apply")
    return y


def affine_coefs(mean, invstd, weight, bias):
    """coefs [4, C] fp32 = mean | invstd | alpha | beta of y = x * alpha + beta (a frozen BatchNorm's constants; one launch)."""
    lib = _lib.load()
    _need(mean, "mean", torch.float32)
    c = mean.numel()
    out = torch.empty((4, c), dtype=torch.float32, device=mean.device)
    check(lib.afan_affine_coefs(_ptr(mean), _ptr(invstd), _ptr(weight), _ptr(bias), c, _ptr(out), _stream(mean)), "afan_affine_coefs")
    return out


def affine_apply(x, coefs, residual=None, relu=False):
    """y = [relu](x * alpha[c] + beta[c] [+ residual]) on a channels-last map with coefficients from affine_coefs."""
    lib = _lib.load()
    _need(x, "x")
    if layout_of(x) != AFAN_NHWC:
        raise ValueError("affine_apply takes channels-last maps")
    if residual is not None:
        _need(residual, "residual", x.dtype)
        _same_layout(x, residual)
    n, c, hw = _nchw(x)
    y = torch.empty_like(x)
    check(lib.afan_affine_apply(_ptr(x), _ptr(residual), _ptr(y), _DT[x.dtype], n, c, hw, _ptr(coefs), int(bool(relu)), _stream(x)),
          "afan_affine_apply")
    return y


def bn_backward(dy, x, y, stats, weight, bias, relu, want_dres, dweight=None, dbias=None, accumulate=False,
                partials=None, dx_out=None, dres_out=None, groups=1):
    """Returns (dx, d_residual|None). dweight/dbias (fp32 [C]) are written/accumulated when given.
    partials: ConvStats written by the dgrad that produced dy (conv_dgrad(..., bn_bwd=...)): skips the reduction pass."""
    lib = _lib.load()
    _need(dy, "dy", x.dtype)
    _need(x, "x")
    if y is not None:
        _need(y, "y", x.dtype)
    _same_layout(x, dy, y)
    n, c, hw = _nchw(x)
    dx = dx_out if dx_out is not None else torch.empty_like(x)
    dres = (dres_out if dres_out is not None else torch.empty_like(x)) if want_dres else None
    _same_layout(x, dx, dres)
    # stand-alone reductions keep the slab + finalize kernels: their blocks all finish together, so the accumulator
    # atomics would arrive as one burst and serialise per address (measured 2x slower); a dgrad epilogue spreads them
    acc = partials.acc if partials is not None else None
    ready = acc is not None
    if groups != 1 and acc is None:
        raise ValueError("grouped BatchNorm backward needs the accumulators of a grouped dgrad")
    if acc is not None:
        check(lib.afan_bn_backward_acc(_ptr(dy), _ptr(x), _ptr(y), _ptr(dx), _ptr(dres), _DT[x.dtype], n, c, hw,
                                       _ptr(stats), int(bool(relu)), _ptr(acc), int(ready), _ptr(dweight), _ptr(dbias),
                                       int(bool(accumulate)), int(groups), _stream(x)), "afan_bn_backward_acc")
        return dx, dres
    ws = _workspace(x, lib.afan_bn_workspace_floats(c), "bn")
    check(lib.afan_bn_backward(_ptr(dy), _ptr(x), _ptr(y), _ptr(dx), _ptr(dres), _DT[x.dtype], layout_of(x), n, c, hw,
                               _ptr(stats), _ptr(weight), _ptr(bias), int(bool(relu)), _ptr(ws), _ptr(dweight),
                               _ptr(dbias), int(bool(accumulate)), _ptr(partials.partials) if partials else None,
                               int(partials.g) if partials else 0, _stream(x)), "afan_bn_backward")
    return dx, dres


# ----------------------------------------------------------------------------------- convolutions
def conv_supported(ci, co, k, stride, dilation=1):
    """Forward + input gradient of this layer run on the library's kernels.  Layers with a 16/32-channel side take the
    small-channel kernel, whose BatchNorm fusions exist in the accumulator form only.  dilation > 1: 3x3 at stride 1."""
    if not _lib.load().afan_conv_supported(int(ci), int(co), int(k), int(stride)):
        return False
    if dilation != 1 and not (k == 3 and stride == 1 and ci >= 40 and co >= 40):
        return False
    return BN_ACC or ci == 3 or (ci % 64 == 0 and co % 64 == 0)


def conv_wgrad_supported(ci, co, k, stride, in_shape=None):
    """The tiled weight-gradient kernel takes co, ci multiples of 8 from 40 up.  With in_shape = (n, hi, wi) also: the image stem
    (ci == 3, afan_conv_stem.hip) and the 3x3 layers with a 16/32-channel side (afan_wgrad_small.hip), which walk whole
    image rows and so depend on the spatial size.  Anything else leaves wgrad to the vendor library."""
    if ci % 8 == 0 and co % 8 == 0 and ci >= 40 and co >= 40:
        return k in (1, 3) and stride in (1, 2)
    if in_shape is None:
        return False
    n, hi, wi = (int(v) for v in in_shape)
    return _lib.load().afan_conv_wgrad_workspace_floats(n, hi, wi, int(ci), int(co), int(k), int(stride)) > 0


def _cl4(t, name):
    _need(t, name, torch.bfloat16)
    if t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError(f"{name}: expected a channels_last 4-D bf16 tensor")
    return t


class ConvStats:
    """BatchNorm sums taken by a convolution's epilogue (consumed by bn_train_forward / bn_backward): either an f64
    accumulator block (`acc`) or per-tile fp32 partials (`partials`, `g` of them per channel)."""
    __slots__ = ("partials", "g", "shift", "acc")

    def __init__(self, partials, g, shift, acc=None):
        self.partials, self.g, self.shift, self.acc = partials, g, shift, acc

    def group(self, i, c):
        """The accumulator block of half-batch i of a grouped launch (conv_fwd / conv_dgrad with groups = 2)."""
        n = acc_block_doubles(c)
        return ConvStats(None, 0, self.shift, self.acc[i * n:(i + 1) * n])


def _conv_acc_ok(c):
    return BN_ACC and bool(_lib.load().afan_bn_acc_supported(AFAN_BF16, int(c)))


def conv_fwd(x, w, stride, stats_shift=None, want_stats=False, stats_buf=None, groups=1, dilation=1):
    """y = conv2d(x, w, padding=dilation*(k//2), stride, dilation): x [N,Ci,H,W], w [Co,Ci,k,k], both bf16 channels_last.
    want_stats=True also returns a ConvStats (moments of y around stats_shift[c], e.g. the BN running mean); None for a
    channel count the statistics fusions do not take (not a multiple of 64 on the tiled kernel)."""
    lib = _lib.load()
    CALLS["conv_fwd"] += 1
    _cl4(x, "x"), _cl4(w, "w")
    n, ci, hi, wi = x.shape
    co, ci2, k, k2 = w.shape
    if ci2 != ci or k != k2:
        raise ValueError("weight shape does not match the input")
    pad = k // 2
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    y = torch.empty((n, co, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    st = None
    if groups != 1 and not (want_stats and _conv_acc_ok(co)):
        raise ValueError("grouped statistics need the accumulator path")
    if want_stats and _conv_acc_ok(co):
        st = ConvStats(None, 0, stats_shift, acc_take(x.device, co, groups))
    elif want_stats and (ci == 3 or co % 64 or ci % 8):
        pass            # the stem kernel sums moments into accumulator blocks only: without them BatchNorm reduces itself
    elif want_stats:
        g = lib.afan_conv_fwd_tiles(n, hi, wi, ci, co, k, stride)
        if stats_buf is None or stats_buf.numel() < 2 * co * g:
            stats_buf = torch.empty(2 * co * g, dtype=torch.float32, device=x.device)
        st = ConvStats(stats_buf, g, stats_shift)
    check(lib.afan_conv_fwd_nhwc_bf16(_ptr(x), _ptr(w), _ptr(y), n, hi, wi, ci, co, k, stride, int(dilation),
                                      _ptr(st.partials) if st else None, _ptr(stats_shift) if st else None,
                                      _ptr(st.acc) if st else None, int(groups), _stream(x)), "afan_conv_fwd_nhwc_bf16")
    return (y, st) if want_stats else y


# ---- convolution + train-mode BatchNorm in one launch (afan_conv_fwd_bn_nhwc_bf16 / afan_conv_dgrad_bn_nhwc_bf16) ----------------
# The launch's workgroups meet at a grid-wide barrier, so they must all be resident: ONE such launch at a time per device — never
# from two streams, never from two processes sharing a GPU (AFAN_GRID_BN=0, or grid_bn(False) around the region, e.g. while an RCCL
# exchange is in flight on another stream: its resident kernels would make the barrier wait for the exchange to end).
GRID_BN = os.environ.get("AFAN_GRID_BN", "1") != "0" and not os.environ.get("AFAN_BENCH_ONE_DEVICE")
_grid_bars = {}
_grid_refused = set()
# GRID_BN (what every call site reads) = GRID_BN_ALLOWED (the process-wide switch: environment, or grid_bn_disable() after a
# barrier gave up) and every enclosing grid_bn(...) context and no gradient exchange in flight (exchange_in_flight)
GRID_BN_ALLOWED = GRID_BN
GRID_BN_ALLOWED_AT_IMPORT = GRID_BN          # (the optimizer's device-side guard is wired whenever the in-launch forms CAN run)
_grid_ctx = True
_grid_exchange = False


def _grid_refresh():
    global GRID_BN
    GRID_BN = bool(GRID_BN_ALLOWED and _grid_ctx and not _grid_exchange)


class grid_bn:
    """Context: switch the in-launch BatchNorm OFF for the launches issued inside (by this process).  Narrowing only: grid_bn(True)
    inside a grid_bn(False) region — or while an exchange is in flight, or after grid_bn_disable() — leaves it off."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global _grid_ctx
        self.old, _grid_ctx = _grid_ctx, _grid_ctx and self.on
        _grid_refresh()
        return self

    def __exit__(self, *exc):
        global _grid_ctx
        _grid_ctx = self.old
        _grid_refresh()
        return False


def exchange_in_flight(on):
    """A gradient exchange (RCCL kernels on the reducer's stream) is / is no longer in flight beside the launches issued from now
    on: its resident channel kernels hold CUs, so a grid barrier would wait for the exchange to END — the in-launch BatchNorm stays
    off until exchange_in_flight(False).  GradAllReducer calls it from its first launch to finish(); the trainers' graph captures
    call it where the replay will start an exchange.  Returns the previous setting."""
    global _grid_exchange
    old, _grid_exchange = _grid_exchange, bool(on)
    _grid_refresh()
    return old


def grid_bn_disable(reason=""):
    """Process-wide and permanent: no convolution + BatchNorm launch takes the in-launch form any more (a grid barrier gave up:
    the launches' workgroups are not co-resident on this GPU — another process's kernels, a preempted queue)."""
    global GRID_BN_ALLOWED
    if GRID_BN_ALLOWED:
        import warnings
        warnings.warn("in-launch BatchNorm switched off for this process" + (": " + reason if reason else "") +
                      " — the two-launch forms run from here on (same bits, ~5 % slower)")
    GRID_BN_ALLOWED = False
    _grid_refresh()


def _grid_barrier(device):
    b = _grid_bars.get(device.index)
    if b is None:
        b = _grid_bars[device.index] = torch.zeros(int(_lib.load().afan_grid_barrier_bytes()) // 4, dtype=torch.int32, device=device)
    return b


def grid_guard_word(device):
    """The grid barrier's error word of `device` as a 1-element int32 view (device memory; non-zero = a barrier's bounded spin gave
    up since the host last cleared it).  afan_sgd_step_guarded / afan_guarded_copy read it on the device."""
    w = int(_lib.load().afan_grid_barrier_error_word())
    return _grid_barrier(torch.device(device))[w:w + 1]


def grid_barrier_error(device=None):
    """True if a grid barrier's bounded spin gave up since the last call (synchronises; results since then are invalid).  Resets."""
    bad = False
    for idx, b in _grid_bars.items():
        if device is not None and torch.device(device).index not in (None, idx):
            continue
        w = int(_lib.load().afan_grid_barrier_error_word())
        if int(b[w].item()) != 0:
            bad = True
            b.zero_()
    return bad


class no_gc_during_capture:
    """Context around a hipGraph capture: Python's CYCLIC garbage collector stays off inside.  A collection in the middle of a capture
    may free an unrelated, unreachable object that owns a hipGraph, a stream or an event (a dropped trainer caught in a reference
    cycle, an exception's traceback): destroying those while a capture is in progress aborts the process (seen once in the -m gpu suite
    of round 6).  Reference-counted frees are unaffected."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        if self.was:
            import gc
            gc.enable()
        return False


def grid_shared(on):
    """Tell the library that kernels of another stream may run beside the convolution + BatchNorm launches from now on (on) or no
    longer (off): resnet_s.wgrad_stream does.  Returns the previous setting."""
    global _GRID_SHARED
    _GRID_SHARED = bool(on)
    return bool(_lib.load().afan_grid_barrier_shared_gpu(int(bool(on))))


_GRID_SHARED = False       # (part of the refused-launch memo's keys: what is declined beside other kernels is taken without them)


_GRID_BN_CHECK = os.environ.get("AFAN_GRID_BN_CHECK", "0") == "1"     # diagnostic: synchronise after every fused launch and name the one
                                                                        # whose barrier gave up (eager passes only)


def _grid_check(dev, what):
    if _GRID_BN_CHECK and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize(dev)
        if grid_barrier_error(dev):
            raise RuntimeError("in-launch BatchNorm: the grid barrier of this launch gave up: " + repr(what))


def _acc_untake(device, blk):
    a = _acc_arena(device)
    if a.off >= blk.numel() and a.buf.data_ptr() + 8 * (a.off - blk.numel()) == blk.data_ptr():
        a.off -= blk.numel()


def conv_fwd_bn(x, w, bn, momentum, residual=None, relu=True, sc=None, dilation=1):
    """(raw, act, stats) = conv(x, w) (3x3 padding 1, or 1x1; stride 1) with the train-mode BatchNorm `bn` (module: weight, bias, eps, running buffers) behind it,
    + residual, + ReLU, in ONE launch; sc = (raw_sc, bn_sc, ConvStats_sc, momentum_sc): act = relu(bn(raw) + bn_sc(raw_sc)) and a
    fourth result stats_sc — the bits of conv_fwd(want_stats) + bn_train_forward[_dual].  None when the launch is not eligible
    (nothing has run: the caller issues the two launches)."""
    if not (GRID_BN and x.is_cuda and BN_ACC):
        return None
    n, ci, hi, wi = x.shape
    co, ci2, k, k2 = w.shape
    key = ("f", n, ci, hi, wi, co, k, int(dilation), x.device.index, _GRID_SHARED)
    if ((k == 1 and dilation != 1) or k not in (1, 3) or k2 != k or ci2 != ci or ci % 64 or co % 64 or key in _grid_refused or not _conv_acc_ok(co) or (k == 1 and not GRID_BN_K1)
            or (dilation != 1 and not GRID_BN_DIL)):
        return None
    lib = _lib.load()
    _cl4(x, "x"), _cl4(w, "w")
    cl = torch.channels_last
    y = torch.empty((n, co, hi, wi), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    a = torch.empty((n, co, hi, wi), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    stats = torch.empty((2, 4, co) if sc is not None else (4, co), dtype=torch.float32, device=x.device)
    st_a = stats[0] if sc is not None else stats
    acc = acc_take(x.device, co)
    if residual is not None:
        _cl4(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("residual must have the output's shape")
    scp = [None] * 10
    if sc is not None:
        raw_sc, bsc, stc, mom_sc = sc
        _cl4(raw_sc, "raw_sc")
        if raw_sc.shape != y.shape or stc.acc is None or residual is not None or not relu:
            raise ValueError("projection form: same-shape raw_sc with accumulator sums, no residual, ReLU")
        scp = [_ptr(raw_sc), _ptr(stc.acc), _ptr(bsc.weight), _ptr(bsc.bias), float(bsc.eps), float(mom_sc), _ptr(stats[1]),
               _ptr(bsc.running_mean), _ptr(bsc.running_var), _ptr(bsc.num_batches_tracked)]
    else:
        scp[4] = scp[5] = 0.0
    rc = lib.afan_conv_fwd_bn_nhwc_bf16(_ptr(x), _ptr(w), _ptr(y), _ptr(a), n, hi, wi, ci, co, k, int(dilation), _ptr(acc), _ptr(bn.running_mean),
                                        _ptr(bn.weight), _ptr(bn.bias), float(bn.eps), float(momentum), _ptr(st_a),
                                        _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(bn.num_batches_tracked), _ptr(residual),
                                        int(bool(relu)), *scp, _ptr(_grid_barrier(x.device)), _stream(x))
    if rc == -3:                                # AFAN_ESHAPE: not this launch — remembered per shape
        _grid_refused.add(key)
        _acc_untake(x.device, acc)
        return None
    check(rc, "afan_conv_fwd_bn_nhwc_bf16")
    _grid_check(x.device, key)
    CALLS["conv_fwd"] += 1
    CALLS["conv_bn_fused"] += 1
    if sc is not None:
        _bn_record(sc[1].running_mean, sc[1].running_var, sc[1].num_batches_tracked, stats[1], n * hi * wi, sc[1].eps, sc[3], 1)
    _bn_record(bn.running_mean, bn.running_var, bn.num_batches_tracked, st_a, n * hi * wi, bn.eps, momentum, 1)
    return (y, a, st_a, stats[1]) if sc is not None else (y, a, st_a)


GRID_BN_K1 = os.environ.get("AFAN_GRID_BN_1X1", "1") != "0"            # ... and for 1x1 convolutions (0: 3x3 only — A/B)
GRID_BN_DIL = os.environ.get("AFAN_GRID_BN_DIL", "1") != "0"           # ... and for atrous 3x3 convolutions
GRID_BN_PAIR = os.environ.get("AFAN_GRID_BN_PAIR", "1") != "0"         # ... and in the stride-2 pair form's input gradient
GRID_BN_SC = os.environ.get("AFAN_GRID_BN", "2") not in ("0", "1")   # ... and the projection shortcut's BatchNorm backward in that launch


def conv_dgrad_bn(dy, wt, in_hw, bn_x, bn_stats, relu, bn_y=None, addend=None, want_dres=False, dweight=None, dbias=None,
                  accumulate=False, dx_out=None, sc=None, dilation=1, pair=None):
    """(dx, dres) = the gradient entering the INPUT of the BatchNorm (+ ReLU) in front of the 3x3 or 1x1 stride-1 convolution whose output
    gradient is dy (and that backward's masked gradient, the shortcut's share, if want_dres): conv_dgrad(bn_bwd=...) + bn_backward
    in ONE launch, the same bits.  None when the launch is not eligible (nothing has run).
    sc = (sc_x, sc_stats, d_sc_out, sc_dweight, sc_dbias) (block-output form, bn_y given): the producing block's projection shortcut's
    BatchNorm (no ReLU; input sc_x, statistics sc_stats) receives the masked gradient as well; its backward runs in the same launch:
    d_sc_out (a tensor of dx's shape) takes the gradient entering its input — bn_backward(dres, sc_x, relu=False) up to the summation
    order of its two channel sums.
    pair = (dy_sc, wt10): the stride-2 pair form (conv_dgrad's sc: a block's first 3x3 / 2 and its 1x1 / 2 projection in one launch;
    block-output form only: bn_y given, no addend — the projection's share arrives through the tenth tap); wt is then ignored."""
    if not (GRID_BN and dy.is_cuda and BN_ACC):
        return None
    if pair is not None:
        return _conv_dgrad_bn_pair(dy, pair, in_hw, bn_x, bn_stats, relu, bn_y, want_dres, dweight, dbias, accumulate)
    n, co, ho, wo = dy.shape
    ci, co2, k, _ = wt.shape
    hi, wi = in_hw
    key = ("b" if sc is None else "bs", n, ci, hi, wi, co, k, int(dilation), dy.device.index, _GRID_SHARED)
    if ((k == 1 and dilation != 1) or k not in (1, 3) or co2 != co or (hi, wi) != (ho, wo) or ci % 64 or co % 64 or key in _grid_refused or not _conv_acc_ok(ci)
            or (k == 1 and not GRID_BN_K1) or (dilation != 1 and not GRID_BN_DIL)):
        return None
    lib = _lib.load()
    _cl4(dy, "dy"), _cl4(wt, "wt"), _cl4(bn_x, "bn_x")
    _need(bn_stats, "bn_stats", torch.float32)
    cl = torch.channels_last
    if dx_out is not None:
        _cl4(dx_out, "dx_out")
        if tuple(dx_out.shape) != (n, ci, hi, wi):
            raise ValueError("dx_out must have dx's shape")
    dx = dx_out if dx_out is not None else torch.empty((n, ci, hi, wi), dtype=torch.bfloat16, device=dy.device, memory_format=cl)
    dres = torch.empty((n, ci, hi, wi), dtype=torch.bfloat16, device=dy.device, memory_format=cl) if want_dres else None
    if bn_x.shape != dx.shape or bn_stats.numel() != 4 * ci:
        raise ValueError("bn_x / bn_stats do not match dx")
    for t, nm in ((bn_y, "bn_y"), (addend, "addend")):
        if t is not None:
            _cl4(t, nm)
            if t.shape != dx.shape:
                raise ValueError(nm + " must have dx's shape")
    scp, acc2 = [None] * 6, None
    if sc is not None:
        sc_x, sc_stats, d_sc, sc_dw, sc_db = sc
        _cl4(sc_x, "sc_x"), _cl4(d_sc, "d_sc")
        _need(sc_stats, "sc_stats", torch.float32)
        if bn_y is None or sc_x.shape != dx.shape or d_sc.shape != dx.shape or sc_stats.numel() != 4 * ci:
            raise ValueError("projection form: block-output launch (bn_y) with sc_x / d_sc of dx's shape")
    acc = acc_take(dy.device, ci)
    if sc is not None:
        acc2 = acc_take(dy.device, ci)
        scp = [_ptr(sc_x), _ptr(sc_stats), _ptr(acc2), _ptr(d_sc), _ptr(sc_dw), _ptr(sc_db)]
    rc = lib.afan_conv_dgrad_bn_nhwc_bf16(_ptr(dy), _ptr(wt), _ptr(dx), _ptr(dres), n, hi, wi, ci, co, k, int(dilation), _ptr(addend), _ptr(bn_x),
                                          _ptr(bn_stats), int(bool(relu)), _ptr(bn_y), _ptr(acc), _ptr(dweight), _ptr(dbias),
                                          int(bool(accumulate)), *scp, _ptr(_grid_barrier(dy.device)), _stream(dy))
    if rc == -3:
        _grid_refused.add(key)
        if acc2 is not None:
            _acc_untake(dy.device, acc2)
        _acc_untake(dy.device, acc)
        return None
    check(rc, "afan_conv_dgrad_bn_nhwc_bf16")
    _grid_check(dy.device, key)
    CALLS["conv_dgrad"] += 1
    CALLS["conv_bn_fused"] += 1
    return dx, dres


def _conv_dgrad_bn_pair(dy, pair, in_hw, bn_x, bn_stats, relu, bn_y, want_dres, dweight, dbias, accumulate):
    dy_sc, wt10 = pair
    n, co, ho, wo = dy.shape
    hi, wi = in_hw
    ci = bn_x.shape[1]
    key = ("bp", n, ci, hi, wi, co, dy.device.index, _GRID_SHARED)
    if (not GRID_BN_PAIR or bn_y is None or (hi, wi) != (2 * ho, 2 * wo) or ci % 64 or co % 64 or key in _grid_refused or not _conv_acc_ok(ci)
            or wt10.numel() != ci * 10 * co or dy_sc.shape != dy.shape):
        return None
    lib = _lib.load()
    _cl4(dy, "dy"), _cl4(dy_sc, "dy_sc"), _cl4(bn_x, "bn_x"), _cl4(bn_y, "bn_y")
    _need(bn_stats, "bn_stats", torch.float32), _need(wt10, "wt10", torch.bfloat16)
    if dy_sc.data_ptr() != dy.data_ptr() + dy.numel() * 2:
        raise ValueError("dy_sc must sit directly behind dy in one allocation")
    if bn_x.shape != (n, ci, hi, wi) or bn_y.shape != bn_x.shape or bn_stats.numel() != 4 * ci:
        raise ValueError("bn_x / bn_y / bn_stats do not match dx")
    cl = torch.channels_last
    dx = torch.empty((n, ci, hi, wi), dtype=torch.bfloat16, device=dy.device, memory_format=cl)
    dres = torch.empty((n, ci, hi, wi), dtype=torch.bfloat16, device=dy.device, memory_format=cl) if want_dres else None
    acc = acc_take(dy.device, ci)
    rc = lib.afan_conv_dgrad_sc_bn_nhwc_bf16(_ptr(dy), _ptr(dy_sc), _ptr(wt10), _ptr(dx), _ptr(dres), n, hi, wi, ci, co, _ptr(bn_x),
                                             _ptr(bn_stats), int(bool(relu)), _ptr(bn_y), _ptr(acc), _ptr(dweight), _ptr(dbias),
                                             int(bool(accumulate)), _ptr(_grid_barrier(dy.device)), _stream(dy))
    if rc == -3:
        _grid_refused.add(key)
        _acc_untake(dy.device, acc)
        return None
    check(rc, "afan_conv_dgrad_sc_bn_nhwc_bf16")
    _grid_check(dy.device, key)
    CALLS["conv_dgrad"] += 1
    CALLS["conv_bn_fused"] += 1
    return dx, dres


def conv_fwd_affine(x, w, stride, coefs, residual=None, relu=False):
    """[relu](bf16(conv2d(x, w)) * alpha + beta [+ residual]) in ONE launch (afan_conv_fwd_affine_nhwc_bf16): a convolution with the
    frozen BatchNorm behind it in its epilogue.  Returns None where the shape belongs to a kernel without that epilogue (the
    caller issues conv_fwd + affine_apply: the same bits)."""
    lib = _lib.load()
    _cl4(x, "x"), _cl4(w, "w")
    n, ci, hi, wi = x.shape
    co, _, k, _ = w.shape
    pad = k // 2
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    y = torch.empty((n, co, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    rc = lib.afan_conv_fwd_affine_nhwc_bf16(_ptr(x), _ptr(w), _ptr(y), n, hi, wi, ci, co, k, int(stride), _ptr(coefs), _ptr(residual),
                                            int(bool(relu)), _stream(x))
    if rc == -3:
        return None
    check(rc, "afan_conv_fwd_affine_nhwc_bf16")
    CALLS["conv_fwd"] += 1
    return y


def conv_fwd_multi_ok(x, ws, stride):
    """Problems on one bf16 channels-last input with one output shape (same Co, k in {1, 3}), on the tiled kernel, BatchNorm
    moments in accumulator blocks."""
    w0 = ws[0]
    lib = _lib.load()
    return (1 <= len(ws) <= 4 and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and layout_of(x) == AFAN_NHWC
            and all(w.shape[:2] == w0.shape[:2] and w.dtype == torch.bfloat16 and w.shape[2] == w.shape[3] and w.shape[2] in (1, 3)
                    and (w.is_contiguous(memory_format=torch.channels_last) or w.shape[2] == 1) for w in ws)
            and w0.shape[1] == x.shape[1] and w0.shape[1] >= 64 and w0.shape[0] >= 64
            and w0.shape[1] % 8 == 0 and _conv_acc_ok(w0.shape[0])
            and (stride == 1 or (x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0) or all(w.shape[2] == w0.shape[2] for w in ws))
            and all(bool(lib.afan_conv_supported(w.shape[1], w.shape[0], w.shape[2], stride)) for w in ws))


def conv_fwd_multi(x, ws, stride, dilations, stats_shifts=None, groups=1):
    """[conv2d(x, w_b, padding=d_b*(k_b//2), stride, dilation=d_b) for b] in ONE launch (afan_conv_fwd_multi_nhwc_bf16).
    Returns ([y_b], [ConvStats_b | None]).  groups = 2: moments per half-batch (ConvStats.group)."""
    lib = _lib.load()
    nb = len(ws)
    CALLS["conv_fwd"] += nb
    _cl4(x, "x")
    n, ci, hi, wi = x.shape
    co, _, k, _ = ws[0].shape
    pad = k // 2
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    yall = torch.empty((nb * n, co, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    ys = [yall[b * n:(b + 1) * n] for b in range(nb)]
    sts = [None] * nb
    arr_p = C.c_void_p * nb
    shifts = accs = None
    if stats_shifts is not None:
        sts = [ConvStats(None, 0, stats_shifts[b], acc_take(x.device, co, groups)) for b in range(nb)]
        shifts = arr_p(*[t.data_ptr() for t in stats_shifts])
        accs = arr_p(*[st.acc.data_ptr() for st in sts])
    check(lib.afan_conv_fwd_multi_nhwc_bf16(_ptr(x), arr_p(*[w.data_ptr() for w in ws]), arr_p(*[y.data_ptr() for y in ys]), nb,
                                            n, hi, wi, ci, co, (C.c_int * nb)(*[int(w.shape[2]) for w in ws]), stride,
                                            (C.c_int * nb)(*[int(d) for d in dilations]), shifts, accs, int(groups if accs is not None else 1),
                                            _stream(x)),
          "afan_conv_fwd_multi_nhwc_bf16")
    return ys, sts


GRID_BN_MULTI = os.environ.get("AFAN_GRID_BN_MULTI", "1") != "0"       # ... and in the two-problem forward launch (first 3x3 / 2 + projection)


def conv_fwd_multi_bn(x, ws, stride, dilations, stats_shifts, bn0, momentum):
    """conv_fwd_multi with problem 0's train-mode BatchNorm + ReLU inside the launch (afan_conv_fwd_multi_bn_nhwc_bf16: a residual
    block's first 3x3 / stride-2 convolution; its projection rides as problem 1): (ys, ConvStats list, act0, stats0), the bits of
    conv_fwd_multi + bn_train_forward on ys[0].  None when the launch is not eligible (nothing has run)."""
    if not (GRID_BN and GRID_BN_MULTI and x.is_cuda and BN_ACC) or stats_shifts is None:
        return None
    nb = len(ws)
    n, ci, hi, wi = x.shape
    co, _, k, _ = ws[0].shape
    key = ("fm", n, ci, hi, wi, co, nb, k, int(stride), x.device.index, _GRID_SHARED)
    if nb < 2 or ci % 64 or co % 64 or key in _grid_refused or not _conv_acc_ok(co):
        return None
    lib = _lib.load()
    _cl4(x, "x")
    pad = k // 2
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    cl = torch.channels_last
    yall = torch.empty((nb * n, co, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    ys = [yall[b * n:(b + 1) * n] for b in range(nb)]
    act0 = torch.empty((n, co, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    stats0 = torch.empty((4, co), dtype=torch.float32, device=x.device)
    arr_p = C.c_void_p * nb
    sts = [ConvStats(None, 0, stats_shifts[b], acc_take(x.device, co, 1)) for b in range(nb)]
    rc = lib.afan_conv_fwd_multi_bn_nhwc_bf16(_ptr(x), arr_p(*[w.data_ptr() for w in ws]), arr_p(*[y.data_ptr() for y in ys]), nb,
                                              n, hi, wi, ci, co, (C.c_int * nb)(*[int(w.shape[2]) for w in ws]), int(stride),
                                              (C.c_int * nb)(*[int(d) for d in dilations]), arr_p(*[t.data_ptr() for t in stats_shifts]),
                                              arr_p(*[st.acc.data_ptr() for st in sts]), _ptr(act0), _ptr(bn0.weight), _ptr(bn0.bias),
                                              float(bn0.eps), float(momentum), _ptr(stats0), _ptr(bn0.running_mean),
                                              _ptr(bn0.running_var), _ptr(bn0.num_batches_tracked), 1, _ptr(_grid_barrier(x.device)),
                                              _stream(x))
    if rc == -3:
        _grid_refused.add(key)
        for st in reversed(sts):
            _acc_untake(x.device, st.acc)
        return None
    check(rc, "afan_conv_fwd_multi_bn_nhwc_bf16")
    _grid_check(x.device, key)
    CALLS["conv_fwd"] += nb
    CALLS["conv_bn_fused"] += 1
    _bn_record(bn0.running_mean, bn0.running_var, bn0.num_batches_tracked, stats0, n * ho * wo, bn0.eps, momentum, 1)
    return ys, sts, act0, stats0


def conv_dgrad(dy, wt, in_hw, stride, addend=None, bn_bwd=None, partials_buf=None, bn_y=None, groups=1, dilation=1, sc=None):
    """dx for y = conv2d(x, w): dy [N,Co,Ho,Wo]; wt = w.permute(1,0,2,3) as [Ci,Co,k,k] channels_last (CRSK memory).
    addend: bf16 tensor of dx's shape added in the epilogue.  bn_bwd = (bn_x, stats[4,Ci], relu): dx is the gradient
    entering that BatchNorm's backward -> also returns a ConvStats with its reduction partials (for bn_backward).
    bn_y: that BatchNorm's output after residual add + ReLU (dx's shape): the ReLU mask is bn_y > 0 instead of recomputed.
    sc = (dy_sc, wt10): the block's 1x1 / stride-2 projection in the same launch (afan_conv_dgrad_sc_nhwc_bf16): dy_sc of
    dy's shape in the same allocation behind dy, wt10 the flat [Ci][10][Co] operand (then `wt` is only read for its shape)."""
    lib = _lib.load()
    CALLS["conv_dgrad"] += 1
    _cl4(dy, "dy"), _cl4(wt, "wt")
    n, co, ho, wo = dy.shape
    ci, co2, k, _ = wt.shape
    if co2 != co:
        raise ValueError("transposed weight shape does not match dy")
    if sc is not None:
        dy_sc, wt10 = sc
        _cl4(dy_sc, "dy_sc")
        _need(wt10, "wt10", torch.bfloat16)
        if dy_sc.shape != dy.shape or wt10.numel() != ci * 10 * co or k != 3 or stride != 2 or dilation != 1 or groups != 1 \
                or addend is not None:
            raise ValueError("fused projection gradient: 3x3 / stride 2 with a same-shape dy_sc and a [Ci][10][Co] operand")
        CALLS["conv_dgrad"] += 1
    hi, wi = in_hw
    dx = torch.empty((n, ci, hi, wi), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    if addend is not None:
        _cl4(addend, "addend")
        if addend.shape != dx.shape:
            raise ValueError("addend must have dx's shape")
    st = bnx = bstats = None
    relu = 0
    if bn_bwd is not None:
        bnx, bstats, relu = bn_bwd
        _cl4(bnx, "bn_x")
        _need(bstats, "bn_stats", torch.float32)
        if bnx.shape != dx.shape or bstats.numel() != 4 * ci * groups:
            raise ValueError("bn_x / bn_stats do not match dx")
        if bn_y is not None:
            _cl4(bn_y, "bn_y")
            if bn_y.shape != dx.shape:
                raise ValueError("bn_y must have dx's shape")
        if groups != 1 and not _conv_acc_ok(ci):
            raise ValueError("grouped statistics need the accumulator path")
        if _conv_acc_ok(ci):
            st = ConvStats(None, 0, None, acc_take(dy.device, ci, groups))
        else:
            g = lib.afan_conv_dgrad_tiles(n, hi, wi, ci, co, k, stride)
            if partials_buf is None or partials_buf.numel() < 2 * ci * g:
                partials_buf = torch.empty(2 * ci * g, dtype=torch.float32, device=dy.device)
            st = ConvStats(partials_buf, g, None)
    if sc is not None:
        check(lib.afan_conv_dgrad_sc_nhwc_bf16(_ptr(dy), _ptr(sc[0]), _ptr(sc[1]), _ptr(dx), n, hi, wi, ci, co, _ptr(bnx),
                                               _ptr(bstats), int(bool(relu)), _ptr(bn_y) if st else None,
                                               _ptr(st.partials) if st else None, _ptr(st.acc) if st else None, _stream(dy)),
              "afan_conv_dgrad_sc_nhwc_bf16")
        return (dx, st) if bn_bwd is not None else dx
    check(lib.afan_conv_dgrad_nhwc_bf16(_ptr(dy), _ptr(wt), _ptr(dx), n, hi, wi, ci, co, k, stride, int(dilation), _ptr(addend),
                                        _ptr(bnx), _ptr(bstats), int(bool(relu)), _ptr(bn_y) if st else None,
                                        _ptr(st.partials) if st else None, _ptr(st.acc) if st else None,
                                        int(groups) if st else 1, _stream(dy)),
          "afan_conv_dgrad_nhwc_bf16")
    return (dx, st) if bn_bwd is not None else dx


def conv_wgrad(x, dy, k, stride, grad=None, accumulate=False, second=None, dilation=1):
    """Weight gradient of y = conv2d(x, w, padding=k//2, stride): returns / adds into an fp32 [Co,Ci,k,k] tensor with
    channels_last strides (KRSC memory, the parameter arena's layout).  second = (x2, dy2): the same layer's operands
    of another pass, summed in the same launch (wgrad_pairable tells when)."""
    lib = _lib.load()
    CALLS["conv_wgrad"] += 1
    _cl4(x, "x"), _cl4(dy, "dy")
    n, ci, hi, wi = x.shape
    co = dy.shape[1]
    if grad is None:
        grad = torch.empty((co, ci, k, k), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        accumulate = False
    _need(grad, "grad", torch.float32)
    if tuple(grad.shape) != (co, ci, k, k) or not (grad.is_contiguous(memory_format=torch.channels_last) or k == 1 or ci == 1):
        raise ValueError("grad must be an fp32 [Co,Ci,k,k] tensor in KRSC (channels_last) memory order")
    n2 = 0
    if second is not None:
        x2, dy2 = second
        _cl4(x2, "x2"), _cl4(dy2, "dy2")
        if x2.shape[1:] != x.shape[1:] or dy2.shape[1:] != dy.shape[1:] or x2.shape[0] != dy2.shape[0]:
            raise ValueError("second operand pair must have the first pair's layer shape")
        n2 = x2.shape[0]
    ws = _workspace(x, lib.afan_conv_wgrad_workspace_floats(n + n2, hi, wi, ci, co, k, stride), "wgrad")
    if second is None:
        check(lib.afan_conv_wgrad_nhwc_bf16(_ptr(x), _ptr(dy), _ptr(grad), n, hi, wi, ci, co, k, stride, int(dilation),
                                            _ptr(ws), int(bool(accumulate)), _stream(x)), "afan_conv_wgrad_nhwc_bf16")
    else:
        check(lib.afan_conv_wgrad2_nhwc_bf16(_ptr(x), _ptr(dy), n, _ptr(x2), _ptr(dy2), n2, _ptr(grad), hi, wi, ci, co, k,
                                             stride, int(dilation), _ptr(ws), int(bool(accumulate)), _stream(x)),
              "afan_conv_wgrad2_nhwc_bf16")
    return grad


def wgrad_pairable(x, dy, k=3, stride=1):
    """conv_wgrad(..., second=) takes this layer: the tiled kernel (channels multiples of 64, first pair's pixels % 64 == 0)
    or the small-channel kernel (any layer it takes); not the image stem."""
    if x.shape[1] % 64 == 0 and dy.shape[1] % 64 == 0:
        return (dy.shape[0] * dy.shape[2] * dy.shape[3]) % 64 == 0
    return x.shape[1] != 3 and conv_wgrad_supported(x.shape[1], dy.shape[1], k, stride, (x.shape[0], x.shape[2], x.shape[3]))


# ---- the general convolution (afan_conv_f32.hip): fp32 arithmetic on v_mfma_f32_32x32x2_f32, any shape / layout ----------
def _w_layout(w):
    """(w as the kernels can address it, AFAN_NHWC for KRSC memory | AFAN_NCHW for KCRS)."""
    if w.dim() != 4 or w.shape[2] != w.shape[3]:
        raise ValueError("convolution weight must be [Co, Ci, k, k]")
    if w.is_contiguous():
        return w, AFAN_NCHW
    if w.is_contiguous(memory_format=torch.channels_last):
        return w, AFAN_NHWC
    return w.contiguous(), AFAN_NCHW


def _conv_out(i, k, stride, pad, dil):
    return (i + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_general_fwd(x, w, bias=None, stride=1, padding=0, dilation=1):
    """y = conv2d(x, w, bias, stride, padding, dilation) with fp32 arithmetic (f32 MFMA): x, w fp32 or bf16 (same dtype),
    x dense NCHW or channels-last (y follows), w [Co,Ci,k,k] in KCRS or KRSC memory."""
    lib = _lib.load()
    CALLS["conv_general"] += 1
    _need(x, "x"), _need(w, "w", x.dtype)
    if x.dim() != 4 or x.dtype not in _DT or w.shape[1] != x.shape[1]:
        raise TypeError("conv_general_fwd: x [N,Ci,H,W] and w [Co,Ci,k,k], both fp32 or both bf16")
    w, wl = _w_layout(w)
    n, ci, hi, wi = x.shape
    co, k = w.shape[0], w.shape[2]
    ho, wo = _conv_out(hi, k, stride, padding, dilation), _conv_out(wi, k, stride, padding, dilation)
    lay = layout_of(x)
    y = torch.empty((n, co, ho, wo), dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if lay == AFAN_NHWC else torch.contiguous_format)
    if bias is not None:
        _need(bias, "bias", torch.float32)
    check(lib.afan_conv_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _DT[x.dtype], lay, wl, n, hi, wi, ci, co, k, int(stride),
                            int(padding), int(dilation), _stream(x)), "afan_conv_fwd")
    return y


def conv_general_dgrad(dy, w, in_hw, stride=1, padding=0, dilation=1):
    """dx [N,Ci,Hi,Wi] for y = conv2d(x, w, ...): dy in the activations' layout and dtype, w the UNTRANSPOSED weight."""
    lib = _lib.load()
    CALLS["conv_general"] += 1
    _need(dy, "dy"), _need(w, "w", dy.dtype)
    if dy.dim() != 4 or dy.dtype not in _DT or w.shape[0] != dy.shape[1]:
        raise TypeError("conv_general_dgrad: dy [N,Co,Ho,Wo] and w [Co,Ci,k,k], both fp32 or both bf16")
    w, wl = _w_layout(w)
    n, co = dy.shape[0], dy.shape[1]
    ci, k = w.shape[1], w.shape[2]
    hi, wi = int(in_hw[0]), int(in_hw[1])
    if (_conv_out(hi, k, stride, padding, dilation), _conv_out(wi, k, stride, padding, dilation)) != tuple(dy.shape[2:]):
        raise ValueError("dy's spatial size does not match the convolution of an input of size in_hw")
    lay = layout_of(dy)
    dx = torch.empty((n, ci, hi, wi), dtype=dy.dtype, device=dy.device,
                     memory_format=torch.channels_last if lay == AFAN_NHWC else torch.contiguous_format)
    check(lib.afan_conv_dgrad(_ptr(dy), _ptr(w), _ptr(dx), _DT[dy.dtype], lay, wl, n, hi, wi, ci, co, k, int(stride),
                              int(padding), int(dilation), _stream(dy)), "afan_conv_dgrad")
    return dx


def conv_wgrad_plan(x, dy, k, stride):
    """Tile-configuration code of the tuned weight-gradient kernel for this problem (0: not on that kernel); equal codes can
    share a conv_wgrad_multi launch."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and dy.dtype == torch.bfloat16 and dy.is_contiguous(memory_format=torch.channels_last)):
        return 0
    n, ci, hi, wi = x.shape
    return int(_lib.load().afan_conv_wgrad_plan(n, hi, wi, ci, dy.shape[1], int(k), int(stride)))


def conv_wgrad_multi(items):
    """items = [(x, dy, k, stride, dilation, grad)] (2..4 layers, equal conv_wgrad_plan codes): grad += wgrad(x, dy) for all of
    them in ONE launch + one reduction launch (afan_conv_wgrad_multi_nhwc_bf16), bit-identical to separate conv_wgrad calls."""
    lib = _lib.load()
    nb = len(items)
    CALLS["conv_wgrad"] += nb
    total = 0
    for (x, dy, k, st, dil, grad) in items:
        _cl4(x, "x"), _cl4(dy, "dy")
        n, ci, hi, wi = x.shape
        co = dy.shape[1]
        if tuple(grad.shape) != (co, ci, k, k) or grad.dtype != torch.float32 or not (grad.is_contiguous(memory_format=torch.channels_last) or k == 1 or ci == 1):
            raise ValueError("grad must be fp32 [Co,Ci,k,k] with channels_last strides")
        total += lib.afan_conv_wgrad_workspace_floats(n, hi, wi, ci, co, k, st)
    ws = _workspace(items[0][0], total, "wgrad")
    ap, al, ai = C.c_void_p * nb, C.c_int64 * nb, C.c_int * nb
    check(lib.afan_conv_wgrad_multi_nhwc_bf16(
        nb, ap(*[i[0].data_ptr() for i in items]), ap(*[i[1].data_ptr() for i in items]), ap(*[i[5].data_ptr() for i in items]),
        al(*[i[0].shape[0] for i in items]), al(*[i[0].shape[2] for i in items]), al(*[i[0].shape[3] for i in items]),
        al(*[i[0].shape[1] for i in items]), al(*[i[1].shape[1] for i in items]), ai(*[int(i[2]) for i in items]),
        ai(*[int(i[3]) for i in items]), ai(*[int(i[4]) for i in items]), _ptr(ws), 1, _stream(items[0][0])),
        "afan_conv_wgrad_multi_nhwc_bf16")


def conv_general_wgrad(x, dy, k, stride=1, padding=0, dilation=1, grad=None, accumulate=False):
    """fp32 weight gradient [Co,Ci,k,k] of y = conv2d(x, w, ...): written / added into `grad` (KCRS or KRSC memory), or
    returned as a new tensor in the memory order that matches the activations' layout."""
    lib = _lib.load()
    CALLS["conv_general"] += 1
    _need(x, "x"), _need(dy, "dy", x.dtype)
    if x.dim() != 4 or x.dtype not in _DT:
        raise TypeError("conv_general_wgrad: x / dy fp32 or bf16 [N,C,H,W]")
    lay = layout_of(x)
    if layout_of(dy) != lay and dy.numel() > 0 and not (dy.shape[1] == 1 or dy.shape[2] * dy.shape[3] == 1):
        dy = dy.contiguous(memory_format=torch.channels_last if lay == AFAN_NHWC else torch.contiguous_format)
    n, ci, hi, wi = x.shape
    co = dy.shape[1]
    if grad is None:
        grad = torch.empty((co, ci, k, k), dtype=torch.float32, device=x.device,
                           memory_format=torch.channels_last if lay == AFAN_NHWC else torch.contiguous_format)
        accumulate = False
    _need(grad, "grad", torch.float32)
    if tuple(grad.shape) != (co, ci, k, k):
        raise ValueError("grad must be [Co,Ci,k,k]")
    if grad.is_contiguous():
        wl = AFAN_NCHW
    elif grad.is_contiguous(memory_format=torch.channels_last):
        wl = AFAN_NHWC
    else:
        raise ValueError("grad must be dense in KCRS or KRSC memory order")
    ws = _workspace(x, lib.afan_conv_wgrad_f32_workspace_floats(n, hi, wi, ci, co, k, int(stride), int(padding), int(dilation)),
                    "wgrad_f32")
    check(lib.afan_conv_wgrad(_ptr(x), _ptr(dy), _ptr(grad), _DT[x.dtype], lay, wl, n, hi, wi, ci, co, k, int(stride),
                              int(padding), int(dilation), _ptr(ws), int(bool(accumulate)), _stream(x)), "afan_conv_wgrad")
    return grad


def transpose_weights(src_arena, dst_arena, desc_dev, n_desc, total_tiles):
    """Batched KRSC -> CRSK transpose of all convolution weights listed in desc_dev (see include/afan_hip.h)."""
    lib = _lib.load()
    _need(src_arena, "src_arena", torch.bfloat16)
    _need(dst_arena, "dst_arena", torch.bfloat16)
    _need(desc_dev, "desc_dev", torch.int64)
    check(lib.afan_transpose_weights(_ptr(src_arena), _ptr(dst_arena), _ptr(desc_dev), int(n_desc), int(total_tiles),
                                     _stream(src_arena)), "afan_transpose_weights")


# ------------------------------------------------------------------ DeepLabv3+ layers (afan_seg.hip, afan_conv_stem7.hip)
def _mf(t):
    return torch.channels_last if layout_of(t) == AFAN_NHWC else torch.contiguous_format


def upsample_bilinear(x, size):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False) for a dense fp32 / bf16 [N,C,H,W] tensor."""
    lib = _lib.load()
    _need(x, "x")
    if x.dim() != 4 or x.dtype not in _DT:
        raise TypeError("upsample_bilinear: 4-D fp32 / bf16 tensor expected")
    n, c, hi, wi = x.shape
    ho, wo = int(size[0]), int(size[1])
    y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=_mf(x))
    check(lib.afan_upsample_bilinear_fwd(_ptr(x), _ptr(y), _DT[x.dtype], layout_of(x), n, c, hi, wi, ho, wo, _stream(x)),
          "afan_upsample_bilinear_fwd")
    return y


def upsample_bilinear_backward(dy, in_hw):
    lib = _lib.load()
    _need(dy, "dy")
    n, c, ho, wo = dy.shape
    hi, wi = int(in_hw[0]), int(in_hw[1])
    dx = torch.empty((n, c, hi, wi), dtype=dy.dtype, device=dy.device, memory_format=_mf(dy))
    check(lib.afan_upsample_bilinear_bwd(_ptr(dy), _ptr(dx), _DT[dy.dtype], layout_of(dy), n, c, hi, wi, ho, wo, _stream(dy)),
          "afan_upsample_bilinear_bwd")
    return dx


def upsample_concat(low, hi):
    """torch.cat([low, F.interpolate(hi, low's size, 'bilinear')], dim=1) for bf16 / fp32 channels-last tensors
    (_deeplab.py:54-56) without the concat's pass over the resized tensor: it is written straight into its channel slice
    (afan_upsample_bilinear_fwd_slice); `low` is copied into the first channels."""
    lib = _lib.load()
    _need(low, "low"), _need(hi, "hi")
    if low.dim() != 4 or hi.dim() != 4 or low.dtype != hi.dtype or low.shape[0] != hi.shape[0] or layout_of(hi) != AFAN_NHWC:
        raise TypeError("upsample_concat: two channels-last 4-D tensors of one dtype and batch size expected")
    n, c0, ho, wo = low.shape
    c, hi_h, hi_w = hi.shape[1], hi.shape[2], hi.shape[3]
    out = torch.empty((n, c0 + c, ho, wo), dtype=hi.dtype, device=hi.device, memory_format=torch.channels_last)
    out[:, :c0].copy_(low)
    es = out.element_size()
    check(lib.afan_upsample_bilinear_fwd_slice(_ptr(hi), out.data_ptr() + c0 * es, _DT[hi.dtype], n, c, hi_h, hi_w, ho, wo,
                                               c0 + c, _stream(hi)), "afan_upsample_bilinear_fwd_slice")
    return out


def upsample_concat_backward(g, c0, in_hw):
    """Gradient of upsample_concat w.r.t. `hi`, read from channels c0.. of the channels-last dense g [N, c0 + c, Ho, Wo]."""
    lib = _lib.load()
    _need(g, "g")
    if layout_of(g) != AFAN_NHWC or not g.is_contiguous(memory_format=torch.channels_last):
        raise TypeError("upsample_concat_backward: dense channels-last gradient expected")
    n, ct, ho, wo = g.shape
    c, hi, wi = ct - c0, int(in_hw[0]), int(in_hw[1])
    dx = torch.empty((n, c, hi, wi), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
    ws = _workspace(g, lib.afan_upsample_bilinear_bwd_workspace_floats(n, c, wi, ho), "upsample_bwd")    # two separable passes
    check(lib.afan_upsample_bilinear_bwd_slice(g.data_ptr() + c0 * g.element_size(), _ptr(dx), _DT[g.dtype], n, c, hi, wi, ho, wo,
                                               ct, _ptr(ws), _stream(g)), "afan_upsample_bilinear_bwd_slice")
    return dx


CE2D_MAX_CLASSES = 32


def ce2d(logits, target, ignore_index=255, grad_scale=1.0, want_grad=True):
    """nn.CrossEntropyLoss(ignore_index, reduction='mean') on [N,C,H,W] fp32 logits and [N,H,W] int64 labels:
    (loss [1], grad_scale * d(loss)/d(logits) in the logits' layout | None)."""
    lib = _lib.load()
    _need(logits, "logits", torch.float32)
    if logits.dim() != 4 or target.dim() != 3 or target.dtype != torch.int64 or not target.is_cuda \
            or tuple(target.shape) != (logits.shape[0], logits.shape[2], logits.shape[3]):
        raise TypeError("ce2d: logits [N,C,H,W] fp32 and target [N,H,W] int64 on the GPU")
    n, c, h, w = logits.shape
    target = target.contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    ws = _workspace(logits, lib.afan_ce2d_workspace_floats(n * h * w), "ce2d")
    check(lib.afan_ce2d(_ptr(logits), _ptr(target), layout_of(logits), n, c, h * w, int(ignore_index), float(grad_scale),
                        _ptr(ws), _ptr(loss), _ptr(dl), _stream(logits)), "afan_ce2d")
    return loss, dl


def ce2d_upsampled_ok(logits, size):
    """The fused resize + cross-entropy kernel takes channels-last fp32 logits of up to 32 classes, resized UP."""
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4 and logits.shape[1] <= CE2D_MAX_CLASSES):
        return False
    h, w = logits.shape[2:]
    if layout_of(logits) != AFAN_NHWC and not (logits.shape[1] == 1 or h * w == 1):
        return False
    return size[0] >= h and size[1] >= w


def ce2d_upsampled(logits, target, ignore_index=255, grad_scale=1.0, want_grad=True):
    """nn.CrossEntropyLoss(ignore_index)(F.interpolate(logits, target.shape[1:], 'bilinear'), target) and its gradient
    w.r.t. the LOW-resolution logits in one kernel: (loss [1], grad_scale * d(loss)/d(logits) | None)."""
    lib = _lib.load()
    _need(logits, "logits", torch.float32)
    n, c, h, w = logits.shape
    if target.dim() != 3 or target.dtype != torch.int64 or not target.is_cuda or target.shape[0] != n:
        raise TypeError("ce2d_upsampled: logits [N,C,h,w] fp32 channels-last and target [N,H,W] int64 on the GPU")
    ho, wo = int(target.shape[1]), int(target.shape[2])
    target = target.contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    ws = _workspace(logits, lib.afan_ce2d_upsampled_workspace_floats(n, c, h, w, ho, wo), "ce2d_up")
    check(lib.afan_ce2d_upsampled(_ptr(logits), _ptr(target), n, c, h, w, ho, wo, int(ignore_index), float(grad_scale), _ptr(ws),
                                  _ptr(loss), _ptr(dl), _stream(logits)), "afan_ce2d_upsampled")
    return loss, dl


def maxpool3x3s2(x):
    lib = _lib.load()
    _need(x, "x")
    n, c, hi, wi = x.shape
    y = torch.empty((n, c, (hi - 1) // 2 + 1, (wi - 1) // 2 + 1), dtype=x.dtype, device=x.device, memory_format=_mf(x))
    check(lib.afan_maxpool3x3s2_fwd(_ptr(x), _ptr(y), _DT[x.dtype], layout_of(x), n, c, hi, wi, _stream(x)),
          "afan_maxpool3x3s2_fwd")
    return y


def maxpool3x3s2_backward(dy, x):
    lib = _lib.load()
    _need(dy, "dy", x.dtype), _need(x, "x")
    if layout_of(dy) != layout_of(x) and dy.numel() > 0:
        raise ValueError("dy must have x's memory layout")
    n, c, hi, wi = x.shape
    dx = torch.empty_like(x)
    check(lib.afan_maxpool3x3s2_bwd(_ptr(dy), _ptr(x), _ptr(dx), _DT[x.dtype], layout_of(x), n, c, hi, wi, _stream(x)),
          "afan_maxpool3x3s2_bwd")
    return dx


def maxpool2d(x, k, stride, pad=0, want_idx=True):
    """nn.MaxPool2d(k, stride, pad) (first maximum wins, NaN wins): (y, idx | None) — idx (uint8, y's shape and layout) holds
    the winner's position inside its window for maxpool2d_backward."""
    lib = _lib.load()
    _need(x, "x")
    if x.dim() != 4 or x.dtype not in _DT:
        raise TypeError("maxpool2d: 4-D fp32 / bf16 tensor expected")
    n, c, hi, wi = x.shape
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=_mf(x))
    idx = torch.empty((n, c, ho, wo), dtype=torch.uint8, device=x.device, memory_format=_mf(x)) if want_idx else None
    check(lib.afan_maxpool2d_fwd(_ptr(x), _ptr(y), _ptr(idx), _DT[x.dtype], layout_of(x), n, c, hi, wi, int(k), int(stride), int(pad),
                                 _stream(x)), "afan_maxpool2d_fwd")
    return y, idx


def maxpool2d_backward(dy, idx, in_shape, k, stride, pad=0, x=None, channels_last=None):
    """dx of maxpool2d for an input of shape in_shape, from the forward's idx (or by re-scanning x when idx is None).
    channels_last: the layout dy / idx are in (default: dy's strides; needed for 1x1 outputs, whose strides do not tell)."""
    lib = _lib.load()
    _need(dy, "dy")
    n, c, hi, wi = (int(v) for v in in_shape)
    lay = layout_of(dy) if channels_last is None else (AFAN_NHWC if channels_last else AFAN_NCHW)
    dx = torch.empty((n, c, hi, wi), dtype=dy.dtype, device=dy.device,
                     memory_format=torch.channels_last if lay == AFAN_NHWC else torch.contiguous_format)
    if channels_last is None and idx is not None and layout_of(idx) != lay and idx.numel() > 0 and not (c == 1 or dy.shape[2] * dy.shape[3] == 1):
        raise ValueError("dy must have the forward output's memory layout")
    check(lib.afan_maxpool2d_bwd(_ptr(dy), _ptr(x), _ptr(idx), _ptr(dx), _DT[dy.dtype], lay, n, c, hi, wi, int(k), int(stride),
                                 int(pad), _stream(dy)), "afan_maxpool2d_bwd")
    return dx


def conv_dgrad_affine(dy, wt, in_hw, stride, alpha, act):
    """dx = bf16((act > 0 ? bf16(dgrad(dy, wt)) : 0) * alpha[c]) in ONE launch (afan_conv_dgrad_affine_nhwc_bf16): an input gradient
    with the backward of the frozen BatchNorm + ReLU it runs into.  None where another kernel owns the shape (the caller issues
    conv_dgrad + affine_relu_backward: the same bits)."""
    lib = _lib.load()
    _cl4(dy, "dy"), _cl4(wt, "wt"), _cl4(act, "act")
    n, co, ho, wo = dy.shape
    ci, _, k, _ = wt.shape
    hi, wi = in_hw
    dx = torch.empty((n, ci, hi, wi), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    rc = lib.afan_conv_dgrad_affine_nhwc_bf16(_ptr(dy), _ptr(wt), _ptr(dx), n, hi, wi, ci, co, k, int(stride), _ptr(alpha), _ptr(act), _stream(dy))
    if rc == -3:
        return None
    check(rc, "afan_conv_dgrad_affine_nhwc_bf16")
    CALLS["conv_dgrad"] += 1
    return dx


def conv_dgrad_dual(dy, wt, in_hw, stride, addend, alpha, act):
    """(d3, dres) = the gradient arriving at a frozen-BatchNorm residual block's output with that block's first backward step
    applied on the way out, in ONE launch (afan_conv_dgrad_dual_nhwc_bf16): g = bf16(bf16(dgrad(dy, wt)) + addend) (addend may be
    None), m = act > 0 ? g : 0, dres = m, d3 = bf16(m * alpha[c]).  None where another kernel owns the shape (the caller issues
    conv_dgrad(addend) + affine_relu_backward with both outputs: the same bits)."""
    lib = _lib.load()
    _cl4(dy, "dy"), _cl4(wt, "wt"), _cl4(act, "act")
    n, co, ho, wo = dy.shape
    ci, _, k, _ = wt.shape
    hi, wi = in_hw
    d3 = torch.empty((n, ci, hi, wi), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    dres = torch.empty_like(d3)
    rc = lib.afan_conv_dgrad_dual_nhwc_bf16(_ptr(dy), _ptr(wt), _ptr(d3), _ptr(dres), n, hi, wi, ci, co, k, int(stride), _ptr(addend), _ptr(alpha),
                                            _ptr(act), _stream(dy))
    if rc == -3:
        return None
    check(rc, "afan_conv_dgrad_dual_nhwc_bf16")
    CALLS["conv_dgrad"] += 1
    return d3, dres


def affine_relu_backward(dy, y, alpha, relu, want_dx=True, want_dres=False):
    """Backward of y = [relu](x * alpha[c] + beta[c] [+ res]) with constant coefficients: (dx | None, d_res | None)."""
    lib = _lib.load()
    _need(dy, "dy")
    if relu:
        _need(y, "y", dy.dtype)
        _same_layout(dy, y)
    n, c, hw = _nchw(dy)
    dx = torch.empty_like(dy) if want_dx else None
    dres = torch.empty_like(dy) if want_dres else None
    if alpha is not None:
        _need(alpha, "alpha", torch.float32)
    check(lib.afan_affine_relu_bwd(_ptr(dy), _ptr(y) if relu else None, _ptr(alpha), _ptr(dx), _ptr(dres), _DT[dy.dtype], layout_of(dy),
                                   n, c, hw, int(bool(relu)), _stream(dy)), "afan_affine_relu_bwd")
    return dx, dres


def avgpool(x, out_fp32=False):
    """nn.AdaptiveAvgPool2d(1): [N,C,H,W] -> [N,C,1,1] (x's dtype, or fp32 with out_fp32; fp32 accumulate)."""
    lib = _lib.load()
    _need(x, "x")
    n, c, hw = _nchw(x)
    y = torch.empty((n, c, 1, 1), dtype=torch.float32 if out_fp32 else x.dtype, device=x.device)
    check(lib.afan_avgpool_fwd(_ptr(x), _ptr(y), _DT[x.dtype], layout_of(x), n, c, hw, int(bool(out_fp32)), _stream(x)),
          "afan_avgpool_fwd")
    return y


def avgpool_backward(dy, like):
    """dx (like's shape / dtype / layout) = dy / HW broadcast; dy [N,C,1,1] in like's dtype or fp32."""
    lib = _lib.load()
    f32 = dy.dtype == torch.float32
    _need(dy, "dy", None if f32 else like.dtype)
    n, c, hw = _nchw(like)
    dx = torch.empty_like(like)
    check(lib.afan_avgpool_bwd(_ptr(dy.reshape(n, c).contiguous()), _ptr(dx), _DT[like.dtype], layout_of(like), n, c, hw,
                               int(f32), _stream(like)), "afan_avgpool_bwd")
    return dx


def linear_small_ok(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.shape[0] <= _lib.load().afan_linear_small_max_rows()
            and (x.shape[0] * weight.shape[0] + 16 * x.shape[0] * 64) * 4 <= 64 * 1024)


def linear_small(x, weight):
    """y[n,co] = sum_ci x[n,ci] * w[co,ci] in fp32 for a handful of rows (the ASPP pooling branch's 1x1 convolution)."""
    lib = _lib.load()
    _need(x, "x", torch.float32), _need(weight, "weight", torch.float32)
    n, ci = x.shape[0], x.numel() // x.shape[0]
    co = weight.shape[0]
    y = torch.empty((n, co), dtype=torch.float32, device=x.device)
    check(lib.afan_linear_small_fwd(_ptr(x), _ptr(weight), _ptr(y), n, ci, co, _stream(x)), "afan_linear_small_fwd")
    return y


def linear_small_backward(dy, x, weight, want_dx, dweight=None, accumulate=False):
    lib = _lib.load()
    _need(dy, "dy", torch.float32)
    n, ci = x.shape[0], x.numel() // x.shape[0]
    co = weight.shape[0]
    dx = torch.empty((n, ci), dtype=torch.float32, device=x.device) if want_dx else None
    check(lib.afan_linear_small_bwd(_ptr(dy.contiguous()), _ptr(x), _ptr(weight), _ptr(dx), _ptr(dweight), n, ci, co,
                                    int(bool(accumulate)), _stream(x)), "afan_linear_small_bwd")
    return dx


def pointwise_supported(x, co):
    return (x.is_cuda and x.dim() == 4 and x.dtype in _DT and (layout_of(x) == AFAN_NHWC or x.shape[2] * x.shape[3] == 1)
            and x.shape[1] % 8 == 0 and co <= _lib.load().afan_pointwise_max_co() and x.shape[1] * co * 4 <= 64 * 1024)


def pointwise_forward(x, weight, bias):
    """1x1 convolution with bias on a channels-last map: fp32 logits [N,Co,H,W] (channels-last memory)."""
    lib = _lib.load()
    _need(x, "x"), _need(weight, "weight", torch.float32)
    n, ci, h, w = x.shape
    co = weight.shape[0]
    y = torch.empty((n, co, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    check(lib.afan_pointwise_fwd(_ptr(x), _DT[x.dtype], _ptr(weight.reshape(co, ci)), _ptr(bias), _ptr(y), n * h * w, ci, co,
                                 _stream(x)), "afan_pointwise_fwd")
    return y


def pointwise_backward(dy, x, weight, want_dx, dweight=None, dbias=None, accumulate=False):
    lib = _lib.load()
    _need(dy, "dy", torch.float32), _need(x, "x")
    n, ci, h, w = x.shape
    co = weight.shape[0]
    m = n * h * w
    if layout_of(dy) != AFAN_NHWC and h * w != 1 and co != 1:
        dy = dy.contiguous(memory_format=torch.channels_last)
    dx = None
    if want_dx:
        dx = torch.empty_like(x)
        check(lib.afan_pointwise_bwd_dx(_ptr(dy), _ptr(weight.reshape(co, ci)), _ptr(dx), _DT[x.dtype], m, ci, co, _stream(x)),
              "afan_pointwise_bwd_dx")
    if dweight is not None:
        ws = _workspace(x, lib.afan_pointwise_workspace_floats(m, ci, co), "pointwise")
        check(lib.afan_pointwise_bwd_dw(_ptr(dy), _ptr(x), _DT[x.dtype], _ptr(dweight), _ptr(dbias), m, ci, co, _ptr(ws),
                                        int(bool(accumulate)), _stream(x)), "afan_pointwise_bwd_dw")
    return dx


_dropout_state = {}


def dropout_state(device):
    """The device-resident generator of afan_dropout for this device (one uint64, seeded from torch's CPU generator)."""
    st = _dropout_state.get(device.index)
    if st is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        st = _dropout_state[device.index] = torch.tensor([seed], dtype=torch.int64, device=device)
    return st


def dropout(x, p, mask=None, used=None):
    """Forward (used=None): returns (y, used) — `used` [1] int64 holds the seed of this call.  Backward: pass the forward's
    `used` (the mask is re-derived).  mask: uint8 tensor of x's shape / layout (host-supplied, for parity tests)."""
    lib = _lib.load()
    _need(x, "x")
    y = torch.empty_like(x)
    fwd = used is None
    if fwd:
        used = torch.empty(1, dtype=torch.int64, device=x.device)
    if mask is not None:
        _need(mask, "mask", torch.uint8)
        _same_layout(x, mask)
    state = dropout_state(x.device) if (fwd and mask is None) else None
    check(lib.afan_dropout(_ptr(x), _ptr(y), _DT[x.dtype], x.numel(), float(p), _ptr(mask), _ptr(state), _ptr(used), 1,
                           _stream(x)), "afan_dropout")
    return y, used


def conv_stem7_ok(x, w, stride, padding):
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] == 3 and layout_of(x) == AFAN_NHWC
            and w.dtype == torch.bfloat16 and tuple(w.shape) == (64, 3, 7, 7) and tuple(stride) == (2, 2)
            and tuple(padding) == (3, 3))


def conv_stem7_im2col(x):
    """[N*Ho*Wo, K] -> as a [N, K, Ho, Wo] channels-last bf16 tensor (K = 152: the 147 taps + 5 zero columns)."""
    lib = _lib.load()
    _cl4(x, "x")
    n, _, hi, wi = x.shape
    k = lib.afan_conv_stem7_im2col_k()
    ho, wo = (hi - 1) // 2 + 1, (wi - 1) // 2 + 1
    cols = torch.empty((n, k, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    check(lib.afan_conv_stem7_im2col(_ptr(x), _ptr(cols), n, hi, wi, _stream(x)), "afan_conv_stem7_im2col")
    return cols


def conv_stem7_col2im(dcols, in_hw):
    """dx [N,3,Hi,Wi] bf16 channels-last from dcols [N,152,Ho,Wo] (the adjoint of conv_stem7_im2col)."""
    lib = _lib.load()
    _cl4(dcols, "dcols")
    n, k = dcols.shape[0], dcols.shape[1]
    hi, wi = int(in_hw[0]), int(in_hw[1])
    if k != lib.afan_conv_stem7_im2col_k() or tuple(dcols.shape[2:]) != ((hi - 1) // 2 + 1, (wi - 1) // 2 + 1) or dcols.dtype != torch.bfloat16:
        raise ValueError("conv_stem7_col2im: dcols must be the bf16 column tensor of an input of size in_hw")
    dx = torch.empty((n, 3, hi, wi), dtype=torch.bfloat16, device=dcols.device, memory_format=torch.channels_last)
    check(lib.afan_conv_stem7_col2im(_ptr(dcols), _ptr(dx), n, hi, wi, _stream(dcols)), "afan_conv_stem7_col2im")
    return dx


def conv_stem7_fwd(x, w):
    """The 7x7 / stride 2 image stem: x [N,3,H,W] bf16 channels-last, w [64,3,7,7] bf16 in KRSC memory."""
    lib = _lib.load()
    CALLS["conv_fwd"] += 1
    _cl4(x, "x")
    _need(w, "w", torch.bfloat16)
    if not (w.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError("stem weight must be in KRSC (channels_last) memory order")
    n, _, hi, wi = x.shape
    y = torch.empty((n, 64, (hi - 1) // 2 + 1, (wi - 1) // 2 + 1), dtype=torch.bfloat16, device=x.device,
                    memory_format=torch.channels_last)
    check(lib.afan_conv_stem7_fwd_nhwc_bf16(_ptr(x), _ptr(w), _ptr(y), n, hi, wi, _stream(x)), "afan_conv_stem7_fwd_nhwc_bf16")
    return y


def conv_stem7_wgrad(x, dy, grad=None, accumulate=False):
    lib = _lib.load()
    CALLS["conv_wgrad"] += 1
    _cl4(x, "x"), _cl4(dy, "dy")
    n, _, hi, wi = x.shape
    if grad is None:
        grad = torch.empty((64, 3, 7, 7), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        accumulate = False
    _need(grad, "grad", torch.float32)
    if not grad.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("grad must be in KRSC (channels_last) memory order")
    ws = _workspace(x, lib.afan_conv_stem7_wgrad_workspace_floats(n, hi, wi), "stem7")
    check(lib.afan_conv_stem7_wgrad_nhwc_bf16(_ptr(x), _ptr(dy), _ptr(grad), n, hi, wi, _ptr(ws), int(bool(accumulate)),
                                              _stream(x)), "afan_conv_stem7_wgrad_nhwc_bf16")
    return grad


# ------------------------------------------------------------------------------------------- SGD
def sgd_step_(param, grad, momentum_buf, lr_dev, momentum, weight_decay, grad_scale=1.0, shadow=None):
    lib = _lib.load()
    _need(param, "param", torch.float32)
    _need(grad, "grad", torch.float32)
    _need(momentum_buf, "momentum_buf", torch.float32)
    _need(lr_dev, "lr_dev", torch.float32)
    if shadow is not None:
        _need(shadow, "shadow", torch.bfloat16)
    if GRID_BN_ALLOWED_AT_IMPORT and param.is_cuda:
        # a step whose in-launch BatchNorm gave up (partial batch totals) must not reach the weights: the update is skipped on the
        # device while the grid barrier's error word is set (sticky until the host has noticed: grid_guard.GridGuard)
        check(lib.afan_sgd_step_guarded(_ptr(param), _ptr(grad), _ptr(momentum_buf), _ptr(shadow), param.numel(),
                                        _ptr(lr_dev), float(momentum), float(weight_decay), float(grad_scale), 0,
                                        _ptr(grid_guard_word(param.device)), _stream(param)), "afan_sgd_step_guarded")
        return
    check(lib.afan_sgd_step(_ptr(param), _ptr(grad), _ptr(momentum_buf), _ptr(shadow), param.numel(),
                            _ptr(lr_dev), float(momentum), float(weight_decay), float(grad_scale), 0,
                            _stream(param)), "afan_sgd_step")


def guarded_copy_(dst, src, counter=None):
    """dst.copy_(src) on the device unless the grid barrier's error word is set (afan_guarded_copy); counter (1-element int32,
    optional) += 1 when the copy happened.  Flat, same dtype and size, a multiple of 16 bytes."""
    lib = _lib.load()
    if dst.dtype != src.dtype or dst.numel() != src.numel() or not (dst.is_contiguous() and src.is_contiguous()):
        raise ValueError("guarded_copy_: two contiguous tensors of one dtype and size")
    check(lib.afan_guarded_copy(_ptr(dst), _ptr(src), dst.numel() * dst.element_size(), _ptr(grid_guard_word(dst.device)),
                                _ptr(counter), _stream(dst)), "afan_guarded_copy")
    return dst


def occupy_cus(workgroups, lds_bytes, microseconds, stream=None):
    """Test / probe aid: park `workgroups` one-wave workgroups with `lds_bytes` of LDS each on `stream` (default: the current one) for
    `microseconds` (afan_occupy_cus)."""
    st = stream if stream is not None else torch.cuda.current_stream()
    check(_lib.load().afan_occupy_cus(int(workgroups), int(lds_bytes), int(microseconds), C.c_void_p(st.cuda_stream)), "afan_occupy_cus")


def normalize_nchw(x, mean, std, out_dtype=torch.float32, channels_last=False):
    """(x - mean[c]) / std[c] of an NCHW fp32 image batch; output optionally bf16 and/or channels_last."""
    lib = _lib.load()
    _need(x, "x", torch.float32)
    if layout_of(x) != AFAN_NCHW:
        x = x.contiguous()
    n, c, hw = _nchw(x)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device,
                    memory_format=torch.channels_last if (channels_last and x.dim() == 4) else torch.contiguous_format)
    check(lib.afan_normalize_nchw(_ptr(x), _ptr(y), _DT[out_dtype], layout_of(y), n, c, hw, _ptr(mean), _ptr(std),
                                  _stream(x)), "afan_normalize_nchw")
    return y


# ------------------------------------------------------------------------------------- measurement
def profile_enable(on=True):
    check(_lib.load().afan_profile_enable(int(bool(on))), "afan_profile_enable")


def profile_event_overhead(n=256):
    """Microseconds an empty (event, event) bracket reads on the current stream: the per-launch bias of profile_collect()'s durations."""
    out = C.c_float(0.0)
    check(_lib.load().afan_profile_event_overhead(int(n), C.byref(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "afan_profile_event_overhead")
    return float(out.value)


def profile_collect(max_kernels=96):
    """{kernel: {"launches", "ms", "bytes"}} for the launches recorded since profile_enable(True)."""
    lib = _lib.load()
    names = C.create_string_buffer(64 * max_kernels)
    launches = (C.c_int64 * max_kernels)()
    ms = (C.c_double * max_kernels)()
    nbytes = (C.c_double * max_kernels)()
    nflops = (C.c_double * max_kernels)()
    k = lib.afan_profile_collect(names, launches, ms, nbytes, nflops, max_kernels)
    out = {}
    for i in range(k):
        name = names.raw[64 * i:64 * (i + 1)].split(b"\0", 1)[0].decode()
        out[name] = {"launches": int(launches[i]), "ms": float(ms[i]), "bytes": float(nbytes[i]),
                     "flops": float(nflops[i])}
    return out
