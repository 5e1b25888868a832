"""Detection operators of the A-FAN Detection step, MI355X-native — same names, argument meaning
and return contract as the reference's extension layer:

    nms(bboxes, scores, threshold) -> LongTensor of kept indices, ascending      Detection/support/layer/nms.py (-> _C.nms)
    roi_align(input, rois, output_size, spatial_scale, sampling_ratio)           Detection/support/layer/roi_align.py:11-48
    ROIAlign(output_size, spatial_scale, sampling_ratio)(input, rois)            ... :51-64
    Pooler.apply(features, proposal_bboxes, proposal_batch_indices, mode)        Detection/roi/pooler.py:10-44
    PGD(x, image_batch, y, model, steps, eps, gamma, idx, randinit, clip)        Detection/attack_algo.py:48-74

The kernels live in libafan_hip.so (afan_det.hip); there is no CPU fallback.  The Faster-RCNN model itself (model.py, rpn/,
roi/) is not part of this slice: `PGD` drives any module that follows the reference's protocol
`model.train().forward({'x','adv','out_idx','flag'}, bboxes, labels) -> 4 loss tensors`."""
import ctypes as C
import os

import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from . import _lib, ops
from ._lib import check
from .attack_algo import get_sample_points, linfball_proj, mix_feature, tensor_clamp  # noqa: F401 (Detection/attack_algo.py:236-265)
from .resnet_s import dgrad_only

_ws = {}


def _workspace_bytes(dev, nbytes):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = _ws[key] = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=dev)
    return w


_arange_cache = {}


def nms(bboxes, scores, threshold, inclusive=False, padded=False, max_keep=0, presorted=False):
    """Kept indices (ascending) of greedy NMS over boxes in descending-score order; IoU with +1 (inclusive corners).
    inclusive=False suppresses at IoU > threshold (the reference's GPU path, nms.cu:49), True at >= (its CPU path).
    padded=True: no host synchronisation — returns (keep [n] int64 on the device, of which the first count[0] are valid,
    count [1] int64 on the device) for callers that can consume a padded result.
    max_keep > 0 (afan_nms_top): the caller passes boxes ALREADY in score order and looks at the first max_keep survivors
    only — the scan stops once it has them (the result may hold up to 63 more); its first max_keep entries are nms()'s.
    presorted=True: the caller vouches that `scores` is already descending (the proposal layer sorts before it slices the
    top pre-NMS candidates): the sort of nms.cu:73-75 is skipped, the order is 0..n-1."""
    if bboxes.device.type != "cuda":
        raise ops.AfanLibraryError("nms: tensors must live on the MI355X (no CPU path in this build)")
    n = bboxes.shape[0] if bboxes.dim() > 0 else 0
    if bboxes.numel() == 0:
        if padded:                                         # the padded contract holds for an empty set too: (keep, count) on the device
            return (torch.empty(0, dtype=torch.int64, device=bboxes.device), torch.zeros(1, dtype=torch.int64, device=bboxes.device))
        return torch.empty(0, dtype=torch.int64)          # nms.h:17-18 returns an empty CPU tensor
    lib = _lib.load()
    boxes = bboxes.detach().float().contiguous()
    if presorted:
        key = (boxes.device.index, n)
        order = _arange_cache.get(key)
        if order is None:
            order = _arange_cache[key] = torch.arange(n, dtype=torch.int64, device=boxes.device)
    else:
        order = torch.sort(scores.detach().float(), dim=0, descending=True)[1].contiguous()      # nms.cu:73-75
    keep = torch.empty(n, dtype=torch.int64, device=boxes.device)
    count = torch.empty(1, dtype=torch.int64, device=boxes.device)
    ws = _workspace_bytes(boxes.device, lib.afan_nms_workspace_bytes(n))
    st = C.c_void_p(torch.cuda.current_stream(boxes.device).cuda_stream)
    if max_keep and max_keep > 0:
        check(lib.afan_nms_top(C.c_void_p(boxes.data_ptr()), C.c_void_p(order.data_ptr()), n, float(threshold), int(bool(inclusive)),
                               C.c_void_p(ws.data_ptr()), C.c_void_p(keep.data_ptr()), C.c_void_p(count.data_ptr()), int(max_keep), st),
              "afan_nms_top")
    else:
        check(lib.afan_nms(C.c_void_p(boxes.data_ptr()), C.c_void_p(order.data_ptr()), n, float(threshold), int(bool(inclusive)),
                           C.c_void_p(ws.data_ptr()), C.c_void_p(keep.data_ptr()), C.c_void_p(count.data_ptr()), st), "afan_nms")
    if padded:
        return keep, count
    return keep[:int(count.item())]       # the result's length is data dependent: one read-back (the reference copies the whole mask)


# ------------------------------------------------------------------------- training targets and losses (afan_det_targets.hip)
def _f32c(t, what):
    if t.device.type != "cuda":
        raise ops.AfanLibraryError(f"{what}: tensors must live on the MI355X (no CPU path in this build)")
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def box_decode_clip(src, transformers, right, bottom):
    """BBox.clip(BBox.apply_transformer(src, transformers), 0, 0, right, bottom) (bbox.py:54-64, :89-92) in one launch."""
    src, t = _f32c(src, "box_decode_clip"), _f32c(transformers, "box_decode_clip")
    if src.shape != t.shape or src.shape[-1] != 4:
        raise ValueError("box_decode_clip: [..., 4] boxes and transformers of one shape")
    out = torch.empty_like(src)
    check(_lib.load().afan_box_decode_clip(_ptr(src), _ptr(t), _ptr(out), src.numel() // 4, float(right), float(bottom), _stream(src.device)),
          "afan_box_decode_clip")
    return out


def box_assign(boxes, gt_bboxes, mode, lo, hi=0.0, gt_classes=None):
    """(labels [B, N] int64, assign [B, N] int64) of boxes [B, N, 4] against gt [B, G, 4]: mode 'anchor'
    (region_proposal_network.py:66-82, thresholds lo / hi) or 'proposal' (model.py:256-264, threshold lo, classes [B, G])."""
    boxes, gt = _f32c(boxes, "box_assign"), _f32c(gt_bboxes, "box_assign")
    B, N, _ = boxes.shape
    G = gt.shape[1]
    if gt.shape[0] != B or gt.shape[2] != 4 or boxes.shape[2] != 4:
        raise ValueError("box_assign: boxes [B, N, 4], gt [B, G, 4]")
    labels = torch.empty((B, N), dtype=torch.int64, device=boxes.device)
    assign = torch.empty((B, N), dtype=torch.int64, device=boxes.device)
    m = {"anchor": 0, "proposal": 1}[mode]
    cls = None
    if m == 1:
        cls = gt_classes.detach().to(torch.int64).contiguous()
        if tuple(cls.shape) != (B, G):
            raise ValueError("box_assign: gt_classes [B, G]")
    ws = _workspace_bytes(boxes.device, B * G * 4) if m == 0 else None
    check(_lib.load().afan_box_assign(_ptr(boxes), _ptr(gt), B, N, G, m, float(lo), float(hi), _ptr(cls), _ptr(labels), _ptr(assign), _ptr(ws),
                                      _stream(boxes.device)), "afan_box_assign")
    return labels, assign


def proposal_rows(cands, keeps, top_n):
    """(padded [B, top_n, 4] fp32, kept [B] int64) from each image's score-ordered candidates `cands[b]` [n, 4] and its padded NMS
    result `keeps[b]` = (keep [n] int64, count [1] int64, both on the device): the first min(count, top_n) survivors, zero rows
    behind them (region_proposal_network.py:255-270 with the count never read) — one launch per image."""
    lib, dev = _lib.load(), cands[0].device
    B = len(cands)
    padded = torch.empty((B, top_n, 4), dtype=torch.float32, device=dev)
    kept = torch.empty(B, dtype=torch.int64, device=dev)
    for b, (sb, (k, c)) in enumerate(zip(cands, keeps)):
        sb = _f32c(sb, "proposal_rows")
        if k.dtype != torch.int64 or c.dtype != torch.int64 or not k.is_contiguous():
            raise ValueError("proposal_rows: keep / count are contiguous int64 tensors")
        check(lib.afan_proposal_rows(_ptr(sb), sb.shape[0], _ptr(k), k.numel(), _ptr(c), top_n, _ptr(padded[b]), _ptr(kept[b:b + 1]),
                                     _stream(dev)), "afan_proposal_rows")
    return padded, kept


def labels_limit_(labels, kept):
    """labels [B, N] int64 in place: columns >= max(kept) become -1 (kept [B] int64 on the device, never read)."""
    if labels.dtype != torch.int64 or kept.dtype != torch.int64 or not labels.is_contiguous() or labels.dim() != 2:
        raise ValueError("labels_limit_: labels [B, N] int64 contiguous, kept int64")
    check(_lib.load().afan_labels_limit(_ptr(labels), labels.shape[0], labels.shape[1], _ptr(kept), kept.numel(), _stream(labels.device)),
          "afan_labels_limit")
    return labels


def sample_lists(labels):
    """The `nonzero()` lists of labels > 0 / == 0 in one launch, nothing read: (fg [M], bg [M], counts [2]) on the device."""
    lib = _lib.load()
    M = labels.numel()
    lists = torch.empty(2 * M + 2, dtype=torch.int64, device=labels.device)
    fg, bg, counts = lists[:M], lists[M:2 * M], lists[2 * M:]
    check(lib.afan_sample_lists(_ptr(labels), M, _ptr(fg), _ptr(bg), _ptr(counts), _stream(labels.device)), "afan_sample_lists")
    return fg, bg, counts


def fg_bg_draw(lists, nf, nb, labels, assign, boxes, gt_bboxes, n_fg, n_total):
    """The three `torch.randperm` draws on the HOST generator over lists of known lengths (nf, nb), composed on the host into
    one position list, and the gather of the sampled rows (see fg_bg_sample)."""
    lib = _lib.load()
    dev = labels.device
    fg, bg, _ = lists
    N, G = labels.shape[1], gt_bboxes.shape[1]
    boxes, gt = _f32c(boxes, "fg_bg_sample"), _f32c(gt_bboxes, "fg_bg_sample")
    p1 = torch.randperm(nf)[:min(nf, n_fg)]
    p2 = torch.randperm(nb)[:n_total - len(p1)]
    pos = torch.cat([p1, -(p2 + 1)])
    pos = pos[torch.randperm(len(pos))].to(dev, non_blocking=True)
    S = pos.numel()
    sel = torch.empty(S, dtype=torch.int64, device=dev)
    out_l = torch.empty(S, dtype=torch.int64, device=dev)
    out_bi = torch.empty(S, dtype=torch.int64, device=dev)
    out_b = torch.empty((S, 4), dtype=torch.float32, device=dev)
    out_d = torch.empty((S, 4), dtype=torch.float32, device=dev)
    if S:
        check(lib.afan_sample_gather(_ptr(fg), _ptr(bg), _ptr(pos), S, _ptr(boxes), _ptr(gt), _ptr(assign), _ptr(labels), N, G, _ptr(sel),
                                     _ptr(out_b), _ptr(out_l), _ptr(out_d), _ptr(out_bi), _stream(dev)), "afan_sample_gather")
    return sel, out_b, out_l, out_d, out_bi


def fg_bg_sample(labels, assign, boxes, gt_bboxes, n_fg, n_total):
    """The reference's sampling (region_proposal_network.py:84-90, model.py:277-282) and what is gathered at the sample right
    after it: three `torch.randperm` draws on the HOST generator (foreground subset, background subset, shuffle) over the
    `nonzero()` lists of labels > 0 / == 0.  One launch builds both lists (sample_lists), ONE host read brings their lengths, the
    three draws are composed on the host into a position list, one launch gathers (fg_bg_draw).  Returns (sel [S] flat positions
    in [B * N), boxes [S, 4], labels [S], regression targets [S, 4] = calc_transformer(box, its ground truth) (bbox.py:41-52),
    batch indices [S])."""
    lists = sample_lists(labels)
    nf, nb = lists[2].tolist()                                   # the iteration's host read for this sampling
    return fg_bg_draw(lists, nf, nb, labels, assign, boxes, gt_bboxes, n_fg, n_total)


class _DetLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, deltas, rows, gt_labels, gt_deltas, batch, batch_size, beta, norm):
        lib = _lib.load()
        if logits.device.type != "cuda":
            raise ops.AfanLibraryError("per_image_losses: tensors must live on the MI355X (no CPU path in this build)")
        lg = logits if (logits.dtype == torch.float32 and logits.is_contiguous()) else logits.float().contiguous()
        dl = deltas if (deltas.dtype == torch.float32 and deltas.is_contiguous()) else deltas.float().contiguous()
        C_ = lg.shape[-1]
        R = lg.numel() // C_
        K = dl.numel() // (R * 4)
        if K * R * 4 != dl.numel() or K not in (1, C_):
            raise ValueError("per_image_losses: deltas must be [R, 4] or [R, C * 4] for logits [R, C]")
        S, B = gt_labels.numel(), int(batch_size)
        dev = lg.device
        gd = gt_deltas.detach().float().contiguous()
        out = torch.empty(2 * B, dtype=torch.float32, device=dev)
        save = torch.empty(S * 4 + S * C_ + 2 * B, dtype=torch.float32, device=dev)
        nrm = (C.c_float * 8)(*norm) if norm is not None else None
        check(lib.afan_det_loss_fwd(_ptr(lg), _ptr(dl), _ptr(rows), _ptr(gt_labels), _ptr(gd), _ptr(batch), S, B, C_, K, float(beta), nrm,
                                    _ptr(out), C.c_void_p(out.data_ptr() + 4 * B), _ptr(save), _stream(dev)), "afan_det_loss_fwd")
        ctx.save_for_backward(save, rows, gt_labels, batch)
        ctx.dims = (S, B, C_, K, R, logits.shape, deltas.shape, logits.dtype, deltas.dtype)
        return out[:B], out[B:]

    @staticmethod
    def backward(ctx, g_ce, g_sl):
        save, rows, gt_labels, batch = ctx.saved_tensors
        S, B, C_, K, R, lshape, dshape, ldt, ddt = ctx.dims
        dev = save.device
        g_ce, g_sl = g_ce.float().contiguous(), g_sl.float().contiguous()
        nl = R * C_
        both = torch.empty(nl + R * K * 4, dtype=torch.float32, device=dev)      # (one allocation: the library zero-fills both with one launch)
        d_l, d_d = both[:nl].view(lshape), both[nl:].view(dshape)
        check(_lib.load().afan_det_loss_bwd(_ptr(g_ce), _ptr(g_sl), _ptr(save), _ptr(rows), _ptr(gt_labels), _ptr(batch), S, B, C_, K, R, _ptr(d_l),
                                            _ptr(d_d), _stream(dev)), "afan_det_loss_bwd")
        return d_l.to(ldt), d_d.to(ddt), None, None, None, None, None, None, None


class _SumOfMeans1(torch.autograd.Function):
    """`l1.mean() + l2.mean() [+ l3.mean() + l4.mean()]` for loss vectors of ONE element each (one image per GPU): the mean is the
    element, the sum keeps the reference's left-to-right order (afan_sum_scalars_f32: one launch); backward: every loss receives the
    incoming gradient as it is (mean of one element: g / 1) — no launch."""

    @staticmethod
    def forward(ctx, *ls):
        lib = _lib.load()
        out = torch.empty((), dtype=torch.float32, device=ls[0].device)
        p = [C.c_void_p(t.data_ptr()) for t in ls] + [None] * (4 - len(ls))
        check(lib.afan_sum_scalars_f32(p[0], p[1], p[2], p[3], C.c_void_p(out.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)), "afan_sum_scalars_f32")
        ctx.shapes = [t.shape for t in ls]
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(g.reshape(sh) for sh in ctx.shapes)


FUSED_LOSS_SUM = os.environ.get("AFAN_DET_LOSS_SUM", "1") != "0"


def sum_of_means(*ls):
    """The reference's `loss1.mean() + loss2.mean() + ...` (Detection/train_aug_sat_muti_advt.py:21-27): as one launch where every loss
    vector holds one image's value (the per-GPU share of BASELINE configs[4]), the plain expression otherwise (same values)."""
    if (FUSED_LOSS_SUM and 2 <= len(ls) <= 4 and all(t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 and t.is_contiguous() for t in ls)):
        return _SumOfMeans1.apply(*ls)
    out = ls[0].mean()
    for t in ls[1:]:
        out = out + t.mean()
    return out


def per_image_losses(logits, deltas, rows, gt_labels, gt_deltas, batch_indices, batch_size, beta, norm=None):
    """region_proposal_network.py:163-185 == model.py:343-367 in one launch (and one for the backward): per image the mean
    cross-entropy of its samples and the beta-smooth-L1 of its foreground samples.  Sample s reads row rows[s] (None: s) of
    logits [..., C] and deltas [..., 4] or [..., C * 4] (then the 4 of the sample's own class); norm: (mean[4] + std[4]) applied
    to the targets (model.py:352-354).  Returns (cross_entropies [B], smooth_l1_losses [B])."""
    return _DetLoss.apply(logits, deltas, rows, gt_labels, gt_deltas, batch_indices, batch_size, beta, norm)


class _ROIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        lib = _lib.load()
        if input.device.type != "cuda":
            raise ops.AfanLibraryError("roi_align: tensors must live on the MI355X (no CPU path in this build)")
        if input.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("roi_align: fp32 or bf16 feature maps")
        ph, pw = _pair(output_size)
        x = input if (input.is_contiguous() or input.is_contiguous(memory_format=torch.channels_last)) else input.contiguous()
        rois = roi.detach().float().contiguous()
        n, c, h, w = x.shape
        lay = ops.layout_of(x)
        y = torch.empty((rois.shape[0], c, ph, pw), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if lay == ops.AFAN_NHWC else torch.contiguous_format)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        check(lib.afan_roi_align_fwd(C.c_void_p(x.data_ptr()), C.c_void_p(rois.data_ptr()), C.c_void_p(y.data_ptr()),
                                     ops._DT[x.dtype], lay, rois.shape[0], c, h, w, ph, pw, float(spatial_scale),
                                     int(sampling_ratio), st), "afan_roi_align_fwd")
        ctx.save_for_backward(rois)
        ctx.geom = (n, c, h, w, ph, pw, float(spatial_scale), int(sampling_ratio), lay, x.dtype)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        lib = _lib.load()
        (rois,) = ctx.saved_tensors
        n, c, h, w, ph, pw, scale, sr, lay, dtype = ctx.geom
        g = grad_output.to(dtype)
        g = g.contiguous(memory_format=torch.channels_last) if lay == ops.AFAN_NHWC else g.contiguous()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device,
                         memory_format=torch.channels_last if lay == ops.AFAN_NHWC else torch.contiguous_format)
        st = C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
        ws = _workspace_bytes(g.device, lib.afan_roi_align_bwd_workspace_bytes(rois.shape[0], h, w))      # the weight tables of the gather form
        check(lib.afan_roi_align_bwd_ws(C.c_void_p(g.data_ptr()), C.c_void_p(rois.data_ptr()), C.c_void_p(dx.data_ptr()),
                                        ops._DT[dtype], lay, rois.shape[0], n, c, h, w, ph, pw, scale, sr, C.c_void_p(ws.data_ptr()), st),
              "afan_roi_align_bwd_ws")
        return dx.to(dtype), None, None, None, None


roi_align = _ROIAlign.apply


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return (f"{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, "
                f"sampling_ratio={self.sampling_ratio})")


class Pooler(object):
    """Detection/roi/pooler.py:10-44, the 'align' mode: ROIAlign to 14 x 14 at scale 1/16 with adaptive sampling, then a
    2 x 2 / stride 2 max pool.  ('pooling' — adaptive max pooling of rounded boxes — is the reference's CPU-only fallback.)"""
    OPTIONS = ["align"]

    @staticmethod
    def apply(features, proposal_bboxes, proposal_batch_indices, mode="align"):
        if getattr(mode, "value", mode) != "align":
            raise ValueError("only Pooler.Mode.ALIGN is built on the MI355X path")
        rois = torch.cat([proposal_batch_indices.view(-1, 1).float(), proposal_bboxes], dim=1)
        pool = ROIAlign((14, 14), spatial_scale=1 / 16, sampling_ratio=0)(features, rois)
        return nn.functional.max_pool2d(input=pool, kernel_size=2, stride=2)


def PGD(x, image_batch, y=None, model=None, steps=3, eps=None, gamma=None, idx=1, randinit=False, clip=False):
    """Detection/attack_algo.py:48-74: K-step sign-gradient ascent on the backbone feature map `x` under the SUM of the four
    detection losses (each a mean).  Returns a new fp32 leaf with requires_grad=True; `x` is not modified."""
    if x.device.type != "cuda":
        raise ops.AfanLibraryError("PGD: x must live on the MI355X (no CPU path in this build)")
    x = x.detach().float()
    x = x if (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))) else x.contiguous()
    x_adv = x.clone()
    if randinit:
        u = torch.rand(x_adv.shape).to(x.device, non_blocking=True)
        if u.stride() != x_adv.stride():
            u = u.contiguous(memory_format=torch.channels_last)
        ops.axpy_noise_(x_adv, u, eps)
    for _ in range(steps):
        xin = x_adv.detach().requires_grad_(True)
        inputs = {"x": image_batch, "adv": xin, "out_idx": idx, "flag": "tail"}
        with dgrad_only():      # only_inputs=True (:66): the library's layers skip (and must not add into) parameter gradients
            l1, l2, l3, l4 = model.train().forward(inputs, y["bb"], y["lb"])
            loss = sum_of_means(l1, l2, l3, l4)
            grad = torch.autograd.grad(loss, xin, only_inputs=True)[0]
        if grad.stride() != x_adv.stride():
            grad = grad.contiguous(memory_format=torch.channels_last) if (x_adv.dim() == 4 and not x_adv.is_contiguous()) else grad.contiguous()
        ops.pgd_step_(x_adv, grad, gamma, x, eps if eps is not None else 0.0, clip)
    return x_adv.requires_grad_(True)
