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
            loss = l1.mean() + l2.mean() + l3.mean() + l4.mean()
            grad = torch.autograd.grad(loss, xin, only_inputs=True)[0]
        if grad.stride() != x_adv.stride():
            grad = grad.contiguous(memory_format=torch.channels_last) if (x_adv.dim() == 4 and not x_adv.is_contiguous()) else grad.contiguous()
        ops.pgd_step_(x_adv, grad, gamma, x, eps if eps is not None else 0.0, clip)
    return x_adv.requires_grad_(True)
