"""The reference's tensor-operation forms of the Faster-RCNN training targets and losses (Detection/bbox.py:41-92,
rpn/region_proposal_network.py:58-105,163-185, model.py:256-282,343-367, extension/functional.py:6-10), restated with torch
operations: the checker for the library's single-launch forms (cv_a-fan_amd/det_ops.py box_assign / fg_bg_sample /
per_image_losses / box_decode_clip).  Test infrastructure: nothing in the package imports this."""
import torch
import torch.nn.functional as F


def centre(b):
    return torch.stack([(b[..., 0] + b[..., 2]) / 2, (b[..., 1] + b[..., 3]) / 2, b[..., 2] - b[..., 0], b[..., 3] - b[..., 1]], dim=-1)


def corners(c):
    return torch.stack([c[..., 0] - c[..., 2] / 2, c[..., 1] - c[..., 3] / 2, c[..., 0] + c[..., 2] / 2, c[..., 1] + c[..., 3] / 2], dim=-1)


def box_deltas(src, dst):
    """bbox.py:41-52 `calc_transformer`."""
    s, d = centre(src), centre(dst)
    return torch.stack([(d[..., 0] - s[..., 0]) / s[..., 2], (d[..., 1] - s[..., 1]) / s[..., 3],
                        torch.log(d[..., 2] / s[..., 2]), torch.log(d[..., 3] / s[..., 3])], dim=-1)


def box_apply(src, t):
    """bbox.py:54-64 `apply_transformer`."""
    s = centre(src)
    return corners(torch.stack([t[..., 0] * s[..., 2] + s[..., 0], t[..., 1] * s[..., 3] + s[..., 1],
                                torch.exp(t[..., 2]) * s[..., 2], torch.exp(t[..., 3]) * s[..., 3]], dim=-1))


def box_clip(b, right, bottom):
    """bbox.py:89-92 (left = top = 0)."""
    b = b.clone()
    b[..., [0, 2]] = b[..., [0, 2]].clamp(min=0, max=right)
    b[..., [1, 3]] = b[..., [1, 3]].clamp(min=0, max=bottom)
    return b


def box_iou(a, b):
    """bbox.py:66-82: [B, Na, 4] x [B, Nb, 4] -> [B, Na, Nb] (no +1: continuous coordinates)."""
    a, b = a.unsqueeze(-2), b.unsqueeze(-3)
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    w = torch.clamp(torch.min(a[..., 2], b[..., 2]) - torch.max(a[..., 0], b[..., 0]), min=0)
    h = torch.clamp(torch.min(a[..., 3], b[..., 3]) - torch.max(a[..., 1], b[..., 1]), min=0)
    inter = w * h
    return inter / (area_a + area_b - inter)


def anchor_labels(in_boxes, gt):
    """region_proposal_network.py:66-82."""
    b = in_boxes.shape[0]
    labels = torch.full((b, in_boxes.shape[1]), -1, dtype=torch.long, device=in_boxes.device)
    ious = box_iou(in_boxes, gt)
    anchor_max, anchor_assign = ious.max(dim=2)
    gt_max, _ = ious.max(dim=1)
    idx = ((ious > 0) & (ious == gt_max.unsqueeze(dim=1))).nonzero()[:, :2].unbind(dim=1)
    labels[anchor_max < 0.3] = 0
    labels[idx] = 1
    labels[anchor_max >= 0.7] = 1
    return labels, anchor_assign


def proposal_labels(proposals, gt, gt_classes):
    """model.py:256-264."""
    b = proposals.shape[0]
    labels = torch.full((b, proposals.shape[1]), -1, dtype=torch.long, device=proposals.device)
    max_ious, assign = box_iou(proposals, gt).max(dim=2)
    labels[max_ious < 0.5] = 0
    fg = (max_ious >= 0.5).nonzero().unbind(dim=1)
    labels[fg] = gt_classes[fg[0], assign[fg]]
    return labels, assign


def fg_bg_sample(labels, n_fg, n_total):
    """region_proposal_network.py:84-90, model.py:277-282: three draws on the default (host) generator."""
    fg, bg = (labels > 0).nonzero(), (labels == 0).nonzero()
    fg = fg[torch.randperm(len(fg))[:min(len(fg), n_fg)]]
    bg = bg[torch.randperm(len(bg))[:n_total - len(fg)]]
    sel = torch.cat([fg, bg], dim=0)
    return sel[torch.randperm(len(sel))].unbind(dim=1)


def beta_smooth_l1(inp, target, beta):
    """extension/functional.py:6-10."""
    d = torch.abs(inp - target)
    return torch.where(d < beta, 0.5 * d ** 2 / beta, d - 0.5 * beta).sum() / (inp.numel() + 1e-8)


def per_image_losses(logits, deltas, gt_labels, gt_deltas, batch_size, batch_indices, beta):
    """region_proposal_network.py:163-185 == model.py:343-367, with the reference's nonzero() index lists."""
    ce = torch.empty(batch_size, dtype=torch.float, device=logits.device)
    sl1 = torch.empty(batch_size, dtype=torch.float, device=logits.device)
    ces, sls = [], []
    for bi in range(batch_size):
        idx = (batch_indices == bi).nonzero().view(-1)
        ces.append(F.cross_entropy(input=logits[idx], target=gt_labels[idx]))
        fg = gt_labels[idx].nonzero().view(-1)
        sls.append(beta_smooth_l1(deltas[idx][fg], gt_deltas[idx][fg], beta))
    del ce, sl1
    return torch.stack(ces), torch.stack(sls)
