"""Single-launch training targets / losses of the Faster-RCNN step (afan_det_targets.hip through det_ops) against the
reference's tensor-operation forms (tests/det_torch_ref.py): boxes, IoU decisions, labels, samples and regression targets bit for
bit; the two loss sums and their gradients to 1e-6 (another summation order)."""
import pytest
import torch

import det_torch_ref as ref

pytestmark = pytest.mark.gpu


def _boxes(g, *shape, span=600.0, lo=8.0, hi=300.0):
    xy = torch.rand(*shape, 2, generator=g) * span
    wh = torch.rand(*shape, 2, generator=g) * (hi - lo) + lo
    return torch.cat([xy, xy + wh], dim=-1)


def test_box_decode_clip_equals_the_tensor_operations(pkg, gpu):
    g = torch.Generator().manual_seed(0)
    for n in (1, 255, 19494):
        anchors = _boxes(g, 2, n).to(gpu)
        t = (torch.randn(2, n, 4, generator=g) * 0.5).to(gpu)
        got = pkg.det_ops.box_decode_clip(anchors, t, 904, 600)
        want = ref.box_clip(ref.box_apply(anchors, t), 904, 600)
        assert torch.equal(got, want)


@pytest.mark.parametrize("B,N,G", [(1, 17000, 6), (3, 2500, 4), (2, 300, 1), (1, 128, 40)])
def test_anchor_labels_equal_the_tensor_operations(pkg, gpu, B, N, G):
    g = torch.Generator().manual_seed(B * 1000 + N + G)
    gt = _boxes(g, B, G, span=500.0, lo=40.0, hi=260.0)
    boxes = _boxes(g, B, N)
    boxes[:, :G] = gt + torch.randn(B, G, 4, generator=g) * 3        # a few anchors close to a ground truth (labels 1 by IoU >= 0.7)
    boxes[:, G:2 * G] = gt                                           # and exact copies (IoU 1: the ties of :76-79)
    if B > 1:
        gt[1, -1] = 0                                                # zero padding of a shorter image (dataset collate)
    boxes, gt = boxes.to(gpu), gt.to(gpu)
    labels, assign = pkg.det_ops.box_assign(boxes, gt, "anchor", 0.3, 0.7)
    want_l, want_a = ref.anchor_labels(boxes, gt)
    assert torch.equal(labels, want_l) and torch.equal(assign, want_a)
    assert (labels == 1).sum() >= G and (labels == 0).sum() > 0 and (labels == -1).sum() > 0


@pytest.mark.parametrize("B,N,G", [(1, 2000, 6), (3, 700, 5)])
def test_proposal_labels_equal_the_tensor_operations(pkg, gpu, B, N, G):
    g = torch.Generator().manual_seed(N + G)
    gt = _boxes(g, B, G, span=500.0, lo=40.0, hi=260.0)
    cls = torch.randint(1, 21, (B, G), generator=g)
    boxes = _boxes(g, B, N)
    boxes[:, :4 * G] = gt.repeat(1, 4, 1) + torch.randn(B, 4 * G, 4, generator=g) * 8
    boxes[:, -50:] = 0                                               # generate_proposals' zero padding: IoU 0 / 0 against ...
    if B > 1:
        gt[2, -2:] = 0                                               # ... zero-padded ground truth is nan: label -1
        cls[2, -2:] = 0
    boxes, gt, cls = boxes.to(gpu), gt.to(gpu), cls.to(gpu)
    labels, assign = pkg.det_ops.box_assign(boxes, gt, "proposal", 0.5, gt_classes=cls)
    want_l, want_a = ref.proposal_labels(boxes, gt, cls)
    assert torch.equal(labels, want_l)
    ok = labels >= 0                                                 # (where every IoU is nan, max's index is whatever the vendor's is)
    assert torch.equal(assign[ok], want_a[ok])
    assert (labels > 0).sum() >= G and (labels == 0).sum() > 0
    if B > 1:
        assert (labels[2, -50:] == -1).all()


@pytest.mark.parametrize("n_fg,n_total", [(128, 256), (32, 128), (4, 16)])
def test_sampling_draws_and_gathers_like_the_reference(pkg, gpu, n_fg, n_total):
    g = torch.Generator().manual_seed(n_total)
    B, N, G = 2, 3000, 5
    gt = _boxes(g, B, G, span=500.0, lo=40.0, hi=260.0)
    boxes = _boxes(g, B, N)
    boxes[:, :20 * G] = gt.repeat(1, 20, 1) + torch.randn(B, 20 * G, 4, generator=g) * 4
    boxes, gt = boxes.to(gpu), gt.to(gpu)
    labels, assign = pkg.det_ops.box_assign(boxes, gt, "anchor", 0.3, 0.7)
    torch.manual_seed(77)
    sel, sb, sl, sd, bi = pkg.det_ops.fg_bg_sample(labels, assign, boxes, gt, n_fg, n_total)
    after = torch.rand(1)
    torch.manual_seed(77)
    want = ref.fg_bg_sample(labels, n_fg, n_total)
    assert torch.equal(after, torch.rand(1))                         # the host generator moved by the same three draws
    assert sel.numel() == n_total and torch.equal(sel, want[0] * N + want[1]) and torch.equal(bi, want[0])
    assert torch.equal(sb, boxes[want]) and torch.equal(sl, labels[want])
    assert torch.equal(sd, ref.box_deltas(boxes[want], gt[want[0], assign[want]]))
    assert (sl > 0).sum() == min(n_fg, int((labels > 0).sum()))


@pytest.mark.parametrize("C,own_class", [(2, False), (21, True)])
@pytest.mark.parametrize("B", [1, 3])
def test_per_image_losses_and_gradients(pkg, gpu, C, own_class, B):
    g = torch.Generator().manual_seed(C + B)
    R, S, beta = 900, 256, 1.0
    K = C if own_class else 1
    logits = torch.randn(R, C, generator=g).to(gpu).requires_grad_(True)
    deltas = (torch.randn(R, K * 4, generator=g) * 1.5).to(gpu).requires_grad_(True)
    rows = torch.randperm(R, generator=g)[:S].to(gpu)
    lab = torch.randint(0, C, (S,), generator=g)
    lab[torch.rand(S, generator=g) < 0.6] = 0
    bi = torch.randint(0, B, (S,), generator=g)
    if B == 3:
        lab[bi == 1] = 0                                             # an image without foreground: smooth-L1 0
    lab, bi = lab.to(gpu), bi.to(gpu)
    gt_d = (torch.randn(S, 4, generator=g) * 0.3).to(gpu)
    norm = (0., 0., 0., 0., .1, .1, .2, .2) if own_class else None
    ce, sl1 = pkg.det_ops.per_image_losses(logits, deltas, rows, lab, gt_d, bi, B, beta, norm=norm)
    w = torch.rand(2, B, generator=g).to(gpu) + 0.5
    (ce * w[0]).sum().add((sl1 * w[1]).sum()).backward()
    got = (ce.detach(), sl1.detach(), logits.grad.clone(), deltas.grad.clone())
    logits.grad = deltas.grad = None
    d_in = deltas[rows].view(S, K, 4)[torch.arange(S), lab if own_class else torch.zeros_like(lab)]
    tgt = (gt_d - torch.tensor(norm[:4], device=gpu)) / torch.tensor(norm[4:], device=gpu) if norm else gt_d
    rce, rsl = ref.per_image_losses(logits[rows], d_in, lab, tgt, B, bi, beta)
    (rce * w[0]).sum().add((rsl * w[1]).sum()).backward()
    for a, b_, what in zip(got, (rce.detach(), rsl.detach(), logits.grad, deltas.grad), ("ce", "sl1", "d logits", "d deltas")):
        assert torch.allclose(a, b_, rtol=1e-6, atol=1e-7), (what, (a - b_).abs().max().item())
    if B == 3:
        assert sl1[1].item() == 0.0


def test_per_image_losses_of_an_image_without_samples_is_nan(pkg, gpu):
    logits = torch.randn(10, 2, device=gpu)
    deltas = torch.randn(10, 4, device=gpu)
    lab = torch.tensor([0, 1, 1, 0], device=gpu)
    bi = torch.tensor([0, 0, 2, 2], device=gpu)
    ce, sl1 = pkg.det_ops.per_image_losses(logits, deltas, torch.arange(4, device=gpu), lab, torch.zeros(4, 4, device=gpu), bi, 3, 1.0)
    assert torch.isnan(ce[1]) and sl1[1].item() == 0.0 and torch.isfinite(ce[[0, 2]]).all()


@pytest.mark.parametrize("ns,counts,P", [((5000,), (1300,), 2000), ((5000, 4000), (2063, 17), 2000), ((90, 300), (90, 0), 128), ((0, 40), (0, 40), 64)])
def test_padded_proposals_equal_the_reference_stack(pkg, gpu, ns, counts, P):
    """region_proposal_network.py:255-270 with the counts on the device: the kept rows, the zero padding, and the -1 labels beyond the
    longest image."""
    g = torch.Generator().manual_seed(sum(ns) + P)
    cands, keeps, want_rows = [], [], []
    for n, c in zip(ns, counts):
        sb = _boxes(g, n).to(gpu) if n else torch.zeros(0, 4, device=gpu)
        k = torch.sort(torch.randperm(n, generator=g)[:c])[0] if n else torch.zeros(0, dtype=torch.int64)
        keep = torch.cat([k, torch.full((n - c,), 7 if n else 0, dtype=torch.int64)]).to(gpu)        # (entries behind the count are junk)
        cands.append(sb)
        keeps.append((keep, torch.tensor([c], device=gpu)))
        want_rows.append(sb[k.to(gpu)][:P])
    padded, kept = pkg.det_ops.proposal_rows(cands, keeps, P)
    assert kept.tolist() == [min(c, P) for c in counts]
    longest = max(len(r) for r in want_rows)
    want = torch.stack([torch.cat([r, torch.zeros(longest - len(r), 4).to(r)]) for r in want_rows])      # :259-270
    assert torch.equal(padded[:, :longest], want) and not padded[:, longest:].any()
    labels = torch.randint(0, 3, (len(ns), P), generator=g).to(gpu)
    got = pkg.det_ops.labels_limit_(labels.clone(), kept)
    assert torch.equal(got[:, :longest], labels[:, :longest]) and (got[:, longest:] == -1).all()


def test_targets_need_the_gpu(pkg):
    with pytest.raises(pkg.AfanLibraryError):
        pkg.det_ops.box_decode_clip(torch.zeros(3, 4), torch.zeros(3, 4), 10, 10)
