"""Detection operators on the GPU through the C-ABI: NMS against the reference's own golden (9770 -> 1934 boxes,
Detection/test/nms/test_nms.py) and the C oracle; ROIAlign forward / backward against the C oracle, both layouts."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ptr

pytestmark = pytest.mark.gpu


def _oracle_nms(lib, boxes, scores, thr, inclusive):
    order = np.argsort(-scores, kind="stable").astype(np.int64)
    keep = np.zeros(len(boxes), dtype=np.int64)
    scratch = np.zeros(max(len(boxes), 1), dtype=np.uint8)
    k = lib.oracle_nms(ptr(np.ascontiguousarray(boxes, np.float32)), ptr(order), len(boxes), thr, inclusive, ptr(keep), ptr(scratch))
    return keep[:k]


def test_nms_reference_golden(pkg, gpu):
    """Detection/test/nms/test_nms.py:39-52"""
    det = torch.from_numpy(np.load(os.path.join(GOLDEN, "det_nms_large_input.npy"))).to(gpu)
    expect = np.load(os.path.join(GOLDEN, "det_nms_large_output.npy"))
    for inclusive in (False, True):
        keep = pkg.det_ops.nms(det[:, 0:4], det[:, 4], 0.7, inclusive=inclusive)
        assert len(keep) == 1934 and keep.dtype == torch.int64
        assert keep.cpu().tolist() == sorted(expect.tolist())               # ascending original indices


def test_nms_small_cases(pkg, gpu):
    """Detection/test/nms/test_nms.py:21-37"""
    nms = pkg.det_ops.nms
    assert len(nms(torch.tensor([], dtype=torch.float, device=gpu), torch.tensor([], dtype=torch.float, device=gpu), 0.7)) == 0
    assert nms(torch.tensor([[5, 5, 10, 10]], dtype=torch.float, device=gpu), torch.tensor([0.8], device=gpu), 0.7).tolist() == [0]
    b = torch.tensor([[5, 5, 10, 10], [5, 5, 10, 10], [5, 5, 30, 30]], dtype=torch.float, device=gpu)
    assert nms(b, torch.tensor([0.6, 0.9, 0.4], device=gpu), 0.7).tolist() == [1, 2]
    b = torch.tensor([[0, 0, 9, 9], [0, 0, 9, 19]], dtype=torch.float, device=gpu)     # IoU exactly 0.5
    s = torch.tensor([0.9, 0.8], device=gpu)
    assert nms(b, s, 0.5).tolist() == [0, 1] and nms(b, s, 0.5, inclusive=True).tolist() == [0]
    with pytest.raises(pkg.AfanLibraryError):
        nms(b.cpu(), s.cpu(), 0.5)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097])
def test_nms_random_vs_c_oracle(pkg, gpu, c_oracle, n):
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 400, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 120, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], axis=1)
    scores = rng.permutation(n).astype(np.float32) / n                    # distinct scores: the order is unambiguous
    for thr, inclusive in ((0.5, 0), (0.3, 1), (0.7, 0)):
        ref = _oracle_nms(c_oracle, boxes, scores, thr, inclusive)
        got = pkg.det_ops.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(scores).to(gpu), thr, inclusive=bool(inclusive))
        assert got.cpu().tolist() == ref.tolist(), (n, thr, inclusive)


def _oracle_roi(lib, x, rois, ph, pw, scale, sr, dy=None):
    n_roi, (N, C, H, W) = len(rois), x.shape
    r = np.ascontiguousarray(rois, np.float32)
    if dy is None:
        y = np.zeros((n_roi, C, ph, pw), np.float32)
        lib.oracle_roi_align(ptr(np.ascontiguousarray(x, np.float32)), ptr(r), ptr(y), n_roi, C, H, W, ph, pw, scale, sr, 0)
        return y
    dx = np.zeros(x.shape, np.float32)
    lib.oracle_roi_align(ptr(dx), ptr(r), ptr(np.ascontiguousarray(dy, np.float32)), n_roi, C, H, W, ph, pw, scale, sr, 1)
    return dx


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("sr", [0, 2])
def test_roi_align_vs_c_oracle(pkg, gpu, c_oracle, nhwc, sr):
    rng = np.random.default_rng(3)
    N, C, H, W = 2, 24, 38, 57                                           # a 600 x 901 image at stride 16 (rpn docstring)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = np.array([[0, 10, 20, 300, 220], [1, 0, 0, 900, 599], [1, 450.5, 100.25, 470.75, 130.5], [0, 880, 580, 905, 610],
                     [0, 33, 44, 34, 45], [1, -30, -10, 50, 80]], np.float32)
    ref = _oracle_roi(c_oracle, x, rois, 14, 14, 1 / 16, sr)
    xt = torch.from_numpy(x).to(gpu)
    xt = xt.contiguous(memory_format=torch.channels_last) if nhwc else xt
    xt.requires_grad_(True)
    y = pkg.det_ops.roi_align(xt, torch.from_numpy(rois).to(gpu), (14, 14), 1 / 16, sr)
    assert tuple(y.shape) == ref.shape
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    y.backward(torch.from_numpy(dy).to(gpu))
    dref = _oracle_roi(c_oracle, x, rois, 14, 14, 1 / 16, sr, dy=dy)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), dref, rtol=1e-4, atol=1e-4)      # atomics: summation order differs
    # the reference's pooler on top (roi/pooler.py:35-44)
    p = pkg.det_ops.Pooler.apply(xt.detach(), torch.from_numpy(rois[:, 1:]).to(gpu), torch.from_numpy(rois[:, 0]).long().to(gpu), "align")
    assert tuple(p.shape) == (len(rois), C, 7, 7)


def test_roi_align_bf16_nhwc(pkg, gpu, c_oracle):
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.standard_normal((1, 64, 20, 30)).astype(np.float32)).bfloat16()
    rois = np.array([[0, 16, 32, 208, 160], [0, 100, 40, 400, 300]], np.float32)
    ref = _oracle_roi(c_oracle, x.float().numpy(), rois, 7, 7, 1 / 16, 0)
    y = pkg.det_ops.roi_align(x.to(gpu).contiguous(memory_format=torch.channels_last), torch.from_numpy(rois).to(gpu), 7, 1 / 16, 0)
    assert y.dtype == torch.bfloat16
    np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)


def test_detection_pgd_protocol(pkg, gpu):
    """Detection/attack_algo.py:48-74 on a stand-in model that follows the reference's forward protocol."""
    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(8, 4, 3, padding=1)

        def forward(self, inputs, bb, lb):
            assert inputs["flag"] == "tail" and inputs["out_idx"] == 2
            o = self.conv(inputs["adv"])
            return o[:, 0].mean(dim=(1, 2)), o[:, 1].abs().mean(dim=(1, 2)), (o[:, 2] ** 2).mean(dim=(1, 2)), o[:, 3].mean(dim=(1, 2))
    torch.manual_seed(0)
    m = Toy().to(gpu)
    x = torch.randn(2, 8, 10, 12, device=gpu)
    x0 = x.clone()
    out = pkg.det_ops.PGD(x, None, y={"bb": None, "lb": None}, model=m, steps=3, eps=2 / 255, gamma=0.5 / 255, idx=2, clip=True)
    assert out.requires_grad and out.is_leaf and torch.equal(x, x0)
    k = ((out.detach() - x) / (0.5 / 255)).round()
    assert float(k.abs().max()) <= 3 and float((out.detach() - x).abs().max()) <= 2 / 255 + 1e-7
