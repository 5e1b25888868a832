"""Detection operators, CPU leg: the C oracle's NMS against the reference's own golden (Detection/test/nms/nms-large-*.npy,
kept as data under tests/golden/) and the small cases of Detection/test/nms/test_nms.py; the ROIAlign restatement against
closed-form cases (the reference holds no vector for it: parity unpinned, see oracle/afan_oracle.c)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ptr


def _nms(lib, boxes, scores, thr, inclusive):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    order = np.argsort(-scores, kind="stable").astype(np.int64)
    keep = np.zeros(len(boxes), dtype=np.int64)
    scratch = np.zeros(max(len(boxes), 1), dtype=np.uint8)
    k = lib.oracle_nms(ptr(boxes), ptr(order), len(boxes), thr, inclusive, ptr(keep), ptr(scratch))
    return keep[:k]


@pytest.mark.parametrize("inclusive", [0, 1])
def test_nms_oracle_reproduces_reference_golden(c_oracle, inclusive):
    det = np.load(os.path.join(GOLDEN, "det_nms_large_input.npy"))
    expect = np.load(os.path.join(GOLDEN, "det_nms_large_output.npy"))
    assert det.shape == (9770, 5) and expect.shape == (1934,)
    keep = _nms(c_oracle, det[:, :4], det[:, 4], 0.7, inclusive)         # test_nms.py:15,39-52
    assert len(keep) == 1934
    assert sorted(keep.tolist()) == sorted(expect.tolist())


def test_nms_oracle_small_cases(c_oracle):
    """test_nms.py:21-37"""
    assert len(_nms(c_oracle, np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.7, 0)) == 0
    assert _nms(c_oracle, np.array([[5, 5, 10, 10]], np.float32), np.array([0.8], np.float32), 0.7, 0).tolist() == [0]
    b = np.array([[5, 5, 10, 10], [5, 5, 10, 10], [5, 5, 30, 30]], np.float32)
    assert _nms(c_oracle, b, np.array([0.6, 0.9, 0.4], np.float32), 0.7, 0).tolist() == [1, 2]
    # the one place the two reference paths differ: a pair at exactly the threshold (IoU = 0.5)
    b = np.array([[0, 0, 9, 9], [0, 0, 9, 19]], np.float32)              # areas 100 and 200, intersection 100
    s = np.array([0.9, 0.8], np.float32)
    assert _nms(c_oracle, b, s, 0.5, 0).tolist() == [0, 1]                # nms.cu:49      IoU > thr  : kept
    assert _nms(c_oracle, b, s, 0.5, 1).tolist() == [0]                   # nms_cpu.cpp:62 IoU >= thr : suppressed


def _roi(lib, x, rois, ph, pw, scale, sr, mode=0, dy=None):
    n_roi, (N, Cc, H, W) = len(rois), x.shape
    if mode == 0:
        y = np.zeros((n_roi, Cc, ph, pw), np.float32)
        xx = np.ascontiguousarray(x, np.float32)
        lib.oracle_roi_align(ptr(xx), ptr(np.ascontiguousarray(rois, np.float32)), ptr(y), n_roi, Cc, H, W, ph, pw, scale, sr, 0)
        return y
    dx = np.zeros(x.shape, np.float32)
    dyc = np.ascontiguousarray(dy, np.float32)
    lib.oracle_roi_align(ptr(dx), ptr(np.ascontiguousarray(rois, np.float32)), ptr(dyc), n_roi, Cc, H, W, ph, pw, scale, sr, 1)
    return dx


def test_roi_align_oracle_closed_forms(c_oracle):
    H, W = 20, 30
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    lin = (0.5 * xx - 0.25 * yy + 3.0)[None, None]                          # bilinear interpolation is exact on a linear map
    x = np.concatenate([np.full((1, 1, H, W), 2.5, np.float32), lin], axis=1)
    rois = np.array([[0, 16.0, 32.0, 208.0, 160.0], [0, 40.0, 40.0, 41.0, 41.0]], np.float32)   # image coords, scale 1/16
    for sr in (2, 0):
        y = _roi(c_oracle, x, rois, 7, 7, 1 / 16, sr)
        np.testing.assert_allclose(y[:, 0], 2.5, rtol=1e-6)                # constant map -> constant
        # bin (ph, pw) of ROI 0 averages symmetric sample points: the value of the linear map at the bin centre
        sw, sh, bw, bh = 1.0, 2.0, 12.0 / 7, 8.0 / 7
        cy = sh + (np.arange(7) + 0.5) * bh
        cx = sw + (np.arange(7) + 0.5) * bw
        np.testing.assert_allclose(y[0, 1], 0.5 * cx[None, :] - 0.25 * cy[:, None] + 3.0, rtol=1e-5)
    # malformed (sub-pixel) ROI is forced to 1 x 1 (ROIAlign_cuda.cu:92-93): bins of 1/7 around (2.5, 2.5)
    y = _roi(c_oracle, x, rois, 7, 7, 1 / 16, 0)
    np.testing.assert_allclose(y[1, 1, 3, 3], 0.5 * (2.5 + 0.5) - 0.25 * (2.5 + 0.5) + 3.0, rtol=1e-5)
    # backward is the adjoint of forward: <dy, fwd(x)> == <bwd(dy), x>
    rng = np.random.default_rng(0)
    xr = rng.standard_normal((2, 3, H, W)).astype(np.float32)
    rois2 = np.array([[1, 5.0, 9.0, 300.0, 200.0], [0, 100.0, 50.0, 470.0, 310.0], [1, -20.0, -8.0, 60.0, 40.0]], np.float32)
    yf = _roi(c_oracle, xr, rois2, 5, 6, 1 / 16, 0)
    dy = rng.standard_normal(yf.shape).astype(np.float32)
    dx = _roi(c_oracle, xr, rois2, 5, 6, 1 / 16, 0, mode=1, dy=dy)
    assert abs(float((dy * yf).sum()) - float((dx * xr).sum())) <= 1e-3 * abs(float((dy * yf).sum()))
