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


def test_roi_align_torch_restatement_equals_c_oracle(orc, c_oracle):
    """oracle.roi_align_torch (the differentiable restatement TinyDetNet uses on the CPU) against oracle_roi_align (C):
    forward values and, through autograd, the input gradient."""
    import torch
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 5, 9, 11)).astype(np.float32)
    rois = np.array([[0, 1.0, 2.0, 30.0, 25.0], [1, 0.0, 0.0, 43.0, 35.0], [1, 10.5, 3.25, 12.0, 4.0], [0, -6.0, -3.0, 20.0, 50.0]],
                    dtype=np.float32)
    for sr in (2, 0):
        xt = torch.from_numpy(x).requires_grad_(True)
        out = orc.roi_align_torch(xt, torch.from_numpy(rois), (3, 2), 0.25, sr)
        ref = _roi(c_oracle, x, rois, 3, 2, 0.25, sr)
        np.testing.assert_allclose(out.detach().numpy(), ref, rtol=1e-5, atol=1e-6)
        dy = rng.standard_normal(out.shape).astype(np.float32)
        out.backward(torch.from_numpy(dy))
        np.testing.assert_allclose(xt.grad.numpy(), _roi(c_oracle, x, rois, 3, 2, 0.25, sr, mode=1, dy=dy), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", ["det_step_tiny_s1", "det_step_tiny_s3"])
def test_detection_step_matches_reference_functions(orc, case):
    """oracle.det_train_step == the loop body of Detection/train_aug_sat_muti_advt.py:70-172 run with the reference's own
    attack_algo functions (oracle/gen_golden.py gen_detection) on oracle.TinyDetNet."""
    import torch
    from conftest import golden
    g = golden(case)
    torch.manual_seed(11)
    model = orc.TinyDetNet()
    model.train()
    ck0 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in model.state_dict().values()])
    np.testing.assert_allclose(ck0, g["ck0"], rtol=1e-12)
    opt = torch.optim.SGD(model.parameters(), 0.01, momentum=0.9, weight_decay=5e-4)
    images = torch.rand(2, 3, 32, 32)
    np.testing.assert_array_equal(images.numpy(), g["images"])          # same generator state as the reference run had
    r = orc.det_train_step(model, opt, images, torch.from_numpy(g["bboxes"]), torch.from_numpy(g["labels"]),
                           loss_settings=int(g["loss_settings"]))
    np.testing.assert_allclose(r["losses"].numpy(), g["losses"], rtol=1e-6)
    np.testing.assert_allclose(float(r["loss"]), float(g["loss"]), rtol=1e-6)
    for k in ("adv_image", "adv1", "adv2", "adv3", "adv_sd"):
        np.testing.assert_allclose(r[k].numpy(), g[k], rtol=0, atol=1e-7, err_msg=k)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("sd1/"):
            np.testing.assert_allclose(sd[k[4:]].numpy(), g[k], rtol=1e-5, atol=1e-7, err_msg=k)
    assert float(sd["layer2.1.running_mean"].abs().max()) == 0.0      # BatchNorm stays frozen (model.py:46-47)
