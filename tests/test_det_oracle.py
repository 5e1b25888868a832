"""Detection operators, CPU leg: the C oracle's NMS against the reference's own golden (Detection/test/nms/nms-large-*.npy,
kept as data under tests/golden/) and the small cases of Detection/test/nms/test_nms.py; the ROIAlign restatement against the
outputs of the reference's own CPU kernel (tests/golden/roi_align_fwd_*.npz: Detection/support/src/cpu/ROIAlign_cpu.cpp:4-219
compiled unedited at fixture time, oracle/gen_golden.py gen_roi_align) bit for bit, its backward by the adjoint identity in f64
(the reference has no CPU backward, Detection/support/src/ROIAlign.h:44), and closed-form cases."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden, oracle_roi, ptr, reference_roialign


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
    return oracle_roi(lib, np.asarray(x, np.float32), rois, ph, pw, scale, sr, mode=mode, dy=dy)


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


# ------------------------------------------------------------------------------------ ROIAlign pinned to the reference's CPU kernel
def test_roi_align_oracle_equals_reference_kernel_small(c_oracle):
    """fp32 and f64, adaptive (sampling_ratio 0, the value roi/pooler.py:36 passes) and fixed grids, square and 7 x 5 bins; boxes
    crossing every border, malformed, whole-image and fully-outside ones: bit for bit."""
    g = golden("roi_align_fwd_small")
    from oracle import afan_oracle as orc
    x = orc.synth_field(tuple(g["x_shape"]), int(g["x_seed"]))
    rois, (ph, pw), scale = g["rois"], g["pooled"], float(g["scale"])
    assert (rois[:, 1] < 0).any() and (rois[:, 3] > x.shape[3] * 16).any() and (rois[:, 3] - rois[:, 1] < 16).any()
    for sr in (0, 2):
        np.testing.assert_array_equal(oracle_roi(c_oracle, x, rois, ph, pw, scale, sr), g[f"y_sr{sr}"])
        y64 = oracle_roi(c_oracle, x.astype(np.float64), rois.astype(np.float64), ph, pw, scale, sr)
        np.testing.assert_array_equal(y64, g[f"y64_sr{sr}"])
        assert np.abs(g[f"y_sr{sr}"] - g[f"y64_sr{sr}"]).max() <= 2e-5          # the two precisions of the reference agree (fp32 sample coordinates ~30: ulp 2e-6 x field slope)
    np.testing.assert_array_equal(oracle_roi(c_oracle, x, rois, 7, 5, scale, 0), g["y_7x5_sr0"])


@pytest.mark.parametrize("case", ["cfg5_r128", "cfg5_r300"])
def test_roi_align_oracle_equals_reference_kernel_cfg5(c_oracle, orc, case):
    """BASELINE configs[4] shapes (C=1024, 38 x 57, 14 x 14, scale 1/16, sampling_ratio 0): four stored channels bit for bit and
    the f64 checksums of the complete fp32 output per ROI / per channel."""
    g = golden("roi_align_fwd_" + case)
    x = orc.synth_field(tuple(g["x_shape"]), int(g["x_seed"]))
    (ph, pw), scale, ch = g["pooled"], float(g["scale"]), g["channels"]
    y = oracle_roi(c_oracle, x, g["rois"], ph, pw, scale, 0)
    np.testing.assert_array_equal(y[:, ch], g["y_sub"])
    np.testing.assert_array_equal(y.astype(np.float64).sum(axis=(1, 2, 3)), g["roi_sums"])
    np.testing.assert_array_equal(y.astype(np.float64).sum(axis=(0, 2, 3)), g["chan_sums"])


def test_roi_align_backward_is_the_adjoint_of_the_pinned_forward(c_oracle, orc):
    """<bwd(dy), x> == <dy, fwd(x)> in f64, fwd = the reference kernel's stored f64 output (a linear map of x): pins the
    backward's samples and weights (ROIAlign_cuda.cu:125-170,178-254) to the pinned forward.  Then the fp32 backward against the
    f64 one."""
    g = golden("roi_align_fwd_small")
    x = orc.synth_field(tuple(g["x_shape"]), int(g["x_seed"])).astype(np.float64)
    rois, (ph, pw), scale = g["rois"].astype(np.float64), g["pooled"], float(g["scale"])
    rng = np.random.default_rng(5)
    for sr in (0, 2):
        yf = g[f"y64_sr{sr}"]
        dy = rng.standard_normal(yf.shape)
        dx = oracle_roi(c_oracle, x, rois, ph, pw, scale, sr, mode=1, dy=dy)
        lhs, rhs = float((dx * x).sum()), float((dy * yf).sum())
        assert abs(lhs - rhs) <= 1e-12 * max(abs(rhs), float(np.abs(dy * yf).sum())), (lhs, rhs)
        # the adjoint identity for a second, independent x: bwd(dy) does not depend on x, fwd is linear
        x2 = rng.standard_normal(x.shape)
        y2 = oracle_roi(c_oracle, x2, rois, ph, pw, scale, sr)
        assert abs(float((dx * x2).sum()) - float((dy * y2).sum())) <= 1e-12 * float(np.abs(dy * y2).sum())
        # every basis direction of dy for one ROI that crosses the border: the complete matrix transposes
        dx32 = oracle_roi(c_oracle, x.astype(np.float32), rois.astype(np.float32), ph, pw, scale, sr, mode=1, dy=dy.astype(np.float32))
        np.testing.assert_allclose(dx32, dx, rtol=0, atol=2e-5 * np.abs(dx).max())


def test_roi_align_backward_matrix_is_the_transpose(c_oracle):
    """The full Jacobian on a tiny case: column j of fwd (x = e_j) and row j of bwd (dy = e_i) are the same matrix, f64."""
    N, C, H, W, ph, pw = 1, 1, 5, 6, 3, 2
    rois = np.array([[0, -10.0, 4.0, 70.0, 60.0], [0, 20.0, 20.0, 21.0, 21.0], [0, 0.0, 0.0, 96.0, 80.0]], np.float64)
    n_in, n_out = H * W, len(rois) * ph * pw
    J = np.zeros((n_out, n_in))
    for j in range(n_in):
        e = np.zeros((N, C, H, W)); e.reshape(-1)[j] = 1.0
        J[:, j] = oracle_roi(c_oracle, e, rois, ph, pw, 1 / 16, 0).reshape(-1)
    Jt = np.zeros((n_in, n_out))
    for i in range(n_out):
        e = np.zeros((len(rois), C, ph, pw)); e.reshape(-1)[i] = 1.0
        Jt[:, i] = oracle_roi(c_oracle, np.zeros((N, C, H, W)), rois, ph, pw, 1 / 16, 0, mode=1, dy=e).reshape(-1)
    np.testing.assert_allclose(Jt, J.T, rtol=0, atol=1e-15)
    assert np.abs(J).sum() > 0


def test_reference_roialign_binary_reproduces_its_vectors(orc):
    """When oracle/_ref/libref_roialign.so is present (built from /root/reference by oracle/Makefile; it travels to the GPU box),
    it must reproduce the committed vectors — i.e. the fixtures are what the reference computes, not a stale copy."""
    ref = reference_roialign()
    if ref is None:
        pytest.skip("oracle/_ref/libref_roialign.so not built (no /root/reference on this machine)")
    g = golden("roi_align_fwd_small")
    x = orc.synth_field(tuple(g["x_shape"]), int(g["x_seed"]))
    np.testing.assert_array_equal(ref(x, g["rois"], 14, 14, 1 / 16, 0), g["y_sr0"])
    np.testing.assert_array_equal(ref(x.astype(np.float64), g["rois"].astype(np.float64), 14, 14, 1 / 16, 2), g["y64_sr2"])
    # and a fresh random case against the C restatement, so the pin is not only the stored boxes
    rng = np.random.default_rng(11)
    xr = rng.standard_normal((3, 5, 17, 23)).astype(np.float32)
    rois = np.concatenate([rng.integers(0, 3, (64, 1)).astype(np.float32), np.sort(rng.uniform(-40, 400, (64, 2, 2)), axis=1)
                           .transpose(0, 1, 2).reshape(64, 4)[:, [0, 1, 2, 3]].astype(np.float32)], axis=1)
    rois[:, [1, 2, 3, 4]] = rois[:, [1, 3, 2, 4]]                         # (x1, y1, x2, y2) with x1 <= x2, y1 <= y2
    from conftest import _c_oracle
    np.testing.assert_array_equal(ref(xr, rois, 7, 7, 1 / 16, 0), oracle_roi(_c_oracle(), xr, rois, 7, 7, 1 / 16, 0))
