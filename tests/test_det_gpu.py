"""Detection operators on the GPU through the C-ABI: NMS against the reference's own golden (9770 -> 1934 boxes,
Detection/test/nms/test_nms.py) and the C oracle; ROIAlign forward against the outputs of the reference's own CPU kernel
(tests/golden/roi_align_fwd_*.npz, incl. BASELINE configs[4] shapes) and the C oracle, backward against the C oracle (the pinned
adjoint) and by the adjoint identity at full size, both layouts."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden, oracle_roi, ptr, reference_roialign

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


@pytest.mark.parametrize("n", [1, 63, 64, 65, 128, 129, 1000, 4097, 12000])
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


def test_nms_thresholds_on_exact_ties(pkg, gpu, c_oracle):
    """The mask kernel decides IoU > t without the division unless inter is within 2e-6 of t * union: small integer boxes give
    thousands of pairs whose quotient IS the threshold (or its fp32 neighbour), where only the reference's division
    (nms.cu:36-49) separates > from >=."""
    rng = np.random.default_rng(5)
    n = 700
    xy = rng.integers(0, 24, (n, 2)).astype(np.float32)
    wh = rng.integers(0, 12, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], axis=1)
    scores = rng.permutation(n).astype(np.float32) / n
    for thr in (0.5, 0.25, 1 / 3, 2 / 3, 0.2, 0.75, 0.6):
        for inclusive in (0, 1):
            ref = _oracle_nms(c_oracle, boxes, scores, np.float32(thr), inclusive)
            got = pkg.det_ops.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(scores).to(gpu), float(np.float32(thr)), inclusive=bool(inclusive))
            assert got.cpu().tolist() == ref.tolist(), (thr, inclusive)
    # and the two modes really differ on this set (ties exist)
    a = pkg.det_ops.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(scores).to(gpu), 0.5, inclusive=False)
    b = pkg.det_ops.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(scores).to(gpu), 0.5, inclusive=True)
    assert a.numel() > b.numel()


def test_nms_more_column_blocks_than_the_default_lds(pkg, gpu, c_oracle):
    """66 000 boxes = 1 032 column blocks: removed[] (8 KB +) beside the scan's 53 KB ring passes the 64 KB a kernel gets by
    default (afan_nms raises the limit for its scan kernel); scattered boxes keep the oracle's greedy loop short."""
    rng = np.random.default_rng(9)
    n = 66000
    xy = rng.uniform(0, 1800, (n, 2)).astype(np.float32)
    wh = rng.uniform(20, 160, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], axis=1)
    scores = rng.permutation(n).astype(np.float32) / n
    ref = _oracle_nms(c_oracle, boxes, scores, 0.3, 0)
    got = pkg.det_ops.nms(torch.from_numpy(boxes).to(gpu), torch.from_numpy(scores).to(gpu), 0.3)
    assert 1000 < len(ref) < n and got.cpu().tolist() == ref.tolist()


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
@pytest.mark.parametrize("C", [24, 192])        # (192 channels-last: the geometry-once-per-bin kernels, both directions)
def test_roi_align_vs_c_oracle(pkg, gpu, c_oracle, nhwc, sr, C):
    rng = np.random.default_rng(3)
    N, H, W = 2, 38, 57                                                  # a 600 x 901 image at stride 16 (rpn docstring)
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


def _hip_roi(pkg, gpu, x, rois, ph, pw, scale, sr, nhwc, grad=False):
    xt = torch.from_numpy(x).to(gpu)
    xt = xt.contiguous(memory_format=torch.channels_last) if nhwc else xt
    if grad:
        xt.requires_grad_(True)
    return xt, pkg.det_ops.roi_align(xt, torch.from_numpy(rois).to(gpu), (ph, pw), scale, sr)


@pytest.mark.parametrize("nhwc", [False, True])
def test_roi_align_fwd_matches_reference_kernel_small(pkg, gpu, orc, nhwc):
    """The reference's own CPU kernel's outputs (fp32): adaptive and fixed sampling grids, 14 x 14 and 7 x 5 bins, boxes over
    every border / malformed / outside.  <= 1e-5 absolute (the values are O(1); measured: bit-identical or 1 ulp)."""
    g = golden("roi_align_fwd_small")
    x = orc.synth_field(tuple(g["x_shape"]), int(g["x_seed"]))
    scale = float(g["scale"])
    for key, ph, pw, sr in (("y_sr0", 14, 14, 0), ("y_sr2", 14, 14, 2), ("y_7x5_sr0", 7, 5, 0)):
        _, y = _hip_roi(pkg, gpu, x, g["rois"], ph, pw, scale, sr, nhwc)
        np.testing.assert_allclose(y.cpu().numpy(), g[key], rtol=0, atol=1e-5, err_msg=key)


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("case", ["cfg5_r128", "cfg5_r300"])
def test_roi_align_fwd_matches_reference_kernel_cfg5(pkg, gpu, orc, case, nhwc):
    """BASELINE configs[4] shapes (2 x 1024 x 38 x 57, 128 / 300 ROIs, 14 x 14, scale 1/16, sampling_ratio 0): the stored channels
    of the reference kernel's output <= 1e-5, its per-ROI and per-channel f64 checksums over the complete output to 1e-6
    relative of the absolute mass; then the backward at this size by the adjoint identity against that pinned forward."""
    g = golden("roi_align_fwd_" + case)
    x = orc.synth_field(tuple(g["x_shape"]), int(g["x_seed"]))
    ph, pw = (int(v) for v in g["pooled"])
    xt, y = _hip_roi(pkg, gpu, x, g["rois"], ph, pw, float(g["scale"]), 0, nhwc, grad=True)
    yd = y.detach()
    np.testing.assert_allclose(yd[:, torch.from_numpy(g["channels"]).to(gpu)].cpu().numpy(), g["y_sub"], rtol=0, atol=1e-5)
    mass = float(yd.double().abs().sum())
    np.testing.assert_allclose(yd.double().sum(dim=(1, 2, 3)).cpu().numpy(), g["roi_sums"], rtol=0, atol=1e-6 * mass / len(g["rois"]))
    np.testing.assert_allclose(yd.double().sum(dim=(0, 2, 3)).cpu().numpy(), g["chan_sums"], rtol=0, atol=1e-6 * mass / x.shape[1])
    gen = torch.Generator().manual_seed(9)
    dy = torch.randn(y.shape, generator=gen).to(gpu)
    dy = dy.contiguous(memory_format=torch.channels_last) if nhwc else dy
    y.backward(dy)
    lhs = float((xt.grad.double() * xt.detach().double()).sum())
    rhs = float((dy.double() * yd.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * float((dy.double() * yd.double()).abs().sum()), (lhs, rhs)
    # bwd(dy) is independent of x: a second field, its forward through the C oracle's ... library forward again
    x2 = torch.randn(xt.shape, generator=gen).to(gpu)
    x2 = x2.contiguous(memory_format=torch.channels_last) if nhwc else x2
    y2 = pkg.det_ops.roi_align(x2, torch.from_numpy(g["rois"]).to(gpu), (ph, pw), float(g["scale"]), 0)
    assert abs(float((xt.grad.double() * x2.double()).sum()) - float((dy.double() * y2.double()).sum())) \
        <= 1e-5 * float((dy.double() * y2.double()).abs().sum())


def test_roi_align_vs_reference_binary_random(pkg, gpu):
    """When the reference's compiled CPU kernel travelled with the snapshot (oracle/_ref/libref_roialign.so): fresh random boxes
    and fields, straight against it."""
    ref = reference_roialign()
    if ref is None:
        pytest.skip("oracle/_ref/libref_roialign.so not present")
    rng = np.random.default_rng(21)
    for (N, C, H, W, R, ph, pw, sr) in ((2, 40, 38, 57, 64, 14, 14, 0), (1, 256, 24, 32, 100, 14, 14, 0), (3, 8, 9, 7, 30, 7, 7, 3)):
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        xy = np.sort(rng.uniform(-32, 16 * max(H, W) + 32, (R, 2, 2)), axis=1)
        rois = np.concatenate([rng.integers(0, N, (R, 1)), xy[:, 0, :], xy[:, 1, :]], axis=1).astype(np.float32)
        want = ref(x, rois, ph, pw, 1 / 16, sr)
        for nhwc in (False, True):
            _, y = _hip_roi(pkg, gpu, x, rois, ph, pw, 1 / 16, sr, nhwc)
            np.testing.assert_allclose(y.cpu().numpy(), want, rtol=0, atol=1e-5)


def test_roi_align_bwd_gather_forms_are_deterministic_and_equal(pkg, gpu, orc):
    """The channels-last backward is a gather by map tile (no atomics): run to run bit-identical, and the two forms of it —
    weights built by the waves (afan_roi_align_bwd) or read from the tables one launch fills (afan_roi_align_bwd_ws, what
    det_ops.roi_align's backward calls) — give the same bits; against the scatter-with-atomics kernel (NCHW path) to rounding."""
    import ctypes as C
    lib = pkg._lib.load()
    g = golden("roi_align_fwd_cfg5_r128")
    rois = torch.from_numpy(g["rois"]).to(gpu)
    N, Cc, H, W = (int(v) for v in g["x_shape"])
    gen = torch.Generator().manual_seed(4)
    for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        dy = torch.randn(len(g["rois"]), 14, 14, Cc, generator=gen).to(gpu).to(dt)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ws = torch.empty(lib.afan_roi_align_bwd_workspace_bytes(len(g["rois"]), H, W), dtype=torch.uint8, device=gpu)
        outs = []
        for w in (None, ws, ws, None):
            dx = torch.full((N, H, W, Cc), float("nan"), device=gpu)         # the gather writes every cell itself: no memset needed
            pkg._lib.check(lib.afan_roi_align_bwd_ws(C.c_void_p(dy.data_ptr()), C.c_void_p(rois.data_ptr()), C.c_void_p(dx.data_ptr()), code, 1,
                                                     len(g["rois"]), N, Cc, H, W, 14, 14, 1 / 16, 0, None if w is None else C.c_void_p(w.data_ptr()), st), "bwd")
            outs.append(dx)
        assert all(torch.equal(outs[0], o) for o in outs[1:]) and torch.isfinite(outs[0]).all()
        # the NCHW scatter on the same values
        dy_nchw = dy.permute(0, 3, 1, 2).contiguous()
        dxs = torch.empty(N, Cc, H, W, device=gpu)
        pkg._lib.check(lib.afan_roi_align_bwd(C.c_void_p(dy_nchw.data_ptr()), C.c_void_p(rois.data_ptr()), C.c_void_p(dxs.data_ptr()), code, 0,
                                              len(g["rois"]), N, Cc, H, W, 14, 14, 1 / 16, 0, st), "bwd nchw")
        ref = dxs.permute(0, 2, 3, 1)
        assert float((outs[0] - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("C", [64, 256])         # (256: 32 channel vectors — the geometry-once-per-bin kernels)
def test_roi_align_bf16_nhwc(pkg, gpu, c_oracle, C):
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.standard_normal((1, C, 20, 30)).astype(np.float32)).bfloat16()
    rois = np.array([[0, 16, 32, 208, 160], [0, 100, 40, 400, 300], [0, -20, 5, 90, 500]], np.float32)
    ref = _oracle_roi(c_oracle, x.float().numpy(), rois, 7, 7, 1 / 16, 0)
    xt = x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = pkg.det_ops.roi_align(xt, torch.from_numpy(rois).to(gpu), 7, 1 / 16, 0)
    assert y.dtype == torch.bfloat16
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref, rtol=1e-2, atol=1e-2)
    dy = torch.from_numpy(rng.standard_normal(ref.shape).astype(np.float32)).bfloat16()
    y.backward(dy.to(gpu).contiguous(memory_format=torch.channels_last))
    dref = _oracle_roi(c_oracle, x.float().numpy(), rois, 7, 7, 1 / 16, 0, dy=dy.float().numpy())
    np.testing.assert_allclose(xt.grad.float().cpu().numpy(), dref, rtol=2e-2, atol=2e-2)      # (the gradient is stored in bf16)


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


@pytest.mark.parametrize("case", ["det_step_tiny_s1", "det_step_tiny_s3"])
def test_detection_step_matches_reference_functions(pkg, orc, gpu, case):
    """det_attack_algo.det_train_step (image PGD with random start, three one-step feature PGDs, fused sample points + mix,
    ROI-feature PGD + mix, eight forwards, weighted loss, SGD) on the GPU against ONE iteration of the reference's own
    Detection/attack_algo.py functions driven through train_aug_sat_muti_advt.py:70-172 (oracle/gen_golden.py
    gen_detection) on oracle.TinyDetNet — here with the library's ROIAlign operator inside the model."""
    g = golden(case)
    torch.manual_seed(11)
    model = orc.TinyDetNet(roi_align=pkg.det_ops.roi_align)
    ck0 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in model.state_dict().values()])
    np.testing.assert_allclose(ck0, g["ck0"], rtol=1e-12)
    model.to(gpu).train()
    opt = torch.optim.SGD(model.parameters(), 0.01, momentum=0.9, weight_decay=5e-4)
    images = torch.rand(2, 3, 32, 32)                   # leaves the CPU generator where the reference's randinit found it
    np.testing.assert_array_equal(images.numpy(), g["images"])
    r = pkg.det_attack_algo.det_train_step(model, opt, images.to(gpu), torch.from_numpy(g["bboxes"]).to(gpu),
                                           torch.from_numpy(g["labels"]).to(gpu), loss_settings=int(g["loss_settings"]))
    np.testing.assert_allclose(r["losses"].cpu().numpy(), g["losses"], rtol=1e-4)
    assert abs(float(r["loss"]) - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"])))
    # perturbations: identical except where sign() flips on a gradient within rounding of zero; the image PGD is clipped to
    # eps = 2/255 and clamped to [0, 1], the feature PGDs take one step of size gamma
    for k, gamma in (("adv1", 0.001 / 255), ("adv2", 0.001 / 255), ("adv3", 1.0 / 255)):
        d_got, d_ref = r[k].cpu().numpy(), g[k]
        frac = float((np.abs(d_got - d_ref) > 0.5 * gamma).mean())
        assert frac <= 2e-2, (k, frac)
    np.testing.assert_allclose(r["adv_image"].cpu().numpy(), g["adv_image"], rtol=0, atol=2 * 2.0 / 255 + 1e-6)
    assert float((np.abs(r["adv_image"].cpu().numpy() - g["adv_image"]) > 1e-6).mean()) <= 5e-2
    assert float(r["adv_image"].min()) >= 0.0 and float(r["adv_image"].max()) <= 1.0
    np.testing.assert_allclose(r["adv_sd"].cpu().numpy(), g["adv_sd"], rtol=1e-3, atol=1e-3)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("sd1/"):
            np.testing.assert_allclose(sd[k[4:]].cpu().numpy(), g[k], rtol=2e-3, atol=2e-5, err_msg=k)


def test_detection_roi_pgd_clip_error_and_rpn_branch(pkg, orc, gpu):
    """rpn_roi_PGD keeps the reference's behaviour at its edges: clip=True raises the NameError of :110 after the first
    step; loss_settings outside 1-4 asserts; the result of the ROI PGD lies on the sign grid."""
    torch.manual_seed(11)
    model = orc.TinyDetNet(roi_align=pkg.det_ops.roi_align).to(gpu).train()
    images = torch.rand(2, 3, 32, 32, device=gpu)
    bb = torch.tensor([[[2., 3., 20., 18.], [10., 12., 30., 31.]], [[0., 0., 15., 15.], [8., 4., 28., 22.]]], device=gpu)
    lb = torch.tensor([[1, 2], [3, 0]], device=gpu)
    y = {"bb": bb, "lb": lb}
    rr = model.train().forward({"x": images, "adv": None, "out_idx": "roi_head", "flag": "clean"}, bb, lb)
    clean = rr["roi_output_dict"]["roi_feature_map"].detach().clone()
    out = pkg.det_attack_algo.rpn_roi_PGD(rpn_roi_output_dict=rr, y=y, model=model, steps=2, eps=1.0, gamma=0.25)
    k = ((out["roi_output_dict"]["roi_feature_map"].detach() - clean) / 0.25).round()
    assert set(k.unique().tolist()) <= {-2.0, 0.0, 2.0} and out["roi_output_dict"]["roi_feature_map"].requires_grad
    rr = model.train().forward({"x": images, "adv": None, "out_idx": "roi_head", "flag": "clean"}, bb, lb)
    with pytest.raises(NameError):
        pkg.det_attack_algo.rpn_roi_PGD(rpn_roi_output_dict=rr, y=y, model=model, steps=1, eps=1.0, gamma=0.25, clip=True)
    with pytest.raises(AssertionError):
        pkg.det_attack_algo.det_train_step(model, torch.optim.SGD(model.parameters(), 0.01), images, bb, lb, loss_settings=7)


def test_nms_padded_form_needs_no_host_read(pkg, gpu):
    """nms(padded=True): (keep [n], count [1]) on the device — the first count entries equal the reference-signature result."""
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(3000, 2, generator=g) * 400
    wh = torch.rand(3000, 2, generator=g) * 60 + 4
    boxes = torch.cat([xy, xy + wh], dim=1).to(gpu)
    scores = torch.rand(3000, generator=g).to(gpu)
    ref = pkg.det_ops.nms(boxes, scores, 0.5)
    keep, count = pkg.det_ops.nms(boxes, scores, 0.5, padded=True)
    assert keep.is_cuda and count.is_cuda and keep.shape == (3000,) and count.shape == (1,)
    assert int(count) == ref.numel() and torch.equal(keep[:int(count)], ref)


@pytest.mark.parametrize("n,spread", [(12000, 900.0), (5000, 250.0), (700, 100.0)])
def test_nms_top_stops_at_the_first_survivors(pkg, gpu, n, spread):
    """nms(max_keep=k) on boxes in score order (the proposal layer's call, region_proposal_network.py:88-93 followed by
    [:post_nms_top_n]): the first k survivors are those of the full scan, and the scan really stopped (fewer survivors
    reported than the full scan finds) whenever the full scan finds more than k + 63."""
    g = torch.Generator().manual_seed(n)
    xy = torch.rand(n, 2, generator=g) * spread
    wh = torch.rand(n, 2, generator=g) * 60 + 4
    boxes = torch.cat([xy, xy + wh], dim=1).to(gpu)
    scores = torch.sort(torch.rand(n, generator=g), descending=True)[0].to(gpu)
    full = pkg.det_ops.nms(boxes, scores, 0.7)
    for k in (1, 50, 300, 2000):
        top = pkg.det_ops.nms(boxes, scores, 0.7, max_keep=k)
        assert torch.equal(pkg.det_ops.nms(boxes, scores, 0.7, max_keep=k, presorted=True), top)     # (scores ARE sorted here)
        m = min(k, full.numel())
        assert torch.equal(top[:m], full[:m]), (n, k)
        assert top.numel() <= max(full.numel(), 0) and (top.numel() < k + 64 or full.numel() < k + 64), (n, k, top.numel(), full.numel())
        if full.numel() >= k:
            assert top.numel() >= k


def test_padded_nms_of_an_empty_set_is_a_pair_on_the_device(pkg, gpu):
    """det_ops.nms(padded=True) answers (keep, count) for every input, the empty set included (RegionProposalNetwork unpacks it)."""
    keep, count = pkg.det_ops.nms(torch.zeros(0, 4, device=gpu), torch.zeros(0, device=gpu), 0.7, padded=True)
    assert keep.is_cuda and count.is_cuda and keep.numel() == 0 and keep.dtype == torch.int64 and int(count.item()) == 0
    assert pkg.det_ops.nms(torch.zeros(0, 4, device=gpu), torch.zeros(0, device=gpu), 0.7).numel() == 0
