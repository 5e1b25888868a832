"""The DeepLabv3+ layer kernels (afan_seg.hip, afan_conv_stem7.hip) and the atrous / ragged-channel convolution paths,
each through the C-ABI against torch's CPU ops on the same values (the arithmetic the reference's network runs:
F.interpolate(bilinear, align_corners=False), nn.CrossEntropyLoss(ignore_index), nn.MaxPool2d(3,2,1),
nn.AdaptiveAvgPool2d(1), nn.Conv2d with dilation, nn.Dropout)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _fmt(t, nhwc):
    return t.contiguous(memory_format=torch.channels_last) if nhwc else t.contiguous()


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("shape,size", [((2, 21, 33, 33), (129, 129)), ((1, 256, 9, 9), (33, 33)), ((2, 8, 1, 1), (33, 33)),
                                        ((1, 5, 17, 23), (40, 31)), ((2, 16, 40, 31), (17, 23)), ((1, 3, 129, 129), (513, 513))])
def test_upsample_bilinear_fp32(pkg, gpu, shape, size, nhwc):
    torch.manual_seed(1)
    x = torch.randn(shape)
    ref = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    got = pkg.ops.upsample_bilinear(_fmt(x.to(gpu), nhwc), size)
    assert got.shape == ref.shape
    # (the interpolation weight is src - floor(src) with src up to ~hi: one ulp of src is ~1e-5 of a weight at 129 -> 513)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6 + 4e-7 * max(shape[2], shape[3]))
    # backward: the adjoint of the same linear map
    g = torch.randn(ref.shape)
    xr = x.clone().requires_grad_(True)
    F.interpolate(xr, size=size, mode="bilinear", align_corners=False).backward(g)
    dx = pkg.ops.upsample_bilinear_backward(_fmt(g.to(gpu), nhwc), shape[2:])
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_upsample_bilinear_bf16_nhwc(pkg, gpu):
    torch.manual_seed(2)
    x = torch.randn(2, 256, 33, 33).bfloat16()
    ref = F.interpolate(x.float(), size=(129, 129), mode="bilinear", align_corners=False)
    got = pkg.ops.upsample_bilinear(_fmt(x.to(gpu), True), (129, 129))
    assert got.dtype == torch.bfloat16 and got.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(got.float().cpu().numpy(), ref.numpy(), rtol=8e-3, atol=8e-3)
    g = torch.randn(2, 256, 129, 129).bfloat16()
    xr = x.float().requires_grad_(True)
    F.interpolate(xr, size=(129, 129), mode="bilinear", align_corners=False).backward(g.float())
    dx = pkg.ops.upsample_bilinear_backward(_fmt(g.to(gpu), True), (33, 33))
    np.testing.assert_allclose(dx.float().cpu().numpy(), xr.grad.numpy(), rtol=1e-2, atol=6e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n,c0,c,lo,hi", [(2, 48, 256, (129, 129), (33, 33)), (1, 8, 16, (40, 31), (17, 23)), (3, 24, 40, (9, 9), (5, 7)),
                                          (1, 4, 6, (7, 7), (3, 3))])
def test_upsample_concat_equals_cat_of_the_resize(pkg, gpu, dtype, n, c0, c, lo, hi):
    """The decoder's concat node (deeplab._UpsampleCatFn: resize written into its channel slice, gradient read from the slice)
    against torch.cat of the library's dense resize (bit for bit) and the dense backward on a copied slice (to rounding)."""
    torch.manual_seed(4)
    low = _fmt(torch.randn(n, c0, *lo, device=gpu).to(dtype), True).requires_grad_(True)
    x = _fmt(torch.randn(n, c, *hi, device=gpu).to(dtype), True).requires_grad_(True)
    out = pkg.deeplab._UpsampleCatFn.apply(low, x)
    ref = torch.cat([low.detach(), pkg.ops.upsample_bilinear(x.detach(), lo)], dim=1)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out, ref)
    g = _fmt(torch.randn(n, c0 + c, *lo, device=gpu).to(dtype), True)
    out.backward(g)
    assert torch.equal(low.grad, g[:, :c0])
    # (the slice entry sums columns, then rows — two separable passes; the dense one sums the products in one pass: rounding)
    want = pkg.ops.upsample_bilinear_backward(g[:, c0:].contiguous(memory_format=torch.channels_last), hi)
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), want.float().cpu().numpy(), rtol=tol, atol=tol)
    xr = x.detach().float().requires_grad_(True)                          # and against torch's own adjoint
    F.interpolate(xr, size=lo, mode="bilinear", align_corners=False).backward(g[:, c0:].float())
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=5 * tol, atol=(6e-2 if dtype == torch.bfloat16 else 1e-5))
    with pytest.raises(TypeError):
        pkg.ops.upsample_concat(low.detach(), x.detach().contiguous())          # NCHW operand: not this entry's case


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("n,c,h,w", [(2, 21, 65, 65), (1, 19, 33, 47), (2, 5, 33, 33), (1, 32, 8, 8)])
def test_ce2d_vs_torch(pkg, gpu, n, c, h, w, nhwc):
    torch.manual_seed(3)
    logits = torch.randn(n, c, h, w) * 3
    target = torch.randint(0, c, (n, h, w))
    target[torch.rand(n, h, w) < 0.07] = 255
    lr = logits.clone().requires_grad_(True)
    ref = nn.CrossEntropyLoss(ignore_index=255)(lr, target)
    ref.backward()
    loss, dl = pkg.ops.ce2d(_fmt(logits.to(gpu), nhwc), target.to(gpu), 255, 1.0)
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    np.testing.assert_allclose(dl.cpu().numpy(), lr.grad.numpy(), rtol=1e-4, atol=1e-8)
    # pre-scaled gradient (the weight of a term in the joint loss)
    _, dl7 = pkg.ops.ce2d(_fmt(logits.to(gpu), nhwc), target.to(gpu), 255, 0.7)
    np.testing.assert_allclose(dl7.cpu().numpy(), 0.7 * lr.grad.numpy(), rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("n,c,h,w,H,W", [(2, 21, 33, 33, 129, 129), (1, 21, 129, 129, 513, 513), (2, 5, 9, 13, 33, 41), (1, 32, 7, 5, 7, 5),
                                         (3, 2, 1, 1, 4, 4)])
def test_ce2d_upsampled_equals_resize_then_ce(pkg, gpu, n, c, h, w, H, W):
    """afan_ce2d_upsampled (resize + cross-entropy + both backward passes in one kernel, gradient at the low resolution)
    against the three-kernel path it replaces — same operations, sums grouped by output tile: equal to rounding — and against torch's CPU F.interpolate + nn.CrossEntropyLoss(ignore_index) with autograd."""
    g = torch.Generator().manual_seed(n * c + h)
    lo = torch.randn(n, c, h, w, generator=g) * 2
    y = torch.randint(0, c, (n, H, W), generator=g)
    y[torch.rand(n, H, W, generator=g) < 0.1] = 255
    lod = lo.to(gpu).contiguous(memory_format=torch.channels_last)
    yd = y.to(gpu)
    loss, dl = pkg.ops.ce2d_upsampled(lod, yd, 255, 0.7)
    up = pkg.ops.upsample_bilinear(lod, (H, W))
    loss3, dup = pkg.ops.ce2d(up, yd, 255, 0.7)
    dl3 = pkg.ops.upsample_bilinear_backward(dup, (h, w))
    # same operations; a source pixel on a tile border sums its (up to four) tiles' parts instead of one running sum
    np.testing.assert_allclose(dl.cpu().numpy(), dl3.cpu().numpy(), rtol=2e-6, atol=1e-9)
    assert abs(float(loss) - float(loss3)) <= 2e-6 * max(1.0, abs(float(loss3)))
    lo64 = lo.double().requires_grad_(True)
    ref = nn.CrossEntropyLoss(ignore_index=255)(F.interpolate(lo64, size=(H, W), mode="bilinear", align_corners=False), y)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    np.testing.assert_allclose(dl.cpu().numpy(), 0.7 * lo64.grad.numpy(), rtol=2e-4, atol=1e-7)
    # through the model-facing callable: a LowResLogits object in, the same loss and gradient out
    crit = pkg.deeplab.seg_criterion(nn.CrossEntropyLoss(ignore_index=255))
    lod2 = lod.clone().requires_grad_(True)
    l2 = crit(pkg.deeplab.LowResLogits(lod2, (H, W)), yd, grad_scale=0.7)
    torch.autograd.backward([l2], [pkg.ops.one(gpu)])
    assert torch.equal(lod2.grad, dl) and float(l2) == float(loss)


def test_ce2d_edge_cases(pkg, gpu):
    logits = torch.randn(1, 4, 3, 3, device=gpu)
    all_ign = torch.full((1, 3, 3), 255, dtype=torch.int64, device=gpu)
    loss, dl = pkg.ops.ce2d(logits, all_ign, 255)
    assert torch.isnan(loss).all() and float(dl.abs().sum()) == 0.0           # torch: 0 / 0 = nan, zero gradient
    bad = torch.zeros((1, 3, 3), dtype=torch.int64, device=gpu)
    bad[0, 1, 1] = 7                                                           # outside [0, C) and not the ignore index
    assert torch.isnan(pkg.ops.ce2d(logits, bad, 255)[0]).all()
    with pytest.raises(TypeError):
        pkg.ops.ce2d(logits, bad.int(), 255)


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 64, 65, 65), (1, 16, 9, 12), (2, 3, 5, 5), (1, 8, 1, 1)])
def test_maxpool_vs_torch_with_ties(pkg, gpu, shape, dtype, nhwc):
    torch.manual_seed(4)
    x = torch.relu(torch.randn(shape)).to(dtype)          # post-ReLU: windows tie at 0 — the gradient must go where ATen's goes
    xr = x.float().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    g = torch.randn(ref.shape).to(dtype)
    ref.backward(g.float())
    y = pkg.ops.maxpool3x3s2(_fmt(x.to(gpu), nhwc))
    np.testing.assert_array_equal(y.float().cpu().numpy(), ref.detach().numpy())
    dx = pkg.ops.maxpool3x3s2_backward(_fmt(g.to(gpu), nhwc), _fmt(x.to(gpu), nhwc))
    tol = dict(rtol=0, atol=0) if dtype == torch.float32 else dict(rtol=8e-3, atol=2e-2)   # bf16: up to 4 addends rounded once
    np.testing.assert_allclose(dx.float().cpu().numpy(), xr.grad.numpy(), **tol)


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_avgpool(pkg, gpu, dtype, nhwc):
    torch.manual_seed(5)
    x = torch.randn(2, 2048, 9, 9).to(dtype)
    y = pkg.ops.avgpool(_fmt(x.to(gpu), nhwc))
    ref = x.float().mean(dim=(2, 3), keepdim=True)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), rtol=1e-5 if dtype == torch.float32 else 8e-3, atol=1e-6 if dtype == torch.float32 else 4e-3)
    y32 = pkg.ops.avgpool(_fmt(x.to(gpu), nhwc), out_fp32=True)          # fp32 pooled side whatever the map's dtype
    assert y32.dtype == torch.float32
    np.testing.assert_allclose(y32.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    g = torch.randn(2, 2048, 1, 1).to(dtype)
    dx = pkg.ops.avgpool_backward(g.to(gpu), _fmt(x.to(gpu), nhwc))
    np.testing.assert_allclose(dx.float().cpu().numpy(), (g.float() / 81).expand(2, 2048, 9, 9).numpy(),
                               rtol=1e-6 if dtype == torch.float32 else 8e-3, atol=1e-7)
    dx32 = pkg.ops.avgpool_backward(g.float().to(gpu), _fmt(x.to(gpu), nhwc))
    assert dx32.dtype == dtype
    np.testing.assert_allclose(dx32.float().cpu().numpy(), (g.float() / 81).expand(2, 2048, 9, 9).numpy(),
                               rtol=1e-6 if dtype == torch.float32 else 8e-3, atol=1e-7)


@pytest.mark.parametrize("n,ci,co", [(2, 2048, 256), (1, 64, 8), (8, 320, 256)])
def test_linear_small(pkg, gpu, n, ci, co):
    torch.manual_seed(12)
    x, w = torch.randn(n, ci), torch.randn(co, ci, 1, 1) * 0.05
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.linear(xr, wr.reshape(co, ci))
    g = torch.randn(n, co)
    ref.backward(g)
    y = pkg.ops.linear_small(x.to(gpu), w.to(gpu))
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    dw = torch.zeros(co, ci, 1, 1, device=gpu)
    dx = pkg.ops.linear_small_backward(g.to(gpu), x.to(gpu), w.to(gpu), True, dw)
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dw.cpu().numpy(), wr.grad.numpy(), rtol=1e-4, atol=1e-5)
    pkg.ops.linear_small_backward(g.to(gpu), x.to(gpu), w.to(gpu), False, dw, accumulate=True)
    np.testing.assert_allclose(dw.cpu().numpy(), 2 * wr.grad.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,ci,co,h,w", [(2, 256, 21, 33, 33), (1, 256, 19, 17, 9), (1, 64, 32, 5, 5)])
def test_pointwise_classifier(pkg, gpu, n, ci, co, h, w, dtype):
    torch.manual_seed(6)
    x = torch.randn(n, ci, h, w).to(dtype)
    wt = torch.randn(co, ci, 1, 1) * 0.1
    b = torch.randn(co)
    xr = x.float().requires_grad_(True)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br)
    g = torch.randn(ref.shape)
    ref.backward(g)
    xg = _fmt(x.to(gpu), True)
    y = pkg.ops.pointwise_forward(xg, wt.to(gpu), b.to(gpu))
    assert y.dtype == torch.float32
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    dw, db = torch.zeros(co, ci, 1, 1, device=gpu), torch.zeros(co, device=gpu)
    dx = pkg.ops.pointwise_backward(_fmt(g.to(gpu), True), xg, wt.to(gpu), True, dw, db, accumulate=False)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(dx.float().cpu().numpy(), xr.grad.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(dw.cpu().numpy(), wr.grad.numpy(), rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol=1e-3)
    # accumulate = 1 adds into the buffers
    pkg.ops.pointwise_backward(_fmt(g.to(gpu), True), xg, wt.to(gpu), False, dw, db, accumulate=True)
    np.testing.assert_allclose(dw.cpu().numpy(), 2 * wr.grad.numpy(), rtol=1e-3, atol=4e-3)


def test_dropout_mask_and_generator(pkg, gpu):
    torch.manual_seed(7)
    x = torch.randn(2, 256, 9, 9, device=gpu)
    mask = (torch.rand(2, 256, 9, 9) > 0.1).to(torch.uint8).to(gpu)
    y, _ = pkg.ops.dropout(x, 0.1, mask)
    np.testing.assert_array_equal(y.cpu().numpy(), (x * mask.float() * (1.0 / (1.0 - 0.1))).cpu().numpy())
    # device generator: ~p dropped, scale 1/(1-p), fresh mask per call, backward re-derives the forward's mask
    big = torch.ones(1 << 20, device=gpu)
    y1, used1 = pkg.ops.dropout(big, 0.1)
    y2, used2 = pkg.ops.dropout(big, 0.1)
    k1, k2 = (y1 != 0), (y2 != 0)
    assert abs(float(k1.float().mean()) - 0.9) < 3e-3 and abs(float(k2.float().mean()) - 0.9) < 3e-3
    assert int(used1) != int(used2) and float((k1 != k2).float().mean()) > 0.1
    np.testing.assert_allclose(y1[k1].cpu().numpy(), 1.0 / 0.9, rtol=1e-6)
    g = torch.full_like(big, 2.0)
    dx, _ = pkg.ops.dropout(g, 0.1, None, used1)
    np.testing.assert_array_equal((dx != 0).cpu().numpy(), k1.cpu().numpy())
    # bf16 tensors take the same path
    yb, _ = pkg.ops.dropout(big.bfloat16(), 0.5)
    assert abs(float((yb != 0).float().mean()) - 0.5) < 3e-3


def _bf(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("n,h,w", [(2, 129, 129), (1, 64, 64), (1, 33, 47), (2, 7, 5)])
def test_stem7_fwd_wgrad(pkg, gpu, n, h, w):
    torch.manual_seed(8)
    x = _bf(torch.randn(n, 3, h, w))
    wt = _bf(torch.randn(64, 3, 7, 7) * 0.1)
    xr, wr = x.clone(), wt.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, 2, 3)
    g = _bf(torch.randn(ref.shape))
    ref.backward(g)
    xg = _fmt(x.to(gpu).bfloat16(), True)
    wg = _fmt(wt.to(gpu).bfloat16(), True)
    y = pkg.ops.conv_stem7_fwd(xg, wg)
    assert tuple(y.shape) == tuple(ref.shape)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.detach().numpy(), rtol=1e-2, atol=2e-2)
    gw = pkg.ops.conv_stem7_wgrad(xg, _fmt(g.to(gpu).bfloat16(), True))
    np.testing.assert_allclose(gw.cpu().numpy(), wr.grad.numpy(), rtol=2e-3, atol=2e-3 * float(wr.grad.abs().max()))
    gw2 = pkg.ops.conv_stem7_wgrad(xg, _fmt(g.to(gpu).bfloat16(), True), gw.clone(), accumulate=True)
    np.testing.assert_allclose(gw2.cpu().numpy(), 2 * wr.grad.numpy(), rtol=2e-3, atol=4e-3 * float(wr.grad.abs().max()))


CONV_CASES = [
    # n, ci, co, h, w, k, stride, dilation
    (2, 256, 256, 33, 33, 3, 1, 2),          # layer4-style atrous
    (2, 512, 256, 17, 17, 3, 1, 6),          # ASPP rate 6
    (1, 2048, 256, 9, 9, 3, 1, 12),          # ASPP rate 12 on a 9x9 map: most taps fall outside
    (1, 512, 256, 33, 33, 3, 1, 18),
    (2, 304, 256, 33, 33, 3, 1, 1),          # decoder: ragged reduction (304 = 4.75 x 64)
    (2, 256, 48, 33, 33, 1, 1, 1),           # low-level projection: ragged output tile
    (2, 64, 256, 33, 33, 1, 1, 1),
    (1, 128, 128, 65, 65, 3, 2, 1),          # odd size, stride 2 (129 -> 65 style)
    (2, 2048, 256, 1, 1, 1, 1, 1),           # ASPP pooling branch: two rows
    (1, 1280, 256, 9, 9, 1, 1, 1),
]


@pytest.mark.parametrize("n,ci,co,h,w,k,stride,dil", CONV_CASES)
def test_conv_atrous_ragged_vs_torch(pkg, gpu, n, ci, co, h, w, k, stride, dil):
    """forward, input gradient and weight gradient of the tiled kernels on the DeepLab shapes against torch's fp32
    convolution of the same bf16 values."""
    torch.manual_seed(9)
    pad = dil * (k // 2)
    x = _bf(torch.randn(n, ci, h, w))
    wt = _bf(torch.randn(co, ci, k, k) * (1.0 / (ci * k * k) ** 0.5))
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride, pad, dil)
    g = _bf(torch.randn(ref.shape))
    ref.backward(g)
    xg, wg = _fmt(x.to(gpu).bfloat16(), True), _fmt(wt.to(gpu).bfloat16(), True)
    y = pkg.ops.conv_fwd(xg, wg, stride, dilation=dil)
    assert tuple(y.shape) == tuple(ref.shape)
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.detach().numpy(), rtol=1e-2, atol=2e-2)
    gg = _fmt(g.to(gpu).bfloat16(), True)
    wtt = _fmt(wt.permute(1, 0, 2, 3).contiguous().to(gpu).bfloat16(), True)
    dx = pkg.ops.conv_dgrad(gg, wtt, (h, w), stride, dilation=dil)
    scale = float(xr.grad.abs().max())
    np.testing.assert_allclose(dx.float().cpu().numpy(), xr.grad.numpy(), rtol=1e-2, atol=1e-2 * scale)
    gw = pkg.ops.conv_wgrad(xg, gg, k, stride, dilation=dil)
    np.testing.assert_allclose(gw.cpu().numpy(), wr.grad.numpy(), rtol=2e-3, atol=2e-3 * float(wr.grad.abs().max()))


def test_conv_stats_fusion_on_ragged_channels_is_declined(pkg, gpu):
    """48 output channels: the moments fusion does not apply — conv_fwd returns no ConvStats and BatchNorm reduces itself."""
    x = torch.randn(1, 256, 9, 9, device=gpu).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(48, 256, 1, 1, device=gpu).bfloat16().contiguous(memory_format=torch.channels_last)
    y, st = pkg.ops.conv_fwd(x, w, 1, stats_shift=torch.zeros(48, device=gpu), want_stats=True)
    assert st is None and tuple(y.shape) == (1, 48, 9, 9)


@pytest.mark.parametrize("n,ci,co,h,w,dils", [(2, 2048, 256, 33, 33, (6, 12, 18)), (1, 512, 128, 17, 21, (2, 5)),
                                              (2, 256, 256, 9, 9, (12, 24, 36, 1))])
def test_conv_fwd_multi_equals_separate_launches(pkg, gpu, n, ci, co, h, w, dils):
    """ASPP's atrous branches as ONE launch (afan_conv_fwd_multi_nhwc_bf16, _deeplab.py:143-150,173-176): outputs bit-identical
    to one launch per branch, BatchNorm moment accumulators equal to accumulation-order noise (f64 atomics), and the module
    path (ASPP.MULTI) leaves the same gradients as the per-branch path."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(n, ci, h, w, generator=g) * 0.5).to(gpu).bfloat16().contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(co, ci, 3, 3, generator=g) * 0.02).to(gpu).bfloat16().contiguous(memory_format=torch.channels_last) for _ in dils]
    shifts = [torch.randn(co, generator=g).to(gpu) * 0.1 for _ in dils]
    assert ops.conv_fwd_multi_ok(x, ws, 1)
    ops.acc_reset(gpu)
    ys, sts = ops.conv_fwd_multi(x, ws, 1, dils, shifts)
    accs = [st.acc.clone() for st in sts]
    ys = [y.clone() for y in ys]
    for b, d in enumerate(dils):
        y1, st1 = ops.conv_fwd(x, ws[b], 1, stats_shift=shifts[b], want_stats=True, dilation=d)
        assert torch.equal(y1, ys[b]), f"branch {b}"
        a1, a2 = st1.acc.double(), accs[b].double()
        assert a1.shape == a2.shape
        # sum over the accumulator copies of (sum, sum of squares) per channel
        k = int(pkg._lib.load().afan_bn_acc_doubles(co))
        body1, body2 = a1[:k], a2[:k]
        # (the tiles' column sums are fp32 before they are added in f64, and the single launch may take another tile shape
        #  than the multi-problem one — e.g. the 128-row x 64-channel halo tile for the dilation-1 branch: fp32 grouping noise)
        torch.testing.assert_close(body1.sum(), body2.sum(), rtol=2e-7, atol=1e-6)
        ref = F.conv2d(x.float(), ws[b].float(), padding=d, dilation=d)
        assert float((y1.float() - ref).norm() / ref.norm()) < 1e-2
    ys2, sts2 = ops.conv_fwd_multi(x, ws, 1, dils, None)
    assert sts2 == [None] * len(dils) and all(torch.equal(a, b_) for a, b_ in zip(ys2, ys))


def test_conv_fwd_multi_with_half_batch_moments(pkg, gpu):
    """The multi-problem launch over two concatenated half-batches (groups = 2: the two sample-point passes of a Segmentation
    iteration as one): outputs bit-identical to the ungrouped launch, each problem's two accumulator blocks equal to the moments a
    grouped single-problem launch sums (33 x 33 maps: the halves are not whole row tiles)."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(4)
    n, ci, co, h, w, dils = 4, 512, 256, 33, 33, (6, 12, 18)
    x = (torch.randn(n, ci, h, w, generator=g) * 0.5).to(gpu).bfloat16().contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(co, ci, 3, 3, generator=g) * 0.02).to(gpu).bfloat16().contiguous(memory_format=torch.channels_last) for _ in dils]
    shifts = [torch.randn(co, generator=g).to(gpu) * 0.1 for _ in dils]
    ops.acc_reset(gpu)
    ys, sts = ops.conv_fwd_multi(x, ws, 1, dils, shifts, groups=2)
    ys = [y.clone() for y in ys]
    k = int(pkg._lib.load().afan_bn_acc_doubles(co))
    for b, d in enumerate(dils):
        y1, st1 = ops.conv_fwd(x, ws[b], 1, stats_shift=shifts[b], want_stats=True, dilation=d, groups=2)
        assert torch.equal(y1, ys[b]), f"branch {b}"
        for grp in range(2):
            a1, a2 = st1.group(grp, co).acc.double()[:k], sts[b].group(grp, co).acc.double()[:k]
            torch.testing.assert_close(a1.sum(), a2.sum(), rtol=2e-7, atol=1e-6)
        # and a BatchNorm over the grouped accumulators normalises each half by its own statistics
        gamma, beta = torch.ones(co, device=gpu), torch.zeros(co, device=gpu)
        o, s_ = ops.bn_train_forward(ys[b], gamma, beta, None, False, 1e-5, 0.1, None, None, None, sts[b], groups=2)
        for grp in range(2):
            half = o[grp * 2:(grp + 1) * 2].float()
            assert abs(float(half.mean())) < 2e-2 and abs(float(half.var(unbiased=False)) - 1) < 5e-2


def test_aspp_multi_launch_equals_per_branch_path(pkg, gpu):
    dl = pkg.deeplab
    res = {}
    old = dl.ASPP.MULTI
    try:
        for multi in (True, False):
            dl.ASPP.MULTI = multi
            torch.manual_seed(5)
            aspp = dl.ASPP(512, (6, 12, 18))
            for m in aspp.modules():
                if isinstance(m, nn.Dropout):
                    m.p = 0.0
                if hasattr(m, "compute_dtype"):
                    m.compute_dtype = torch.bfloat16
            aspp.to(gpu).train()
            pkg.arena.ParamArena(aspp, skip=())
            g = torch.Generator().manual_seed(7)
            x = torch.randn(2, 512, 33, 33, generator=g).relu().to(gpu).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            pkg.ops.acc_reset(gpu)
            before = pkg.ops.CALLS["conv_fwd"]
            y = aspp(x)
            gy = (torch.randn(y.shape, generator=g) * 1e-2).to(gpu).bfloat16().contiguous(memory_format=torch.channels_last)
            y.backward(gy)
            torch.cuda.synchronize()
            res[multi] = (y.detach().float(), x.grad.float(), {k: p.grad.float().clone() for k, p in aspp.named_parameters()},
                          {k: b.clone() for k, b in aspp.named_buffers()})
            assert pkg.ops.CALLS["conv_fwd"] - before == 5        # 1x1, three atrous, projection (the pooling branch is a linear layer)
    finally:
        dl.ASPP.MULTI = old
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0])
    # the input gradient: three dgrads chained through the epilogue addend vs summed by autograd — bf16 rounding of partial sums
    assert float((a[1] - b[1]).norm() / b[1].norm()) < 6e-3
    # parameter gradients: the per-branch launches may take other tiles than the multi-problem launch (ops: 64 x 64 for launches that
    # leave half the chip idle), so the BatchNorm backward's column sums are added in another order and single bf16 gradients
    # round the other way: relative to each tensor's norm
    for k in b[2]:
        assert float((a[2][k] - b[2][k]).norm() / (b[2][k].norm() + 1e-12)) < 2e-3, k
    for k in b[3]:
        torch.testing.assert_close(a[3][k].float(), b[3][k].float(), rtol=1e-5, atol=1e-6, msg=k)


def test_stem_conv_with_an_image_gradient(pkg, gpu):
    """StemConv on an image that requires a gradient (Detection's image-level perturbation): forward through the im2col +
    MFMA path like the gradient-free case (bit-equal to it), input gradient from the general kernel — against an fp32
    convolution of the same bf16 values."""
    torch.manual_seed(5)
    m = pkg.deeplab.StemConv(3, 64, kernel_size=7, stride=2, padding=3, bias=False).to(gpu)
    m.compute_dtype = torch.bfloat16
    x = torch.randn(2, 3, 37, 53, device=gpu).bfloat16().contiguous(memory_format=torch.channels_last)
    y0 = m(x)
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    assert torch.equal(y, y0)
    g = torch.randn_like(y)
    y.backward(g)
    w = m.lp_weight().float()
    xf = x.float().requires_grad_(True)
    ref = F.conv2d(xf, w, None, 2, 3)
    ref.backward(g.float())
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(xr.grad.float().cpu().numpy(), xf.grad.cpu().numpy(), rtol=2e-2, atol=2e-2 * float(xf.grad.abs().max()))
