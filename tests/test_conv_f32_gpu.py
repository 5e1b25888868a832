"""The general fp32-arithmetic convolutions (afan_conv_fwd / afan_conv_dgrad / afan_conv_wgrad, csrc/afan_conv_f32.hip) against
torch's CPU convolution in float64 on the same inputs — forward, input gradient, weight gradient — over the layouts,
storage types and layer shapes the path uses (Classification/resnet_s.py:88-106, Segmentation/network/backbone/resnet.py:
143, _deeplab.py:33-45,146-155), plus ragged and degenerate ones.  fp32 storage: the f32 MFMA is a k-ordered fmaf chain, so
the bound is accumulation-order noise: |d| <= 2e-6 * sum|a*b| (checked as rtol on the f64 result of |x|,|w|).  These
kernels are what fp32 parity mode (north_star's 1e-4 bar) now runs on; there is no vendor convolution to compare with."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (n, ci, hi, wi, co, k, stride, pad, dilation)
SHAPES = [
    (2, 3, 32, 32, 16, 3, 1, 1, 1),        # CIFAR stem (resnet_s.py:88)
    (2, 16, 32, 32, 16, 3, 1, 1, 1),
    (2, 16, 32, 32, 32, 3, 2, 1, 1),       # stage transition, stride 2
    (3, 32, 16, 16, 64, 3, 2, 1, 1),
    (2, 64, 8, 8, 64, 3, 1, 1, 1),
    (2, 64, 9, 7, 128, 1, 2, 0, 1),        # 1x1 / 2 projection shortcut on an odd map: a parity class with no tap
    (2, 128, 4, 4, 128, 3, 1, 1, 1),
    (1, 3, 33, 37, 64, 7, 2, 3, 1),        # ImageNet / DeepLab stem, odd sizes
    (2, 64, 17, 17, 64, 3, 1, 2, 2),       # atrous
    (1, 40, 13, 11, 48, 3, 1, 6, 6),       # atrous rate 6, ragged channels
    (2, 304, 9, 9, 256, 3, 1, 1, 1),       # decoder 3x3 304 -> 256 (_deeplab.py:41)
    (2, 256, 9, 9, 21, 1, 1, 0, 1),        # classifier (bias)
    (2, 20, 6, 5, 10, 3, 2, 1, 1),         # nothing aligned
    (1, 5, 5, 5, 7, 5, 1, 2, 1),
    (4, 2048, 1, 1, 256, 1, 1, 0, 1),      # pooled ASPP branch / linear layers as 1x1
]


def _ref(x, w, b, stride, pad, dil):
    x64 = x.double().cpu().requires_grad_(True)
    w64 = w.double().cpu().requires_grad_(True)
    y = F.conv2d(x64, w64, None if b is None else b.double().cpu(), stride, pad, dil)
    return x64, w64, y


def _bound(x, w, stride, pad, dil):
    """sum |a*b| per output element: the scale accumulation-order noise is relative to."""
    return F.conv2d(x.double().cpu().abs(), w.double().cpu().abs(), None, stride, pad, dil)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_general_conv_matches_float64(pkg, gpu, shape, layout, dtype):
    n, ci, hi, wi, co, k, stride, pad, dil = shape
    g = torch.Generator().manual_seed(hash(shape) % (1 << 31))
    x = torch.randn(n, ci, hi, wi, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    b = torch.randn(co, generator=g) if co == 21 else None
    x, w = x.to(dtype), w.to(dtype)                      # bf16: the stored values ARE the operands
    mf = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    xd = x.to(gpu).contiguous(memory_format=mf)
    wd = w.to(gpu).contiguous(memory_format=mf)
    bd = None if b is None else b.to(gpu)
    x64, w64, y64 = _ref(x.float(), w.float(), b, stride, pad, dil)
    scale = _bound(x.float(), w.float(), stride, pad, dil)
    tol_out = 2e-6 if dtype == torch.float32 else 4e-3     # bf16: one rounding of the stored result
    y = pkg.ops.conv_general_fwd(xd, wd, bd, stride, pad, dil)
    assert y.dtype == dtype and y.shape == y64.shape
    assert y.is_contiguous(memory_format=mf) or y.numel() == 0
    err = (y.double().cpu() - y64.detach()).abs()
    lim = tol_out * (scale + y64.detach().abs() + 1e-3)
    assert bool((err <= lim).all()), f"fwd: max err {float(err.max()):.3e}, max of bound {float(lim.max()):.3e}"

    gy = torch.randn(y64.shape, generator=g).to(dtype)
    y64.backward(gy.double())
    gyd = gy.to(gpu).contiguous(memory_format=mf)
    dx = pkg.ops.conv_general_dgrad(gyd, wd, (hi, wi), stride, pad, dil)
    x0 = torch.zeros_like(x64, requires_grad=True)              # sum |dy * w| per input element (the noise scale)
    F.conv2d(x0, w.double().abs(), None, stride, pad, dil).backward(gy.double().abs())
    sx = x0.grad
    err = (dx.double().cpu() - x64.grad).abs()
    lim = tol_out * (sx + x64.grad.abs() + 1e-3)
    assert dx.shape == x.shape and bool((err <= lim).all()), f"dgrad: max err {float(err.max()):.3e}"

    dw = pkg.ops.conv_general_wgrad(xd, gyd, k, stride, pad, dil)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (co, ci, k, k)
    ref = w64.grad
    err = (dw.double().cpu() - ref).abs()
    npix = n * y64.shape[2] * y64.shape[3]
    lim = 4e-6 * (ref.abs() + float(npix) ** 0.5 * 3.0)     # fp32 accumulate over npix products of O(1) terms, sliced sums
    assert bool((err <= lim).all()), f"wgrad: max err {float(err.max()):.3e} (limit {float(lim.max()):.3e})"
    # accumulate into an existing gradient tensor of the OTHER memory order
    omf = torch.contiguous_format if layout == "nhwc" else torch.channels_last
    acc = torch.ones((co, ci, k, k), device=gpu).contiguous(memory_format=omf)
    pkg.ops.conv_general_wgrad(xd, gyd, k, stride, pad, dil, grad=acc, accumulate=True)
    assert torch.allclose(acc - 1.0, dw, rtol=1e-5, atol=1e-5)


def test_general_conv_is_deterministic_and_counts_no_vendor_call(pkg, gpu):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 64, 16, 16, generator=g).to(gpu).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 64, 3, 3, generator=g).to(gpu).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(8, 64, 16, 16, generator=g).to(gpu).contiguous(memory_format=torch.channels_last)
    a = [pkg.ops.conv_general_fwd(x, w, None, 1, 1, 1), pkg.ops.conv_general_dgrad(gy, w, (16, 16), 1, 1, 1),
         pkg.ops.conv_general_wgrad(x, gy, 3, 1, 1, 1)]
    b = [pkg.ops.conv_general_fwd(x, w, None, 1, 1, 1), pkg.ops.conv_general_dgrad(gy, w, (16, 16), 1, 1, 1),
         pkg.ops.conv_general_wgrad(x, gy, 3, 1, 1, 1)]
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert pkg.ops.CALLS["vendor_conv"] == 0


def test_module_fp32_conv_autograd_matches_torch_cpu(pkg, gpu):
    """resnet_s.Conv2d in fp32 (both layouts) through autograd: output, input gradient and weight gradient against the
    same module's arithmetic on the CPU in float64; the 7x7 stem and an atrous layer included."""
    for (ci, co, k, s, p, d, hw) in ((3, 64, 7, 2, 3, 1, 33), (16, 32, 3, 2, 1, 1, 16), (64, 64, 3, 1, 4, 4, 12), (48, 24, 1, 1, 0, 1, 9)):
        for cl in (False, True):
            torch.manual_seed(ci + co)
            m = pkg.resnet_s.Conv2d(ci, co, k, stride=s, padding=p, dilation=d, bias=False).to(gpu)
            x = torch.randn(2, ci, hw, hw)
            xd = x.to(gpu)
            if cl:
                xd = xd.contiguous(memory_format=torch.channels_last)
            xd.requires_grad_(True)
            y = m(xd)
            gy = torch.randn(y.shape)
            y.backward(gy.to(gpu))
            x64 = x.double().requires_grad_(True)
            w64 = m.weight.detach().double().cpu().requires_grad_(True)
            y64 = F.conv2d(x64, w64, None, s, p, d)
            y64.backward(gy.double())
            assert torch.allclose(y.double().cpu(), y64.detach(), rtol=1e-5, atol=1e-5)
            assert torch.allclose(xd.grad.double().cpu(), x64.grad, rtol=1e-5, atol=1e-5)
            assert torch.allclose(m.weight.grad.double().cpu(), w64.grad, rtol=1e-5, atol=5e-4)      # sums of ~600 O(1) products
    assert pkg.ops.CALLS["vendor_conv"] == 0


@pytest.mark.parametrize("n,ci,co", [(64, 2048, 1000), (128, 2048, 84), (5, 1024, 21), (16, 64, 10), (3, 300, 700)])
def test_linear_on_the_general_kernels_matches_float64(pkg, gpu, n, ci, co):
    """resnet_s._LinearFn (nn.Linear without a vendor GEMM): output, input gradient, weight and bias gradients against
    float64 — including the two re-readings as weight-gradient problems (few rows with a long reduction: the forward for
    ci >= 1024, the input gradient for co >= 512)."""
    torch.manual_seed(n + ci + co)
    lin = torch.nn.Linear(ci, co).to(gpu)
    x = torch.randn(n, ci, device=gpu, requires_grad=True)
    y = pkg.resnet_s._LinearFn.apply(x, lin.weight, lin.bias, True)
    g = torch.randn(n, co, device=gpu)
    y.backward(g)
    x64 = x.detach().double().cpu().requires_grad_(True)
    w64, b64 = lin.weight.detach().double().cpu().requires_grad_(True), lin.bias.detach().double().cpu().requires_grad_(True)
    y64 = F.linear(x64, w64, b64)
    y64.backward(g.double().cpu())
    assert torch.allclose(y.double().cpu(), y64.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(x.grad.double().cpu(), x64.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(lin.weight.grad.double().cpu(), w64.grad, rtol=1e-5, atol=1e-4)
    assert torch.allclose(lin.bias.grad.double().cpu(), b64.grad, rtol=1e-5, atol=1e-4)
    assert pkg.ops.CALLS["vendor_conv"] == 0
