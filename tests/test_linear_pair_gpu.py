"""The two-heads-on-one-input linear kernels (afan_linear.hip through ops.linear_pair_*) against float64 tensor arithmetic: the ROI
head's shapes (128 x 2048 -> 21 | 84: split reduction), the RPN's (38 x 57 pixels x 512 -> 18 | 36), ragged ones; fp32 FMA chains in
another order than the reference's GEMM: 2e-5 of the result's scale.  Deterministic: two runs are the same bits."""
import pytest
import torch

pytestmark = pytest.mark.gpu
SHAPES = [(128, 2048, 21, 84), (2166, 512, 18, 36), (37, 36, 5, 3), (1, 4, 1, 1), (300, 260, 64, 64), (4332, 512, 18, 36)]


def _close(got, want, tol=2e-5):
    scale = float(want.abs().max()) + 1e-30
    assert float((got.double() - want).abs().max()) <= tol * scale, (float((got.double() - want).abs().max()), scale)


@pytest.mark.parametrize("M,K,n1,n2", SHAPES)
def test_forward_input_gradient_and_parameter_gradients(pkg, gpu, M, K, n1, n2):
    ops = pkg.ops
    g = torch.Generator().manual_seed(M + K + n1)
    x = torch.randn(M, K, generator=g).to(gpu)
    w1, w2 = (torch.randn(n1, K, generator=g) * 0.05).to(gpu), (torch.randn(n2, K, generator=g) * 0.05).to(gpu)
    b1, b2 = torch.randn(n1, generator=g).to(gpu), torch.randn(n2, generator=g).to(gpu)
    assert ops.linear_pair_ok(x, w1, w2)
    y1, y2 = ops.linear_pair_fwd(x, w1, b1, w2, b2)
    _close(y1, x.double() @ w1.double().t() + b1.double())
    _close(y2, x.double() @ w2.double().t() + b2.double())
    again = ops.linear_pair_fwd(x, w1, b1, w2, b2)
    assert torch.equal(again[0], y1) and torch.equal(again[1], y2)
    nb1, nb2 = ops.linear_pair_fwd(x, w1, None, w2, None)
    _close(nb1, x.double() @ w1.double().t())
    g1, g2 = torch.randn(M, n1, generator=g).to(gpu), torch.randn(M, n2, generator=g).to(gpu)
    gx = ops.linear_pair_dgrad(g1, g2, w1, w2)
    _close(gx, g1.double() @ w1.double() + g2.double() @ w2.double())
    gw1, gw2, gb1, gb2 = torch.full_like(w1, 7.0), torch.full_like(w2, 7.0), torch.full_like(b1, 7.0), torch.full_like(b2, 7.0)
    ops.linear_pair_wgrad(g1, g2, x, gw1, gb1, gw2, gb2, False)
    _close(gw1, g1.double().t() @ x.double())
    _close(gw2, g2.double().t() @ x.double())
    _close(gb1, g1.double().sum(0))
    _close(gb2, g2.double().sum(0))
    first = gw1.clone()
    ops.linear_pair_wgrad(g1, g2, x, gw1, gb1, gw2, None, True)                 # accumulate; the second bias absent
    _close(gw1, 2 * (g1.double().t() @ x.double()))
    _close(gb1, 2 * g1.double().sum(0))
    _close(gb2, g2.double().sum(0))
    gw1.copy_(first)
    ops.linear_pair_wgrad(g1, g2, x, gw1, gb1, gw2, gb2, False)
    assert torch.equal(gw1, first)


def test_the_autograd_function_equals_two_linear_layers(pkg, gpu):
    """det_model._linear_pair on nn.Linear and on 1x1 nn.Conv2d parameters: outputs and every gradient against the layers in float64."""
    dm = pkg.det_model
    torch.manual_seed(3)
    for conv in (False, True):
        K, n1, n2, M = 512, 18, 36, 2166
        if conv:
            l1, l2 = torch.nn.Conv2d(K, n1, 1).to(gpu), torch.nn.Conv2d(K, n2, 1).to(gpu)
        else:
            l1, l2 = torch.nn.Linear(K, n1).to(gpu), torch.nn.Linear(K, n2).to(gpu)
        x = torch.randn(M, K, device=gpu, requires_grad=True)
        ys = dm._linear_pair(x, l1, l2)
        assert ys is not None
        u, v = torch.randn(M, n1, device=gpu), torch.randn(M, n2, device=gpu)
        ((ys[0] * u).sum() + (ys[1] * v).sum()).backward()
        xd = x.detach().double().requires_grad_(True)
        W1, W2 = l1.weight.detach().double().reshape(n1, K).requires_grad_(True), l2.weight.detach().double().reshape(n2, K).requires_grad_(True)
        B1, B2 = l1.bias.detach().double().requires_grad_(True), l2.bias.detach().double().requires_grad_(True)
        r1, r2 = xd @ W1.t() + B1, xd @ W2.t() + B2
        ((r1 * u.double()).sum() + (r2 * v.double()).sum()).backward()
        _close(ys[0].detach(), r1.detach()), _close(ys[1].detach(), r2.detach())
        _close(x.grad, xd.grad)
        _close(l1.weight.grad.reshape(n1, K), W1.grad), _close(l2.weight.grad.reshape(n2, K), W2.grad)
        _close(l1.bias.grad, B1.grad), _close(l2.bias.grad, B2.grad)
        assert l1.weight.grad.shape == l1.weight.shape


def test_shapes_outside_the_kernels_are_refused(pkg, gpu):
    ops = pkg.ops
    x = torch.randn(8, 10, device=gpu)                       # K % 4 != 0
    assert not ops.linear_pair_ok(x, torch.randn(3, 10, device=gpu), torch.randn(3, 10, device=gpu))
    x = torch.randn(8, 16, device=gpu)
    assert not ops.linear_pair_ok(x, torch.randn(100, 16, device=gpu), torch.randn(29, 16, device=gpu))      # 129 features
    assert not ops.linear_pair_ok(x.cpu(), torch.randn(3, 16), torch.randn(3, 16))
    lib = pkg._lib.load()
    assert lib.afan_linear_pair_workspace_floats(0, 8, 100, 29, 16) == -1


def test_one_layer_and_argument_errors_through_the_c_abi(pkg, gpu):
    """n2 == 0 (one layer: w2 / y2 / g2 NULL) and the entry points' argument checks (AFAN_E* before any launch)."""
    import ctypes as C
    lib = pkg._lib.load()
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(gpu).cuda_stream)
    g = torch.Generator().manual_seed(11)
    M, K, n1 = 70, 96, 9
    x, w1, b1 = torch.randn(M, K, generator=g).to(gpu), torch.randn(n1, K, generator=g).to(gpu), torch.randn(n1, generator=g).to(gpu)
    y1 = torch.empty(M, n1, device=gpu)
    nws = lib.afan_linear_pair_workspace_floats(0, M, n1, 0, K)
    ws = torch.empty(max(nws, 1), device=gpu)
    assert lib.afan_linear_pair_fwd_f32(ptr(x), ptr(w1), ptr(b1), None, None, ptr(y1), None, M, n1, 0, K, ptr(ws), st) == 0
    _close(y1, x.double() @ w1.double().t() + b1.double())
    g1 = torch.randn(M, n1, generator=g).to(gpu)
    gx = torch.empty(M, K, device=gpu)
    assert lib.afan_linear_pair_dgrad_f32(ptr(g1), None, ptr(w1), None, ptr(gx), M, n1, 0, K, st) == 0
    _close(gx, g1.double() @ w1.double())
    gw, gb = torch.empty(n1, K, device=gpu), torch.empty(n1, device=gpu)
    ws2 = torch.empty(lib.afan_linear_pair_workspace_floats(1, M, n1, 0, K), device=gpu)
    assert lib.afan_linear_pair_wgrad_f32(ptr(g1), None, ptr(x), ptr(gw), ptr(gb), None, None, 0, M, n1, 0, K, ptr(ws2), st) == 0
    _close(gw, g1.double().t() @ x.double()), _close(gb, g1.double().sum(0))
    E_NULL, E_SHAPE, E_ALIGN = -4, -3, -2          # include/afan_hip.h: AFAN_ENULL, AFAN_ESHAPE, AFAN_EALIGN
    assert lib.afan_linear_pair_fwd_f32(None, ptr(w1), None, None, None, ptr(y1), None, M, n1, 0, K, ptr(ws), st) == E_NULL
    assert lib.afan_linear_pair_fwd_f32(ptr(x), ptr(w1), None, None, None, ptr(y1), None, M, n1, 3, K, ptr(ws), st) == E_NULL      # second layer announced, not given
    assert lib.afan_linear_pair_fwd_f32(ptr(x), ptr(w1), None, None, None, ptr(y1), None, M, n1, 0, K + 2, ptr(ws), st) == E_SHAPE   # K % 4
    assert lib.afan_linear_pair_fwd_f32(ptr(x), ptr(w1), None, None, None, ptr(y1), None, M, 129, 0, K, ptr(ws), st) == E_SHAPE
    off = torch.empty(M * K + 1, device=gpu)[1:]                        # 4-byte aligned only
    assert lib.afan_linear_pair_fwd_f32(ptr(off), ptr(w1), None, None, None, ptr(y1), None, M, n1, 0, K, ptr(ws), st) == E_ALIGN
    assert lib.afan_linear_pair_wgrad_f32(ptr(g1), None, ptr(x), ptr(gw), None, None, None, 0, M, n1, 0, K, None, st) == E_NULL     # no workspace
    # a split forward (few rows) without its workspace
    xs = torch.randn(8, 2048, generator=g).to(gpu)
    wl = torch.randn(n1, 2048, generator=g).to(gpu)
    assert lib.afan_linear_pair_workspace_floats(0, 8, n1, 0, 2048) > 0
    assert lib.afan_linear_pair_fwd_f32(ptr(xs), ptr(wl), None, None, None, ptr(torch.empty(8, n1, device=gpu)), None, 8, n1, 0, 2048, None, st) == E_NULL
    torch.cuda.synchronize()


def test_a_rows_result_does_not_depend_on_the_batch_it_is_in(pkg, gpu):
    """The forward splits its reduction per staging chunk for every M: rows computed alone, in a batch of 128 and in a batch of 896 are
    the same bits (the Faster-RCNN heads run pass by pass or for several passes at once); the input gradient is a per-row sum as well."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(2)
    for K, n1, n2 in ((2048, 21, 84), (512, 18, 36)):
        x = torch.randn(896, K, generator=g).to(gpu)
        w1, w2 = (torch.randn(n1, K, generator=g) * 0.05).to(gpu), (torch.randn(n2, K, generator=g) * 0.05).to(gpu)
        b1, b2 = torch.randn(n1, generator=g).to(gpu), torch.randn(n2, generator=g).to(gpu)
        big = ops.linear_pair_fwd(x, w1, b1, w2, b2)
        for lo, hi in ((0, 128), (128, 256), (640, 896), (5, 6)):
            part = ops.linear_pair_fwd(x[lo:hi].contiguous(), w1, b1, w2, b2)
            assert torch.equal(part[0], big[0][lo:hi]) and torch.equal(part[1], big[1][lo:hi])
        g1, g2 = torch.randn(896, n1, generator=g).to(gpu), torch.randn(896, n2, generator=g).to(gpu)
        gx = ops.linear_pair_dgrad(g1, g2, w1, w2)
        assert torch.equal(ops.linear_pair_dgrad(g1[128:256].contiguous(), g2[128:256].contiguous(), w1, w2), gx[128:256])
