"""HIP kernels (through the C-ABI, via cv_a-fan_amd.ops) against the oracle on identical inputs.
Bit-exact for the PGD update / projection / noise / clamp / lerp; stated tolerances for reductions."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import assert_close_frac, golden, ptr

pytestmark = pytest.mark.gpu


def _dev(a, gpu, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    return t if dtype is None else t.to(dtype)


# ------------------------------------------------------------------------------------------- PGD step
@pytest.mark.parametrize("case", ["pgd_trace_r20s_k3", "pgd_trace_r20s_k3_clip", "pgd_trace_r18_k5"])
def test_pgd_step_golden_trace(pkg, gpu, case):
    """Reference x_adv(t), reference gradient -> x_adv(t+1): bit for bit, every step of the reference run (ResNet-20s, and
    the headline network's 64 x 32 x 32 feature map at K = 5: given the gradient, the perturbation is the reference's)."""
    g = golden(case)
    gamma, eps = float(g["gamma_eps"][0]) / 255, float(g["gamma_eps"][1]) / 255
    clip = bool(int(g["clip"]))
    fm = _dev(g["fm"], gpu)
    for t in range(g["grads"].shape[0]):
        xa = _dev(g["snaps"][t], gpu)
        pkg.ops.pgd_step_(xa, _dev(g["grads"][t], gpu), gamma, fm, eps, clip)
        np.testing.assert_array_equal(xa.cpu().numpy(), g["snaps"][t + 1])
    # last step fused with the norms
    t = g["grads"].shape[0] - 1
    xa = _dev(g["snaps"][t], gpu)
    l2, linf = pkg.ops.pgd_step_norms_(xa, _dev(g["grads"][t], gpu), gamma, fm, eps, clip)
    np.testing.assert_array_equal(xa.cpu().numpy(), g["snaps"][t + 1])
    d = (torch.from_numpy(g["snaps"][t + 1]) - torch.from_numpy(g["fm"])).reshape(fm.shape[0], -1)
    np.testing.assert_allclose(l2.cpu().numpy(), torch.norm(d, p=2, dim=1).numpy(), rtol=1e-5)
    np.testing.assert_array_equal(linf.cpu().numpy(), torch.norm(d, p=float("inf"), dim=1).numpy())


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 63, 64, 257, 4099, 1 << 20])
@pytest.mark.parametrize("clip", [False, True])
@pytest.mark.parametrize("gdt", [torch.float32, torch.bfloat16])
def test_pgd_step_vs_c_oracle(pkg, gpu, c_oracle, n, clip, gdt):
    rng = np.random.default_rng(n + 7 * clip)
    x = rng.standard_normal(n).astype(np.float32)
    xa = (x + rng.uniform(-0.02, 0.02, n)).astype(np.float32)
    gr = rng.standard_normal(n).astype(np.float32)
    if n > 8:
        gr[1], gr[2], gr[3] = 0.0, -0.0, np.nan   # sign(0)=0, sign(NaN)=NaN
        xa[5] = np.nan
    g_t = _dev(gr, gpu, gdt)
    gr_used = g_t.float().cpu().numpy()        # what the kernel sees after the bf16 rounding
    gamma, eps = np.float32(0.5 / 255), np.float32(2 / 255)
    ref = xa.copy()
    c_oracle.oracle_pgd_step(ptr(ref), ptr(gr_used), ptr(x), n, gamma, eps, int(clip))
    xa_t = _dev(xa, gpu)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=gpu)
    pkg.ops.pgd_step_(xa_t, g_t, float(gamma), _dev(x, gpu), float(eps), clip, shadow)
    out = xa_t.cpu().numpy()
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))
    # bf16 shadow == torch's RNE cast of the fp32 result
    # (compared as values: a NaN stays a NaN, its payload bits are not part of the contract)
    np.testing.assert_array_equal(shadow.float().cpu().numpy(),
                                  torch.from_numpy(ref).to(torch.bfloat16).float().numpy())


def test_pgd_step_unaligned_views(pkg, gpu, c_oracle):
    """Views that start 4 bytes off a 16-byte boundary take the scalar path; same bits."""
    n = 1001
    rng = np.random.default_rng(5)
    base = _dev(rng.standard_normal(n + 1).astype(np.float32), gpu)
    gbase = _dev(rng.standard_normal(n + 1).astype(np.float32), gpu)
    xa, gr = base[1:], gbase[1:]
    ref = xa.cpu().numpy().copy()
    c_oracle.oracle_pgd_step(ptr(ref), ptr(gr.cpu().numpy().copy()), None, n, np.float32(0.01), np.float32(0), 0)
    pkg.ops.pgd_step_(xa, gr, 0.01)
    np.testing.assert_array_equal(xa.cpu().numpy(), ref)


@pytest.mark.parametrize("batch,per", [(1, 1), (3, 7), (4, 4096), (5, 4100), (2, 16384 + 4), (256, 1024)])
def test_norms_vs_c_oracle(pkg, gpu, c_oracle, batch, per):
    rng = np.random.default_rng(batch * 131 + per)
    x = rng.standard_normal((batch, per)).astype(np.float32)
    xa = (x + rng.uniform(-0.01, 0.01, (batch, per))).astype(np.float32)
    l2r, lir = np.zeros(batch, np.float32), np.zeros(batch, np.float32)
    c_oracle.oracle_perturb_norms(ptr(xa), ptr(x), batch, per, ptr(l2r), ptr(lir))
    l2, linf = pkg.ops.perturb_norms(_dev(xa, gpu), _dev(x, gpu))
    np.testing.assert_allclose(l2.cpu().numpy(), l2r, rtol=2e-6)   # fp32 tree sum vs double sum
    np.testing.assert_array_equal(linf.cpu().numpy(), lir)         # max is order independent: exact


def test_norms_nan_propagates(pkg, gpu):
    x = torch.zeros(2, 5000, device=gpu)
    xa = x.clone()
    xa[1, 4321] = float("nan")
    l2, linf = pkg.ops.perturb_norms(xa, x)
    assert float(l2[0]) == 0.0 and float(linf[0]) == 0.0
    assert torch.isnan(l2[1]) and torch.isnan(linf[1])


def test_randinit_noise_golden(pkg, gpu):
    g = golden("step_r20s_k3_clip_rand")
    ref = torch.from_numpy(g["feature_map"].copy())
    ref += (2.0 * torch.from_numpy(g["u"]) - 1.0) * (2.0 / 255)   # attack_algo.py:44 on the reference's draw
    xa = _dev(g["feature_map"], gpu)
    pkg.ops.axpy_noise_(xa, _dev(g["u"], gpu), 2.0 / 255)
    np.testing.assert_array_equal(xa.cpu().numpy(), ref.numpy())


def test_clamp_edges_golden(pkg, gpu):
    g = golden("clamp_edges")
    t = _dev(g["t"], gpu)
    out = pkg.attack_algo.linfball_proj(_dev(g["c"], gpu), float(g["radius"]), t, in_place=True)
    assert out.data_ptr() == t.data_ptr()
    np.testing.assert_array_equal(out.cpu().numpy(), g["out"])
    t2 = _dev(g["t"], gpu)
    c = _dev(g["c"], gpu)
    out2 = pkg.attack_algo.tensor_clamp(t2, c - 0.5, c + 0.5, in_place=False)
    assert out2.data_ptr() != t2.data_ptr()
    np.testing.assert_array_equal(out2.cpu().numpy(), g["out"])
    np.testing.assert_array_equal(t2.cpu().numpy(), g["t"])


# --------------------------------------------------------------------------------- mix_feature / lerp
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_mix_feature_golden(pkg, gpu, tag):
    g = golden("seg_ops")
    out = pkg.attack_algo.mix_feature(_dev(g[f"mix_{tag}_clean"], gpu), _dev(g[f"mix_{tag}_adv"], gpu))
    np.testing.assert_allclose(out.cpu().numpy(), g[f"mix_{tag}_out"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 1024, 33, 33), (1, 256, 33, 33), (1, 304, 20, 129), (2, 2, 3, 3),
                                   (1, 2048, 5, 7), (1, 6000, 2, 3), (128, 2048, 1, 1), (3, 19, 1, 1)])   # (.., 1, 1): Detection's pooled ROI feature
def test_mix_feature_vs_c_oracle(pkg, gpu, c_oracle, shape):
    rng = np.random.default_rng(sum(shape))
    clean = (rng.standard_normal(shape) * 1.3 + 4.0).astype(np.float32)   # |mean| >> std stresses the variance
    adv = (clean + rng.standard_normal(shape) * 0.1).astype(np.float32)
    ref = np.zeros_like(clean)
    n, c = shape[:2]
    c_oracle.oracle_mix_feature(ptr(clean), ptr(adv), ptr(ref), n, c, clean.size // (n * c), np.float32(1e-5))
    out = pkg.ops.mix_feature(_dev(clean, gpu), _dev(adv, gpu))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    # idempotence-like property of the op: mixing x with itself returns x (up to rounding)
    same = pkg.ops.mix_feature(_dev(clean, gpu), _dev(clean, gpu))
    np.testing.assert_allclose(same.cpu().numpy(), clean, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 1024, 33, 33), (1, 304, 20, 129), (3, 19, 7, 9), (1, 2048, 5, 7), (2, 64, 16, 16),
                                   (1, 6000, 2, 3)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_mix_feature_channels_last(pkg, gpu, c_oracle, shape, dt):
    """The channels-last kernel (one wave per pixel) against the C oracle / the NCHW kernel on the same values, and on
    the golden vectors of the reference's mix_feature."""
    rng = np.random.default_rng(sum(shape))
    clean = (rng.standard_normal(shape) * 1.3 + 4.0).astype(np.float32)
    adv = (clean + rng.standard_normal(shape) * 0.1).astype(np.float32)
    tc = torch.from_numpy(clean).to(gpu, dt).contiguous(memory_format=torch.channels_last)
    ta = torch.from_numpy(adv).to(gpu, dt).contiguous(memory_format=torch.channels_last)
    out = pkg.ops.mix_feature(tc, ta)
    assert out.stride() == tc.stride() and out.dtype == dt
    n, c = shape[:2]
    cin, ain = tc.float().cpu().contiguous().numpy(), ta.float().cpu().contiguous().numpy()   # the values the kernel saw
    ref = np.zeros_like(cin)
    c_oracle.oracle_mix_feature(ptr(cin), ptr(ain), ptr(ref), n, c, cin.size // (n * c), np.float32(1e-5))
    tol = dict(rtol=2e-5, atol=2e-5) if dt == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, **tol)
    np.testing.assert_allclose(out.float().cpu().numpy(), pkg.ops.mix_feature(tc.contiguous(), ta.contiguous()).float().cpu().numpy(),
                               **tol)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_mix_feature_channels_last_golden(pkg, gpu, tag):
    g = golden("seg_ops")
    cl = lambda a: _dev(a, gpu).contiguous(memory_format=torch.channels_last)
    out = pkg.attack_algo.mix_feature(cl(g[f"mix_{tag}_clean"]), cl(g[f"mix_{tag}_adv"]))
    np.testing.assert_allclose(out.cpu().numpy(), g[f"mix_{tag}_out"], rtol=1e-5, atol=1e-5)


def test_mix_feature_c1_is_nan(pkg, gpu):
    x = torch.randn(1, 1, 4, 4, device=gpu)
    assert torch.isnan(pkg.ops.mix_feature(x, x + 1)).all()   # unbiased variance over one channel: 0/0, as the reference


def test_mix_feature_bf16(pkg, gpu, orc):
    torch.manual_seed(0)
    clean = torch.randn(2, 64, 9, 9)
    adv = clean + 0.1 * torch.randn_like(clean)
    ref = orc.mix_feature(clean.bfloat16().float(), adv.bfloat16().float())
    out = pkg.ops.mix_feature(clean.bfloat16().to(gpu), adv.bfloat16().to(gpu))
    assert out.dtype == torch.bfloat16
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("shape", [(2, 1024, 33, 33), (1, 304, 20, 129), (3, 19, 7, 9), (1, 2048, 5, 7), (2, 2, 3, 3), (1, 6000, 2, 3),
                                   (128, 2048, 1, 1)])
@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("number,mix", [(3, (True, True)), (3, (True, False)), (3, (False, True)), (5, (True, False, True, True)),
                                        (2, (True,))])
def test_fused_sample_points_and_mix(pkg, gpu, orc, shape, nhwc, number, mix):
    """afan_lerp_mix (one launch) == get_sample_points followed by mix_feature on the flagged points, bit for bit against
    the product's separate launches and within the mix_feature tolerance against the reference restatement."""
    rng = np.random.default_rng(sum(shape) + number)
    clean = torch.from_numpy((rng.standard_normal(shape) * 1.3 + 4.0).astype(np.float32))
    adv = clean + torch.from_numpy((rng.standard_normal(shape) * 0.1).astype(np.float32))
    fmt = (lambda t: t.to(gpu).contiguous(memory_format=torch.channels_last)) if nhwc else (lambda t: t.to(gpu))
    x, y = fmt(clean), fmt(adv)
    pts = pkg.attack_algo.sample_points_mixed(x, y, number, mix)
    assert len(pts) == number and pts[0] is x
    sep = pkg.attack_algo.get_sample_points(x, y, number)
    ref = orc.get_sample_points(clean, adv, number)
    for j in range(1, number):
        want = pkg.attack_algo.mix_feature(x, sep[j]) if mix[j - 1] else sep[j]
        assert pts[j].stride() == x.stride()
        assert torch.equal(pts[j], want), (j, float((pts[j] - want).abs().max()))
        r = orc.mix_feature(clean, ref[j]) if mix[j - 1] else ref[j]
        np.testing.assert_allclose(pts[j].cpu().numpy(), r.numpy(), rtol=2e-5, atol=2e-5)
    if not mix[-1]:
        assert pts[-1] is y or pts[-1].data_ptr() == y.data_ptr()


@pytest.mark.parametrize("n", [1, 7, 8, 1000, 65536 * 4 + 3])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("want_shadow", [False, True])
def test_pgd_init_one_launch(pkg, gpu, n, dt, want_shadow):
    """afan_pgd_init: fp32 copy, clone and bf16 shadow of the feature map in one pass == .float(), .clone(), .bfloat16()."""
    torch.manual_seed(n)
    x = (torch.randn(n, device=gpu) * 3).to(dt)
    x32, x_adv, shadow = pkg.ops.pgd_init(x, want_shadow)
    assert x32.dtype == torch.float32 and torch.equal(x32, x.float()) and (x32 is x) == (dt == torch.float32)
    assert x_adv.dtype == torch.float32 and torch.equal(x_adv, x.float()) and x_adv.data_ptr() != x32.data_ptr()
    assert (shadow is None) == (not want_shadow)
    if want_shadow:
        assert torch.equal(shadow, x.float().bfloat16())
    xc = torch.randn(2, 16, 5, 5, device=gpu).to(dt).contiguous(memory_format=torch.channels_last)
    a, b, _ = pkg.ops.pgd_init(xc)
    assert a.stride() == xc.stride() and b.stride() == xc.stride() and torch.equal(b, xc.float())


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("npts", [3, 5])
def test_lerp_points_golden(pkg, gpu, tag, npts):
    g = golden("seg_ops")
    pts = pkg.attack_algo.get_sample_points(_dev(g[f"mix_{tag}_clean"], gpu), _dev(g[f"mix_{tag}_adv"], gpu), npts)
    assert len(pts) == npts
    got = np.stack([p.cpu().numpy() for p in pts])
    ref = g[f"lerp_{tag}_{npts}"]
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[-1], ref[-1])
    np.testing.assert_allclose(got, ref, rtol=1.2e-7, atol=1e-7)   # <= 1 ulp (ATen's scalar tail rounds twice)
    assert (got == ref).mean() > 0.95


# ------------------------------------------------------------------------------------------ BatchNorm
BN_SHAPES = [(4, 16, 32, 32), (2, 64, 16, 16), (3, 5, 7, 9), (8, 512, 4, 4), (2, 3, 33, 33), (1, 8, 1, 1024),
             (64, 64, 32, 32)]


@pytest.mark.parametrize("shape", BN_SHAPES)
@pytest.mark.parametrize("res,relu", [(False, False), (False, True), (True, True)])
def test_bn_forward_backward_fp32_vs_c_oracle(pkg, gpu, c_oracle, shape, res, relu):
    rng = np.random.default_rng(sum(shape) + 2 * res + relu)
    n, c = shape[:2]
    hw = shape[2] * shape[3]
    x = (rng.standard_normal(shape) * 2.0 + 3.0).astype(np.float32)
    r = rng.standard_normal(shape).astype(np.float32) if res else None
    w = rng.uniform(0.5, 1.5, c).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    rm, rv = rng.standard_normal(c).astype(np.float32), rng.uniform(0.5, 2, c).astype(np.float32)
    dy = rng.standard_normal(shape).astype(np.float32)
    # oracle
    y_ref, mean_ref, is_ref = np.zeros_like(x), np.zeros(c, np.float32), np.zeros(c, np.float32)
    rm_ref, rv_ref = rm.copy(), rv.copy()
    c_oracle.oracle_bn_train_forward(ptr(x), ptr(r), ptr(y_ref), n, c, hw, np.float32(1e-5), np.float32(0.1),
                                     ptr(w), ptr(b), int(relu), ptr(mean_ref), ptr(is_ref), ptr(rm_ref), ptr(rv_ref))
    dx_ref, dres_ref = np.zeros_like(x), np.zeros_like(x)
    dw_ref, db_ref = np.zeros(c, np.float32), np.zeros(c, np.float32)
    c_oracle.oracle_bn_backward(ptr(dy), ptr(x), ptr(y_ref), ptr(dx_ref), ptr(dres_ref) if res else None, n, c, hw,
                                ptr(mean_ref), ptr(is_ref), ptr(w), int(relu), ptr(dw_ref), ptr(db_ref))
    # HIP
    xt, rt = _dev(x, gpu), (_dev(r, gpu) if res else None)
    wt, bt, rmt, rvt = _dev(w, gpu), _dev(b, gpu), _dev(rm, gpu), _dev(rv, gpu)
    nbt = torch.zeros((), dtype=torch.int64, device=gpu)
    y, stats = pkg.ops.bn_train_forward(xt, wt, bt, rt, relu, 1e-5, 0.1, rmt, rvt, nbt)
    mean, invstd = stats[0], stats[1]
    assert int(nbt) == 1
    np.testing.assert_allclose(mean.cpu().numpy(), mean_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(invstd.cpu().numpy(), is_ref, rtol=1e-5)
    np.testing.assert_allclose(rmt.cpu().numpy(), rm_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvt.cpu().numpy(), rv_ref, rtol=1e-5)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref, rtol=1e-5, atol=3e-5)
    dwb = torch.zeros(2, c, device=gpu)
    y_for_mask = y if (relu and res) else None
    dx, dres = pkg.ops.bn_backward(_dev(dy, gpu), xt, y_for_mask, stats, wt, bt, relu, res, dwb[0], dwb[1])
    scale = max(1.0, float(np.abs(dw_ref).max()))
    np.testing.assert_allclose(dwb[0].cpu().numpy(), dw_ref, rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(dwb[1].cpu().numpy(), db_ref, rtol=1e-4, atol=1e-4 * scale)
    # a ReLU mask can flip where |bn(x)| is within fp32 rounding of 0 (oracle computes it in double): allow 1e-5 of elements
    assert_close_frac(dx.cpu().numpy(), dx_ref, 1e-4, 2e-5, 1e-5, 'dx')
    if res:
        assert_close_frac(dres.cpu().numpy(), dres_ref, 0, 0, 1e-5, 'dres')


def test_bn_matches_torch_autograd_fp32(pkg, gpu):
    """Fused module forward/backward vs torch.nn.BatchNorm2d + relu on CPU (the arithmetic the reference runs)."""
    torch.manual_seed(1)
    x = torch.randn(6, 12, 10, 10) * 1.5 + 0.7
    res = torch.randn_like(x)
    ref_bn = torch.nn.BatchNorm2d(12)
    ref_bn.weight.data.uniform_(0.5, 1.5)
    ref_bn.bias.data.normal_()
    bn = pkg.resnet_s.BatchNorm2d(12)
    bn.load_state_dict(ref_bn.state_dict())
    bn.to(gpu)
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True)
    yr = torch.relu(ref_bn(xr) + rr)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    xg = x.to(gpu).requires_grad_(True)
    rg = res.to(gpu).requires_grad_(True)
    yg = bn.fused(xg, rg, True)
    yg.backward(gy.to(gpu))
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(rg.grad.cpu().numpy(), rr.grad.numpy())
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), ref_bn.weight.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), ref_bn.bias.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref_bn.running_var.numpy(), rtol=1e-5)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref_bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1


def test_bn_bf16_close_to_fp32(pkg, gpu):
    torch.manual_seed(2)
    x = (torch.randn(8, 32, 16, 16) + 0.5).bfloat16()
    w, b = torch.rand(32) + 0.5, torch.randn(32)
    y32, s32 = pkg.ops.bn_train_forward(x.float().to(gpu), w.to(gpu), b.to(gpu), None, True, 1e-5, 0.1, None, None, None)
    y16, s16 = pkg.ops.bn_train_forward(x.to(gpu), w.to(gpu), b.to(gpu), None, True, 1e-5, 0.1, None, None, None)
    (m32, i32), (m16, i16) = (s32[0], s32[1]), (s16[0], s16[1])
    np.testing.assert_allclose(m16.cpu().numpy(), m32.cpu().numpy(), rtol=1e-5, atol=1e-6)  # stats are fp32 in both
    np.testing.assert_allclose(i16.cpu().numpy(), i32.cpu().numpy(), rtol=1e-5)
    np.testing.assert_allclose(y16.float().cpu().numpy(), y32.cpu().numpy(), rtol=8e-3, atol=8e-3)


def test_bn_stats_standalone(pkg, gpu):
    torch.manual_seed(3)
    x = torch.randn(5, 7, 6, 6, device=gpu) * 3 + 10
    mean, invstd = pkg.ops.bn_stats(x)
    ref_mean = x.double().mean(dim=(0, 2, 3))
    ref_var = x.double().var(dim=(0, 2, 3), unbiased=False)
    np.testing.assert_allclose(mean.cpu().numpy(), ref_mean.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(invstd.cpu().numpy(), (1 / torch.sqrt(ref_var + 1e-5)).cpu().numpy(), rtol=1e-5)


# ------------------------------------------------------------------------------------------------ SGD
@pytest.mark.parametrize("n", [1, 5, 1024, 100003])
def test_sgd_vs_c_oracle_and_torch(pkg, gpu, c_oracle, n):
    rng = np.random.default_rng(n)
    p0 = rng.standard_normal(n).astype(np.float32)
    p_ref, m_ref = p0.copy(), np.zeros(n, np.float32)
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    topt = torch.optim.SGD([pt], 0.1, momentum=0.9, weight_decay=5e-4)
    pg, mg = _dev(p0, gpu), torch.zeros(n, device=gpu)
    lr = torch.zeros(1, device=gpu)
    shadow = torch.zeros(n, dtype=torch.bfloat16, device=gpu)
    for it, lr_v in enumerate([0.0, 0.025, 0.1]):   # warm-up starts at lr = 0 (main_perturb.py:288-293)
        g = rng.standard_normal(n).astype(np.float32)
        c_oracle.oracle_sgd_step(ptr(p_ref), ptr(g), ptr(m_ref), n, np.float32(lr_v), np.float32(0.9),
                                 np.float32(5e-4), np.float32(1.0))
        for grp in topt.param_groups:
            grp["lr"] = lr_v
        pt.grad = torch.from_numpy(g.copy())
        topt.step()
        lr.fill_(lr_v)
        pkg.ops.sgd_step_(pg, _dev(g, gpu), mg, lr, 0.9, 5e-4, 1.0, shadow)
        np.testing.assert_array_equal(pg.cpu().numpy(), p_ref)                      # C oracle: bit-exact
        np.testing.assert_allclose(pg.cpu().numpy(), pt.detach().numpy(), rtol=1e-6, atol=1e-7)  # torch: FMA-level
    np.testing.assert_array_equal(shadow.view(torch.int16).cpu().numpy(),
                                  pg.to(torch.bfloat16).view(torch.int16).cpu().numpy())


def test_ops_reject_cpu_tensors_and_missing_fallback(pkg):
    with pytest.raises(pkg.AfanLibraryError):
        pkg.ops.pgd_step_(torch.zeros(4), torch.zeros(4), 0.1)
    with pytest.raises(pkg.AfanLibraryError):
        pkg.attack_algo.PGD(torch.zeros(1, 2, 2, 2), None, model=None, gamma=0.1)


# ------------------------------------------------------------------------- channels-last (NHWC) variants
def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


NHWC_SHAPES = [(4, 16, 32, 32), (2, 64, 16, 16), (8, 512, 4, 4), (3, 304, 5, 7), (2, 3, 9, 9), (64, 128, 16, 16),
               (2, 1024, 3, 3), (1, 8, 1, 1),
               (2, 48, 33, 29), (5, 200, 6, 5)]     # generic mapping, several rows per block pass (DeepLab's 48-channel projection)


@pytest.mark.parametrize("shape", NHWC_SHAPES)
@pytest.mark.parametrize("res,relu", [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_bn_nhwc_matches_nchw_kernels(pkg, gpu, bn_mode, shape, res, relu, dt):
    """The channels-last kernels against the NCHW kernels (already pinned to the C oracle) on the same values."""
    torch.manual_seed(sum(shape) + relu)
    n, c = shape[:2]
    x = (torch.randn(shape) * 2 + 3).to(gpu, dt)
    r = torch.randn(shape).to(gpu, dt) if res else None
    w, b = (torch.rand(c) + 0.5).to(gpu), torch.randn(c).to(gpu)
    dy = torch.randn(shape).to(gpu, dt)
    outs = []
    for cl in (False, True):
        conv = _cl if cl else (lambda t: t)
        rm, rv = torch.zeros(c, device=gpu), torch.ones(c, device=gpu)
        nbt = torch.zeros((), dtype=torch.int64, device=gpu)
        xx = conv(x)
        y, stats = pkg.ops.bn_train_forward(xx, w, b, None if r is None else conv(r), relu, 1e-5, 0.1, rm, rv, nbt)
        mean, invstd = stats[0], stats[1]
        assert y.stride() == xx.stride() and int(nbt) == 1
        dwb = torch.zeros(2, c, device=gpu)
        dx, dres = pkg.ops.bn_backward(conv(dy), xx, y if (relu and res) else None, stats, w, b, relu, res,
                                       dwb[0], dwb[1])
        outs.append([t.float().cpu().numpy() for t in (y, mean, invstd, rm, rv, dx, dwb)] +
                    [dres.float().cpu().numpy() if res else None])
    a, bb = outs
    lo = dt == torch.bfloat16
    tol = dict(rtol=2e-2, atol=2e-2) if lo else dict(rtol=1e-4, atol=1e-4)
    for i, name in enumerate(["y", "mean", "invstd", "rmean", "rvar", "dx", "dwb"]):
        stat = name in ("mean", "invstd", "rmean", "rvar")
        t = dict(rtol=2e-5, atol=2e-6) if stat else tol
        if name == "dwb":
            t = dict(rtol=tol["rtol"], atol=tol["atol"] * max(1.0, float(np.abs(a[i]).max())))
        if name in ("dx", "y") and not stat:
            assert_close_frac(bb[i], a[i], t["rtol"], t["atol"], 1e-4, name)
        else:
            np.testing.assert_allclose(bb[i], a[i], err_msg=name, **t)
    if res:
        assert_close_frac(bb[7], a[7], 0, 0, 1e-4, "dres")


@pytest.mark.parametrize("shape", [(4, 64, 16, 16), (2, 256, 8, 8), (64, 128, 16, 16)])
def test_bn_backward_acc_self_reducing_matches_slab_path(pkg, gpu, shape):
    """afan_bn_backward_acc with acc_ready = 0 (takes its own sums with f64 atomics) against the slab + finalize path."""
    import ctypes
    torch.manual_seed(sum(shape))
    c = shape[1]
    x = _cl((torch.randn(shape) * 2 + 1).to(gpu, torch.bfloat16))
    dy = _cl(torch.randn(shape).to(gpu, torch.bfloat16))
    w, b = (torch.rand(c) + 0.5).to(gpu), torch.randn(c).to(gpu)
    old = pkg.ops.BN_ACC
    pkg.ops.BN_ACC = False
    try:
        y, stats = pkg.ops.bn_train_forward(x, w, b, None, True, 1e-5, 0.1, None, None, None)
        dwb0 = torch.zeros(2, c, device=gpu)
        dx0, _ = pkg.ops.bn_backward(dy, x, None, stats, w, b, True, False, dwb0[0], dwb0[1])
    finally:
        pkg.ops.BN_ACC = old
    lib = pkg._lib.load()
    acc = torch.zeros(lib.afan_bn_acc_doubles(c), dtype=torch.float64, device=gpu)
    dx1, dwb1 = torch.empty_like(x), torch.ones(2, c, device=gpu)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.afan_bn_backward_acc(P(dy), P(x), None, P(dx1), None, pkg._lib.AFAN_BF16, shape[0], c, shape[2] * shape[3],
                                  P(stats), 1, P(acc), 0, P(dwb1[0]), P(dwb1[1]), 1, 1,
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    scale = max(1.0, float(dwb0.abs().max()))
    np.testing.assert_allclose(dwb1.cpu().numpy() - 1.0, dwb0.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale)   # accumulate=1
    assert_close_frac(dx1.float().cpu().numpy(), dx0.float().cpu().numpy(), 2e-2, 2e-3, 1e-4, "dx")


def test_bn_nhwc_module_autograd_vs_torch(pkg, gpu, bn_mode):
    torch.manual_seed(4)
    x = torch.randn(6, 32, 10, 10) * 1.5 + 0.7
    res = torch.randn_like(x)
    ref_bn = torch.nn.BatchNorm2d(32)
    ref_bn.weight.data.uniform_(0.5, 1.5)
    ref_bn.bias.data.normal_()
    bn = pkg.resnet_s.BatchNorm2d(32)
    bn.load_state_dict(ref_bn.state_dict())
    bn.to(gpu)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    yr = torch.relu(ref_bn(xr) + rr)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    xg = _cl(x.to(gpu)).requires_grad_(True)
    rg = res.to(gpu).requires_grad_(True)            # NCHW residual: converted to the conv output's layout
    yg = bn.fused(xg, rg, True)
    assert yg.is_contiguous(memory_format=torch.channels_last)
    yg.backward(gy.to(gpu))                          # NCHW incoming gradient: converted too
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(rg.grad.cpu().numpy(), rr.grad.numpy())
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), ref_bn.weight.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref_bn.running_var.numpy(), rtol=1e-5)


def test_elementwise_kernels_accept_channels_last(pkg, gpu, c_oracle):
    """PGD step / norms / noise are layout-agnostic as long as all operands share strides; mixed strides raise."""
    torch.manual_seed(5)
    x = _cl(torch.randn(3, 8, 5, 5, device=gpu))
    g = _cl(torch.randn(3, 8, 5, 5, device=gpu))
    xa = x.clone()
    assert xa.stride() == x.stride()
    pkg.ops.pgd_step_(xa, g, 0.01, x, 0.02, True)
    ref = x.cpu().contiguous().numpy().copy()
    c_oracle.oracle_pgd_step(ptr(ref), ptr(g.cpu().contiguous().numpy()), ptr(x.cpu().contiguous().numpy()), ref.size,
                             np.float32(0.01), np.float32(0.02), 1)
    np.testing.assert_array_equal(xa.cpu().numpy(), ref)
    l2, linf = pkg.ops.perturb_norms(xa, x)
    d = (xa - x).reshape(3, -1)
    np.testing.assert_allclose(l2.cpu().numpy(), d.norm(dim=1).cpu().numpy(), rtol=1e-6)
    with pytest.raises(ValueError):
        pkg.ops.pgd_step_(xa, g.contiguous(), 0.01)
    img = torch.rand(2, 3, 6, 6, device=gpu)
    m, s = torch.tensor([0.4914, 0.4822, 0.4465], device=gpu), torch.tensor([0.2470, 0.2435, 0.2616], device=gpu)
    a = pkg.ops.normalize_nchw(img, m, s)
    b = pkg.ops.normalize_nchw(img, m, s, torch.float32, channels_last=True)
    assert b.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    np.testing.assert_array_equal(a.cpu().numpy(), ((img - m[None, :, None, None]) / s[None, :, None, None]).cpu().numpy())


@pytest.mark.parametrize("shape,k", [((32, 512, 4, 4), 10), ((6, 64, 8, 8), 10), ((3, 2048, 7, 7), 16), ((4, 64, 1, 1), 7),
                                     ((512, 512, 4, 4), 10), ((5, 24, 3, 3), 3)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fused_classifier_head_matches_torch(pkg, gpu, shape, k, dt):
    """afan_head_forward / _backward against pool -> flatten -> linear in torch fp32 on the same values."""
    torch.manual_seed(sum(shape) + k)
    x = torch.randn(shape, device=gpu).to(dt).contiguous(memory_format=torch.channels_last)
    lin = nn.Linear(shape[1], k).to(gpu)
    xr = x.float().detach().requires_grad_(True)
    ref = lin(xr.mean(dim=(2, 3)))
    g = torch.randn_like(ref)
    ref.backward(g)
    logits, pooled = pkg.ops.head_forward(x, lin.weight.detach(), lin.bias.detach())
    np.testing.assert_allclose(logits.cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    dw, db = torch.ones_like(lin.weight), torch.ones_like(lin.bias)
    dx = pkg.ops.head_backward(g, lin.weight.detach(), pooled, x, True, dw, db, accumulate=True)
    assert dx.dtype == dt and dx.stride() == x.stride()
    tol = dict(rtol=1e-5, atol=1e-6) if dt == torch.float32 else dict(rtol=1e-2, atol=1e-4)
    np.testing.assert_allclose(dx.float().cpu().numpy(), xr.grad.cpu().numpy(), **tol)
    np.testing.assert_allclose(dw.cpu().numpy() - 1, lin.weight.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy() - 1, lin.bias.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,k", [(256, 10), (512, 10), (7, 3), (1, 10), (64, 1000), (3, 100), (65, 64), (17, 63)])
def test_fused_cross_entropy_matches_torch(pkg, gpu, n, k):
    """afan_cross_entropy (loss and d(loss)/d(logits) in one launch) against nn.CrossEntropyLoss on the same logits."""
    torch.manual_seed(n + k)
    logits = (torch.randn(n, k, device=gpu) * 4).requires_grad_(True)
    y = torch.randint(0, k, (n,), device=gpu)
    ref = nn.CrossEntropyLoss()(logits, y)
    (gref,) = torch.autograd.grad(ref * 0.5, logits)
    loss, dl = pkg.ops.cross_entropy(logits.detach(), y)
    np.testing.assert_allclose(float(loss.detach()), float(ref.detach()), rtol=2e-6)
    # softmax - onehot cancels where the target's probability is near 1: absolute tolerance of a few ulps of 1/n
    np.testing.assert_allclose(dl.cpu().numpy() * 0.5, gref.cpu().numpy(), rtol=1e-5, atol=1e-6 / n)
    # through autograd, as the step uses it (criterion pattern-matched on the bf16 path only)
    class _M:
        compute_dtype = torch.bfloat16
    crit = pkg.resnet_s.fused_criterion(nn.CrossEntropyLoss(), _M())
    l2 = crit(logits, y)
    (g2,) = torch.autograd.grad(l2 * 0.5, logits)
    np.testing.assert_allclose(g2.cpu().numpy(), gref.cpu().numpy(), rtol=1e-5, atol=1e-6 / n)
    _M.compute_dtype = torch.float32
    assert isinstance(pkg.resnet_s.fused_criterion(nn.CrossEntropyLoss(), _M()), nn.CrossEntropyLoss)
    assert isinstance(pkg.resnet_s.fused_criterion(nn.CrossEntropyLoss(label_smoothing=0.1), _M()), nn.CrossEntropyLoss)
