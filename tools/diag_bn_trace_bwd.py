"""Backward trace: BatchNorm backward calls of the channels-last fp32 DeepLab iteration, NHWC forward kernels vs NCHW forward kernels:
the first call whose incoming gradient agrees and whose outgoing gradient does not (diagnostic)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, load_pkg  # noqa: E402
import test_deeplab_gpu as T  # noqa: E402

pkg = load_pkg()
ops = pkg.ops
gpu = torch.device("cuda:0")
g = golden("seg_dl101_aspp_k3_damped")
f0, b0 = ops.bn_train_forward, ops.bn_backward
CL = torch.channels_last
trace, MODE, idx = [], ["rec"], [0]


def fwd(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    if MODE[0] == "swap" and x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32:
        y, st = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm, rv, nb)
        st = st.clone()
        st[2].copy_(st[1] * weight)
        st[3].copy_(torch.addcmul(bias, st[0], st[2], value=-1.0))
        return y.contiguous(memory_format=CL), st
    return f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bwd(dy, x, y, stats, weight, bias, relu, want_dres, dweight=None, dbias=None, accumulate=False, partials=None, dx_out=None, dres_out=None, groups=1):
    dx, dres = b0(dy, x, y, stats, weight, bias, relu, want_dres, dweight, dbias, accumulate, partials, dx_out, dres_out, groups)
    i = idx[0]
    idx[0] += 1
    if i < 120:
        if MODE[0] == "rec":
            trace.append((dy.clone(), x.clone(), None if y is None else y.clone(), stats.clone(), dx.clone()))
        else:
            dy0, x0, y0, st0, dx0 = trace[i]
            if i == 0:
                mA = torch.addcmul(st0[3].view(1, -1, 1, 1), x, st0[2].view(1, -1, 1, 1)) > 0
                mB = torch.addcmul(stats[3].view(1, -1, 1, 1), x, stats[2].view(1, -1, 1, 1)) > 0
                dA, _ = b0(dy, x, y, st0, weight, bias, relu, want_dres)
                dB, _ = b0(dy, x, y, stats, weight, bias, relu, want_dres)
                print("   call 0: masks from the two statistics blocks differ on", float((mA != mB).double().mean()), "; dx(A stats) vs dx(B stats) on B's tensors:", rel(dA, dB),
                      "; dx(A stats, B tensors) vs recorded A dx:", rel(dA, dx0), " weight given:", weight is not None)
                d1, _ = b0(dy0, x, y, st0, weight, bias, relu, want_dres)
                d2, _ = b0(dy, x0, y, st0, weight, bias, relu, want_dres)
                print("   only x from B:", rel(d1, dx0), " only dy from B:", rel(d2, dx0), " max|dy-dy0|/max|dy0|:", float((dy - dy0).abs().max() / dy0.abs().max()),
                      " |dy| mean", float(dy0.abs().mean()), "max", float(dy0.abs().max()), " |dx| mean", float(dx0.abs().mean()), "max", float(dx0.abs().max()))
                gm = (dy0 * mA).double()
                print("   per-channel |mean g| / mean |g| (median over channels):", float((gm.mean(dim=(0, 2, 3)).abs() / gm.abs().mean(dim=(0, 2, 3)).clamp_min(1e-30)).median()))
                print("   stats A rows:", [st0[k][:3].tolist() for k in range(4)], "\n   stats B rows:", [stats[k][:3].tolist() for k in range(4)])
            print(f"bwd call {i} {tuple(x.shape)} y={'yes' if y is not None else 'no'} relu={relu}: dy {rel(dy0, dy):.2e} x {rel(x0, x):.2e} "
                  f"y {0.0 if y is None else rel(y0, y):.2e} mean {rel(st0[0], stats[0]):.2e} invstd {rel(st0[1], stats[1]):.2e} "
                  f"alpha {rel(st0[2], stats[2]):.2e} beta {rel(st0[3], stats[3]):.2e} -> dx {rel(dx0, dx):.2e}")
    return dx, dres


ops.bn_train_forward, ops.bn_backward = fwd, bwd
images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
for mode in ("rec", "swap"):
    MODE[0], idx[0] = mode, 0
    model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
    tr.step(images, labels)
    torch.cuda.synchronize()

# ---- call 0 in detail: both statistics blocks applied to the SAME (dy, x) by the NHWC backward kernel, against float64
dy0, x0, y0, st0, dx0 = trace[0]
print("call 0 detail:", tuple(x0.shape))
w_ = None
for m in model.modules():
    if isinstance(m, pkg.resnet_s.BatchNorm2d) and m.num_features == x0.shape[1]:
        w_ = m
x64, dy64 = x0.double(), dy0.double()
mu = x64.mean(dim=(0, 2, 3), keepdim=True)
var = x64.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
is64 = 1 / torch.sqrt(var + 1e-5)
alpha_k, beta_k = st0[2].double().view(1, -1, 1, 1), st0[3].double().view(1, -1, 1, 1)
act_k = x64 * alpha_k + beta_k
print("   fraction of |act| < 1e-5:", float((act_k.abs() < 1e-5).double().mean()), " < 1e-3:", float((act_k.abs() < 1e-3).double().mean()),
      " exactly-equal x values (first channel):", int(x0[:, 0].numel() - x0[:, 0].unique().numel()))
act32 = torch.addcmul(st0[3].view(1, -1, 1, 1), x0, st0[2].view(1, -1, 1, 1))
print("   mask(fp32 fma-free recompute) vs mask(f64 with kernel alpha/beta) differ on", float(((act32 > 0) != (act_k > 0)).double().mean()))
w = (alpha_k / st0[1].double().view(1, -1, 1, 1))
g = dy64 * (act_k > 0)
xh = (x64 - st0[0].double().view(1, -1, 1, 1)) * st0[1].double().view(1, -1, 1, 1)
M = x64.numel() / x64.shape[1]
dx64 = (g - g.sum(dim=(0, 2, 3), keepdim=True) / M - xh * (g * xh).sum(dim=(0, 2, 3), keepdim=True) / M) * alpha_k
print("   recorded NHWC dx vs float64 formula (same mask rule):", float((dx0.double() - dx64).norm() / dx64.norm()))
dxb, _ = b0(dy0, x0, None, st0, None, None, True, False)
print("   kernel re-run on the recorded tensors vs recorded dx:", float((dxb - dx0).norm() / dx0.norm()))
dxn, _ = b0(dy0.contiguous(), x0.contiguous(), None, st0, None, None, True, False)
print("   NCHW backward kernel on the same tensors vs float64:", float((dxn.double() - dx64).norm() / dx64.norm()))
