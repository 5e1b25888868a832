"""Which channels-last fp32 BatchNorm calls cost the DeepLab step its perturbation agreement?  Route subsets of them through the
NCHW kernels (diagnostic)."""
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
SEL = {"fwd": lambda x, res: False, "bwd": lambda x, res: False}


def fwd(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    if x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and conv_stats is None and SEL["fwd"](x, residual is not None):
        y, st = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm, rv, nb)
        return y.contiguous(memory_format=CL), st
    return f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)


def bwd(dy, x, y, stats, weight, bias, relu, want_dres, dweight=None, dbias=None, accumulate=False, partials=None, dx_out=None, dres_out=None, groups=1):
    if x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and partials is None and SEL["bwd"](x, y is not None):
        dx, dres = b0(dy.contiguous(), x.contiguous(), None if y is None else y.contiguous(), stats, weight, bias, relu, want_dres, dweight, dbias, accumulate)
        return dx.contiguous(memory_format=CL), None if dres is None else dres.contiguous(memory_format=CL)
    return b0(dy, x, y, stats, weight, bias, relu, want_dres, dweight, dbias, accumulate, partials, dx_out, dres_out, groups)


ops.bn_train_forward, ops.bn_backward = fwd, bwd
images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
gam = float(g["gammas"][0]) / 255
k_ref = np.rint((g["adv_se"] - g["fm_se"]) / gam)
yes, no = (lambda x, r: True), (lambda x, r: False)
big = lambda x, r: x.shape[1] > 1024
small = lambda x, r: x.shape[1] <= 1024
for name, sf, sb in (("none", no, no), ("fwd all", yes, no), ("bwd all", no, yes), ("fwd C>1024", big, no), ("fwd C<=1024", small, no),
                     ("bwd C>1024", no, big), ("bwd C<=1024", no, small), ("bwd with res", no, lambda x, r: r), ("bwd no res", no, lambda x, r: not r),
                     ("fwd with res", lambda x, r: r, no), ("fwd no res", lambda x, r: not r, no)):
    SEL["fwd"], SEL["bwd"] = sf, sb
    model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
    r = tr.step(images, labels)
    k_got = np.rint((r["adv_se"].float().cpu().numpy() - r["fm_se"].float().cpu().numpy()) / gam)
    print(f"NCHW kernels for [{name}]: agreement {float((k_got == k_ref).mean()):.5f}")

# ---- ideal forward: the NHWC kernels' statistics, but y computed in float64 and rounded once
def fwd_ideal(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    y, st = f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)
    if x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and conv_stats is None and MODE[0]:
        x64 = x.double()
        if MODE[0] == "f64stats":
            mu = x64.mean(dim=(0, 2, 3), keepdim=True)
            var = x64.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            y64 = (x64 - mu) / torch.sqrt(var + eps) * weight.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
        else:                                    # the kernel's own fp32 mean / invstd, exact arithmetic from there
            y64 = (x64 - st[0].double().view(1, -1, 1, 1)) * st[1].double().view(1, -1, 1, 1) * weight.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
        if residual is not None:
            y64 = y64 + residual.double()
        if relu:
            y64 = y64.relu()
        y = y64.float().contiguous(memory_format=CL)
    return y, st


MODE = [None]
ops.bn_train_forward, ops.bn_backward = fwd_ideal, b0
for mode in ("f64stats", "kernelstats"):
    MODE[0] = mode
    model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
    r = tr.step(images, labels)
    k_got = np.rint((r["adv_se"].float().cpu().numpy() - r["fm_se"].float().cpu().numpy()) / gam)
    print(f"ideal BatchNorm forward [{mode}]: agreement {float((k_got == k_ref).mean()):.5f}")

# ---- NHWC kernels' y, but the statistics block handed to the backward taken from elsewhere
def fwd_stats(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    y, st = f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)
    if x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and conv_stats is None:
        if MODE[0] == "nchw_stats":
            _, st2 = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm.clone(), rv.clone(), nb.clone())
            st = st.clone()
            st[0].copy_(st2[0]); st[1].copy_(st2[1])
            st[2].copy_(st[1] * weight); st[3].copy_(bias - st[0] * st[2])
        elif MODE[0] == "f64_stats":
            x64 = x.double()
            mu = x64.mean(dim=(0, 2, 3)); var = x64.var(dim=(0, 2, 3), unbiased=False)
            is_ = (1.0 / torch.sqrt(var + eps)).float()
            st = st.clone()
            st[0].copy_(mu.float()); st[1].copy_(is_)
            st[2].copy_(is_ * weight); st[3].copy_(bias - mu.float() * st[2])
    return y, st


ops.bn_train_forward = fwd_stats
for mode in ("nchw_stats", "f64_stats"):
    MODE[0] = mode
    model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
    r = tr.step(images, labels)
    k_got = np.rint((r["adv_se"].float().cpu().numpy() - r["fm_se"].float().cpu().numpy()) / gam)
    print(f"NHWC forward values, statistics block from [{mode}]: agreement {float((k_got == k_ref).mean()):.5f}")

# ---- full swap to the NCHW kernels, with / without ALSO executing the NHWC kernel (result discarded): a side effect?
def fwd_side(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    if x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and conv_stats is None:
        if MODE[0] == "both":
            f0(x, weight, bias, residual, relu, eps, momentum, rm.clone(), rv.clone(), nb.clone())
        y, st = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm, rv, nb)
        return y.contiguous(memory_format=CL), st
    return f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)


ops.bn_train_forward = fwd_side
SEL["bwd"] = yes
ops.bn_backward = bwd
for mode in ("swap", "both"):
    MODE[0] = mode
    model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
    r = tr.step(images, labels)
    k_got = np.rint((r["adv_se"].float().cpu().numpy() - r["fm_se"].float().cpu().numpy()) / gam)
    print(f"full swap to NCHW kernels [{mode}]: agreement {float((k_got == k_ref).mean()):.5f}")

# ---- NCHW forward kernels (statistics block completed with alpha / beta), NHWC backward kernels
def fwd_fix(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    if x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and conv_stats is None:
        y, st = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm, rv, nb)
        st = st.clone()
        print_once(st)
        st[2].copy_(st[1] * weight)
        st[3].copy_(torch.addcmul(bias, st[0], st[2], value=-1.0))
        return y.contiguous(memory_format=CL), st
    return f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)


_seen = [False]


def print_once(st):
    if not _seen[0]:
        _seen[0] = True
        print("   NCHW forward's statistics block rows 2, 3 (first 4):", st[2][:4].tolist(), st[3][:4].tolist())


ops.bn_train_forward, ops.bn_backward = fwd_fix, b0
model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
r = tr.step(images, labels)
k_got = np.rint((r["adv_se"].float().cpu().numpy() - r["fm_se"].float().cpu().numpy()) / gam)
print(f"NCHW forward + NHWC backward: agreement {float((k_got == k_ref).mean()):.5f}")
