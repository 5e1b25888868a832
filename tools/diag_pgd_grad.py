"""The feature gradient PGD takes the sign of, at every step, in three fp32 runs of the DeepLab golden iteration: channels-last,
channels-last with BatchNorm forward by the NCHW kernels, NCHW.  How far apart are the gradients themselves? (diagnostic)"""
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
f0 = ops.bn_train_forward
p0, pn0 = ops.pgd_step_, ops.pgd_step_norms_
CL = torch.channels_last
MODE, grads = ["plain"], {}


def fwd(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    if MODE[0] == "swap" and x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32:
        y, st = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm, rv, nb)
        st = st.clone()
        st[2].copy_(st[1] * weight)
        st[3].copy_(torch.addcmul(bias, st[0], st[2], value=-1.0))
        return y.contiguous(memory_format=CL), st
    return f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)


def rec(grad, x_adv):
    grads.setdefault(KEY[0], []).append((grad.detach().float().contiguous().cpu().numpy().copy(), x_adv.detach().float().contiguous().cpu().numpy().copy()))


def pstep(x_adv, grad, *a, **k):
    rec(grad, x_adv)
    return p0(x_adv, grad, *a, **k)


def pstepn(x_adv, grad, *a, **k):
    rec(grad, x_adv)
    return pn0(x_adv, grad, *a, **k)


ops.bn_train_forward, ops.pgd_step_, ops.pgd_step_norms_ = fwd, pstep, pstepn
for m in ("attack_algo", "seg_attack_algo", "det_attack_algo"):
    mod = getattr(pkg, m, None)
    if mod is not None and hasattr(mod, "ops"):
        pass    # (they call ops.pgd_step_ through the module attribute)
KEY = [None]
images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
for key, nhwc, mode in (("nhwc", True, "plain"), ("nhwc+nchwBN", True, "swap"), ("nchw", False, "plain")):
    KEY[0], MODE[0] = key, mode
    model, tr = T._build(pkg, g, torch.float32, nhwc, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
    tr.step(images, labels)
    torch.cuda.synchronize()
    print(key, "recorded", len(grads[key]), "PGD steps", [a.shape for a, _ in grads[key]][:2])
ref = grads["nchw"]
for key in ("nhwc", "nhwc+nchwBN"):
    for i, ((ga, xa), (gb, xb)) in enumerate(zip(grads[key], ref)):
        if ga.shape != gb.shape:
            continue
        dx = np.abs(xa - xb).max()
        rel = np.linalg.norm((ga - gb).ravel()) / np.linalg.norm(gb.ravel())
        flips = float((np.sign(ga) != np.sign(gb)).mean())
        sc = np.abs(gb).mean()
        bad = np.sign(ga) != np.sign(gb)
        print(f"{key} vs nchw, PGD call {i} {ga.shape}: iterate max diff {dx:.2e}; gradient rel l2 diff {rel:.2e}; sign flips {flips:.2e}; "
              f"median |g| at flips / mean |g| {np.median(np.abs(gb[bad])) / sc if bad.any() else 0:.2e}")
