"""Two channels-last fp32 DeepLab iterations from the same state, BatchNorm forward by the NHWC kernels vs by the NCHW kernels:
call by call, where do the outputs first part ways? (diagnostic)"""
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
CL = torch.channels_last
trace, MODE, idx = [], ["rec"], [0]


def fwd(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    hooked = x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32
    if hooked and MODE[0] == "swap":
        y, st = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm, rv, nb)
        st = st.clone()
        st[2].copy_(st[1] * weight)
        st[3].copy_(torch.addcmul(bias, st[0], st[2], value=-1.0))
        y = y.contiguous(memory_format=CL)
    else:
        y, st = f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)
    if hooked:
        i = idx[0]
        idx[0] += 1
        if MODE[0] == "rec":
            if i < 400:
                trace.append((x.detach().clone(), y.detach().clone()))
        elif i < len(trace):
            x0, y0 = trace[i]
            sc = float(y0.abs().max()) + 1e-30
            dx = float((x - x0).abs().max()) / (float(x0.abs().max()) + 1e-30)
            dy = float((y - y0).abs().max()) / sc
            nz = float(((y > 0) != (y0 > 0)).float().mean())
            if i < 12 or dy > 3e-6 or i % 40 == 0:
                print(f"call {i} {tuple(x.shape)} res={residual is not None}: input diff {dx:.2e} output diff {dy:.2e} relu-mask diff {nz:.2e}")
    return y, st


ops.bn_train_forward = fwd
images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
for mode in ("rec", "swap"):
    MODE[0], idx[0] = mode, 0
    model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
    r = tr.step(images, labels)
    torch.cuda.synchronize()
