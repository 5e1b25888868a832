"""Every fp32 channels-last BatchNorm forward of one DeepLab iteration against float64, for the NHWC and the NCHW kernels (diagnostic)."""
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
rows = []


def fwd(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
    cmp = x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and conv_stats is None
    if cmp:
        rm2, rv2, nb2 = rm.clone(), rv.clone(), nb.clone()
        y2, st2 = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm2, rv2, nb2)
    y, st = f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)
    if cmp:
        x64 = x.double()
        mu = x64.mean(dim=(0, 2, 3), keepdim=True)
        var = x64.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        y64 = (x64 - mu) / torch.sqrt(var + eps) * weight.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1)
        pre = y64
        if residual is not None:
            y64 = y64 + residual.double()
        if relu:
            y64 = y64.relu()
        scale = pre.std(dim=(0, 2, 3)).clamp_min(1e-30)                      # per-channel scale of the normalised signal
        e1 = ((y.double() - y64).abs().amax(dim=(0, 2, 3)) / scale)
        e2 = ((y2.double() - y64).abs().amax(dim=(0, 2, 3)) / scale)
        flips1 = float(((y > 0) != (y64 > 0)).double().mean()) if relu else 0.0
        flips2 = float(((y2 > 0) != (y64 > 0)).double().mean()) if relu else 0.0
        c = int(e1.argmax())
        rows.append((tuple(x.shape), residual is not None, float(e1.max()), float(e2.max()), float(e1.mean()), float(e2.mean()), flips1, flips2,
                     float(mu.flatten()[c] / var.flatten()[c].sqrt()), float(var.flatten()[c])))
    return y, st


ops.bn_train_forward = fwd
model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
r = tr.step(images, labels)
torch.cuda.synchronize()
print(len(rows), "forward calls;  columns: shape res | max err/scale NHWC, NCHW | mean err/scale NHWC, NCHW | relu flips NHWC, NCHW | mean/std, var of worst channel")
for t in sorted(rows, key=lambda t: -t[2])[:14]:
    print(t[0], t[1], "%.2e %.2e | %.2e %.2e | %.2e %.2e | %.2f %.3e" % t[2:])
print("totals: mean of max err NHWC %.3e NCHW %.3e; mean err NHWC %.3e NCHW %.3e; flips NHWC %.3e NCHW %.3e" % (
    np.mean([t[2] for t in rows]), np.mean([t[3] for t in rows]), np.mean([t[4] for t in rows]), np.mean([t[5] for t in rows]),
    np.mean([t[6] for t in rows]), np.mean([t[7] for t in rows])))
