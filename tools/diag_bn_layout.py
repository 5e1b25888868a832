"""fp32 BatchNorm (+residual, ReLU) forward / backward: error of the NHWC and the NCHW kernels against float64 (diagnostic)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
ops = pkg.ops
gpu = torch.device("cuda:0")


def rel(a, b):
    return float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))


for (n, c, h, w, off, res, relu) in ((4, 64, 65, 65, 0.0, False, True), (4, 256, 33, 33, 3.0, True, True), (4, 2048, 9, 9, 1.0, True, True),
                                     (4, 256, 9, 9, 0.5, False, True), (4, 48, 33, 33, 5.0, False, True), (2, 16, 32, 32, 0.0, True, True)):
    g = torch.Generator().manual_seed(n * c + h)
    x = torch.randn(n, c, h, w, generator=g) * (torch.rand(1, c, 1, 1, generator=g) + 0.2) + off * torch.randn(1, c, 1, 1, generator=g)
    r = torch.randn(n, c, h, w, generator=g) if res else None
    wt, bs = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    dy = torch.randn(n, c, h, w, generator=g)
    # float64 reference
    x64 = x.double().requires_grad_(True)
    w64, b64 = wt.double().requires_grad_(True), bs.double().requires_grad_(True)
    r64 = None if r is None else r.double().requires_grad_(True)
    mu = x64.mean(dim=(0, 2, 3), keepdim=True)
    var = x64.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    y64 = (x64 - mu) / torch.sqrt(var + 1e-5) * w64.view(1, c, 1, 1) + b64.view(1, c, 1, 1)
    if r64 is not None:
        y64 = y64 + r64
    if relu:
        y64 = y64.relu()
    y64.backward(dy.double())
    for cl in (False, True):
        mf = torch.channels_last if cl else torch.contiguous_format
        xd = x.to(gpu).contiguous(memory_format=mf)
        rd = None if r is None else r.to(gpu).contiguous(memory_format=mf)
        rm, rv, nb = torch.zeros(c, device=gpu), torch.ones(c, device=gpu), torch.zeros((), dtype=torch.int64, device=gpu)
        y, st = ops.bn_train_forward(xd, wt.to(gpu), bs.to(gpu), rd, relu, 1e-5, 0.1, rm, rv, nb)
        dw, db = torch.zeros(c, device=gpu), torch.zeros(c, device=gpu)
        dx, dres = ops.bn_backward(dy.to(gpu).contiguous(memory_format=mf), xd, y if (relu and rd is not None) else None, st, wt.to(gpu), bs.to(gpu),
                                   relu, rd is not None, dw, db)
        print(f"[{n},{c},{h},{w}] off={off} res={res} cl={cl}: y {rel(y, y64.detach()):.2e} mean {rel(st[0], mu.flatten().detach()):.2e} "
              f"invstd {rel(st[1], (1 / torch.sqrt(var + 1e-5)).flatten().detach()):.2e} dx {rel(dx, x64.grad):.2e} "
              f"dw {rel(dw, w64.grad):.2e} db {rel(db, b64.grad):.2e}")
