"""bf16 / fp32 channels-last BatchNorm backward: slab + finalize path vs accumulator path vs float64 (probe)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
ops, lib = pkg.ops, pkg._lib.load()
gpu = torch.device("cuda:0")
CL = torch.channels_last


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30))


for dtype in (torch.bfloat16, torch.float32):
    for (n, c, h, w, have_y) in ((16, 128, 16, 16, True), (16, 64, 32, 32, False), (16, 512, 4, 4, True)):
        g = torch.Generator().manual_seed(c)
        x = (torch.randn(n, c, h, w, generator=g) * 0.7 + 0.3).to(dtype).to(gpu).contiguous(memory_format=CL)
        res = torch.randn(n, c, h, w, generator=g).to(dtype).to(gpu).contiguous(memory_format=CL) if have_y else None
        wt, bs = (torch.rand(c, generator=g) + 0.5).to(gpu), (torch.randn(c, generator=g) * 0.1).to(gpu)
        dy = (torch.randn(n, c, h, w, generator=g) * 1e-3).to(dtype).to(gpu).contiguous(memory_format=CL)
        rm, rv, nb = torch.zeros(c, device=gpu), torch.ones(c, device=gpu), torch.zeros((), dtype=torch.int64, device=gpu)
        y, st = ops.bn_train_forward(x, wt, bs, res, True, 1e-5, 0.1, rm, rv, nb)
        yy = y if have_y else None
        dw1, db1 = torch.zeros(c, device=gpu), torch.zeros(c, device=gpu)
        dx1, _ = ops.bn_backward(dy, x, yy, st, wt, bs, True, have_y, dw1, db1)              # slab + finalize
        acc = ops.acc_take(gpu, c)
        dx2 = torch.empty_like(x)
        dres2 = torch.empty_like(x) if have_y else None
        dw2, db2 = torch.zeros(c, device=gpu), torch.zeros(c, device=gpu)
        pkg._lib.check(lib.afan_bn_backward_acc(ops._ptr(dy), ops._ptr(x), ops._ptr(yy), ops._ptr(dx2), ops._ptr(dres2), ops._DT[dtype], n, c, h * w,
                                                ops._ptr(st), 1, ops._ptr(acc), 0, ops._ptr(dw2), ops._ptr(db2), 0, 1, ops._stream(x)), "acc")
        # float64 on the stored values
        x64, dy64 = x.double(), dy.double()
        mu, isd, al = st[0].double().view(1, -1, 1, 1), st[1].double().view(1, -1, 1, 1), st[2].double().view(1, -1, 1, 1)
        mask = (y > 0) if have_y else ((x64 * al + st[3].double().view(1, -1, 1, 1)) > 0)
        gm = dy64 * mask
        xh = (x64 - mu) * isd
        M = n * h * w
        dx64 = (gm - gm.sum(dim=(0, 2, 3), keepdim=True) / M - xh * (gm * xh).sum(dim=(0, 2, 3), keepdim=True) / M) * al
        print(f"{dtype} [{n},{c},{h},{w}] y={have_y}: slab dx vs f64 {rel(dx1, dx64):.2e}  acc dx vs f64 {rel(dx2, dx64):.2e}  slab vs acc {rel(dx1, dx2):.2e} "
              f"elements differing {float((dx1 != dx2).float().mean()):.2e}  dw {rel(dw1, dw2):.1e} db {rel(db1, db2):.1e}")
