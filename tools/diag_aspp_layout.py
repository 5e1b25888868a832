"""fp32 ASPP / decoder pieces: channels-last vs NCHW product results on identical data (diagnostic)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
dl = pkg.deeplab
gpu = torch.device("cuda:0")
torch.manual_seed(0)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def run(mod, x, gy, cl):
    mod.zero_grad()
    xd = x.clone().to(gpu)
    if cl:
        xd = xd.contiguous(memory_format=torch.channels_last)
    xd.requires_grad_(True)
    y = mod(xd)
    y.backward(gy.to(gpu))
    return y.detach().cpu(), xd.grad.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in mod.named_parameters() if p.grad is not None}


aspp = dl.ASPP(2048, (6, 12, 18)).to(gpu).train()
aspp.project[3].p = 0.0
x = torch.randn(4, 2048, 9, 9).relu()
y0 = None
res = {}
for cl in (False, True):
    gy = torch.randn(4, 256, 9, 9, generator=torch.Generator().manual_seed(1))
    res[cl] = run(aspp, x, gy, cl)
print("ASPP y", rel(res[True][0], res[False][0]), "dx", rel(res[True][1], res[False][1]))
for n in res[True][2]:
    e = rel(res[True][2][n], res[False][2][n])
    if e > 1e-4:
        print("   param grad", n, e)
# branch by branch
for bi in range(5):
    br = aspp.convs[bi]
    f = (lambda t, br=br: dl._cbr(br[0], br[1], t)) if bi == 0 else br

    class W(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.br = br

        def forward(self, t):
            return f(t)
    w = W()
    r = {}
    for cl in (False, True):
        gy = torch.randn(4, 256, 9, 9, generator=torch.Generator().manual_seed(2))
        r[cl] = run(w, x, gy, cl)
    print(f"branch {bi}: y {rel(r[True][0], r[False][0]):.2e} dx {rel(r[True][1], r[False][1]):.2e}",
          {n: f"{rel(r[True][2][n], r[False][2][n]):.1e}" for n in r[True][2]})
# raw dgrad, both layouts, against float64 on the CPU
import torch.nn.functional as F
for (ci, co, k, dil, hw) in ((2048, 256, 3, 6, 9), (2048, 256, 3, 12, 9), (2048, 256, 1, 1, 9), (1280, 256, 1, 1, 9), (304, 256, 3, 1, 33)):
    wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
    gy = torch.randn(2, co, hw, hw)
    pad = dil * (k // 2)
    x0 = torch.zeros(2, ci, hw, hw, dtype=torch.float64, requires_grad=True)
    F.conv2d(x0, wt.double(), None, 1, pad, dil).backward(gy.double())
    for cl in (False, True):
        mf = torch.channels_last if cl else torch.contiguous_format
        dx = pkg.ops.conv_general_dgrad(gy.to(gpu).contiguous(memory_format=mf), wt.to(gpu).contiguous(memory_format=mf), (hw, hw), 1, pad, dil)
        print(f"dgrad ci={ci} co={co} k={k} dil={dil} cl={cl}: rel err {rel(dx.cpu(), x0.grad):.2e}")
