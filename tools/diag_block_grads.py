"""Bottleneck chains of the DeepLab backbone, bf16 channels-last (resnet_s._BlockFn) against the fp32 CPU oracle: forward,
input gradient and every parameter gradient, for 1..3 blocks and several shapes."""
import importlib
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cv_a-fan_amd")
from oracle import afan_oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
dl = pkg.deeplab


def chain(kind, inpl, planes, nblk, stride, dil, damp):
    def mk(B, conv, bn):
        blocks, cin = [], inpl
        for i in range(nblk):
            ds = None
            st = stride if i == 0 else 1
            if st != 1 or cin != planes * 4:
                ds = nn.Sequential(conv(cin, planes * 4, 1, st, bias=False) if kind == "ref" else conv(cin, planes * 4, kernel_size=1, stride=st, bias=False), bn(planes * 4))
            blocks.append(B(cin, planes, st, ds, dil))
            cin = planes * 4
        return nn.Sequential(*blocks)
    if kind == "ref":
        return mk(orc.SegBottleneck, nn.Conv2d, nn.BatchNorm2d)
    return mk(dl.Bottleneck, pkg.resnet_s.Conv2d, pkg.resnet_s.BatchNorm2d)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / max(b.norm().item(), 1e-12)).item()


for (inpl, planes, nblk, stride, dil, hw, damp) in [(256, 64, 1, 1, 1, 33, 0.1), (64, 64, 1, 1, 1, 33, 0.1), (256, 64, 2, 1, 1, 33, 0.1),
                                                    (256, 64, 3, 1, 1, 33, 1.0), (256, 128, 2, 2, 1, 33, 0.1), (1024, 512, 3, 1, 2, 9, 0.1)]:
    torch.manual_seed(1)
    ref = chain("ref", inpl, planes, nblk, stride, dil, damp)
    m = chain("own", inpl, planes, nblk, stride, dil, damp)
    m.load_state_dict(ref.state_dict())
    for b in list(ref) + list(m):
        b.bn3.weight.data.mul_(damp)
    ref.train()

    class W(nn.Module):
        def __init__(s, seq):
            super().__init__()
            s.seq = seq
            s.compute_dtype, s.channels_last = torch.bfloat16, True
    w = W(m)
    for mod in m.modules():
        if isinstance(mod, pkg.resnet_s.Conv2d):
            mod.compute_dtype = torch.bfloat16
    w.to(dev).train()
    arena = pkg.arena.ParamArena(w, skip=())
    x = torch.relu(torch.randn(2, inpl, hw, hw))
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    xg = x.to(dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    pkg.ops.acc_reset(dev)
    arena.zero_grad()
    y = m(xg)
    y.backward(gy.to(dev).bfloat16().contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    print(f"in {inpl} planes {planes} blocks {nblk} stride {stride} dil {dil} hw {hw} damp {damp}: y {rel(y, yr):.3e}  dx {rel(xg.grad, xr.grad):.3e}")
    pr = dict(ref.named_parameters())
    worst = sorted(((rel(p.grad, pr[n].grad), n) for n, p in m.named_parameters()), reverse=True)[:4]
    print("   worst param grads:", [(f"{e:.3e}", n) for e, n in worst])
