"""Error statistics of the tiled convolution (forward, input gradient) on DeepLab's shapes against an fp32 convolution of the
same bf16 operands; run under two builds (AFAN_HIP_LIB) to see what a change of the K order does: rounding flips only
(profiles/r03j_conv_order_check.txt).
    [AFAN_HIP_LIB=...] python tools/probe/conv_order_check.py"""
import importlib, sys, os, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
# n, ci, co, h, k, stride, dilation
SH = [(2, 304, 256, 129, 3, 1, 1), (2, 256, 256, 129, 3, 1, 1), (2, 256, 48, 129, 1, 1, 1), (2, 2048, 256, 33, 3, 1, 12), (2, 2048, 256, 33, 1, 1, 1),
      (2, 512, 512, 33, 3, 1, 2), (2, 512, 512, 33, 3, 1, 4), (2, 256, 256, 33, 3, 1, 1), (2, 128, 128, 65, 3, 1, 1), (2, 128, 128, 129, 3, 2, 1),
      (2, 1280, 256, 33, 1, 1, 1), (2, 64, 64, 129, 3, 1, 1), (2, 1024, 2048, 33, 1, 1, 1), (2, 256, 512, 65, 1, 2, 1), (2, 1024, 256, 33, 1, 1, 1)]
for (n, ci, co, h, k, s, d) in SH:
    torch.manual_seed(ci + co + h + k + d)
    x = cl(torch.randn(n, ci, h, h, device=dev).bfloat16())
    w = cl((torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5).bfloat16())
    y = pkg.ops.conv_fwd(x, w, s, dilation=d) if d > 1 else pkg.ops.conv_fwd(x, w, s)
    ref = F.conv2d(x.float(), w.float(), None, s, d * (k // 2), d)
    e = (y.float() - ref)
    dy = cl(torch.randn_like(ref).bfloat16())
    wt = cl(w.permute(1, 0, 2, 3))
    dx = pkg.ops.conv_dgrad(dy, wt, (h, h), s, dilation=d)
    refd = torch.nn.grad.conv2d_input((n, ci, h, h), w.float(), dy.float(), stride=s, padding=d * (k // 2), dilation=d)
    ed = dx.float() - refd
    print(f"{(n, ci, co, h, k, s, d)}: fwd max {e.abs().max().item():.4e} mean {e.abs().mean().item():.4e} sum {y.float().sum().item():.6e} | dgrad max {ed.abs().max().item():.4e} mean {ed.abs().mean().item():.4e} sum {dx.float().sum().item():.6e}")
