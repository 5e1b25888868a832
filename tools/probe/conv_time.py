"""Launch time of the ResNet-18 tail's stride-1 convolutions (plain forward / input gradient, back to back, events around 50 launches).
    [AFAN_HIP_LIB=tools/probe/_bin/libafan_hip_<name>.so] [AFAN_CONV_TP2=0] python tools/probe/conv_time.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("lib:", os.environ.get("AFAN_HIP_LIB", "tree"), " TP2:", os.environ.get("AFAN_CONV_TP2", "1"))
for ci, co, h, n in ((128, 128, 16, 256), (256, 256, 8, 256), (512, 512, 4, 256)):
    x = cl(torch.randn(n, ci, h, h, device=dev).bfloat16())
    w = cl((torch.randn(co, ci, 3, 3, device=dev) * 0.05).bfloat16())
    wt = cl(w.permute(1, 0, 2, 3))
    dy = cl(torch.randn(n, co, h, h, device=dev).bfloat16())
    gf = 2.0 * n * h * h * co * ci * 9 / 1e9
    tf = timeit(lambda: ops.conv_fwd(x, w, 1))
    td = timeit(lambda: ops.conv_dgrad(dy, wt, (h, h), 1))
    print(f"{ci:4d}->{co:4d} {h:2d}x{h:<2d}  fwd {tf:6.1f} us ({gf / tf * 1e3:6.1f} TFLOP/s)   dgrad {td:6.1f} us ({gf / td * 1e3:6.1f} TFLOP/s)")
