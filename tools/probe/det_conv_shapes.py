"""Launch time of the Faster-RCNN ResNet-101 bottleneck convolutions at the bench's 600x904 image (one image per GPU): rows = pixels of
the feature map, plain forward / input gradient, back to back, events around 50 launches; the workgroup count of each launch beside it.
    python tools/probe/det_conv_shapes.py"""
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


SHAPES = (("layer2", 1, 75, 113, ((512, 128, 1), (128, 128, 3), (128, 512, 1))),
          ("layer3", 1, 38, 57, ((1024, 256, 1), (256, 256, 3), (256, 1024, 1))),
          ("layer4 (128 ROIs)", 128, 4, 4, ((2048, 512, 1), (512, 512, 3), (512, 2048, 1))))
for name, n, h, w_, convs in SHAPES:
    for ci, co, k in convs:
        x = cl(torch.randn(n, ci, h, w_, device=dev).bfloat16())
        w = cl((torch.randn(co, ci, k, k, device=dev) * 0.05).bfloat16())
        wt = cl(w.permute(1, 0, 2, 3))
        dy = cl(torch.randn(n, co, h, w_, device=dev).bfloat16())
        gf = 2.0 * n * h * w_ * co * ci * k * k / 1e9
        tf = timeit(lambda: ops.conv_fwd(x, w, 1))
        td = timeit(lambda: ops.conv_dgrad(dy, wt, (h, w_), 1))
        print(f"{name:18s} {n * h * w_:6d} rows {ci:4d}->{co:4d} {k}x{k}  fwd {tf:6.1f} us ({gf / tf * 1e3:6.1f} TFLOP/s)   "
              f"dgrad {td:6.1f} us ({gf / td * 1e3:6.1f} TFLOP/s)")
