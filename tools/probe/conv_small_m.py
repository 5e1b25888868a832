"""Convolutions whose output has few rows (one 600 x 904 image at stride 8 / 16, two 513 x 513 images at stride 16: 2 166 -
8 475 pixels): the grid does not fill 256 CUs and every workgroup walks its whole reduction alone.  Per shape: the library's
forward and input-gradient launch (50 back to back between two events), MIOpen's for reference, and the number of workgroups
and K-steps of the launch.
    python tools/probe/conv_small_m.py > gpurun_out/conv_small_m.txt"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


# (images, H, W, Ci, Co, k, dilation)
shapes = [(1, 38, 57, 1024, 256, 1, 1), (1, 38, 57, 256, 256, 3, 1), (1, 38, 57, 256, 1024, 1, 1),
          (1, 75, 113, 512, 128, 1, 1), (1, 75, 113, 128, 128, 3, 1), (1, 75, 113, 128, 512, 1, 1),
          (1, 150, 226, 256, 64, 1, 1), (1, 150, 226, 64, 64, 3, 1), (1, 150, 226, 64, 256, 1, 1),
          (2, 33, 33, 1024, 256, 1, 1), (2, 33, 33, 256, 256, 3, 1), (2, 33, 33, 256, 1024, 1, 1),
          (2, 33, 33, 2048, 512, 1, 1), (2, 33, 33, 512, 512, 3, 2), (2, 33, 33, 512, 2048, 1, 1),
          (128, 14, 14, 1024, 512, 1, 1), (128, 7, 7, 512, 512, 3, 1), (128, 7, 7, 512, 2048, 1, 1)]
print(f"{'shape':>34} {'rows':>7} {'fwd us':>8} {'TF':>6} {'dgrad us':>9} {'TF':>6} {'miopen fwd':>10} {'miopen dgrad':>12}")
for (n, h, w_, ci, co, k, dil) in shapes:
    x = cl(torch.randn(n, ci, h, w_, device=dev).bfloat16())
    w = cl((torch.randn(co, ci, k, k, device=dev) * 0.05).bfloat16())
    p = dil * (k // 2)
    y = torch.ops.aten.convolution(x, w, None, (1, 1), (p, p), (dil, dil), False, (0, 0), 1)
    dy = cl(torch.randn_like(y))
    wt = cl(w.permute(1, 0, 2, 3))
    flops = 2.0 * n * h * w_ * co * ci * k * k
    af = timeit(lambda: pkg.ops.conv_fwd(x, w, 1, dilation=dil))
    ad = timeit(lambda: pkg.ops.conv_dgrad(dy, wt, (h, w_), 1, dilation=dil))
    mf = timeit(lambda: torch.ops.aten.convolution(x, w, None, (1, 1), (p, p), (dil, dil), False, (0, 0), 1))
    md = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (p, p), (dil, dil), False, (0, 0), 1, [True, False, False]))
    print(f"{n:3d}x{h:3d}x{w_:3d} {ci:4d}->{co:4d} k{k} d{dil} {n * h * w_:7d} {af:8.1f} {flops / af / 1e6:6.1f} {ad:9.1f} {flops / ad / 1e6:6.1f} {mf:10.1f} {md:12.1f}")

# ---- what a launch costs as a function of its reduction length, operands cold (24 rotating operand sets) -----------------
print()
print("kernel time (event bracket minus its overhead) with 24 rotating operand sets (weights and activations not in L2):")
ov = pkg.ops.profile_event_overhead()
print(f"event bracket overhead {ov:.2f} us")
for (n, h, w_, co, k) in [(1, 38, 57, 256, 1), (1, 38, 57, 1024, 1), (1, 38, 57, 256, 3), (2, 33, 33, 256, 1), (1, 75, 113, 128, 1), (1, 150, 226, 64, 1)]:
    for ci in ((64, 256, 512, 1024, 2048) if k == 1 else (64, 128, 256, 512)):
        sets = []
        for _ in range(24):
            sets.append((cl(torch.randn(n, ci, h, w_, device=dev).bfloat16()), cl((torch.randn(co, ci, k, k, device=dev) * 0.05).bfloat16())))
        for x, w in sets[:4]:
            pkg.ops.conv_fwd(x, w, 1)
        torch.cuda.synchronize()
        pkg.ops.profile_enable(True)
        for r in range(2):
            for x, w in sets:
                pkg.ops.conv_fwd(x, w, 1)
        torch.cuda.synchronize()
        prof = pkg.ops.profile_collect()
        pkg.ops.profile_enable(False)
        (name, v), = [(a, b) for a, b in prof.items() if "conv" in a] or [("?", {"launches": 1, "ms": 0.0})]
        us = v["ms"] * 1e3 / v["launches"] - ov
        ks = k * k * ((ci + 63) // 64)
        print(f"  {n}x{h}x{w_} {ci:4d}->{co:4d} k{k}: {ks:3d} K-steps  {us:6.1f} us  ({2.0 * n * h * w_ * co * ci * k * k / us / 1e6:6.1f} TF)  {name}")
