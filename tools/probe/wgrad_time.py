"""Weight-gradient launches of the headline step (ResNet-18 / CIFAR at batch 256, joint pass = two operand pairs per tail layer) and a
few ResNet-50 shapes, one at a time: 20 back-to-back ops.conv_wgrad calls between two events.  AFAN_HIP_LIB selects a build
(cv_a-fan_amd/exp/*.so: other prefetch depths).
    python tools/probe/wgrad_time.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
print("library:", pkg.LIB_PATH)
g = torch.Generator().manual_seed(0)
shapes = [(256, 64, 64, 32, 3, 1, True), (256, 128, 128, 16, 3, 1, True), (256, 256, 256, 8, 3, 1, True), (256, 512, 512, 4, 3, 1, True),
          (256, 64, 128, 32, 3, 2, True), (256, 64, 128, 32, 1, 2, True), (256, 256, 512, 8, 3, 2, True), (256, 64, 64, 32, 3, 1, False),
          (64, 256, 64, 56, 1, 1, False), (64, 64, 64, 56, 3, 1, False), (64, 512, 128, 28, 1, 1, False), (64, 1024, 256, 14, 1, 1, False)]
for (n, ci, co, hw, k, st, pair) in shapes:
    cl = torch.channels_last
    x = torch.randn(n, ci, hw, hw, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
    ho = (hw - 1) // st + 1
    dy = torch.randn(n, co, ho, ho, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
    grad = torch.zeros(co, ci, k, k, device=dev).contiguous(memory_format=cl)
    second = (x.clone(), dy.clone()) if pair else None
    run = lambda: pkg.ops.conv_wgrad(x, dy, k, st, grad, accumulate=True, second=second)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * (n * (2 if pair else 1)) * ho * ho * co * ci * k * k
    print(f"n={n * (2 if pair else 1):4d} {ci:4d}->{co:4d} {hw:3d}x{hw:<3d} k{k} s{st}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
