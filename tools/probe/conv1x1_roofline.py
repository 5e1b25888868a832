"""The 1x1 convolutions of ResNet-50 (BASELINE configs[2] share: 64 images of 224 x 224) and of DeepLabv3+ / ResNet-101 (configs[3] share:
2 images of 513 x 513, output stride 16) one shape at a time on the tuned kernel: forward and input gradient, 20 back-to-back launches
between two events; algorithmic bytes (input + output + weights, bf16) and FLOPs per launch, and which roofline the launch sits under
(HBM 8 TB/s vs dense bf16 MFMA 2.5 PFLOP/s: the time each would need).  Is a tap-free GEMM kernel what these layers lack?
    python tools/probe/conv1x1_roofline.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cl = torch.channels_last
shapes = [("R50 layer1 conv1", 64, 256, 64, 56, 1), ("R50 layer1 conv3", 64, 64, 256, 56, 1), ("R50 layer2 conv1", 64, 512, 128, 28, 1),
          ("R50 layer2 conv3", 64, 128, 512, 28, 1), ("R50 layer2 proj/2", 64, 256, 512, 56, 2), ("R50 layer3 conv1", 64, 1024, 256, 14, 1),
          ("R50 layer3 conv3", 64, 256, 1024, 14, 1), ("R50 layer4 conv1", 64, 2048, 512, 7, 1), ("R50 layer4 conv3", 64, 512, 2048, 7, 1),
          ("DL layer1 conv1", 2, 256, 64, 129, 1), ("DL layer1 conv3", 2, 64, 256, 129, 1), ("DL layer2 conv3", 2, 128, 512, 65, 1),
          ("DL layer3 conv1", 2, 1024, 256, 33, 1), ("DL layer3 conv3", 2, 256, 1024, 33, 1), ("DL layer4 conv3", 2, 512, 2048, 33, 1),
          ("DL aspp 1x1", 2, 2048, 256, 33, 1), ("DL aspp project", 2, 1280, 256, 33, 1)]
print(f"{'layer':20s} {'M x K -> N':>22s} {'GFLOP':>7s} {'MB':>7s} | {'fwd us':>7s} {'TB/s':>5s} {'TF/s':>6s} | {'dgrad us':>8s} {'TB/s':>5s} {'TF/s':>6s} | HBM-time us  MFMA-time us")
for name, n, ci, co, hw, st in shapes:
    x = torch.randn(n, ci, hw, hw, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5).to(dev).bfloat16().contiguous(memory_format=cl)
    wt = w.permute(1, 0, 2, 3).contiguous(memory_format=cl)
    y = pkg.ops.conv_fwd(x, w, st)
    dy = torch.randn_like(y)
    ho = y.shape[2]

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    tf = timeit(lambda: pkg.ops.conv_fwd(x, w, st))
    td = timeit(lambda: pkg.ops.conv_dgrad(dy, wt, (hw, hw), st))
    M = n * ho * ho
    fl = 2.0 * M * ci * co
    by = 2.0 * (n * hw * hw * ci / (st * st if st > 1 else 1) + M * co + ci * co)
    print(f"{name:20s} {M:>8d} x {ci:>4d} -> {co:>4d} {fl / 1e9:7.2f} {by / 1e6:7.1f} | {tf:7.1f} {by / tf / 1e6:5.2f} {fl / tf / 1e6:6.0f} | {td:8.1f} "
          f"{by / td / 1e6:5.2f} {fl / td / 1e6:6.0f} | {by / 8e6:11.1f} {fl / 2.5e9:13.1f}")
