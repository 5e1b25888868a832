"""ROIAlign backward (channels-last bf16, 128 rois x 14 x 14 x 1024 channels on a 38 x 57 map) by roi size: is its time the number
of fp32 atomics (4 per sample point, samples = ceil(roi / 14)^2 per bin)?
    python tools/probe/roi_bwd_time.py > gpurun_out/roi_bwd_time.txt"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 1024, 38, 57, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
for size in (64, 128, 224, 400, 800):
    x0 = torch.rand(128, 1, generator=g) * max(900 - size, 1)
    y0 = torch.rand(128, 1, generator=g) * max(600 - size, 1)
    rois = torch.cat([torch.zeros(128, 1), x0, y0, (x0 + size).clamp(max=903), (y0 + size).clamp(max=599)], dim=1).to(dev)
    y = pkg.det_ops.roi_align(x, rois, (14, 14), 1 / 16, 0)
    dy = torch.randn_like(y)
    for _ in range(2):
        x.grad = None
        y.backward(dy, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        x.grad = None
        y.backward(dy, retain_graph=True)
    e1.record()
    e1.synchronize()
    cells = size / 16
    samples = int(-(-cells // 14)) ** 2
    print(f"roi {size:4d} px = {cells:5.1f} cells, {int(samples)} samples per bin: backward {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us (incl. memset + bf16 cast of the map)")
