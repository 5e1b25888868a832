"""Time of the device NMS (mask + scan + compact) on 12 000 boxes in score order, full scan against the top-2000 early stop, for
box sets of different overlap (how many survive decides where the scan may stop).
    python tools/probe/nms_time.py > gpurun_out/nms_time.txt"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
n = 12000
for spread in (300.0, 900.0, 3000.0):
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(n, 2, generator=g) * spread
    wh = torch.rand(n, 2, generator=g) * 200 + 30
    boxes = torch.cat([xy, xy + wh], dim=1).to(dev)
    scores = torch.sort(torch.rand(n, generator=g), descending=True)[0].to(dev)
    for mk in (0, 2000):
        keep, count = pkg.det_ops.nms(boxes, scores, 0.7, padded=True, max_keep=mk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            pkg.det_ops.nms(boxes, scores, 0.7, padded=True, max_keep=mk)
        e1.record()
        e1.synchronize()
        print(f"spread {spread:6.0f}  max_keep {mk:5d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call (with the sort), survivors reported {int(count)}")
