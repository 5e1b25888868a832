"""Host-side profile of the Faster-RCNN A-FAN iteration (bench.py --arch fasterrcnn_resnet101 is host-bound: 559 ms wall vs
149 ms of kernels): cProfile over 3 iterations, top functions by cumulative and by own time.
    python tools/probe/det_host_profile.py > gpurun_out/det_host_profile.txt"""
import cProfile
import importlib
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
torch.manual_seed(3)
model = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode="align")
for b in model.modules():
    if isinstance(b, pkg.det_model.Bottleneck):
        b.bn3.weight.data.mul_(0.2)
model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
arena = pkg.arena.ParamArena(model, skip=())
opt = pkg.arena.ArenaSGD(arena, lr=0.001, momentum=0.9, weight_decay=0.0005)
g = torch.Generator().manual_seed(3)
side = (600, 904)
x = torch.rand(1, 3, *side, generator=g).to(dev)
x0 = torch.rand(1, 6, 1, generator=g) * (side[1] - 260)
y0 = torch.rand(1, 6, 1, generator=g) * (side[0] - 260)
wh = 60 + torch.rand(1, 6, 2, generator=g) * 200
bb = torch.cat([x0, y0, x0 + wh[..., :1], y0 + wh[..., 1:]], dim=-1).to(dev)
lb = torch.randint(1, 21, (1, 6), generator=g).to(dev)


def step():
    return pkg.det_attack_algo.det_train_step(model, opt, x, bb, lb, loss_settings=1)


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per iteration (unprofiled)")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
