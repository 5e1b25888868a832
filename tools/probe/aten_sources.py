"""Which ops launch the library-external (aten / copy) kernels of an A-FAN iteration?  torch.profiler over one eager iteration,
GPU time of every aten op grouped by its chain of enclosing ops (autograd node / custom Function names) and input shapes.
    python tools/probe/aten_sources.py [deeplab|r18] > gpurun_out/aten_sources_<arch>.txt"""
import collections
import importlib
import os
import sys

import torch
import torch.nn as nn
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cv_a-fan_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "deeplab"
dev = torch.device("cuda:0")
torch.manual_seed(3)
g = torch.Generator().manual_seed(3)
if which == "deeplab":
    model = pkg.deeplab.MODELS["deeplabv3plus_resnet101"](num_classes=21, output_stride=16)
    model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
    trainer = pkg.seg_trainer.SegTrainer(model, nn.CrossEntropyLoss(ignore_index=255, reduction="mean"), steps=3, eps=2.0,
                                         gamma_se=0.5, gamma_sd=0.5, pertub_idx_se=3, pertub_idx_sd="aspp", mix_layer="11",
                                         mix_sd=True, lr=0.01, use_graph=False)
    x = torch.rand(2, 3, 513, 513, generator=g).to(dev)
    y = torch.randint(0, 21, (2, 513, 513), generator=g).to(dev)
else:
    ctor, idx = pkg.resnet_s.ARCHS["resnet18"]
    model = ctor()
    model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
    trainer = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=5, gamma=0.5, eps=2.0, perturb_idx=idx, lr=0.1,
                                         use_graph=False)
    x = torch.rand(256, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, 10, (256,), generator=g).to(dev)

for _ in range(3):
    trainer.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.step(x, y)
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    t = getattr(e, "self_device_time_total", 0) or 0
    if t <= 0 or not e.name.startswith("aten::"):
        continue
    chain, q = [], e.cpu_parent
    while q is not None:
        chain.append(q.name.replace("autograd::engine::evaluate_function: ", "eval:"))
        q = q.cpu_parent
    shapes = [tuple(s_) for s_ in (e.input_shapes or []) if s_]
    where = " < ".join(chain[:3]) + "   " + str(shapes[:2])
    a = agg[(e.name, where)]
    a[0] += 1
    a[1] += t
tot = sum(v[1] for v in agg.values())
print(f"{which}: {sum(v[0] for v in agg.values())} aten ops with GPU time, {tot / 1e3:.3f} ms")
for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{t / 1e3:8.3f} ms  n={n:4d}  {name:28s} {where}")
