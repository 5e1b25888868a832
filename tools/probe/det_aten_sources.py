"""Which Python lines launch the library-external (aten / copy / fill) kernels of ONE Faster-RCNN A-FAN iteration?
torch.profiler with stacks over one eager iteration: aten ops that own GPU time, grouped by the innermost frame inside
cv_a-fan_amd/ and the op's name: count, GPU time.      python tools/probe/det_aten_sources.py"""
import collections, importlib, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
torch.manual_seed(3)
model = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode="align")
model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
tr = pkg.det_trainer.DetTrainer(model, lr=0.001, noise_ahead=True)
g = torch.Generator().manual_seed(3)
side = (600, 904)
x = torch.rand(1, 3, *side, generator=g).to(dev)
x0 = torch.rand(1, 6, 1, generator=g) * (side[1] - 260)
y0 = torch.rand(1, 6, 1, generator=g) * (side[0] - 260)
wh = 60 + torch.rand(1, 6, 2, generator=g) * 200
bb = torch.cat([x0, y0, x0 + wh[..., :1], y0 + wh[..., 1:]], dim=-1).to(dev)
lb = torch.randint(1, 21, (1, 6), generator=g).to(dev)
for _ in range(4):
    tr.step(x, bb, lb)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(x, bb, lb)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    t = getattr(e, "self_device_time_total", 0) or 0
    if t <= 0 or not (e.name.startswith("aten::") or "Memcpy" in e.name or "Memset" in e.name):
        continue
    where = "?"
    q = e
    while q is not None and where == "?":
        for fr in (q.stack or []):
            if "cv_a-fan_amd" in fr:
                where = fr.split("cv_a-fan_amd/")[-1]
                break
        q = q.cpu_parent
    a = agg[(where, e.name)]
    a[0] += 1
    a[1] += t
print(sum(v[0] for v in agg.values()), "aten / copy ops with GPU time,", round(sum(v[1] for v in agg.values()) / 1e3, 3), "ms")
by_line = collections.defaultdict(lambda: [0, 0.0, []])
for (where, name), (n, t) in agg.items():
    b = by_line[where]
    b[0] += n
    b[1] += t
    b[2].append(f"{name.replace('aten::', '')} x{n}")
for where, (n, t, names) in sorted(by_line.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"n={n:4d} {t / 1e3:7.3f} ms  {where[:70]:70s} {', '.join(sorted(names))[:150]}")
