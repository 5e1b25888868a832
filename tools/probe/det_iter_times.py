"""Per-iteration wall times of the Faster-RCNN A-FAN iteration (bench.py's workload), 30 iterations after 5: is the 62-70 ms run-to-run
spread a few slow iterations (allocator, garbage collection) or a level?"""
import importlib, os, sys, time, gc
import torch
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
for _ in range(5):
    tr.step(x, bb, lb)
torch.cuda.synchronize()
for label, prep in (("default", lambda: None), ("gc off", gc.disable)):
    prep()
    ts = []
    m0 = torch.cuda.memory_stats(dev)
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.step(x, bb, lb)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    m1 = torch.cuda.memory_stats(dev)
    ts_s = sorted(ts)
    print(label, "median %.1f  min %.1f  max %.1f  mean %.1f ms" % (ts_s[15], ts_s[0], ts_s[-1], sum(ts) / 30),
          "| hipMalloc calls", m1["num_device_alloc"] - m0["num_device_alloc"], "frees", m1["num_device_free"] - m0["num_device_free"],
          "| slowest five", [round(v, 1) for v in ts_s[-5:]])
