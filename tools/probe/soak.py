"""Soak: many iterations of the three trainers on synthetic data; every loss finite, BatchNorm buffers finite at the end."""
import importlib, os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
torch.manual_seed(3)
def finite(model): return all(torch.isfinite(v).all() for v in model.state_dict().values() if v.is_floating_point())
m = pkg.resnet_s.ARCHS["resnet18"][0](); m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=5, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.05)
xs = [torch.rand(256, 3, 32, 32, device=dev) for _ in range(4)]; ys = [torch.randint(0, 10, (256,), device=dev) for _ in range(4)]
t0 = time.time(); worst = 0.0
for i in range(600):
    r = tr.step(xs[i % 4], ys[i % 4])
    if i % 50 == 0: l = float(r["loss"]); assert l == l, i; worst = max(worst, l)
torch.cuda.synchronize(); print(f"resnet18 600 iterations {time.time()-t0:.1f}s, final loss {float(r['loss']):.4f}, max sampled {worst:.3f}, buffers finite {finite(m)}", flush=True)
m = pkg.deeplab.deeplabv3plus_resnet101(num_classes=21, output_stride=16); m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
tr = pkg.seg_trainer.SegTrainer(m, steps=3, lr=0.01, total_itrs=400)
g = torch.Generator().manual_seed(3)
x = torch.rand(2, 3, 513, 513, generator=g).to(dev); y = torch.randint(0, 21, (2, 513, 513), generator=g).to(dev)
t0 = time.time()
for i in range(200):
    r = tr.step(x, y); tr.scheduler.step()
    if i % 25 == 0: l = float(r["loss"]); assert l == l, i
torch.cuda.synchronize(); print(f"deeplab 200 iterations {time.time()-t0:.1f}s, final loss {float(r['loss']):.4f}, graph {tr._graph is not None}, buffers finite {finite(m)}", flush=True)
