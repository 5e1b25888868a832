"""GPU probe: learnable multi-layer A-FAN step time (ResNet-56s, batch 128, K = 3, bf16), eager vs hipGraph."""
import importlib, os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
for use_graph in (False, True):
    torch.manual_seed(3)
    m = pkg.resnet_s.resnet56(init_weight_eta=1 / 9)
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
    tr = pkg.learnable.LearnableTrainer(m, nn.CrossEntropyLoss(), steps=3, gamma=1.0, eps=2.0, use_graph=use_graph)
    x, y = torch.rand(128, 3, 32, 32, device=dev), torch.randint(0, 10, (128,), device=dev)
    for _ in range(6):
        r = tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        r = tr.step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"use_graph={use_graph} graph={'yes' if tr._graph is not None else 'no'} {dt*1e3:.1f} ms/step {128/dt:.0f} img/s "
          f"loss {float(r['loss']):.4f} w_sum {float(r['w'].sum()):.6f} failed={tr._graph_failed}")
