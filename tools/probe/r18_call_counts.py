"""Library calls of ONE ResNet-18 A-FAN step (batch 256, K = 5, folded schedule, eager) by kind — how many convolution launches carry a
BatchNorm inside them (ops.CALLS["conv_bn_fused"]) — and which shapes the in-launch form declined.
    python tools/probe/r18_call_counts.py        (AFAN_GRID_BN_MULTI=0 / AFAN_GRID_BN_PAIR=0 / AFAN_GRID_BN_1X1=0 to compare)"""
import importlib, os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
m = pkg.resnet_s.ARCHS["resnet18"][0]()
m.set_compute_dtype(torch.bfloat16); m.to(dev).train(); m.set_channels_last(True)
tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=5, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.05, use_graph=False, fold_clean=True, share_head=True)
x, y = torch.rand(256, 3, 32, 32, device=dev), torch.randint(0, 10, (256,), device=dev)
tr.step(x, y)
b = dict(ops.CALLS)
tr.step(x, y)
print({k: ops.CALLS[k] - b[k] for k in ops.CALLS if ops.CALLS[k] != b[k]})
print(sorted(k for k in ops._grid_refused))
