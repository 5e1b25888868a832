"""Which pass of the segmentation step breaks the bf16 gradients?  Gradient norms of the clean pass alone, then with other
forward passes interleaved, against the fp32 CPU oracle on the same (damped) weights."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cv_a-fan_amd")
from oracle import afan_oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
side = 129
mode = sys.argv[1] if len(sys.argv) > 1 else "clean"


def build():
    torch.manual_seed(3)
    ref = orc.deeplabv3plus_resnet101(21, 16)
    ref.classifier.aspp.project[3].p = 0.0
    for m in ref.backbone.modules():
        if isinstance(m, orc.SegBottleneck):
            m.bn3.weight.data.mul_(0.1)
    ref.train()
    m = pkg.deeplab.deeplabv3plus_resnet101(21, 16)
    m.load_state_dict(ref.state_dict())
    m.classifier.aspp.project[3].p = 0.0
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
    return ref, m


ref, m = build()
arena = pkg.arena.ParamArena(m, skip=())
torch.manual_seed(0)
x = torch.rand(2, 3, side, side)
y = torch.randint(0, 21, (2, side, side))
crit = nn.CrossEntropyLoss(ignore_index=255)
fcrit = pkg.deeplab.seg_criterion(crit)


def run(model, xx, yy, c, own):
    if own:
        pkg.ops.acc_reset(dev)
        arena.zero_grad()
    if mode == "clean":
        loss = c(model({"x": xx, "adv": None, "out_idx": 0, "flag": "clean"}), yy)
    elif mode == "head_tail":        # head pass, then tail from its (non-detached) output: same function as clean
        h = model({"x": xx, "adv": None, "out_idx": 3, "flag": "head"})
        loss = c(model({"x": xx, "adv": h["out"], "out_idx": 3, "flag": "tail", "low_level_feat": h["low_level"]}), yy)
    elif mode == "two":              # two clean passes, both in the loss
        l1 = c(model({"x": xx, "adv": None, "out_idx": 0, "flag": "clean"}), yy)
        l2 = c(model({"x": xx, "adv": None, "out_idx": 0, "flag": "clean"}), yy)
        loss = 0.7 * l1 + 0.3 * l2
    elif mode == "extra_fwd":        # a forward pass that is never backpropagated, before the clean pass
        with torch.no_grad():
            model({"x": xx, "adv": None, "out_idx": 0, "flag": "clean"})
        loss = c(model({"x": xx, "adv": None, "out_idx": 0, "flag": "clean"}), yy)
    loss.backward()
    return float(loss)


lr = run(ref, x, y, crit, False)
lo = run(m, x.to(dev), y.to(dev), fcrit, True)
torch.cuda.synchronize()
print(mode, "loss", lo, lr)
pr = dict(ref.named_parameters())
rat = []
for i, n in enumerate(arena.names):
    g = float(arena.view(arena.grad, i).double().norm())
    r = float(pr[n].grad.double().norm())
    rat.append(g / max(r, 1e-30))
    if i % 40 == 0 or i > 280:
        print(f"{n:50s} {g:.4e} {r:.4e} ratio {rat[-1]:.3f}")
rat = np.array(rat)
print("ratio quantiles backbone:", np.quantile(rat[:313], [0.05, 0.5, 0.95]), " classifier:", np.quantile(rat[313:], [0.05, 0.5, 0.95]))
