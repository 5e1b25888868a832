"""Stage-by-stage comparison of the bf16 channels-last DeepLabv3+ forward (library kernels) with the fp32 CPU oracle on the
same weights: relative error after the stem, every residual stage, ASPP and the decoder.  `python tools/diag_deeplab_layers.py [side]`"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cv_a-fan_amd")
from oracle import afan_oracle as orc  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 129
RESYNC = len(sys.argv) > 2 and sys.argv[2] == "resync"
dev = torch.device("cuda:0")
torch.manual_seed(3)
ref = orc.deeplabv3plus_resnet101(21, 16)
ref.classifier.aspp.project[3].p = 0.0
ref.train()
m = pkg.deeplab.deeplabv3plus_resnet101(21, 16)
m.load_state_dict(ref.state_dict())
m.classifier.aspp.project[3].p = 0.0
m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
arena = pkg.arena.ParamArena(m, skip=())
x = torch.rand(2, 3, side, side)


def rel(a, b, name):
    a = a.detach().float().cpu()
    b = b.detach().float()
    e = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-9)
    r = ((a - b).norm() / max(b.norm().item(), 1e-9)).item()
    print(f"{name:28s} shape {tuple(b.shape)}  max-rel {e:.3e}  l2-rel {r:.3e}")


with torch.no_grad():
    rb, mb = ref.backbone, m.backbone
    a = rb.normal(x)
    b = mb.normal(x.to(dev))
    rel(b, a, "normal")
    a = rb.conv1(a)
    b = mb.conv1(b)
    rel(b, a, "conv1 (7x7/2)")
    a = rb.relu(rb.bn1(a))
    b = mb.bn1.fused(b, None, True)
    rel(b, a, "bn1+relu")
    a = rb.maxpool(a)
    b = mb.maxpool(b)
    rel(b, a, "maxpool")
    for li in (1, 2, 3, 4):
        la, lb = getattr(rb, f"layer{li}"), getattr(mb, f"layer{li}")
        for bi, (ba, bb) in enumerate(zip(la, lb)):
            if RESYNC:      # every block from the oracle's input: per-block error instead of the accumulated one
                b = a.to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
            a = ba(a)
            b = bb(b)
            if bi in (0, 1) or bi == len(la) - 1:
                rel(b, a, f"layer{li}.{bi}")
        if li == 1:
            low_a, low_b = a, b
    if RESYNC:
        b = a.to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
        low_b = low_a.to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    ya = ref.classifier.aspp(a)
    yb = m.classifier.aspp(b)
    rel(yb, ya, "aspp")
    for i in range(5):
        ca = ref.classifier.aspp.convs[i](a)
        if i == 4:
            ca = torch.nn.functional.interpolate(ca, size=a.shape[-2:], mode="bilinear", align_corners=False)
        cb = m.classifier.aspp.convs[i](b) if i else pkg.deeplab._cbr(m.classifier.aspp.convs[0][0], m.classifier.aspp.convs[0][1], b)
        rel(cb, ca, f"aspp.convs.{i}")
    pa = ref.classifier.project(low_a)
    pb = pkg.deeplab._cbr(m.classifier.project[0], m.classifier.project[1], low_b)
    rel(pb, pa, "decoder project (48)")
    oa = ref.classifier({"low_level": low_a, "out": a})
    ob = m.classifier({"low_level": low_b, "out": b})
    rel(ob, oa, "logits (129)")
