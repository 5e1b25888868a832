"""fp32 DeepLab tail (layer4 -> ASPP -> decoder -> loss), channels-last against NCHW on the same weights and the same SE feature map:
every module's forward output and (through full backward hooks) output / input gradients, layout against layout — the first module
whose result differs by more than rounding is where the channels-last path loses accuracy (tools/diag_dl_flip_onset.py: 0.54 % of the
first PGD step's signs differ from the reference's in channels-last, 0.016 % in NCHW).
    python tools/diag_dl_layout_tail.py"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, load_pkg  # noqa: E402
import test_deeplab_gpu as T  # noqa: E402

pkg = load_pkg()
gpu = torch.device("cuda:0")
g = golden("seg_dl101_aspp_k3_damped")
images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
crit = pkg.deeplab.seg_criterion(nn.CrossEntropyLoss(ignore_index=255, reduction="mean"))


def nchw(t):
    return t.detach().float().contiguous().cpu().double() if torch.is_tensor(t) and t.dim() == 4 else None


def capture(nhwc):
    model, tr = T._build(pkg, g, torch.float32, nhwc, gpu, use_graph=False)
    rec, order = {}, []
    names = {m: n for n, m in model.named_modules()}

    def fwd_hook(m, inp, out):
        o = out if torch.is_tensor(out) else None
        if o is not None and o.dim() == 4:
            n = names[m]
            if n not in rec:
                order.append(n)
            rec[n] = {"out": nchw(o)}

    def bwd_hook(m, gin, gout):
        n = names[m]
        if n in rec:
            rec[n]["gout"] = nchw(gout[0]) if gout and gout[0] is not None else None
            rec[n]["gin"] = nchw(gin[0]) if gin and gin[0] is not None else None
    hs = []
    for n, m in model.named_modules():
        if n.startswith(("backbone.layer4", "classifier")) and n:
            hs.append(m.register_forward_hook(fwd_hook))
            hs.append(m.register_full_backward_hook(bwd_hook))
    head = model({"x": images, "adv": None, "out_idx": 3, "flag": "head"})
    fm = head["out"].detach().float()
    if os.environ.get("DIAG_OWN_FM", "0") != "1":
        fm = torch.from_numpy(g["fm_se"]).to(gpu)                  # the SAME feature map in both layouts: the reference's
        if nhwc:
            fm = fm.contiguous(memory_format=torch.channels_last)
    xin = fm.clone().requires_grad_(True)
    with pkg.resnet_s.dgrad_only():
        out = model({"x": images, "adv": xin, "out_idx": 3, "flag": "tail", "low_level_feat": head["low_level"], "low_res": False})
        loss = crit(out, labels)
        grad = torch.autograd.grad(loss, xin)[0]
    for h in hs:
        h.remove()
    return rec, order, nchw(grad), float(loss)


ra, order, ga, la = capture(False)
rb, _, gb, lb = capture(True)
base = golden("ref_noise_floor")["seg_dl101_aspp_k3_damped/base_dk_per_step"][0]
for tag, gr in (("NCHW", ga), ("NHWC", gb)):
    print(f"{tag}: signs of d(loss)/d(fm) that differ from the reference baseline's first step: {float((np.sign(gr.numpy()) != base).mean()):.5f}")
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-300))
print(f"loss NCHW {la:.8f}  NHWC {lb:.8f}   d(loss)/d(fm): layout against layout {rel(gb, ga):.3e}")
print("sign differences of that gradient:", float((torch.sign(ga) != torch.sign(gb)).double().mean()))
d = (gb - ga).abs()
big = d > 1e-3 * ga.abs().max()
print("elements whose gradient differs by > 1e-3 of the largest entry:", int(big.sum()), "of", big.numel(), " |g|max", float(ga.abs().max()))
if int(big.sum()):
    idx = big.nonzero()
    for ax, nm in enumerate(("image", "channel", "row", "column")):
        vals, cnt = torch.unique(idx[:, ax], return_counts=True)
        top = sorted(zip(cnt.tolist(), vals.tolist()), reverse=True)[:12]
        print(f"   by {nm}: {len(vals)} distinct; most frequent (count, index): {top}")
    j = idx[0].tolist()
    print("   e.g.", j, "NCHW", float(ga[tuple(j)]), "NHWC", float(gb[tuple(j)]))
print(f"{'module':58s} {'forward out':>12s} {'grad out':>12s} {'grad in':>12s}")
for n in order:
    a, b = ra[n], rb.get(n)
    if b is None:
        continue
    f = rel(b["out"], a["out"]) if a["out"].shape == b["out"].shape else float("nan")
    go = rel(b["gout"], a["gout"]) if a.get("gout") is not None and b.get("gout") is not None and a["gout"].shape == b["gout"].shape else float("nan")
    gi = rel(b["gin"], a["gin"]) if a.get("gin") is not None and b.get("gin") is not None and a["gin"].shape == b["gin"].shape else float("nan")
    print(f"{n:58s} {f:12.3e} {go:12.3e} {gi:12.3e}")
