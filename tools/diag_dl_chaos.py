"""Is the 0.54 % of first-step PGD signs by which the fp32 channels-last DeepLab differs from the reference (contractive golden) a
property of the channels-last kernels or a draw of a bistable function?  The SE feature map's own rounding noise (1.5e-6 relative
after 91 layers, in EITHER layout) decides a handful of ReLU masks in the decoder and the batch-of-4 BatchNorm of ASPP's pooling
branch; here the images are perturbed by 1e-6 relative noise (eight draws) and the first-step gradient signs of each layout are
compared with the same layout's unperturbed signs and with the other layout's.
    python tools/diag_dl_chaos.py"""
import os
import sys

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
labels = torch.from_numpy(g["labels"]).to(gpu)
crit = pkg.deeplab.seg_criterion(nn.CrossEntropyLoss(ignore_index=255, reduction="mean"))
base = torch.from_numpy(golden("ref_noise_floor")["seg_dl101_aspp_k3_damped/base_dk_per_step"][0]).float()


def grad_sign(nhwc, images):
    model, tr = T._build(pkg, g, torch.float32, nhwc, gpu, use_graph=False)
    head = model({"x": images, "adv": None, "out_idx": 3, "flag": "head"})
    xin = head["out"].detach().float().clone().requires_grad_(True)
    with pkg.resnet_s.dgrad_only():
        out = model({"x": images, "adv": xin, "out_idx": 3, "flag": "tail", "low_level_feat": head["low_level"], "low_res": False})
        gr = torch.autograd.grad(crit(out, labels), xin)[0]
    return torch.sign(gr.detach().float().contiguous().cpu())


img0 = torch.from_numpy(g["images"])
gen = torch.Generator().manual_seed(1)
ref = {}
for d in range(9):
    images = (img0 if d == 0 else img0 * (1 + 1e-6 * torch.randn(img0.shape, generator=gen))).to(gpu)
    s = {lay: grad_sign(lay == "NHWC", images) for lay in ("NCHW", "NHWC")}
    if d == 0:
        ref = s
    print(f"draw {d}: " + "  ".join(f"{lay} vs reference {float((s[lay] != base).float().mean()):.5f} vs own unperturbed {float((s[lay] != ref[lay]).float().mean()):.5f}"
                                   for lay in s) + f"   NHWC vs NCHW {float((s['NHWC'] != s['NCHW']).float().mean()):.5f}")
