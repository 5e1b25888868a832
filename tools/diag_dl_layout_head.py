"""fp32 DeepLab head (stem .. layer3), channels-last against NCHW on the same weights and images: every stage / block output, layout
against layout, and the SE feature map of each against the reference's (golden).  Companion of tools/diag_dl_layout_tail.py.
    python tools/diag_dl_layout_head.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, load_pkg  # noqa: E402
import test_deeplab_gpu as T  # noqa: E402

pkg = load_pkg()
gpu = torch.device("cuda:0")
g = golden("seg_dl101_aspp_k3_damped")
images = torch.from_numpy(g["images"]).to(gpu)


def capture(nhwc):
    model, tr = T._build(pkg, g, torch.float32, nhwc, gpu, use_graph=False)
    rec, order = {}, []
    names = {m: n for n, m in model.named_modules()}

    def hook(m, inp, out):
        if torch.is_tensor(out) and out.dim() == 4:
            n = names[m]
            if n not in rec:
                order.append(n)
            rec[n] = out.detach().float().contiguous().cpu().double()
    hs = [m.register_forward_hook(hook) for n, m in model.named_modules() if n.startswith("backbone") and n.count(".") <= 2]
    with torch.no_grad():
        head = model({"x": images, "adv": None, "out_idx": 3, "flag": "head"})
    for h in hs:
        h.remove()
    return rec, order, head["out"].detach().float().contiguous().cpu().double()


ra, order, fa = capture(False)
rb, _, fb = capture(True)
ref = torch.from_numpy(g["fm_se"]).double()
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-300))
print(f"fm_se against the reference's: NCHW l2 {rel(fa, ref):.3e} max {float((fa - ref).abs().max()):.2e}   NHWC l2 {rel(fb, ref):.3e} max {float((fb - ref).abs().max()):.2e}"
      f"   layout against layout {rel(fb, fa):.3e}")
for n in order:
    if n in rb and ra[n].shape == rb[n].shape:
        print(f"{n:40s} {tuple(ra[n].shape)!s:22s} layout against layout {rel(rb[n], ra[n]):.3e}")
