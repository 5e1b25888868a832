"""GPU diagnostic: how many exact zeros does the feature gradient contain (bf16 vs fp32 backbone)?"""
import importlib, sys, os
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
gpu = torch.device("cuda:0")
for dtype in (torch.float32, torch.bfloat16):
    torch.manual_seed(3)
    model = pkg.resnet_s.resnet18().set_compute_dtype(dtype).to(gpu).train()
    torch.manual_seed(0)
    x = torch.rand(256, 3, 32, 32, device=gpu); y = torch.randint(0, 10, (256,), device=gpu)
    with torch.no_grad():
        fm = model(x, end_point=6, start_point=0)
    xin = fm.detach().clone().requires_grad_(True)
    with pkg.resnet_s.dgrad_only():
        out = model(xin, end_point=15, start_point=6)
        loss = nn.CrossEntropyLoss()(out, y)
        g = torch.autograd.grad(loss, xin)[0]
    gf = g.float()
    per_sample_zero = (gf == 0).reshape(256, -1).float().mean(1)
    print(dtype, "loss", float(loss), "zero frac", float((gf == 0).float().mean()), "|g| median", float(gf.abs().median()),
          "min nonzero", float(gf.abs()[gf != 0].min()), "samples all-zero", int((per_sample_zero == 1).sum()),
          "max per-sample zero frac", float(per_sample_zero.max()), "logit absmax", float(out.abs().max()))
    # where are zeros? per channel
    zc = (gf == 0).float().mean(dim=(0, 2, 3))
    print(" per-channel zero frac: max", float(zc.max()), "n channels > 0.5:", int((zc > 0.5).sum()))
    fz = (fm.float() == 0).float().mean()
    print(" feature-map zero frac", float(fz), " grad zero where fm zero:", float(((gf == 0) & (fm.float() == 0)).float().sum() / max(1.0, float((gf == 0).float().sum()))))
