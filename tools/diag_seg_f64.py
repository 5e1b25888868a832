"""How deterministic is the reference's own fp32 result?  The DeepLab golden iteration in float64 on the CPU oracle against the
fp32 golden: fraction of K-step perturbation elements that agree, per-tensor gradient distance (CPU only)."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa: E402
from oracle import afan_oracle as orc  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "seg_dl101_aspp_k3_damped"
g = golden(case)
steps, se_idx, mix_sd = [int(v) for v in g["meta"]]
gamma_se, gamma_sd, eps = [float(v) for v in g["gammas"]]
torch.manual_seed(int(g["seed"]))
net = orc.deeplabv3plus_resnet101(num_classes=21, output_stride=16)
net.classifier.aspp.project[3].p = 0.0
if float(g["damp"]) != 1.0:
    for m in net.backbone.modules():
        if isinstance(m, orc.SegBottleneck):
            m.bn3.weight.data.mul_(float(g["damp"]))
net = net.double()
opt = orc.seg_make_optimizer(net, lr=float(g["lr"]), weight_decay=1e-4)
net.train()
crit = nn.CrossEntropyLoss(ignore_index=255, reduction="mean")
images, labels = torch.from_numpy(g["images"]).double(), torch.from_numpy(g["labels"])
r = orc.seg_train_step(net, opt, crit, images, labels, steps=steps, eps=eps, gamma_se=gamma_se, gamma_sd=gamma_sd,
                       pertub_idx_se=se_idx, pertub_idx_sd=str(g["sd_idx"]), mix_layer=str(g["mix_layer"]), mix_sd=bool(mix_sd))
gam = gamma_se / 255
k64 = np.rint((r["adv_se"].numpy() - r["fm_se"].numpy()) / gam)
k32 = np.rint((g["adv_se"] - g["fm_se"]) / gam)
np.save(os.path.join(ROOT, "tools", "probe", "_k64_" + case + ".npy"), k64.astype(np.int8))
print(case, "float64 vs the fp32 golden: perturbation agreement", float((k64 == k32).mean()), "loss", float(r["loss"]), float(g["loss"]))
params = dict(net.named_parameters())
for k in g.files:
    if k.startswith("grad/"):
        a = params[k[5:]].grad.numpy()
        print("   ", k, "rel dist fp32 golden vs f64: %.3e" % (np.linalg.norm((a - g[k]).ravel()) / np.linalg.norm(a.ravel())))
