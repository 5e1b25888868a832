"""Where do the step-1 sign flips of the fp32 channels-last DeepLab iteration come from?  Contractive golden, K = 1: fraction of
SE-perturbation elements that differ from the reference baseline's first step (ref_noise_floor.npz), per configuration switch.
    python tools/diag_dl_flip_onset.py"""
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
case = "seg_dl101_aspp_k3_damped"
g, fl = golden(case), golden("ref_noise_floor")
base = fl[case + "/base_dk_per_step"][0]
images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
gam = float(g["gammas"][0]) / 255
floor = max(float(fl[f"{case}/{k}/per_step"][0]) for k in ("f64", "nomkldnn", "cl", "t", "t_nomkldnn", "t_cl"))
print(f"reference-vs-reference after 1 step: {floor:.5f}")


def run(tag, nhwc, env=None, **kw):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        model, tr = T._build(pkg, g, torch.float32, nhwc, gpu, use_graph=False, **kw)
        tr.kw["steps"] = 1
        r = tr.step(images, labels)
        k_got = np.rint((r["adv_se"].float() - r["fm_se"].float()).cpu().numpy() / gam).astype(np.int8)
        fm_err = float(np.abs(r["fm_se"].float().cpu().numpy() - g["fm_se"]).max())
        print(f"{tag:60s} flips {float((k_got != base).mean()):.5f}   max |fm_se - ref| {fm_err:.2e}   fold_clean={r.get('fold_clean')} fold_pgd0={r.get('fold_pgd0')}")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


run("NCHW", False)
run("NHWC default", True)
run("NHWC, resize before the loss (AFAN_CE_LOWRES=0)", True, {"AFAN_CE_LOWRES": "0"})
run("NHWC, reference schedule (fold_clean=False)", True, fold_clean=False)
run("NHWC, folded clean pass, PGD's first pass separate", True, fold_pgd0=False)
run("NHWC, BatchNorm slab path (AFAN_BN_ACC=0)", True, {"AFAN_BN_ACC": "0"})
