"""How far does the bf16 DeepLab iteration's loss move under 1e-3 of image noise?  The golden cases of tests/test_deeplab_gpu.py,
six draws each (profiles/r03j_deeplab_loss_spread.txt: the bound on the chaotic case is taken from this spread).
    [AFAN_HIP_LIB=...] python tools/probe/dl_loss_spread.py"""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_deeplab_gpu as T
pkg = importlib.import_module("cv_a-fan_amd")
gpu = torch.device("cuda:0")
for case in ["seg_dl101_aspp_k1", "seg_dl101_aspp_k3_damped"]:
    g = T.golden(case)
    out = []
    for seed in range(6):
        model, tr = T._build(pkg, g, torch.bfloat16, True, gpu, use_graph=False)
        im = torch.from_numpy(g["images"]).to(gpu)
        if seed:
            gen = torch.Generator(device=gpu).manual_seed(seed)
            im = im + 1e-3 * torch.randn(im.shape, device=gpu, generator=gen)      # far below one bf16 ulp of the image values' scale
        r = tr.step(im, torch.from_numpy(g["labels"]).to(gpu))
        out.append(float(r["loss"]))
    print(case, "golden", float(g["loss"]), "bf16 losses (seed 0 = the golden images, others + 1e-3 noise):", [round(v, 4) for v in out])
