"""DeepLab fold vs three passes: per-parameter gradient difference after one fp32 iteration, undamped and damped network."""
import importlib, os, sys
import numpy as np, torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("cv_a-fan_amd")
import test_deeplab_gpu as T
gpu = torch.device("cuda:0")
for case in sys.argv[1:] or ["seg_dl101_aspp_k1", "seg_dl101_aspp_k3_damped"]:
    g = T.golden(case)
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    res = {}
    for tag, fold in (("a", False), ("b", False), ("f", True)):
        model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=fold)
        r = tr.step(images, labels)
        res[tag] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        print(case, tag, "losses", [f"{float(v):.7f}" for v in r["losses"]], flush=True)
    for x, y in (("a", "b"), ("a", "f")):
        rel = sorted(((float((res[y][n] - v).norm() / v.norm().clamp_min(1e-20)), n) for n, v in res[x].items()), reverse=True)
        import collections
        reg = collections.defaultdict(list)
        for r_, n in rel:
            reg[".".join(n.split(".")[:2])].append(r_)
        print("   by region:", {k: f"{sorted(v)[len(v)//2]:.1e}" for k, v in reg.items()}, flush=True)
        print(case, f"{x} vs {y}: worst", [(f"{r:.2e}", n) for r, n in rel[:4]], "median %.2e" % rel[len(rel) // 2][0], flush=True)
