"""Per-tensor gradient error of the fp32 DeepLab step against the reference golden, by layout and fold (diagnostic)."""
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
case = sys.argv[1] if len(sys.argv) > 1 else "seg_dl101_aspp_k3_damped"
g = golden(case)
for nhwc in (False, True):
    for fold in ((None,) if not nhwc else (False, True)):
        kw = {} if fold is None else dict(fold_clean=fold, fold_pgd0=fold)
        model, tr = T._build(pkg, g, torch.float32, nhwc, gpu, use_graph=False, **kw)
        images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
        r = tr.step(images, labels)
        torch.cuda.synchronize()
        names = [str(k) for k in g["param_names"]]
        got = np.array([float(tr.arena.view(tr.arena.grad, i).double().norm()) for i in range(len(names))])
        ref = g["grad_norms"]
        rel = np.abs(got - ref) / (ref + 1e-7 * ref.max())
        worst = np.argsort(-rel)[:5]
        print(f"nhwc={nhwc} fold={fold} loss={float(r['loss']):.6f} ref={float(g['loss']):.6f} worst norm errs:",
              [(names[i], float(rel[i])) for i in worst])
        for k in g.files:
            if k.startswith("grad/"):
                i = names.index(k[5:])
                a = tr.arena.view(tr.arena.grad, i).float().cpu().numpy()
                e = np.linalg.norm((a - g[k]).ravel()) / max(np.linalg.norm(g[k].ravel()), 1e-12)
                print(f"   {k}: rel err {e:.3e}")
        gam = float(g["gammas"][0]) / 255
        d = r["adv_se"].float().cpu().numpy() - r["fm_se"].float().cpu().numpy()
        k_got, k_ref = np.rint(d / gam), np.rint((g["adv_se"] - g["fm_se"]) / gam)
        k64p = os.path.join(ROOT, "tools", "probe", "_k64_" + case + ".npy")
        if os.path.exists(k64p):
            print("   agreement with the float64 run: product %.5f, reference fp32 %.5f" % (float((k_got == np.load(k64p)).mean()), float((k_ref == np.load(k64p)).mean())))
        print("   SE perturbation agreement", float((k_got == k_ref).mean()), "losses", r["losses"].cpu().numpy(), g["losses"])
        fm = r["fm_se"].float().cpu().numpy()
        print("   fm_se max abs err", float(np.abs(fm - g["fm_se"]).max()), "calls", dict(pkg.ops.CALLS))

# ---- which op family carries the extra channels-last noise?  Re-run NHWC with BatchNorm routed through the NCHW kernels
if os.environ.get("DIAG_BN_NCHW", "1") == "1":
    ops = pkg.ops
    f0, b0 = ops.bn_train_forward, ops.bn_backward
    CL = torch.channels_last

    def fwd(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats=None, out=None, stats_out=None, groups=1):
        if x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and conv_stats is None:
            y, st = f0(x.contiguous(), weight, bias, None if residual is None else residual.contiguous(), relu, eps, momentum, rm, rv, nb)
            return y.contiguous(memory_format=CL), st
        return f0(x, weight, bias, residual, relu, eps, momentum, rm, rv, nb, conv_stats, out, stats_out, groups)

    def bwd(dy, x, y, stats, weight, bias, relu, want_dres, dweight=None, dbias=None, accumulate=False, partials=None, dx_out=None, dres_out=None, groups=1):
        if x.dim() == 4 and ops.layout_of(x) == ops.AFAN_NHWC and x.dtype == torch.float32 and partials is None:
            dx, dres = b0(dy.contiguous(), x.contiguous(), None if y is None else y.contiguous(), stats, weight, bias, relu, want_dres, dweight, dbias, accumulate)
            return dx.contiguous(memory_format=CL), None if dres is None else dres.contiguous(memory_format=CL)
        return b0(dy, x, y, stats, weight, bias, relu, want_dres, dweight, dbias, accumulate, partials, dx_out, dres_out, groups)

    ops.bn_train_forward, ops.bn_backward = fwd, bwd
    model, tr = T._build(pkg, g, torch.float32, True, gpu, use_graph=False, fold_clean=False, fold_pgd0=False)
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    r = tr.step(images, labels)
    gam = float(g["gammas"][0]) / 255
    d = r["adv_se"].float().cpu().numpy() - r["fm_se"].float().cpu().numpy()
    k_got, k_ref = np.rint(d / gam), np.rint((g["adv_se"] - g["fm_se"]) / gam)
    print("NHWC with NCHW BatchNorm kernels: SE perturbation agreement", float((k_got == k_ref).mean()))
