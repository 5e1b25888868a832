"""DeepLab: ONE iteration's gradients with the two sample-point forwards batched (seg_train_phases batch_tails=True) against the two
passes — next to the same comparison between two runs of the two-pass schedule whose images differ by 1e-6: the scale the step's
own discontinuities (sign steps) set.  NB=8 SIDE=65 python tools/diag_dl_batch_tails.py"""
import importlib, os, sys, numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
gpu = torch.device("cuda:0")
N, S = int(os.environ.get("NB", 8)), int(os.environ.get("SIDE", 65))
g = torch.Generator().manual_seed(5)
images = torch.rand(N, 3, S, S, generator=g).to(gpu)
labels = torch.randint(0, 21, (N, S, S), generator=g).to(gpu)
res = {}
for mode in ("two passes", "batched", "two passes, images + 1e-6 noise"):
    torch.manual_seed(3)
    model = pkg.deeplab.deeplabv3plus_resnet50(num_classes=21, output_stride=16)
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    tr = pkg.seg_trainer.SegTrainer(model, steps=2, lr=0.0, momentum=0.0, weight_decay=0.0, use_graph=False, batch_tails=(mode == "batched"))
    x = images + (torch.rand(images.shape, generator=g).to(gpu) * 1e-6 if "noise" in mode else 0)
    r = tr.step(x, labels)
    torch.cuda.synchronize()
    res[mode] = (torch.cat([p.grad.float().flatten() for p in tr.arena.params]), r["losses"].float().cpu().numpy(),
                 {n: p.grad.float().clone() for n, p in zip(tr.arena.names, tr.arena.params)})
ref = res["two passes"]
for mode in ("batched", "two passes, images + 1e-6 noise"):
    a = res[mode]
    cos = float(torch.dot(a[0], ref[0]) / (a[0].norm() * ref[0].norm()))
    rel = float((a[0] - ref[0]).norm() / ref[0].norm())
    print(f"{mode:35s} losses {a[1]}  grad cosine {cos:.6f}  |diff| / |grad| {rel:.4f}")
    rows = sorted(((float((a[2][n] - ref[2][n]).norm() / (ref[2][n].norm() + 1e-12)), n) for n in ref[2]), reverse=True)
    print("    worst:", [("%.3f" % v, n) for v, n in rows[:6]])
