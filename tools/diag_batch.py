"""GPU probe: per-parameter gradient difference between the batched and the separate final passes (K = 0)."""
import importlib, os, sys
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
from oracle import afan_oracle as orc
dev = torch.device("cuda:0")
res = {}
MODES = [(False, True), (True, True), (False, False)]      # (batched, acc)
for batched, acc in MODES:
    pkg.ops.BN_ACC = acc
    torch.manual_seed(3)
    ARCH = os.environ.get("ARCH", "resnet18")
    ref = orc.resnet50(num_classes=16) if ARCH == "resnet50" else orc.resnet18_cifar()
    m = pkg.resnet_s.resnet50(num_classes=16) if ARCH == "resnet50" else pkg.resnet_s.resnet18()
    m.load_state_dict(ref.state_dict())
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
    tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=0, gamma=0.5, eps=2.0, perturb_idx=8 if ARCH == "resnet50" else 6, lr=0.1,
                                    use_graph=False, batch_final=batched)
    torch.manual_seed(0)
    side = 64 if ARCH == "resnet50" else 32
    x, y = torch.rand(32, 3, side, side, device=dev), torch.randint(0, 10, (32,), device=dev)
    r = tr.step(x, y)
    g = {}
    names = [n for n, _ in m.named_parameters() if n != "w"]
    for n, p in m.named_parameters():
        if n != "w":
            g[n] = p.grad.detach().float().cpu().numpy().copy()
    res[(batched, acc)] = (float(r["loss"]), g)
for other in MODES[1:]:
    print("== separate/acc vs", other, "loss", res[MODES[0]][0], res[other][0])
    tot_a = tot_d = 0.0
    for n in res[MODES[0]][1]:
        a, b = res[MODES[0]][1][n], res[other][1][n]
        tot_a += float((a * a).sum()); tot_d += float(((a - b) ** 2).sum())
        rel = np.linalg.norm(a - b) / (np.linalg.norm(a) + 1e-12)
        if n.endswith("conv1.weight") or "14" in n:
            print(f"{n:50s} rel {rel:.4f}")
    print("whole arena rel", (tot_d / tot_a) ** 0.5)
