"""Segmented vs one-piece folded step (bf16, ResNet-18) by the number of PGD steps: a kernel difference or a sign() avalanche?"""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402
import test_train_step_gpu as T  # noqa: E402
from oracle import afan_oracle as orc  # noqa: E402

pkg = load_pkg()
gpu = torch.device("cuda:0")
for K in (0, 1, 2, 3):
    res = {}
    for seg in (False, True):
        m = T._build(pkg, orc, "resnet18", gpu, dtype=torch.bfloat16)
        m.set_channels_last(True)
        tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=K, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.05, use_graph=False,
                                        segmented=seg, fold_clean=True if K else None)
        torch.manual_seed(0)
        x, y = torch.rand(16, 3, 32, 32, device=gpu), torch.randint(0, 10, (16,), device=gpu)
        r = tr.step(x, y)
        res[seg] = (tr.arena.grad.clone(), r["x_adv"].clone(), float(r["loss"]))
    g0, g1 = res[False][0], res[True][0]
    print(f"K={K}: grad rel diff {float((g1 - g0).norm() / g0.norm()):.3e}  perturbation elements differing {float((res[True][1] != res[False][1]).float().mean()):.3e} "
          f"losses {res[False][2]:.5f} {res[True][2]:.5f}")
