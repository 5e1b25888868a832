"""GPU diagnostic (not a test): prints parity numbers of every golden step case in fp32."""
import importlib, sys, os
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden, load_pkg
from oracle import afan_oracle as orc
pkg = load_pkg()
gpu = torch.device("cuda:0")
ARCH = {"r20s": "resnet20s", "r56s": "resnet56s", "r18": "resnet18"}
sd20 = {k[4:]: torch.from_numpy(golden("step_r20s_k1")[k]) for k in golden("step_r20s_k1").files if k.startswith("sd0/")}
for case in ["step_r20s_k1", "step_r20s_k5", "step_r20s_k5_clip", "step_r20s_k3_clip_rand", "step_r56s_k5", "step_r18_k5"]:
    g = golden(case)
    K, idx, ln, randinit, clip = [int(v) for v in g["meta"]]
    gamma, eps = [float(v) for v in g["gamma_eps"]]
    arch = ARCH[case.split("_")[1]]
    torch.manual_seed(3)
    model = pkg.resnet_s.ARCHS[arch][0]()
    if arch == "resnet20s":
        model.load_state_dict(sd20)
    model.to(gpu).train()
    tr = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=K, gamma=gamma, eps=eps, perturb_idx=idx, layer_number=ln, randinit=bool(randinit), clip=bool(clip), lr=0.1)
    if randinit:
        torch.manual_seed(3); _ = orc.ARCHS[arch][0](); _ = torch.rand(g["x"].shape), torch.randint(0, 10, (g["x"].shape[0],))
    r = tr.step(torch.from_numpy(g["x"]).to(gpu), torch.from_numpy(g["y"]).to(gpu))
    d_got = (r["x_adv"] - r["feature_map"]).cpu().numpy(); d_ref = g["x_adv"] - g["feature_map"]
    bad = np.abs(d_got - d_ref) > 2e-6
    fm_err = np.abs(r["feature_map"].cpu().numpy() - g["feature_map"]).max()
    print(f"{case}: dloss={float(r['loss'])-float(g['loss']):+.2e} dadv={float(r['loss_adv'])-float(g['loss_adv']):+.2e} "
          f"dclean={float(r['loss_clean'])-float(g['loss_clean']):+.2e} loss={float(g['loss']):.4f} flipfrac={bad.mean():.3e} "
          f"fm_maxerr={fm_err:.2e} l2rel={np.abs(r['l2'].cpu().numpy()/g['l2']-1).max():.2e} "
          f"linf_err={np.abs(r['linf'].cpu().numpy()-g['linf']).max():.2e} out_clean_err={np.abs(r['out_clean'].cpu().numpy()-g['out_clean']).max():.2e}", flush=True)
# NaN norms
x = torch.zeros(2, 5000, device=gpu); xa = x.clone(); xa[1, 4321] = float("nan")
print("nan norms", pkg.ops.perturb_norms(xa, x))
