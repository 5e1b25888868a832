import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_det_model_gpu as T
from conftest import load_pkg
pkg = load_pkg(); gpu = torch.device("cuda:0")
for mode in ("pooling", "align"):
    g = T._golden_for(mode)
    m = T._build(pkg, g, gpu, torch.float32, True, mode)
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    arena = pkg.arena.ParamArena(m, skip=())
    opt = pkg.arena.ArenaSGD(arena, lr=0.001, momentum=0.9, weight_decay=0.0005)
    torch.manual_seed(102)
    r = pkg.det_attack_algo.det_train_step(m, opt, images, bboxes, labels, loss_settings=1)
    L = r["losses"].float().cpu().numpy()
    print(mode, "losses rel", np.abs(L - g["step_losses"]) / np.abs(g["step_losses"]))
    print(mode, "loss rel", abs(float(r["loss"]) - float(g["step_loss"])) / float(g["step_loss"]))
    d = (r["adv_image"][:, :, ::4, ::4].float().cpu().numpy() - g["adv_image_sub"])
    print(mode, "adv pixels off", float((np.abs(d) > 1e-6).mean()))
    ck1 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in m.state_dict().values()])
    print(mode, "ck rel", float((np.abs(ck1[:, 1] - g["ck1"][:, 1]) / (np.abs(g["ck1"][:, 1]) + 1e-4)).max()))
