"""GPU probe: which epoch-boundary action of main_perturb.py poisons the next training step (NaN loss)."""
import importlib, os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
mp = importlib.import_module("cv_a-fan_amd.main_perturb")
dev = torch.device("cuda:0")
acts = os.environ.get("ACTS", "VSK")
args = mp.parser.parse_args(["--seed", "3", "--synthetic", "5120", "--batch_size", "256", "--arch", "resnet18", "--perturb_idx", "6",
                             "--steps", "5", "--gamma", "0.5", "--print_freq", "100"])
if "D" in acts:
    mp.setup_seed(3)
else:
    torch.manual_seed(3)
model = pkg.resnet_s.resnet18()
model.set_compute_dtype(torch.bfloat16).to(dev)
crit = nn.CrossEntropyLoss()
tr = pkg.train_step.AfanTrainer(model, crit, steps=5, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.1)
sched = torch.optim.lr_scheduler.MultiStepLR(tr.optimizer, milestones=[50, 150], gamma=0.1)
train_loader = mp.SyntheticLoader(5120, 256, dev)
val_loader = mp.SyntheticLoader(512, 256, dev)
log = lambda *a: None
mp.train(train_loader, tr, tr.optimizer, 0, args, log)
print("epoch 0 done; graph", tr._graph is not None, flush=True)
if "V" in acts:
    mp.validate(val_loader, model, crit, args, log)
if "S" in acts:
    sched.step()
if "K" in acts:
    torch.save({"state_dict": model.state_dict(), "optimizer": tr.optimizer.state_dict(), "scheduler": sched.state_dict()}, "/tmp/ck_diag.pt")
model.train()
for i, (x, y) in enumerate(train_loader):
    r = tr.step(x, y)
    print("epoch1 step", i, "loss", float(r["loss"]), "adv", float(r["loss_adv"]), "clean", float(r["loss_clean"]),
          "l2max", float(r["l2"].max()), flush=True)
    if i >= 2:
        break
