"""GPU probe: first non-finite loss when cycling 8 fixed synthetic batches (the soak of main_perturb.py --synthetic)."""
import importlib, os, sys, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
variant = os.environ.get("VARIANT", "default")
if os.environ.get("DETERMINISTIC"):
    torch.backends.cudnn.deterministic = True
torch.manual_seed(3)
m = pkg.resnet_s.resnet18()
dtype = torch.float32 if variant == "fp32" else torch.bfloat16
m.set_compute_dtype(dtype).set_channels_last(True).to(dev).train()
kw = {}
if variant == "nobatch": kw["batch_final"] = False
if variant == "nograph": kw["use_graph"] = False
if variant == "lr01": pass
tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=5, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.1, **kw)
g = torch.Generator().manual_seed(3)
xs = [torch.rand(256, 3, 32, 32, generator=g).to(dev) for _ in range(8)]
ys = [torch.randint(0, 10, (256,), generator=g).to(dev) for _ in range(8)]
t0 = time.time()
first = None
for i in range(int(os.environ.get("ITERS", 400))):
    if i < 200:   # epoch-0 warm-up lr of main_perturb.py
        for grp in tr.optimizer.param_groups: grp["lr"] = min(i * 0.1 / 199, 0.1)
    r = tr.step(xs[i % 8], ys[i % 8])
    if i % 20 == 0 or first is None:
        l = float(r["loss"])
        if not (l == l and abs(l) < 1e30) and first is None:
            first = i
            print(variant, "first non-finite loss at step", i, flush=True)
            break
        if i % 20 == 0: print(variant, i, round(l, 4), "wmax", float(tr.arena.param.abs().max()), flush=True)
torch.cuda.synchronize()
print(variant, "done", first, "time", round(time.time() - t0, 1), "graph", tr._graph is not None)

sus = os.environ.get("SUSPECT", "")
if sus:
    sched = torch.optim.lr_scheduler.MultiStepLR(tr.optimizer, milestones=[50, 150], gamma=0.1)
    if "A" in sus: sched.step()
    if "B" in sus: sd = tr.optimizer.state_dict()
    if "C" in sus: torch.save({"state_dict": m.state_dict()}, "/tmp/diag_ck.pt")
    if "D" in sus: torch.save({"optimizer": tr.optimizer.state_dict(), "scheduler": sched.state_dict()}, "/tmp/diag_ck2.pt")
    print("suspects", sus, "lr now", tr.optimizer.param_groups[0]["lr"], flush=True)
if os.environ.get("WITH_EVAL"):
    # epoch boundary of main_perturb.py: eval-mode passes, then training resumes on the captured graph
    m.eval()
    with torch.no_grad():
        for k in range(4):
            out = m(xs[k], end_point=m.layer_number, start_point=0)
    print("eval ok", float(out.float().abs().max()), flush=True)
    m.train()
    for i in range(400, 700):
        r = tr.step(xs[i % 8], ys[i % 8])
        if i % 20 == 0:
            l = float(r["loss"])
            print("after-eval", i, round(l, 4), "wmax", float(tr.arena.param.abs().max()), flush=True)
            if not (l == l):
                break
