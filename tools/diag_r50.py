import importlib, os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
from oracle import afan_oracle as orc
gpu = torch.device("cuda:0")
torch.manual_seed(3)
ref = orc.resnet50(num_classes=16); ref.train()
sd = {k: v.clone() for k, v in ref.state_dict().items()}
x, y = torch.rand(4, 3, 64, 64), torch.randint(0, 16, (4,))
with torch.no_grad():
    feats = []
    t = x
    for i, L in enumerate(ref.sequential_model[:21]):
        t = L(t); feats.append(t)
m = pkg.resnet_s.resnet50(num_classes=16); m.load_state_dict(sd)
m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
with torch.no_grad():
    t = x.to(gpu)
    for i, L in enumerate(m.sequential_model[:21]):
        if isinstance(L, pkg.resnet_s.BatchNorm2d):
            t = L.fused(t)
        else:
            t = L(t)
        r = feats[i]
        d = (t.float().cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-9)
        print(i, type(L).__name__, tuple(t.shape), "rel max err", round(d, 4), flush=True)
