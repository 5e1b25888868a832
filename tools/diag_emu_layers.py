"""Layer-by-layer: bf16 channels-last product forward vs the bf16-emulating oracle (orc.emulate_bf16) on the same weights."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cv_a-fan_amd")
from oracle import afan_oracle as orc
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
dev = torch.device("cuda:0")
torch.manual_seed(3)
ref = orc.ARCHS[arch][0](); ref.train()
m = pkg.resnet_s.ARCHS[arch][0](); m.load_state_dict(ref.state_dict())
m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
arena = pkg.arena.ParamArena(m)
torch.manual_seed(5)
x = torch.rand(16, 3, 32, 32)
n = len(ref.sequential_model)
with torch.no_grad(), orc.emulate_bf16():
    for e in range(1, n + 1):
        if isinstance(ref.sequential_model[e - 1], (torch.nn.BatchNorm2d,)) and e < n and isinstance(ref.sequential_model[e], torch.nn.ReLU):
            continue
        a = ref(x, end_point=e, start_point=0)
        pkg.ops.acc_reset(dev)
        b = m(x.to(dev), end_point=e, start_point=0).float().cpu()
        d = (a - b)
        nz = (d != 0).float().mean().item()
        print(f"end {e:2d} {type(ref.sequential_model[e-1]).__name__:18s} rel-l2 {float(d.norm()/a.norm()):.3e}  frac differing {nz:.3f}  max {float(d.abs().max()):.3e}")
