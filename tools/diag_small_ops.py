"""GPU probe: which framework (aten) ops does one eager A-FAN step issue besides the library's kernels, with shapes."""
import importlib, os, sys, collections
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
torch.manual_seed(3)
m = pkg.resnet_s.resnet18()
m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=5, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.1, use_graph=False)
x, y = torch.rand(256, 3, 32, 32, device=dev), torch.randint(0, 10, (256,), device=dev)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.step(x, y)
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::cat", "aten::add", "aten::mul",
                                                   "aten::_to_copy", "aten::zeros_like", "aten::ones_like", "aten::add_", "aten::div", "aten::sum",
                                                   "aten::contiguous", "aten::empty_like", "aten::mul_"):
        cnt[(e.name, str(e.input_shapes)[:90])] += 1
for (n, s), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
    print("%3d  %-18s %s" % (c, n, s))
