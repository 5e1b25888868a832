"""Which SOURCE LINES of the Detection iteration launch the library-external (aten / copy / fill) kernels?  torch.profiler with Python
stacks over one eager iteration of bench.py's Faster-RCNN workload; GPU kernel count and time per innermost frame inside cv_a-fan_amd/.
    python tools/probe/det_aten_lines.py > gpurun_out/r06/det_aten_lines.txt"""
import collections, importlib, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
torch.manual_seed(3)
g = torch.Generator().manual_seed(3)
model = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode="align")
for b_ in model.modules():
    if isinstance(b_, pkg.det_model.Bottleneck):
        b_.bn3.weight.data.mul_(0.2)
model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
trainer = pkg.det_trainer.DetTrainer(model, lr=0.001, momentum=0.9, weight_decay=0.0005, loss_settings=1, noise_ahead=True)
side = (600, 904)
x = torch.rand(1, 3, side[0], side[1], generator=g).to(dev)
x0 = torch.rand(1, 6, 1, generator=g) * (side[1] - 260)
y0 = torch.rand(1, 6, 1, generator=g) * (side[0] - 260)
wh = 60 + torch.rand(1, 6, 2, generator=g) * 200
bb = torch.cat([x0, y0, x0 + wh[..., :1], y0 + wh[..., 1:]], dim=-1).to(dev)
lb = torch.randint(1, 21, (1, 6), generator=g).to(dev)
for _ in range(3):
    trainer.step(x, bb, lb)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.select", "aten.slice", "aten.expand", "aten.permute", "aten.t.",
        "aten.transpose", "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.reshape", "aten.empty", "aten.unbind", "aten.split",
        "aten.lift_fresh", "aten._local_scalar_dense", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.numel", "aten.narrow",
        "aten.chunk", "aten.new_empty", "aten.empty_like", "aten.result_type", "aten.unfold", "aten.view_as")
agg = collections.defaultdict(lambda: [0, collections.Counter()])
reads = collections.Counter()


class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "cv_a-fan_amd" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            if "_local_scalar_dense" in name or name.startswith("aten.item"):
                reads[where] += 1
            else:
                agg[where][0] += 1
                agg[where][1][name.replace("aten.", "")] += 1
        elif "_local_scalar_dense" in name:
            for fr in reversed(traceback.extract_stack()):
                if "cv_a-fan_amd" in fr.filename:
                    reads[f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"] += 1
                    break
        return func(*args, **(kwargs or {}))


with Tracer():
    trainer.step(x, bb, lb)
    torch.cuda.synchronize()
tot = sum(v[0] for v in agg.values())
print(f"faster-rcnn iteration: {tot} aten operator calls that launch work (views / allocations not counted; backward ops run by the autograd engine are attributed to the line that called backward)")
for where, (n, names) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:100]:
    print(f"n={n:4d}  {where[:70]:70s} {dict(names.most_common(5))}")
print("host reads (.item() / bool() of a device tensor):")
for where, n in reads.most_common(30):
    print(f"n={n:4d}  {where}")
