"""The proposal layer's NMS (afan_nms_top at 0.7, 2 000 survivors wanted) on the boxes the headline Faster-RCNN step really
hands it: one trainer step of bench.py's detection workload with det_model.nms wrapped to keep its inputs, then the three
launches (mask, scan, compaction) alone between two events, 30 calls.  Prints how far the scan had to go (the block holding
the 2 000th survivor) and a checksum of the survivors, so two builds / two scan forms can be compared across processes:
    python tools/probe/nms_time.py > gpurun_out/nms_time.txt"""
import importlib
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
torch.manual_seed(3)
g = torch.Generator().manual_seed(3)
model = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode="align")
for b_ in model.modules():
    if isinstance(b_, pkg.det_model.Bottleneck):
        b_.bn3.weight.data.mul_(0.2)
model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
trainer = pkg.det_trainer.DetTrainer(model, lr=0.001, momentum=0.9, weight_decay=0.0005, loss_settings=1)
x = torch.rand(1, 3, 600, 904, generator=g).to(dev)
x0 = torch.rand(1, 6, 1, generator=g) * (904 - 260)
y0 = torch.rand(1, 6, 1, generator=g) * (600 - 260)
wh = 60 + torch.rand(1, 6, 2, generator=g) * 200
boxes = torch.cat([x0, y0, x0 + wh[..., :1], y0 + wh[..., 1:]], dim=-1).to(dev)
labels = torch.randint(1, 21, (1, 6), generator=g).to(dev)

seen = []
real = pkg.det_model.nms


def spy(b, s, thr, **kw):
    seen.append((b.detach().clone(), None if s is None else s.detach().clone(), thr, dict(kw)))
    return real(b, s, thr, **kw)


pkg.det_model.nms = spy
for _ in range(2):
    trainer.step(x, boxes, labels)
pkg.det_model.nms = real
torch.cuda.synchronize()
print(f"{len(seen)} NMS calls in two steps; scan form:", "first (AFAN_NMS_SCAN_V1=1)" if os.environ.get("AFAN_NMS_SCAN_V1") == "1" else "barrier-free (default)")
for idx in (0, len(seen) // 2, len(seen) - 1):
    b, s, thr, kw = seen[idx]
    n = b.shape[0]
    keep = real(b, s, thr, **kw)
    full = real(b, s, thr, presorted=True)
    kk = keep.cpu()
    last = int(full.cpu()[min(kw.get("max_keep", 0), len(full)) - 1]) if len(full) else 0
    crc = zlib.crc32(kk.numpy().tobytes())
    for _ in range(3):
        real(b, s, thr, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        real(b, s, thr, **kw)       # ends in the host's read of the count, like the model's call
    e1.record()
    e1.synchronize()
    print(f"call {idx:3d}: n {n:6d} ({(n + 63) // 64} blocks), kw {kw}, survivors returned {len(kk)} of {len(full)} unbounded; "
          f"the {kw.get('max_keep', 0)}th sits in block {last // 64}; crc {crc:08x}; {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us per call "
          f"(mask + scan + compaction + the count's read)")

# synthetic sets of different overlap (how many survive decides where the scan may stop), full scan against the early stop
n = 12000
for spread in (300.0, 900.0, 3000.0):
    g = torch.Generator().manual_seed(1)
    xy = torch.rand(n, 2, generator=g) * spread
    wh = torch.rand(n, 2, generator=g) * 200 + 30
    sboxes = torch.cat([xy, xy + wh], dim=1).to(dev)
    scores = torch.sort(torch.rand(n, generator=g), descending=True)[0].to(dev)
    for mk in (0, 2000):
        keep, count = pkg.det_ops.nms(sboxes, scores, 0.7, padded=True, max_keep=mk)
        torch.cuda.synchronize()
        crc = zlib.crc32(keep[:int(count)].cpu().numpy().tobytes())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            pkg.det_ops.nms(sboxes, scores, 0.7, padded=True, max_keep=mk)
        e1.record()
        e1.synchronize()
        print(f"synthetic spread {spread:6.0f}  max_keep {mk:5d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call (with the sort, no host read), "
              f"survivors reported {int(count)}, crc {crc:08x}")
