"""Time of the fused resize + cross-entropy kernel against the three-kernel path (probe)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
ops = pkg.ops
gpu = torch.device("cuda:0")
for n in (2, 8):
    lo = torch.randn(n, 21, 129, 129, device=gpu).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 21, (n, 513, 513), device=gpu)

    def fused():
        return ops.ce2d_upsampled(lo, y, 255, 0.7)

    def three():
        up = ops.upsample_bilinear(lo, (513, 513))
        loss, dup = ops.ce2d(up, y, 255, 0.7)
        return loss, ops.upsample_bilinear_backward(dup, (129, 129))
    for name, f in (("fused", fused), ("three kernels", three)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"batch {n}: {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
