"""ROIAlign backward (channels-last bf16, 128 rois x 14 x 14 x 1024 channels on a 38 x 57 map) by roi size, the kernel alone through
the C-ABI (afan_roi_align_bwd, 20 launches between two events): the scatter (AFAN_ROI_BWD_GATHER=0: one fp32 atomic per sample corner
and channel) against the atomic-free gather by map tile (default).
    for v in 0 1; do AFAN_ROI_BWD_GATHER=$v python tools/probe/roi_bwd_time.py; done > gpurun_out/roi_bwd_time.txt"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
lib = pkg._lib.load()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, Cc, H, W, R = 1, 1024, 38, 57, 128
print("AFAN_ROI_BWD_GATHER =", os.environ.get("AFAN_ROI_BWD_GATHER", "1 (default)"), " weight tables (AFAN_ROI_BWD_WS) =", os.environ.get("AFAN_ROI_BWD_WS", "1 (default)"))
for size in (64, 128, 224, 336, 400, 800, "mixed"):
    if size == "mixed":        # log-uniform 32..900 px, like an untrained RPN's surviving proposals
        wh = torch.exp(torch.rand(R, 2, generator=g) * (6.8 - 3.5) + 3.5)
    else:
        wh = torch.full((R, 2), float(size))
    x0 = torch.rand(R, 1, generator=g) * (903 - wh[:, :1]).clamp(min=1)
    y0 = torch.rand(R, 1, generator=g) * (599 - wh[:, 1:]).clamp(min=1)
    rois = torch.cat([torch.zeros(R, 1), x0, y0, (x0 + wh[:, :1]).clamp(max=903), (y0 + wh[:, 1:]).clamp(max=599)], dim=1).to(dev)
    dy = torch.randn(R, 14, 14, Cc, generator=g).to(dev).bfloat16()
    dx = torch.empty(N, H, W, Cc, device=dev, dtype=torch.float32)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(lib.afan_roi_align_bwd_workspace_bytes(R, H, W), dtype=torch.uint8, device=dev)
    use_ws = os.environ.get("AFAN_ROI_BWD_WS", "1") != "0"

    def run():
        pkg._lib.check(lib.afan_roi_align_bwd_ws(C.c_void_p(dy.data_ptr()), C.c_void_p(rois.data_ptr()), C.c_void_p(dx.data_ptr()), 1, 1, R, N, Cc,
                                                 H, W, 14, 14, 1 / 16, 0, C.c_void_p(ws.data_ptr()) if use_ws else None, st), "afan_roi_align_bwd_ws")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    e1.synchronize()
    print(f"roi {size!s:>5} px: backward {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per launch (memset included where the scatter needs it)")
