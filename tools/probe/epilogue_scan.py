"""Epilogue cost of the tiled convolution (library built with AFAN_CONV_ABLATE=5: no K loop) against output size."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for (N, c, h) in [(32, 128, 16), (64, 128, 16), (128, 128, 16), (256, 128, 16), (512, 128, 16), (256, 256, 8), (512, 256, 8), (256, 512, 4)]:
    x = cl(torch.randn(N, c, h, h, device=dev).bfloat16()); w = cl((torch.randn(c, c, 3, 3, device=dev) * 0.05).bfloat16())
    for _ in range(3): pkg.ops.conv_fwd(x, w, 1)
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        pkg.ops.conv_fwd(x, w, 1)
        with torch.cuda.graph(g, stream=st):
            for _ in range(40): pkg.ops.conv_fwd(x, w, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    g.replay()
    for _ in range(4):
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
    mb = N * c * h * h * 2 / 1e6
    print(f"N{N:4d} c{c:4d} h{h:3d}: output {mb:6.1f} MB, {N*h*h//128 * (c//128):5d} workgroups: {best:6.1f} us  ({mb / max(best - 2.2, 0.1):.2f} TB/s after 2.2 us of launch + prologue)", flush=True)
