"""GPU probe: conv fwd time with hot vs cold operands (rotating over NBUF input/output buffers)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
N = 256
for (ci, co, h, k, s) in [(128, 128, 16, 3, 1), (256, 256, 8, 3, 1), (512, 512, 4, 3, 1), (64, 64, 32, 3, 1)]:
    w = cl((torch.randn(co, ci, k, k, device=dev) * 0.05).bfloat16())
    for nbuf in (1, 4, 16, 48):
        xs = [cl(torch.randn(N, ci, h, h, device=dev).bfloat16()) for _ in range(nbuf)]
        for x in xs: pkg.ops.conv_fwd(x, w, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for i in range(96): pkg.ops.conv_fwd(xs[i % nbuf], w, s)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 96 * 1e3)
        mb = N * ci * h * h * 2 / 1e6
        print(f"ci{ci} co{co} h{h}: nbuf={nbuf:2d} (input set {mb*nbuf:7.1f} MB) {best:6.1f} us", flush=True)
