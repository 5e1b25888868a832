"""GPU probe (not product code): MIOpen conv fwd/dgrad/wgrad time for the ResNet-18-CIFAR layer shapes at batch 256,
NCHW vs channels_last, bf16/fp32, benchmark on/off.  Decides the backbone's layout.  Writes gpurun_out/conv_probe.json"""
import json, os, sys, time
import torch

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 256))
shapes = [  # (cin, cout, hw_in, k, stride)
    (3, 64, 32, 3, 1), (64, 64, 32, 3, 1), (64, 128, 32, 3, 2), (128, 128, 16, 3, 1), (64, 128, 32, 1, 2),
    (128, 256, 16, 3, 2), (256, 256, 8, 3, 1), (128, 256, 16, 1, 2), (256, 512, 8, 3, 2), (512, 512, 4, 3, 1),
    (256, 512, 8, 1, 2)]

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

res = []
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for dt in (torch.bfloat16, torch.float32):
        for cl in (False, True):
            tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}
            for (ci, co, hw, k, s) in shapes:
                x = torch.randn(N, ci, hw, hw, device=dev, dtype=dt)
                w = torch.randn(co, ci, k, k, device=dev, dtype=dt)
                if cl:
                    x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
                p = k // 2
                y = torch.ops.aten.convolution(x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1)
                gy = torch.randn_like(y)
                f = timeit(lambda: torch.ops.aten.convolution(x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1))
                d = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1, [True, False, False]))
                wg = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1, [False, True, False]))
                flops = 2.0 * N * co * ci * k * k * (hw // s) ** 2
                res.append(dict(bench=bench, dtype=str(dt), channels_last=cl, shape=[ci, co, hw, k, s], fwd_us=round(f, 1), dgrad_us=round(d, 1), wgrad_us=round(wg, 1),
                                fwd_TF=round(flops / f / 1e6, 1), dgrad_TF=round(flops / d / 1e6, 1), wgrad_TF=round(flops / wg / 1e6, 1)))
                tot["fwd"] += f; tot["dgrad"] += d; tot["wgrad"] += wg
            print(f"bench={bench} {dt} channels_last={cl}: fwd {tot['fwd']:.0f} us dgrad {tot['dgrad']:.0f} us wgrad {tot['wgrad']:.0f} us", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/conv_probe.json", "w"), indent=0)
