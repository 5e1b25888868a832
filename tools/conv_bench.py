"""GPU microbench (not product code): afan implicit-GEMM conv vs MIOpen (channels_last bf16) per ResNet-18 layer shape."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
N = int(os.environ.get("N", 256))
shapes = [(64, 64, 32, 3, 1), (64, 128, 32, 3, 2), (128, 128, 16, 3, 1), (64, 128, 32, 1, 2), (128, 256, 16, 3, 2),
          (256, 256, 8, 3, 1), (128, 256, 16, 1, 2), (256, 512, 8, 3, 2), (512, 512, 4, 3, 1), (256, 512, 8, 1, 2)]
if os.environ.get("ONLY_FIRST"):
    shapes = shapes[:1]
cl = lambda t: t.contiguous(memory_format=torch.channels_last)

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best

tot = dict(af=0, mf=0, ad=0, md=0)
for (ci, co, h, k, s) in shapes:
    x = cl(torch.randn(N, ci, h, h, device=dev).bfloat16()); w = cl((torch.randn(co, ci, k, k, device=dev) * 0.05).bfloat16())
    p = k // 2
    y = torch.ops.aten.convolution(x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1)
    dy = cl(torch.randn_like(y)); wt = cl(w.permute(1, 0, 2, 3))
    flops = 2.0 * N * co * ci * k * k * (h // s) ** 2
    af = timeit(lambda: pkg.ops.conv_fwd(x, w, s))
    mf = timeit(lambda: torch.ops.aten.convolution(x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1)) if not os.environ.get("NO_MIOPEN") else 0.0
    ad = timeit(lambda: pkg.ops.conv_dgrad(dy, wt, (h, h), s))
    md = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1, [True, False, False])) if not os.environ.get("NO_MIOPEN") else 0.0
    if os.environ.get("FUSED"):
        shift = torch.zeros(co, device=dev)
        afs = timeit(lambda: pkg.ops.conv_fwd(x, w, s, stats_shift=shift, want_stats=True))
        gamma, beta = torch.ones(ci, device=dev), torch.zeros(ci, device=dev)
        _, st4 = pkg.ops.bn_train_forward(x, gamma, beta, None, True, 1e-5, 0.1, None, None, None)
        add = cl(torch.randn_like(x))
        ads = timeit(lambda: pkg.ops.conv_dgrad(dy, wt, (h, h), s, bn_bwd=(x, st4, True)))
        ada = timeit(lambda: pkg.ops.conv_dgrad(dy, wt, (h, h), s, addend=add, bn_bwd=(x, st4, True)))
        print(f"      fused: fwd+moments {afs:7.1f}us  dgrad+bn_bwd {ads:7.1f}us  dgrad+addend+bn_bwd {ada:7.1f}us")
    aw = timeit(lambda: pkg.ops.conv_wgrad(x, dy, k, s))
    mw = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1, [False, True, False])) if not os.environ.get("NO_MIOPEN") else 0.0
    tot["af"] += af; tot["mf"] += mf; tot["ad"] += ad; tot["md"] += md; tot["aw"] = tot.get("aw", 0) + aw; tot["mw"] = tot.get("mw", 0) + mw
    print(f"      wgrad afan {aw:7.1f}us ({flops/aw/1e6:6.1f} TF) miopen {mw:7.1f}us")
    print(f"ci{ci:4d} co{co:4d} h{h:3d} k{k} s{s}: fwd afan {af:7.1f}us ({flops/af/1e6:6.1f} TF) miopen {mf:7.1f}us ({flops/max(mf,1e-9)/1e6:6.1f} TF) | "
          f"dgrad afan {ad:7.1f}us ({flops/ad/1e6:6.1f} TF) miopen {md:7.1f}us ({flops/max(md,1e-9)/1e6:6.1f} TF)", flush=True)
print("sum:", {k: round(v) for k, v in tot.items()})
