"""Microbench + correctness of the register-streamed-weights convolution variant (afan_conv_breg.hip) against the tiled kernel."""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
lib = pkg._lib.load()
dev = torch.device("cuda:0")
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

def pack(w):
    co, ci, k, _ = w.shape
    n = lib.afan_pack_weights_elems(co, k * k, ci)
    out = torch.empty(n, dtype=torch.bfloat16, device=dev)
    pkg._lib.check(lib.afan_pack_weights(C.c_void_p(w.data_ptr()), C.c_void_p(out.data_ptr()), co, k * k, ci, None), "pack")
    return out

shapes = [(256, 128, 128, 16, 3), (256, 256, 256, 8, 3), (256, 512, 512, 4, 3), (2, 256, 256, 33, 3), (2, 1024, 256, 33, 1), (2, 256, 1024, 33, 1),
          (2, 512, 512, 33, 3), (64, 128, 128, 28, 3), (64, 256, 256, 14, 3)]
for (N, ci, co, h, k) in shapes:
    x = cl(torch.randn(N, ci, h, h, device=dev).bfloat16()); w = cl((torch.randn(co, ci, k, k, device=dev) * 0.05).bfloat16())
    wp = pack(w)
    y0 = pkg.ops.conv_fwd(x, w, 1)
    flops = 2.0 * N * co * ci * k * k * h * h
    t0 = timeit(lambda: pkg.ops.conv_fwd(x, w, 1))
    line = f"N{N:4d} ci{ci:5d} co{co:5d} h{h:3d} k{k}: tiled {t0:7.1f}us ({flops/t0/1e6:6.1f} TF)"
    for var in (0, 1, 2, 3):
        y = torch.empty_like(y0)
        def run():
            rc = lib.afan_conv_fwd_breg_exp(C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(y.data_ptr()), N, h, h, ci, co, k, var, None)
            assert rc == 0, rc
        run(); torch.cuda.synchronize()
        err = float((y.float() - y0.float()).abs().max())
        t = timeit(run)
        line += f" | v{var} {t:7.1f}us ({flops/t/1e6:6.1f} TF) err {err:.1e}"
    print(line, flush=True)
