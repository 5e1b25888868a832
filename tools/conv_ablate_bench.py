"""Time the tiled convolution forward with parts of its K loop removed (libraries from tools/build_ablate.sh).

Parent: one child process per library build (a process loads the library once).  Child: the ResNet-18 / DeepLab shapes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "product library", 1: "no DMA in loop", 2: "no LDS read/MFMA", 3: "MFMA only (const frags)", 4: "no DMA, no barrier", 5: "no K loop", 6: "no epilogue", 7: "empty kernel", 8: "MFMA only, no DMA/barrier"}

def child():
    import importlib, torch
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("cv_a-fan_amd")
    dev = torch.device("cuda:0")
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    out = []
    for (N, ci, co, h, k) in [(256, 128, 128, 16, 3), (256, 256, 256, 8, 3), (256, 512, 512, 4, 3), (2, 512, 512, 33, 3), (2, 256, 1024, 33, 1)]:
        x = cl(torch.randn(N, ci, h, h, device=dev).bfloat16()); w = cl((torch.randn(co, ci, k, k, device=dev) * 0.05).bfloat16())
        for _ in range(5): pkg.ops.conv_fwd(x, w, 1)
        torch.cuda.synchronize()
        # 40 launches in one hipGraph: the host's ~10 us per Python call is out of the measurement
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            pkg.ops.conv_fwd(x, w, 1)
            with torch.cuda.graph(g, stream=st):
                for _ in range(40): pkg.ops.conv_fwd(x, w, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        g.replay()
        for _ in range(4):
            e0.record()
            g.replay()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
        out.append(f"{best:7.1f}")
    print(" ".join(out), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        print("shapes: N256 128>128 h16 | 256>256 h8 | 512>512 h4 | N2 512>512 h33 | N2 256>1024 h33 k1   (us)")
        names = sys.argv[1:] or ["", "abl1", "abl2", "abl3", "abl4", "abl5", "abl6", "abl7"]
        for name in names:
            env = dict(os.environ)
            if name: env["AFAN_HIP_LIB"] = os.path.join(ROOT, "cv_a-fan_amd", "exp", f"libafan_hip_{name}.so")
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            label = NAMES.get(int(name[3:]) if name.startswith("abl") else (0 if not name else -1), name)
            print(f"{label:28s} {r.stdout.strip()} {r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
