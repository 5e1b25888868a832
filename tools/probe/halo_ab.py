"""A/B of the LDS-resident-halo form of the tiled convolution (AFAN_CONV_HALO=1) against the per-tap form, one child process per
setting (the knob is read once per process): forward and input gradient of the step's 3x3 / stride 1 shapes, 40 launches in
one hipGraph each, plus the largest |difference| to an fp32 convolution of the same bf16 operands.
    python tools/probe/halo_ab.py > gpurun_out/halo_ab.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(256, 256, 256, 8), (256, 512, 512, 4), (2, 256, 256, 33), (8, 256, 256, 33), (256, 128, 128, 16), (64, 512, 512, 7), (3, 128, 256, 19),
          (128, 256, 256, 14)]


def child():
    import importlib
    import torch
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("cv_a-fan_amd")
    dev = torch.device("cuda:0")
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    torch.manual_seed(1)

    def graph_time(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn()
            with torch.cuda.graph(g, stream=st):
                for _ in range(40):
                    fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        g.replay()
        for _ in range(4):
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
        return best

    for (N, ci, co, h) in SHAPES:
        x = cl(torch.randn(N, ci, h, h, device=dev).bfloat16())
        w = cl((torch.randn(co, ci, 3, 3, device=dev) * 0.05).bfloat16())
        y = pkg.ops.conv_fwd(x, w, 1)
        ref = torch.nn.functional.conv2d(x.float(), w.float(), padding=1)
        ef = (y.float() - ref).abs().max().item() / ref.abs().max().item()
        dy = cl(torch.randn(N, co, h, h, device=dev).bfloat16())
        wt = cl(w.permute(1, 0, 2, 3))
        dx = pkg.ops.conv_dgrad(dy, wt, (h, h), 1)
        refd = torch.nn.grad.conv2d_input((N, ci, h, h), w.float(), dy.float(), padding=1)
        ed = (dx.float() - refd).abs().max().item() / refd.abs().max().item()
        tf = graph_time(lambda: pkg.ops.conv_fwd(x, w, 1))
        td = graph_time(lambda: pkg.ops.conv_dgrad(dy, wt, (h, h), 1))
        print(f"N{N:4d} {ci:4d}>{co:4d} h{h:3d}: fwd {tf:7.1f} us (err {ef:.2e})  dgrad {td:7.1f} us (err {ed:.2e})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for halo, h256 in (("0", "0"), ("1", "0"), ("1", "1")) * 2:       # (HALO256: 256-row halo tiles for the 385..768-workgroup launches)
            env = dict(os.environ, AFAN_CONV_HALO=halo, AFAN_CONV_HALO256=h256)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(f"--- AFAN_CONV_HALO={halo} AFAN_CONV_HALO256={h256}\n{r.stdout}{r.stderr[-1500:] if r.returncode else ''}", flush=True)
