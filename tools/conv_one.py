"""GPU probe: run ONE conv shape (fwd + dgrad) a few times, for rocprofv3 --pmc passes."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
ci, co, h, k, s = [int(v) for v in os.environ.get("SHAPE", "128,128,16,3,1").split(",")]
N = int(os.environ.get("N", 256))
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
x = cl(torch.randn(N, ci, h, h, device=dev).bfloat16()); w = cl((torch.randn(co, ci, k, k, device=dev) * 0.05).bfloat16())
for _ in range(int(os.environ.get("ITERS", 10))):
    y = pkg.ops.conv_fwd(x, w, s)
torch.cuda.synchronize()
