import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd"); ops = pkg.ops
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
class BN:
    def __init__(self, c):
        self.weight, self.bias = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        self.running_mean, self.running_var = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        self.num_batches_tracked, self.eps = torch.zeros((), dtype=torch.int64, device=dev), 1e-5
for shared in (False, True):
    ops.grid_shared(shared)
    for n, ci, co, h, k in ((8, 1024, 256, 33, 1), (8, 256, 256, 33, 3), (8, 1024, 512, 33, 1)):
        x = cl(torch.randn(n, ci, h, h, device=dev).bfloat16()); w = cl((torch.randn(co, ci, k, k, device=dev) * 0.03).bfloat16())
        ops.acc_reset(dev)
        r = ops.conv_fwd_bn(x, w, BN(co), 0.1)
        print("shared", shared, (n, ci, co, h, k), "taken" if r is not None else "declined")
