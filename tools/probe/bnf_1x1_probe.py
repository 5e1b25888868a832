"""Which 1x1 launches of DeepLab's layer3 / ResNet-50's tails does the in-launch BatchNorm take?  (None = declined)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)


class BN:
    def __init__(self, c):
        self.weight, self.bias = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        self.running_mean, self.running_var = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        self.num_batches_tracked, self.eps = torch.zeros((), dtype=torch.int64, device=dev), 1e-5


for n, ci, co, h, k in ((2, 1024, 256, 33, 1), (2, 256, 1024, 33, 1), (2, 256, 256, 33, 3), (2, 2048, 512, 33, 1), (2, 512, 2048, 33, 1),
                        (64, 1024, 256, 14, 1), (64, 256, 1024, 14, 1)):
    x = cl(torch.randn(n, ci, h, h, device=dev).bfloat16())
    w = cl((torch.randn(co, ci, k, k, device=dev) * 0.03).bfloat16())
    ops.acc_reset(dev)
    r = ops.conv_fwd_bn(x, w, BN(co), 0.1)
    dy = cl(torch.randn(n, co, h, h, device=dev).bfloat16())
    y, st = ops.bn_train_forward(x, torch.ones(ci, device=dev), torch.zeros(ci, device=dev), None, True, 1e-5, 0.1, None, None, None)
    ops.acc_reset(dev)
    d = ops.conv_dgrad_bn(dy, cl(w.permute(1, 0, 2, 3)), (h, h), x, st, True)
    print((n, ci, co, h, k), "forward", "taken" if r is not None else "declined", "| input gradient", "taken" if d is not None else "declined")
