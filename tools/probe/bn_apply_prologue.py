"""What do the stand-alone BatchNorm kernels pay for deriving ALL channels' coefficients in every block's prologue?
apply_acc (coefficients from the f64 accumulators, in the kernel) against afan_affine_apply (coefficients given) on the same
tensors: ResNet-50's 14 x 14 x 64-image maps at 1 024 / 256 channels, DeepLab's 33 x 33 x 2 at 1 024 / 256."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
ops = pkg.ops
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n, c, h in ((64, 1024, 14), (64, 256, 14), (64, 2048, 7), (2, 1024, 33), (2, 256, 33), (64, 256, 56)):
    x = cl(torch.randn(n, c, h, h, device=dev).bfloat16())
    res = cl(torch.randn(n, c, h, h, device=dev).bfloat16())
    w = cl((torch.randn(c, 64, 1, 1, device=dev) * 0.1).bfloat16())
    xin = cl(torch.randn(n, 64, h, h, device=dev).bfloat16())
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    ops.acc_reset(dev)
    raw, st = ops.conv_fwd(xin, w, 1, stats_shift=rm, want_stats=True)

    def acc_form():
        ops.bn_train_forward(raw, gamma, beta, res, True, 1e-5, 0.1, None, None, None, st)
    y, stats = ops.bn_train_forward(raw, gamma, beta, res, True, 1e-5, 0.1, None, None, None, st)
    coefs = stats                                   # [4, C]: mean | invstd | alpha | beta

    def given_form():
        ops.affine_apply(raw, coefs, residual=res, relu=True)
    ta, tg = timeit(acc_form), timeit(given_form)
    mb = 3 * raw.numel() * 2 / 1e6
    print(f"{n} x {c} x {h} x {h}: {mb:6.1f} MB moved | apply_acc {ta:6.1f} us ({mb / ta * 1e-3 * 1e3:5.2f} TB/s) | coefficients given {tg:6.1f} us ({mb / tg:5.2f} TB/s)")
