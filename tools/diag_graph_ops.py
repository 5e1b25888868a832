"""GPU probe: each library op captured alone in a hipGraph must be self-contained — replay, fill every byte the caching
allocator can reach with NaN, replay: outputs may not change."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
ops = importlib.import_module("cv_a-fan_amd.ops")
dev = torch.device("cuda:0")
CL = torch.channels_last


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16).contiguous(memory_format=CL)


def poison():
    keep, sz = [], 1 << 28
    base = torch.cuda.memory_reserved()
    while sz >= 512:
        t = torch.empty(sz // 4, device=dev)
        if torch.cuda.memory_reserved() > base:
            del t
            base = torch.cuda.memory_reserved()
            sz //= 2
            continue
        t.fill_(float("nan"))
        keep.append(t)
    torch.cuda.synchronize()
    return keep


def check(name, fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        outs = fn()
    outs = [o for o in (outs if isinstance(outs, (list, tuple)) else [outs]) if torch.is_tensor(o)]
    g.replay()
    torch.cuda.synchronize()
    ref = [o.detach().clone() for o in outs]
    keep = poison()
    g.replay()
    torch.cuda.synchronize()
    same = [torch.equal(o.contiguous().view(-1).view(torch.uint8), r.contiguous().view(-1).view(torch.uint8)) for o, r in zip(outs, ref)]
    nans = [int(torch.isnan(o.float()).sum()) for o in outs]
    print("OP %-34s %s same %s nan %s" % (name, "ok " if all(same) else "BAD", same, nans), flush=True)
    del keep, g


torch.manual_seed(0)
N = 256
tail = [(64, 64, 3, 1, 32), (64, 128, 3, 2, 32), (64, 128, 1, 2, 32), (128, 128, 3, 1, 16), (128, 256, 3, 2, 16),
        (128, 256, 1, 2, 16), (256, 256, 3, 1, 8), (256, 512, 3, 2, 8), (256, 512, 1, 2, 8), (512, 512, 3, 1, 4)]
for ci, co, k, st, hw in tail:
    x = rnd(N, ci, hw, hw)
    w = rnd(co, ci, k, k, scale=0.05)
    wt = w.permute(1, 0, 2, 3).contiguous(memory_format=CL)
    ho = hw // st
    dy = rnd(N, co, ho, ho)
    shift = torch.zeros(co, device=dev)
    tag = "%d->%d k%d s%d %d" % (ci, co, k, st, hw)
    check("conv_fwd " + tag, lambda: ops.conv_fwd(x, w, st))

    def fwd_stats():
        ops.acc_reset(dev)
        y, stt = ops.conv_fwd(x, w, st, stats_shift=shift, want_stats=True)
        return [y, stt.acc if stt.acc is not None else stt.partials]
    check("conv_fwd+stats " + tag, fwd_stats)
    check("conv_dgrad " + tag, lambda: ops.conv_dgrad(dy, wt, (hw, hw), st))
    if ops.conv_wgrad_supported(ci, co, k, st):
        check("conv_wgrad " + tag, lambda: ops.conv_wgrad(x, dy, k, st))

for c, hw in [(64, 32), (128, 16), (256, 8), (512, 4)]:
    x = rnd(N, c, hw, hw)
    dy = rnd(N, c, hw, hw)
    wgt, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    res = rnd(N, c, hw, hw)

    def bn_f():
        ops.acc_reset(dev)
        y, stats = ops.bn_train_forward(x, wgt, b, res, True, 1e-5, 0.1, rm, rv, nbt)
        return [y, stats]
    check("bn_fwd c%d" % c, bn_f)
    y, stats = ops.bn_train_forward(x, wgt, b, res, True, 1e-5, 0.1, rm, rv, nbt)
    dw, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    check("bn_bwd c%d" % c, lambda: [*ops.bn_backward(dy, x, y, stats, wgt, b, True, True, dw, db), dw, db])
    check("bn_bwd dgrad-only c%d" % c, lambda: list(ops.bn_backward(dy, x, y, stats, wgt, b, True, True)))

x = rnd(N, 512, 4, 4)
hw_, hb = torch.randn(10, 512, device=dev) * 0.05, torch.zeros(10, device=dev)
check("head_forward", lambda: ops.head_forward(x, hw_, hb))
xa = torch.randn(N, 64, 32, 32, device=dev).contiguous(memory_format=CL)
xc = xa.clone()
gr = rnd(N, 64, 32, 32)
sh = torch.empty_like(xa, dtype=torch.bfloat16)
check("pgd_step", lambda: [ops.pgd_step_(xa, gr, 0.002, xc, 0.008, False, sh), sh])
check("pgd_step_norms", lambda: [*ops.pgd_step_norms_(xa, gr, 0.002, xc, 0.008, False, sh), sh])
