import importlib, sys, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("cv_a-fan_amd")
dev = torch.device("cuda:0")
which = sys.argv[1]
x = (torch.randn(1, 6000, 2, 3) * 1.3 + 4).to(dev).contiguous(memory_format=torch.channels_last)
y = x + 0.1 * torch.randn_like(x)
if which == "unfused":
    o = pkg.ops.mix_feature(x, y); torch.cuda.synchronize(); print("unfused ok", float(o.sum()))
elif which == "lerp":
    o = pkg.ops.lerp_points(x, y, 3); torch.cuda.synchronize(); print("lerp ok", float(o[0].sum()))
else:
    o = pkg.ops.lerp_mix(x, y, 3, [True, True]); torch.cuda.synchronize(); print("fused ok", float(o[0].sum()), float(o[1].sum()))
