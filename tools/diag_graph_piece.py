"""GPU probe: capture ONE piece of the A-FAN step as a hipGraph, replay, fill every byte the caching allocator can reach
with NaN, replay again.  A piece whose outputs change reads memory the graph does not own."""
import importlib, os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
from importlib import import_module
resnet_s = import_module("cv_a-fan_amd.resnet_s")
attack = import_module("cv_a-fan_amd.attack_algo")
ops = import_module("cv_a-fan_amd.ops")
dev = torch.device("cuda:0")
piece = os.environ.get("PIECE", "pgd1")
torch.manual_seed(3)
if os.environ.get("FUSION") == "0":
    resnet_s._Flags.block_fusion = False
model = resnet_s.resnet18()
model.set_compute_dtype(torch.bfloat16).to(dev)
crit = nn.CrossEntropyLoss()
tr = pkg.train_step.AfanTrainer(model, crit, steps=5, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.1, use_graph=False)
g0 = torch.Generator().manual_seed(4)
x = torch.rand(256, 3, 32, 32, generator=g0).to(dev)
y = torch.randint(0, 10, (256,), generator=g0).to(dev)
model.train()
for _ in range(3):
    tr.step(x, y)
idx, ln = 6, model.layer_number


TRACE = []
if os.environ.get("TRACE"):
    def wrap(name):
        fn = getattr(ops, name)

        def w(*a, **k):
            r = fn(*a, **k)
            ins = [(i, t) for i, t in enumerate(a) if torch.is_tensor(t)] + [(kk, t) for kk, t in k.items() if torch.is_tensor(t)]
            outs_ = [t for t in (r if isinstance(r, (tuple, list)) else [r]) if torch.is_tensor(t)]
            TRACE.append((name, ins, outs_))
            return r
        setattr(ops, name, w)
    for n in ("conv_dgrad", "bn_backward", "head_backward", "conv_fwd", "bn_train_forward", "head_forward"):
        wrap(n)

if os.environ.get("TRACE"):
    blk8 = model.sequential_model[8]
    xs = torch.randn(256, 128, 16, 16, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wl = blk8.conv1.lp_weight()
    print("DEBUG own_ok", resnet_s._own_conv_ok(xs, wl.detach(), blk8.conv1.stride, blk8.conv1.padding), "w", wl.dtype, tuple(wl.shape),
          wl.stride(), "fast", resnet_s._block_fast_path_ok(blk8, xs), "fusion", resnet_s._Flags.block_fusion,
          "compute", blk8.conv1.compute_dtype, "ops is", ops is resnet_s.ops, ops.conv_fwd.__name__, flush=True)

DY = (torch.randn(256, 256, 8, 8, device=dev)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def body():
    ops.acc_reset(dev)
    if piece == "headfwd":
        with torch.no_grad():
            return [model(x, end_point=idx, start_point=0)]
    with torch.no_grad():
        fm = model(x, end_point=idx, start_point=0)
    if piece == "tailfwd":
        with torch.no_grad():
            return [model(fm, end_point=ln, start_point=idx)]
    if piece.startswith("dg8"):
        blk = model.sequential_model[8]
        conv = blk.shortcut[0] if "sc" in piece else blk.conv1
        wt = conv.lp_weight_t()
        if "copy" in piece:
            wt = wt.clone(memory_format=torch.preserve_format)
        return [ops.conv_dgrad(DY, wt, (16, 16), 2)]
    if piece == "ce":
        lg = torch.randn(256, 10, device=dev).requires_grad_(True)
        return [torch.autograd.grad(crit(lg, y), lg)[0]]
    if piece.startswith("from"):
        j = int(piece[4:])
        with torch.no_grad():
            mid = model(fm, end_point=j, start_point=idx) if j > idx else fm
        xin = mid.detach().requires_grad_(True)
        with resnet_s.dgrad_only():
            out = model(xin, end_point=ln, start_point=j)
            loss = crit(out, y)
            return [torch.autograd.grad(loss, xin)[0], out.detach()]
    if piece == "tailgrad":
        xin = fm.detach().requires_grad_(True)
        with resnet_s.dgrad_only():
            out = model(xin, end_point=ln, start_point=idx)
            loss = crit(out, y)
            return [torch.autograd.grad(loss, xin)[0], out.detach()]
    if piece.startswith("pgd"):
        k = int(piece[3:])
        xa = attack.PGD(fm.float(), crit, y=y, model=model, steps=k, gamma=0.5 / 255, start_idx=idx, layer_number=ln,
                        eps=2 / 255, with_norms=True)
        return [xa.detach(), *attack.last_norms()]
    if piece == "final":
        out = tr._forward_backward(x, y, overlap_allreduce=False)
        return [out["loss"], out["x_adv"], tr.arena.grad]
    raise SystemExit("unknown piece")


torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
if os.environ.get("SIDEWARM"):
    with torch.cuda.stream(s):
        for _ in range(2):
            body()
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    outs = body()
g.replay()
torch.cuda.synchronize()
ref = [o.detach().clone() for o in outs]
g.replay()
torch.cuda.synchronize()
same = [torch.equal(o.contiguous().view(-1).view(torch.uint8), r.contiguous().view(-1).view(torch.uint8)) for o, r in zip(outs, ref)]
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
g.replay()
torch.cuda.synchronize()
after = [torch.equal(o.contiguous().view(-1).view(torch.uint8), r.contiguous().view(-1).view(torch.uint8)) for o, r in zip(outs, ref)]
print("PIECE", piece, "replay-reproducible", same, "unchanged-after-poison", after,
      "nan", [int(torch.isnan(o.float()).sum()) for o in outs], flush=True)
nn_ = lambda t: int(torch.isnan(t.float()).sum())
for name, ins, outs_ in TRACE[-len(TRACE) // 3 if False else 0:]:
    pass
import collections
print("TRACE len", len(TRACE), dict(collections.Counter(t[0] for t in TRACE)), flush=True)
for name, ins, outs_ in TRACE[-6:]:
    print("TRACE tail", name, [(tuple(t.shape), hex(t.data_ptr()), nn_(t)) for t in outs_], flush=True)
print("TRACE final out", tuple(outs[0].shape), hex(outs[0].data_ptr()), flush=True)
for name, ins, outs_ in TRACE:
    bad_in = [(k, tuple(t.shape), nn_(t)) for k, t in ins if nn_(t)]
    bad_out = [(tuple(t.shape), nn_(t)) for t in outs_ if nn_(t)]
    if bad_in or bad_out:
        print("TRACE", name, "in:", bad_in, "out:", bad_out, "| all-in shapes", [(k, tuple(t.shape)) for k, t in ins], flush=True)
o = outs[0].float()
if o.dim() == 4 and torch.isnan(o).any():
    m = torch.isnan(o)
    print("NAN per image (nonzero count):", int((m.sum(dim=(1, 2, 3)) > 0).sum()), "first images", (m.sum(dim=(1, 2, 3)) > 0).nonzero().flatten()[:8].tolist())
    print("NAN channels:", (m.sum(dim=(0, 2, 3)) > 0).nonzero().flatten().tolist()[:40], "count", int((m.sum(dim=(0, 2, 3)) > 0).sum()))
    print("NAN by (h%2, w%2):", [[int(m[:, :, a::2, b::2].sum()) for b in range(2)] for a in range(2)])
    print("NAN rows h:", (m.sum(dim=(0, 1, 3)) > 0).nonzero().flatten().tolist(), "cols w:", (m.sum(dim=(0, 1, 2)) > 0).nonzero().flatten().tolist())
