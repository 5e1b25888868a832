"""GPU probe: run main_perturb.main() in-process with pieces of the epoch boundary replaced (NaN-after-validate hunt).
SKIP: V = no validation, T = no torch.save; POISON=1: instead of validating, fill every free cached block with NaN
(a graph that still points at a freed eager tensor then reads NaN); SEED=0: no --seed; GRAPH=0: eager steps."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("cv_a-fan_amd")
mp = importlib.import_module("cv_a-fan_amd.main_perturb")
skip = os.environ.get("SKIP", "")


def poison(*a, **k):
    keep = []
    for sz in [2 ** i for i in range(9, 28)]:
        for _ in range(6 if sz < (1 << 24) else 2):
            keep.append(torch.full((sz // 4,), float("nan"), device="cuda"))
    torch.cuda.synchronize()
    del keep
    return 0.0, 0.0


VAR = os.environ.get("VAR", "")


def variant(val_loader, model, criterion, args, log):
    """A: mode toggles only; B: train-mode forward under no_grad; C: eval forward of the head slice; D: eval forward
    of the tail slice; E: full eval forward, one batch."""
    inp, target = next(iter(val_loader))
    if VAR == "A":
        model.eval()
    elif VAR == "B":
        with torch.no_grad():
            model(inp, end_point=model.layer_number, start_point=0)
    else:
        model.eval()
        with torch.no_grad():
            if VAR == "C":
                model(inp, end_point=6, start_point=0)
            elif VAR == "D":
                model.train()
                f = model(inp, end_point=6, start_point=0)
                model.eval()
                model(f, end_point=model.layer_number, start_point=6)
            elif VAR == "E":
                model(inp, end_point=model.layer_number, start_point=0)
            elif VAR in "FGH":
                for i, (inp, target) in enumerate(val_loader):
                    out = model(inp, end_point=model.layer_number, start_point=0)
                    if VAR in "GH":
                        loss = criterion(out, target)
                    if VAR == "H":
                        loss.float().item()
    torch.cuda.synchronize()
    return 0.0, 0.0


TRAINERS = []
_init0 = pkg.train_step.AfanTrainer.__init__


def _init_keep(self, *a, **k):
    if os.environ.get("BATCHFINAL") == "0":
        k["batch_final"] = False
    if os.environ.get("FUSION") == "0":
        pkg.resnet_s._Flags.block_fusion = False
    _init0(self, *a, **k)
    TRAINERS.append(self)


pkg.train_step.AfanTrainer.__init__ = _init_keep


def persistent_state():
    tr = TRAINERS[0]
    out = {}
    for k, v in tr.model.state_dict().items():
        out["model." + k] = v
    for k, v in vars(tr.arena).items():
        if torch.is_tensor(v):
            out["arena." + k] = v
    for k, v in pkg.ops._acc_arenas.items():
        out["acc.%s" % (k,)] = v.buf
    for k, v in pkg.ops._ws_cache.items():
        out["ws.%s" % (k,)] = v
    for n, m in tr.model.named_modules():
        for a in ("_stats_buf", "_bwd_buf", "_lp", "_wt"):
            v = getattr(m, a, None)
            if torch.is_tensor(v):
                out["mod.%s.%s" % (n, a)] = v
    for i, v in enumerate(tr._static_in or ()):
        out["static_in.%d" % i] = v
    for k, v in (tr._static_out or {}).items():
        if torch.is_tensor(v):
            out["static_out." + k] = v
    return out


def live_cuda_tensors():
    import gc
    out = {}
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                st = o.untyped_storage()
                out[st.data_ptr()] = (st.nbytes(), tuple(o.shape), str(o.dtype))
        except Exception:
            pass
    return out


def checked(val_loader, model, criterion, args, log):
    torch.cuda.synchronize()
    live0 = live_cuda_tensors()
    ptr0 = {k: v.data_ptr() for k, v in persistent_state().items()}
    before = {k: v.detach().clone() for k, v in persistent_state().items()}
    model.eval()
    with torch.no_grad():
        for i, (inp, target) in enumerate(val_loader):
            model(inp, end_point=model.layer_number, start_point=0)
    torch.cuda.synchronize()
    after = persistent_state()
    live1 = live_cuda_tensors()
    for ptr, info in live0.items():
        if ptr not in live1:
            print("STATE FREED during validation", hex(ptr), info, flush=True)
    for k, v in after.items():
        if ptr0.get(k) != v.data_ptr():
            print("STATE MOVED", k, flush=True)
    for k in sorted(set(before) | set(after)):
        if k not in before or k not in after:
            print("STATE", k, "appeared/disappeared", flush=True)
            continue
        a, b = before[k], after[k]
        if a.shape != b.shape or not torch.equal(a.contiguous().view(-1).view(torch.uint8), b.detach().contiguous().view(-1).view(torch.uint8)):
            print("STATE CHANGED", k, tuple(a.shape), a.dtype, flush=True)
    print("STATE check done", len(before), flush=True)
    return 0.0, 0.0


def replay_probe(val_loader, model, criterion, args, log):
    """Same batch through the captured step before and after two eval forwards, state restored in between."""
    tr = TRAINERS[0]
    if tr._graph is None:
        return 0.0, 0.0
    torch.cuda.synchronize()
    st = persistent_state()
    saved = {k: v.detach().clone() for k, v in st.items() if k.startswith(("model.", "arena."))}

    def restore():
        for k, v in saved.items():
            st[k].copy_(v)

    def show(tag, r):
        msg = [tag]
        for k in ("loss_adv", "loss_clean", "l2", "x_adv", "feature_map", "out_clean"):
            v = r[k].float()
            msg.append("%s nan=%d max=%.4g" % (k, int(torch.isnan(v).sum()), float(torch.nan_to_num(v).abs().max())))
        print("PROBE", " | ".join(msg), flush=True)

    it = iter(val_loader)
    x0, y0 = next(it)
    model.train()
    show("before", tr.step(x0, y0)); restore()
    show("again ", tr.step(x0, y0)); restore()
    model.eval()
    with torch.no_grad():
        for i, (inp, target) in enumerate(val_loader):
            model(inp, end_point=model.layer_number, start_point=0)
    model.train()
    show("after ", tr.step(x0, y0)); restore()
    show("after2", tr.step(x0, y0)); restore()
    torch.cuda.synchronize()
    show("synced", tr.step(x0, y0)); restore()
    return 0.0, 0.0


def poison_all(*a, **k):
    """Fill every cached-but-free byte of the default pool with NaN: allocate, largest first, whatever fits without
    growing the reservation."""
    torch.cuda.synchronize()
    keep, sz = [], 1 << 28
    base = torch.cuda.memory_reserved()
    while sz >= 512:
        t = torch.empty(sz // 4, device="cuda")
        if torch.cuda.memory_reserved() > base:
            del t
            torch.cuda.empty_cache() if False else None
            base = torch.cuda.memory_reserved()   # the new segment stays cached; it is clean, fill it as well below
            sz //= 2
            continue
        t.fill_(float("nan"))
        keep.append(t)
    torch.cuda.synchronize()
    print("POISONED", sum(t.numel() * 4 for t in keep) >> 20, "MiB in", len(keep), "blocks", flush=True)
    del keep
    return 0.0, 0.0


_hunted = []


def uaf_hunt(val_loader, model, criterion, args, log):
    """Which cached-but-free block of the default pool does the captured step still read?  Take every free block,
    NaN-fill halves of the list, replay, bisect; then print the allocation history of the block found."""
    tr = TRAINERS[0]
    if tr._graph is None or _hunted:
        return 0.0, 0.0
    _hunted.append(1)
    torch.cuda.synchronize()
    st = persistent_state()
    saved = {k: v.detach().clone() for k, v in st.items() if k.startswith(("model.", "arena."))}
    x0, y0 = next(iter(val_loader))
    x0, y0 = x0.clone(), y0.clone()
    keep, sz = [], 1 << 28
    base = torch.cuda.memory_reserved()
    while sz >= 512:
        t = torch.empty(sz // 4, device="cuda")
        if torch.cuda.memory_reserved() > base:
            del t
            base = torch.cuda.memory_reserved()
            sz //= 2
            continue
        keep.append(t)

    def bad(idx):
        for i, t in enumerate(keep):
            t.zero_()
        for i in idx:
            keep[i].fill_(float("nan"))
        for k, v in saved.items():
            st[k].copy_(v)
        model.train()
        r = tr._step_graph(x0, y0)
        return bool(torch.isnan(r["loss"]).item())

    cand = list(range(len(keep)))
    print("HUNT blocks", len(cand), "all-zero bad:", bad([]), "all-nan bad:", bad(cand), flush=True)
    while len(cand) > 1:
        half = cand[:len(cand) // 2]
        cand = half if bad(half) else cand[len(cand) // 2:]
    t = keep[cand[0]]
    lo, hi = t.data_ptr(), t.data_ptr() + t.numel() * 4
    print("HUNT culprit block addr %#x size %d confirm %s" % (lo, hi - lo, bad(cand)), flush=True)
    snap = torch.cuda.memory._snapshot()
    seen, sigs = 0, {}
    for trace in snap["device_traces"]:
        for ev in trace:
            if ev["action"] in ("alloc",) and ev["addr"] < hi + (1 << 21) and ev["addr"] + ev["size"] > lo - (1 << 21):
                fr = [f for f in ev.get("frames", []) if "cv_a-fan_amd" in f["filename"] or "tools/" in f["filename"]]
                where = " <- ".join("%s:%d %s" % (os.path.basename(f["filename"]), f["line"], f["name"]) for f in fr[:6])
                sig = (ev["addr"], ev["size"], where)
                if sig not in sigs:
                    sigs[sig] = 0
                sigs[sig] += 1
                seen += 1
    rows = [r for r in sorted(sigs.items()) if "uaf_hunt" not in r[0][2]]
    below = [r for r in rows if r[0][0] < lo][-25:]
    above = [r for r in rows if r[0][0] >= lo][:8]
    for (a, n, where), cnt in below + above:
        print("HUNT ev alloc off %+d size %d x%d | %s" % (a - lo, n, cnt, where), flush=True)
    if True:
        if True:
            if True:
                pass
    segs = sorted(snap["segments"], key=lambda g: g["address"])
    for i, g in enumerate(segs):
        if g["address"] <= lo < g["address"] + g["total_size"]:
            for h in segs[max(i - 3, 0):i + 3]:
                print("HUNT seg %#x size %d pool %s stream %s type %s gap-to-culprit-seg %d" % (
                    h["address"], h["total_size"], h.get("segment_pool_id"), h.get("stream"), h.get("segment_type"),
                    g["address"] - (h["address"] + h["total_size"])), flush=True)
                if h is not g:
                    for b in h["blocks"][-4:]:
                        fr = [f for f in b.get("frames", []) if "cv_a-fan_amd" in f["filename"]]
                        print("HUNT   tail block %#x size %d state %s | %s" % (
                            b.get("address", 0), b["size"], b["state"],
                            " <- ".join("%s:%d %s" % (os.path.basename(f["filename"]), f["line"], f["name"]) for f in fr[:6])), flush=True)
            print("HUNT culprit offset in its segment %d" % (lo - g["address"]), flush=True)
    print("HUNT events", seen, "trace lengths", [len(t) for t in snap["device_traces"]], flush=True)
    for trace in snap["device_traces"]:
        for ev in trace[:3] + trace[-3:]:
            print("HUNT sample", {k: v for k, v in ev.items() if k != "frames"}, flush=True)
    for k, v in saved.items():
        st[k].copy_(v)
    del keep
    return 0.0, 0.0


if VAR == "U":
    torch.cuda.memory._record_memory_history(enabled="all", context="all", stacks="python", max_entries=1000000)
    mp.validate = uaf_hunt
elif VAR == "S":
    mp.validate = checked
elif VAR == "R":
    mp.validate = replay_probe
elif VAR:
    mp.validate = variant
elif os.environ.get("POISON") == "2":
    mp.validate = poison_all
elif os.environ.get("POISON"):
    mp.validate = poison
elif "V" in skip:
    mp.validate = lambda *a, **k: (0.0, 0.0)
if "T" in skip:
    torch.save = lambda *a, **k: None
if os.environ.get("GRAPH", "1") == "0":
    _init = pkg.train_step.AfanTrainer.__init__

    def init(self, *a, **k):
        k["use_graph"] = False
        _init(self, *a, **k)
    pkg.train_step.AfanTrainer.__init__ = init
if os.environ.get("NOMIOPEN"):
    torch.backends.cudnn.enabled = False
seed = os.environ.get("SEED", "3")
argv = ["--synthetic", "5120", "--batch_size", "256", "--arch", "resnet18", "--perturb_idx", "6", "--steps", "5",
        "--gamma", "0.5", "--epochs", "2", "--print_freq", "5", "--save_dir", "/tmp/diag_main"]
if seed != "0":
    argv += ["--seed", seed]
mp.main(argv)
