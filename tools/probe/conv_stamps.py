"""Where a halo-form convolution's K loop spends its cycles: per-tap stamps (diagnostic build, tools/build_stamp.sh) of workgroup
(0,0,0)'s first MFMA wave (arrive at barrier | released | MFMAs issued) and first producer wave (loop top | data landed | released).
  AFAN_HIP_LIB=tools/probe/_bin/libafan_hip_stamp.so python tools/probe/conv_stamps.py"""
import ctypes as C, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
lib = C.CDLL(pkg._lib.LIB_PATH)
dev = torch.device("cuda:0")
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
BNF_RD = getattr(lib, "afan_conv_bnf_stamps", None)      # (None: the stamped in-launch-BatchNorm unit did not compile, tools/build_stamp.sh)
SEC = os.environ.get("CONV_STAMPS", "all")     # all | taps | r18 | deeplab | r50
if SEC in ("all", "taps"):
    for ci, co, h, n in ((128, 128, 16, 256), (256, 256, 8, 256), (512, 512, 4, 256)):
        x = cl(torch.randn(n, ci, h, h, device=dev).bfloat16())
        w = cl((torch.randn(co, ci, 3, 3, device=dev) * 0.05).bfloat16())
        for dgrad in (False,):
            for _ in range(5):
                y = pkg.ops.conv_fwd(x, w, 1)
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                y = pkg.ops.conv_fwd(x, w, 1)
            t1.record(); torch.cuda.synchronize()
            buf = (C.c_ulonglong * (2 * 96 * 3))()
            assert lib.afan_conv_stamps(buf) == 0
            s = np.array(buf, dtype=np.uint64).reshape(2, 96, 3).astype(np.int64)
            taps = int((s[0, :, 0] > 0).sum())
            m, p = s[0, :taps], s[1, :taps]
            base = min(m[0, 0], p[0, 0])
            print(f"== {ci}->{co} {h}x{h} batch {n}: {t0.elapsed_time(t1) / 20 * 1e3:.1f} us per launch (eager, stamped build); {taps} taps; "
                  f"loop {(m[-1, 2] - m[0, 0])} cycles = {(m[-1, 2] - m[0, 0]) / taps:.0f} per tap")
            mw, mc = m[:, 1] - m[:, 0], m[:, 2] - m[:, 1]
            gap = np.concatenate([[0], m[1:, 0] - m[:-1, 2]])
            if not p.any():
                print(f"   MFMA wave:  barrier wait mean {mw[1:].mean():.0f}, reads+MFMA issue mean {mc.mean():.0f} (producer stamps: only in the AFAN_CONV_HPIPE build)")
                continue
            pw, pb = p[:, 1] - p[:, 0], p[:, 2] - p[:, 1]
            pi = np.concatenate([p[1:, 0] - p[:-1, 2], [0]])
            print(f"   MFMA wave:  barrier wait mean {mw.mean():.0f} (min {mw.min()} max {mw.max()}), reads+MFMA issue mean {mc.mean():.0f} (min {mc.min()} max {mc.max()}), between {gap.mean():.0f}")
            print(f"   producer:   vmcnt wait mean {pw.mean():.0f} (max {pw.max()}), barrier wait mean {pb.mean():.0f}, DMA issue mean {pi.mean():.0f} (max {pi.max()})")
            print("   tap: Mwait Mcomp | Pvmcnt Pbar Pissue")
            for i in range(min(taps, 20)):
                print(f"   {i:3d}: {mw[i]:6d} {mc[i]:6d} | {pw[i]:6d} {pb[i]:6d} {pi[i]:6d}   (M arrive {m[i,0]-base:7d}, P top {p[i,0]-base:7d})")


# ---- phases of whole launches (thread 0 of workgroup 0): plain / with the BatchNorm's sums / with the BatchNorm inside the launch
PH = ("entry", "K loop starts", "K loop done", "tile in LDS", "pass 1 done", "sums added", "barrier passed", "pass 2 done", "exit")


def phases(fn, reader):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2 * 96 * 3))()
    assert reader(buf) == 0
    both = np.array(buf, dtype=np.uint64).reshape(2, 96, 3).astype(np.int64)[1, :9, :2]
    s, l = both[:, 0], both[:, 1]
    t0 = s[0]
    have = [(PH[i], int(s[i] - t0)) for i in range(9) if s[i] >= t0 and (i == 0 or s[i] > 0)]
    phases.last = [(PH[i], int(l[i] - l[0])) for i in range(9) if l[i] >= l[0] and (i == 0 or l[i] > 0)]   # the grid's LAST workgroup
    return e0.elapsed_time(e1) / 10 * 1e3, have


class _BN:
    def __init__(self, c):
        self.weight, self.bias = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        self.running_mean, self.running_var = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        self.num_batches_tracked, self.eps = torch.zeros((), dtype=torch.int64, device=dev), 1e-5


ops = pkg.ops
if SEC in ("all", "r18"):
    for ci, co, h, n in ((128, 128, 16, 256), (256, 256, 8, 256), (512, 512, 4, 256)):
        x = cl(torch.randn(n, ci, h, h, device=dev).bfloat16())
        w = cl((torch.randn(co, ci, 3, 3, device=dev) * 0.05).bfloat16())
        wt = cl(w.permute(1, 0, 2, 3))
        dy = cl(torch.randn(n, co, h, h, device=dev).bfloat16())
        res = cl(torch.randn(n, ci, h, h, device=dev).bfloat16())
        bn = _BN(co)
        y, stats = ops.bn_train_forward(x, bn.weight[:ci] if ci <= co else torch.ones(ci, device=dev), torch.zeros(ci, device=dev), None, True, 1e-5, 0.1, None, None, None)
        print(f"== {ci}->{co} {h}x{h} batch {n}: phases in ticks from entry (thread 0 of workgroup 0), launch time eager")
        for name, fn, rd in (
                ("forward, plain", lambda: ops.conv_fwd(x, w, 1), lib.afan_conv_stamps),
                ("forward + BatchNorm sums", lambda: (ops.acc_reset(dev), ops.conv_fwd(x, w, 1, stats_shift=bn.running_mean, want_stats=True)), lib.afan_conv_stamps),
                ("forward + BatchNorm in the launch", lambda: (ops.acc_reset(dev), ops.conv_fwd_bn(x, w, bn, 0.1, relu=True)), BNF_RD),
                ("input gradient, plain", lambda: ops.conv_dgrad(dy, wt, (h, h), 1), lib.afan_conv_stamps),
                ("input gradient + sums (block-output form)", lambda: (ops.acc_reset(dev), ops.conv_dgrad(dy, wt, (h, h), 1, addend=res, bn_bwd=(x, stats, True), bn_y=y)), lib.afan_conv_stamps),
                ("input gradient + BatchNorm backward in the launch (block-output form)", lambda: (ops.acc_reset(dev), ops.conv_dgrad_bn(dy, wt, (h, h), x, stats, True, bn_y=y, addend=res, want_dres=True)), BNF_RD)):
            if rd is None:
                print(f"   {name:72s} (no stamps in this build)")
                continue
            us, ph = phases(fn, rd)
            print(f"   {name:72s} {us:6.1f} us   " + "  ".join(f"{k} {v}" for k, v in ph[1:]))


if SEC in ("all", "deeplab"):
    # ---- DeepLab's layer3 at two 513^2 images (2 x 33 x 33 = 2 178 rows: 35 row tiles of 64 on 256 CUs)
    x1 = cl(torch.randn(2, 1024, 33, 33, device=dev).bfloat16())
    x2 = cl(torch.randn(2, 256, 33, 33, device=dev).bfloat16())
    for name, xx, ci, co, k in (("1x1 1024 -> 256", x1, 1024, 256, 1), ("1x1 256 -> 1024", x2, 256, 1024, 1), ("3x3 256 -> 256", x2, 256, 256, 3)):
        w = cl((torch.randn(co, ci, k, k, device=dev) * 0.03).bfloat16())
        rm = torch.zeros(co, device=dev)
        print(f"== DeepLab layer3 {name}, 2 x 33 x 33")
        for nm, fn in (("forward, plain", lambda: ops.conv_fwd(xx, w, 1)),
                       ("forward + BatchNorm sums", lambda: (ops.acc_reset(dev), ops.conv_fwd(xx, w, 1, stats_shift=rm, want_stats=True)))):
            us, ph = phases(fn, lib.afan_conv_stamps)
            print(f"   {nm:40s} {us:6.1f} us   " + "  ".join(f"{k_} {v}" for k_, v in ph[1:]))


if SEC in ("all", "r50"):
    # ---- ResNet-50 at 64 x 224^2 (configs[2] share): layer1 / layer2 1x1 convolutions on the two-stage kernel (thousands of workgroups,
    # two per CU: workgroup 0's phases are ONE workgroup's life, the launch is ~a dozen of them per CU back to back)
    for name, n, ci, co, h, k in (("layer1 1x1 256 -> 64", 64, 256, 64, 56, 1), ("layer1 1x1 64 -> 256", 64, 64, 256, 56, 1),
                                  ("layer2 1x1 512 -> 128", 64, 512, 128, 28, 1), ("layer2 3x3 128 -> 128", 64, 128, 128, 28, 3)):
        xx = cl(torch.randn(n, ci, h, h, device=dev).bfloat16())
        w = cl((torch.randn(co, ci, k, k, device=dev) * 0.03).bfloat16())
        wt = cl(w.permute(1, 0, 2, 3))
        dy = cl(torch.randn(n, co, h, h, device=dev).bfloat16())
        rm = torch.zeros(co, device=dev)
        y_, st_ = ops.bn_train_forward(xx, torch.ones(ci, device=dev), torch.zeros(ci, device=dev), None, True, 1e-5, 0.1, None, None, None)
        print(f"== ResNet-50 {name}, 64 x {h} x {h}")
        for nm, fn in (("forward, plain", lambda: ops.conv_fwd(xx, w, 1)),
                       ("forward + BatchNorm sums", lambda: (ops.acc_reset(dev), ops.conv_fwd(xx, w, 1, stats_shift=rm, want_stats=True))),
                       ("input gradient, plain", lambda: ops.conv_dgrad(dy, wt, (h, h), 1)),
                       ("input gradient + BatchNorm-backward sums", lambda: (ops.acc_reset(dev), ops.conv_dgrad(dy, wt, (h, h), 1, bn_bwd=(xx, st_, True))))):
            us, ph = phases(fn, lib.afan_conv_stamps)
            print(f"   {nm:44s} {us:6.1f} us   " + "  ".join(f"{k_} {v}" for k_, v in ph[1:]))
            print(f"   {'   (the last workgroup of the grid)':44s}             " + "  ".join(f"{k_} {v}" for k_, v in phases.last[1:]))


if SEC in ("all", "s2"):
    # ---- round 6: the stride-2 launches of the ResNet-18 tail (a stage's first 3x3 / 2 + its 1x1 / 2 projection as a two-problem forward
    # launch; their input gradient as the four-parity-class pair form): ~35 launches of ~40 us per step for half a stride-1 layer's FLOPs
    for ci, co, h, n in ((64, 128, 32, 256), (128, 256, 16, 256), (256, 512, 8, 256)):
        ho = h // 2
        x = cl(torch.randn(n, ci, h, h, device=dev).bfloat16())
        w3 = cl((torch.randn(co, ci, 3, 3, device=dev) * 0.05).bfloat16())
        w1 = cl((torch.randn(co, ci, 1, 1, device=dev) * 0.05).bfloat16())
        bn, bnsc = _BN(co), _BN(co)
        dyall = torch.randn(2 * n, co, ho, ho, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        dy, dy_sc = dyall[:n], dyall[n:]
        wt = cl(w3.permute(1, 0, 2, 3))
        wt10 = torch.cat([w3.permute(1, 2, 3, 0).reshape(ci, 9, co), w1.permute(1, 2, 3, 0).reshape(ci, 1, co)], dim=1).contiguous()
        y, stats = ops.bn_train_forward(x, torch.ones(ci, device=dev), torch.zeros(ci, device=dev), None, True, 1e-5, 0.1, None, None, None)
        print(f"== stride 2: {ci}->{co} {h}x{h} -> {ho}x{ho} batch {n}: phases in ticks from entry (thread 0 of workgroup 0), launch time eager")
        for name, fn, rd in (
                ("forward 3x3/2 + projection, plain", lambda: ops.conv_fwd_multi(x, [w3, w1], 2, [1, 1]), lib.afan_conv_stamps),
                ("forward 3x3/2 + projection + sums", lambda: (ops.acc_reset(dev), ops.conv_fwd_multi(x, [w3, w1], 2, [1, 1], stats_shifts=[bn.running_mean, bnsc.running_mean])), lib.afan_conv_stamps),
                ("forward 3x3/2 + projection + BatchNorm in the launch", lambda: (ops.acc_reset(dev), ops.conv_fwd_multi_bn(x, [w3, w1], 2, [1, 1], [bn.running_mean, bnsc.running_mean], bn, 0.1)), BNF_RD),
                ("input gradient pair, plain", lambda: ops.conv_dgrad(dy, wt, (h, h), 2, sc=(dy_sc, wt10)), lib.afan_conv_stamps),
                ("input gradient pair + sums (block-output form)", lambda: (ops.acc_reset(dev), ops.conv_dgrad(dy, wt, (h, h), 2, bn_bwd=(x, stats, True), bn_y=y, sc=(dy_sc, wt10))), lib.afan_conv_stamps),
                ("input gradient pair + BatchNorm backward in the launch", lambda: (ops.acc_reset(dev), ops.conv_dgrad_bn(dy, wt, (h, h), x, stats, True, bn_y=y, want_dres=True, pair=(dy_sc, wt10))), BNF_RD)):
            if rd is None:
                print(f"   {name:60s} (no stamps in this build)")
                continue
            try:
                us, ph = phases(fn, rd)
            except Exception as e:  # noqa: BLE001
                print(f"   {name:60s} failed: {type(e).__name__}: {e}")
                continue
            print(f"   {name:60s} {us:6.1f} us   " + "  ".join(f"{k} {v}" for k, v in ph[1:]))
            print(f"   {'   (the last workgroup of the grid)':60s}             " + "  ".join(f"{k_} {v}" for k_, v in phases.last[1:]))
