"""Golden-vector generator — runs ONLY in the build container, where /root/reference exists.

It imports the reference's own Python (Classification/attack_algo.py, Classification/resnet_s.py,
Classification/main_learnable.py, Segmentation/attack_algo.py, Segmentation/network/) with arithmetic-neutral shims
(SURVEY.md §8c):
  1. a stand-in for `advertorch.utils.NormalizeByChannelMeanStd` (absent from the image; 3-line formula),
  2. `.cuda()` -> identity, because Classification/attack_algo.py:44-46 hard-codes it,
  3. an empty `torchvision.models.utils.load_state_dict_from_url` (Segmentation/network/backbone/resnet.py:3; never called
     with pretrained_backbone=False),
runs fixed-seed cases through the REFERENCE functions and writes inputs + outputs to tests/golden/*.npz.
Only those arrays travel to the GPU box; no reference source does.  `main_perturb.py` itself cannot be
imported (top-level torchvision/matplotlib imports, main_perturb.py:15-23), so its loop body (lines
173-201, 288-293) is driven here line by line around the reference's PGD and ResNet.

The one compiled piece of the reference on this path, Detection's CPU ROIAlign forward (Detection/support/src/cpu/ROIAlign_cpu.cpp:
4-219, plain templates), is built unedited from where it lies by oracle/Makefile (-> oracle/_ref/libref_roialign.so) and produces
tests/golden/roi_align_fwd_*.npz and the `align`-mode Faster-RCNN fixture.

Usage:  python oracle/gen_golden.py            (rewrites tests/golden/ except the two Faster-RCNN files)
        python oracle/gen_golden.py frcnn      (det_frcnn_r101.npz, det_frcnn_r101_align.npz; own process: Detection/ on sys.path)
        python oracle/gen_golden.py roialign   (only roi_align_fwd_*.npz)
        python oracle/gen_golden.py floor      (ref_noise_floor.npz: the reference against itself in float64 / other fp32 summation orders)
        python oracle/gen_golden.py lossfloor  (ref_loss_floor.npz: the same for the losses; + traj_r18_damped.npz)
        python oracle/gen_golden.py bf16floor  (ref_bf16_floor.npz: the reference's own step under bf16 autocast against its fp32 run, round 6)
        python oracle/gen_golden.py detfloor   (ref_det_floor.npz: the reference's Detection iteration against itself, round 6)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("AFAN_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def _shims():
    class NormalizeByChannelMeanStd(nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            self.register_buffer("mean", torch.tensor(mean))
            self.register_buffer("std", torch.tensor(std))

        def forward(self, t):
            return (t - self.mean[None, :, None, None]) / self.std[None, :, None, None]

    adv = types.ModuleType("advertorch")
    adv_utils = types.ModuleType("advertorch.utils")
    adv_utils.NormalizeByChannelMeanStd = NormalizeByChannelMeanStd
    adv.utils = adv_utils
    sys.modules["advertorch"] = adv
    sys.modules["advertorch.utils"] = adv_utils
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _np(t):
    return t.detach().cpu().numpy().copy()  # copy: live parameters are updated in place by the SGD step


def _state_np(model, prefix="sd/"):
    return {prefix + k: _np(v) for k, v in model.state_dict().items()}


def _checksums(model):
    """per-tensor (sum, abs-sum) in float64 — a compact fingerprint of the whole state_dict"""
    keys = list(model.state_dict().keys())
    vals = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in model.state_dict().values()])
    return keys, vals


def _ref_step(ref_attack, model, optimizer, criterion, inp, target, steps, gamma, eps, idx, layer_number,
              randinit, clip):
    """Classification/main_perturb.py:173-201, the reference's PGD and model called exactly as there."""
    feature_map = model(inp, end_point=idx, start_point=0).detach()
    feature_map_adv = ref_attack.PGD(feature_map, criterion, y=target, model=model, steps=steps,
                                     gamma=(gamma / 255), start_idx=idx, layer_number=layer_number,
                                     eps=(eps / 255), randinit=randinit, clip=clip)
    batch_size = inp.shape[0]
    perturbation = (feature_map_adv - feature_map).clone()
    perturbation = perturbation.detach().cpu().reshape(batch_size, -1)
    l2 = torch.norm(perturbation, p=2, dim=1)
    linf = torch.norm(perturbation, p=float("inf"), dim=1)
    output_adv = model(feature_map_adv, end_point=layer_number, start_point=idx)
    output_clean = model(inp, end_point=layer_number, start_point=0)
    loss_adv = criterion(output_adv, target)
    loss_clean = criterion(output_clean, target)
    loss = (loss_adv + loss_clean) / 2
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return dict(feature_map=feature_map, x_adv=feature_map_adv.detach(), l2=l2, linf=linf, loss=loss.detach(),
                loss_adv=loss_adv.detach(), loss_clean=loss_clean.detach(), out_clean=output_clean.detach())


def _pgd_trace(ref_attack, model, criterion, fm, target, steps, gamma, eps, idx, layer_number, clip):
    """Per-step gradients of the reference PGD (no randinit): re-run it with steps=1..K from identical model
    buffers is not possible (BN running stats move), so instead wrap the model to record what PGD feeds it and
    derive grads from consecutive x_adv snapshots.  Returns list of x_adv inputs seen by the tail forward."""
    seen = []

    class Spy(nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x, end_point=None, start_point=0):
            seen.append(x.detach().clone())
            return self.m(x, end_point=end_point, start_point=start_point)

    out = ref_attack.PGD(fm, criterion, y=target, model=Spy(model), steps=steps, gamma=gamma / 255,
                         start_idx=idx, layer_number=layer_number, eps=eps / 255, randinit=False, clip=clip)
    seen.append(out.detach().clone())
    return seen  # K+1 snapshots: x_adv before step 0 ... after step K-1


def gen_pgd_traces(ref_attack, build, crit, only=None):
    """x_adv before / after every step of the reference's PGD with the gradient it used: ResNet-20s (K = 3, with and
    without clipping) and the headline network (ResNet-18, K = 5, one image: the 64 x 32 x 32 feature map of idx 6)."""
    for name, arch, nimg, K, gamma, clip in (("pgd_trace_r20s_k3", "resnet20s", 2, 3, 0.5, False),
                                            ("pgd_trace_r20s_k3_clip", "resnet20s", 2, 3, 1.5, True),
                                            ("pgd_trace_r18_k5", "resnet18", 1, 5, 0.5, False)):
        if only is not None and name not in only:
            continue
        torch.manual_seed(3)
        model, idx, ln = build(arch)
        model.train()
        x = torch.rand(nimg, 3, 32, 32)
        y = torch.randint(0, 10, (nimg,))
        fm = model(x, end_point=idx, start_point=0).detach()
        import copy
        model_b = copy.deepcopy(model)
        snaps = _pgd_trace(ref_attack, model, crit, fm, y, K, gamma, 2.0, idx, ln, clip)
        # gradients the reference saw: recompute with an identical twin model fed the recorded inputs in order
        grads = []
        for t in range(K):
            xin = snaps[t].clone().requires_grad_(True)
            loss = crit(model_b(xin, end_point=ln, start_point=idx), y)
            grads.append(torch.autograd.grad(loss, xin)[0])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), fm=_np(fm), y=_np(y),
                            gamma_eps=np.array([gamma, 2.0]), clip=np.array(int(clip)),
                            snaps=np.stack([_np(s) for s in snaps]), grads=np.stack([_np(g) for g in grads]))
        print(name, "ok")


def gen_damped_r18(ref_attack, orc):
    """A CONTRACTIVE ResNet-18 (CIFAR stem) step for the end-to-end comparison of the bf16 kernels: every
    block's last-BatchNorm weight x 0.1 (weights are data), so that a residual block is identity + a small correction and
    rounding noise is damped instead of amplified; batch 32, K = 5.  The reference's own PGD (Classification/attack_algo.py)
    and the loop body of main_perturb.py:173-201 drive the build-defined ResNet-18 module (the reference ships none).  Stored:
    inputs, the three losses, the perturbation as int8 multiples of gamma, every parameter's gradient norm, a few gradient
    tensors, every BatchNorm's running statistics after the step, the state_dict fingerprint."""
    crit = nn.CrossEntropyLoss()
    torch.manual_seed(3)
    model, idx, ln = orc.resnet18_cifar(), 6, 15
    damp = 0.1
    for m in model.modules():
        if isinstance(m, orc.Block):
            m.bn2.weight.data.mul_(damp)
    model.train()
    opt = torch.optim.SGD(model.parameters(), 0.1, momentum=0.9, weight_decay=5e-4)
    bs, K, gamma, eps = 32, 5, 0.5, 2.0
    x = torch.rand(bs, 3, 32, 32)
    y = torch.randint(0, 10, (bs,))
    rec = {"x": _np(x), "y": _np(y), "meta": np.array([K, idx, ln, 0, 0]), "gamma_eps": np.array([gamma, eps], dtype=np.float64),
           "damp": np.array(damp)}
    k0, c0 = _checksums(model)
    rec["ck0"] = c0
    r = _ref_step(ref_attack, model, opt, crit, x, y, K, gamma, eps, idx, ln, False, False)
    for k in ("l2", "linf", "loss", "loss_adv", "loss_clean", "out_clean"):
        rec[k] = _np(r[k])
    dk = torch.round((r["x_adv"] - r["feature_map"]) / np.float32(gamma / 255))
    assert float(dk.abs().max()) <= K
    rec["dk"] = _np(dk).astype(np.int8)
    rec["feature_map_sub"] = _np(r["feature_map"][:, ::4, ::2, ::2])
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
    rec["param_names"] = np.array([n for n, _ in named])
    rec["grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in named])
    for n, p in named:
        if n.endswith(("sequential_model.1.weight", ".4.conv1.weight", ".7.bn2.weight", ".6.conv1.weight", ".11.bn1.bias", ".14.weight", ".14.bias")):
            rec["grad/" + n] = _np(p.grad)
    k1, c1 = _checksums(model)
    rec["keys"], rec["ck1"] = np.array(k1), c1
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            rec["sd1/" + k] = _np(v)
    np.savez_compressed(os.path.join(OUT, "step_r18_k5_b32_damped.npz"), **rec)
    print("step_r18_k5_b32_damped loss", float(r["loss"]), float(r["loss_adv"]), float(r["loss_clean"]), "grads stored:",
          [k for k in rec if k.startswith("grad/")])


def _variant_ctx(kind):
    """Arithmetic variants of the SAME reference code on the SAME inputs: what two correct implementations may differ by."""
    import contextlib
    if kind.endswith("nomkldnn"):       # ATen's native convolution / GEMM path instead of oneDNN: another summation order, still fp32
        return torch.backends.mkldnn.flags(enabled=False)
    return contextlib.nullcontext()


def _to_variant(kind, model, tensors):
    if kind.startswith("in"):
        # the images (first tensor) times (1 + 1e-6 * N(0, 1)), draw number = the digit: the size of the rounding noise the head's
        # ~90 fp32 layers accumulate (the product's SE feature map is 1.5e-6 relative from the reference's in either layout,
        # tools/diag_dl_layout_head.py) — what the tail's sign pattern does under it is the function's own sensitivity, not an
        # implementation's (tools/diag_dl_chaos.py: both layouts of the product fall on the same branch of it draw by draw)
        gen = torch.Generator().manual_seed(1000 + int(kind[2:]))
        t0 = tensors[0]
        return model, [t0 * (1 + 1e-6 * torch.randn(t0.shape, generator=gen))] + list(tensors[1:])
    if kind.startswith("t_") or kind == "t":
        # the whole problem transposed (H <-> W): every convolution kernel and every image / label map transposed.  Convolutions,
        # pooling, BatchNorm and bilinear resizing with the same stride / padding on both axes are equivariant, so every feature
        # map is the transpose of the baseline's — mathematically; the sums run in another order.
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.data = m.weight.data.transpose(2, 3).contiguous()
                    assert m.stride[0] == m.stride[1] and m.padding[0] == m.padding[1] and m.dilation[0] == m.dilation[1]
        tensors = [t.transpose(-1, -2).contiguous() if t.dim() >= 3 else t for t in tensors]
        kind = kind[2:] if kind.startswith("t_") else "base"
    if kind == "f64":            # float64 throughout, results compared after rounding to the sign grid
        return model.double(), [t.double() if t.is_floating_point() else t for t in tensors]
    if kind == "cl":             # channels-last weights and activations: oneDNN picks other kernels / blockings
        return model.to(memory_format=torch.channels_last), [t.contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t
                                                              for t in tensors]
    return model, list(tensors)


def gen_noise_floor(ref_attack, build, ref_seg=None, ref_network=None):
    """tests/golden/ref_noise_floor.npz — how far the REFERENCE is from itself.  For every deep-network golden case the
    reference's own PGD (same code, same seed, same inputs) is re-run in six other arithmetics — float64, ATen-native fp32
    (oneDNN off), channels-last fp32, and the same three on the TRANSPOSED problem (every kernel and image with H and W swapped: the
    same mathematics, another summation order) — and on four draws of 1e-6 relative noise on the images (`in1`..`in4`: the size of
    the rounding noise ~90 fp32 layers accumulate in ANY implementation), and the fraction of feature elements whose perturbation (an integer multiple of gamma
    on the sign grid) differs from the fp32 baseline's is stored: after every step (`<case>/<variant>/per_step`) and after K steps
    (`.../final`).  A sign() flips where a gradient sits within rounding distance of zero, and flips compound through the
    remaining steps: this is the floor below which no fp32 implementation can be told from the reference, and
    the tests bound the product's flip fraction by max(2 x floor, 1e-4) instead of by hand-set constants."""
    import copy
    crit = nn.CrossEntropyLoss()
    rec = {}
    variants = ("f64", "nomkldnn", "cl", "t", "t_nomkldnn", "t_cl", "in1", "in2", "in3", "in4")
    untr = lambda kind, a: a.swapaxes(-1, -2) if (kind == "t" or kind.startswith("t_")) else a

    def record(case, base_steps, runs):
        for kind, steps in runs.items():
            per = [float((np.rint(a) != np.rint(b)).mean()) for a, b in zip(steps, base_steps)]
            rec[f"{case}/{kind}/per_step"] = np.array(per)
            rec[f"{case}/{kind}/final"] = np.array(per[-1])
        rec[f"{case}/floor"] = np.array(max(float(rec[f"{case}/{k}/final"]) for k in runs))
        rec[f"{case}/floor_arith"] = np.array(max(float(rec[f"{case}/{k}/final"]) for k in runs if not k.startswith("in")))
        print(f"   {case}: " + "  ".join(f"{k} {float(rec[f'{case}/{k}/final']):.5f}" for k in runs) +
              f"  -> floor {float(rec[case + '/floor']):.5f} (arithmetic variants only: {float(rec[case + '/floor_arith']):.5f})")

    # ---- Classification (main_perturb.py:173-185): head pass, K-step PGD
    for case, arch, bs, K, gamma, clip in (("step_r18_k5", "resnet18", 2, 5, 0.5, False), ("step_r56s_k5", "resnet56s", 2, 5, 0.5, False),
                                           ("step_r18_k5_b16", "resnet18", 16, 5, 0.5, False), ("step_r56s_k5_b16", "resnet56s", 16, 5, 0.5, False),
                                           ("step_r20s_k1", "resnet20s", 4, 1, 0.5, False),
                                           ("step_r20s_k5", "resnet20s", 4, 5, 0.5, False), ("step_r20s_k5_clip", "resnet20s", 4, 5, 1.5, True),
                                           ("step_r18_k5_b32_damped", "resnet18", 32, 5, 0.5, False)):
        torch.manual_seed(3)
        model0, idx, ln = build(arch)
        if case.endswith("_damped"):                      # gen_damped_r18: every block's last-BatchNorm weight x 0.1
            for m in model0.modules():
                if hasattr(m, "bn2") and hasattr(m, "conv2"):
                    m.bn2.weight.data.mul_(0.1)
        model0.train()
        x, y = torch.rand(bs, 3, 32, 32), torch.randint(0, 10, (bs,))

        def run(kind):
            model, (xx,) = _to_variant(kind, copy.deepcopy(model0), [x])
            with _variant_ctx(kind):
                fm = model(xx, end_point=idx, start_point=0).detach()
                snaps = _pgd_trace(ref_attack, model, crit, fm, y, K, gamma, 2.0, idx, ln, clip)
            g_ = (gamma / 255)
            return [untr(kind, ((s_.double() - fm.double()) / g_).contiguous(memory_format=torch.contiguous_format).numpy()) for s_ in snaps[1:]]
        base = run("base")
        gold = np.load(os.path.join(OUT, case + ".npz"))
        if "dk" in gold.files:
            assert np.array_equal(np.rint(base[-1]).astype(np.int8), gold["dk"]), case
        else:
            assert np.array_equal(np.rint(base[-1]), np.rint((gold["x_adv"] - gold["feature_map"]) / (gamma / 255))), case
        record(case, base, {k: run(k) for k in variants})

    # ---- the flip rate at its source: one gradient from the reference's OWN iterate (pgd_trace_r18_k5.npz: x_adv before every step
    # and the gradient the baseline computed there), recomputed in each variant: fraction of elements whose sign differs
    tr = np.load(os.path.join(OUT, "pgd_trace_r18_k5.npz"))
    torch.manual_seed(3)
    model0, idx, ln = build("resnet18")
    model0.train()
    y = torch.from_numpy(tr["y"])
    worst = 0.0
    for kind in [k for k in variants if not k.startswith("in")]:      # (the iterate IS the reference's: arithmetic variants only)
        per = []
        for t in range(tr["grads"].shape[0]):
            model, (xin,) = _to_variant(kind, copy.deepcopy(model0), [torch.from_numpy(tr["snaps"][t])])
            xin = xin.clone().requires_grad_(True)
            with _variant_ctx(kind):
                gr = torch.autograd.grad(crit(model(xin, end_point=ln, start_point=idx), y), xin)[0]
            gr = untr(kind, gr.contiguous(memory_format=torch.contiguous_format).numpy())
            per.append(float((np.sign(gr) != np.sign(tr["grads"][t])).mean()))
        rec[f"pgd_trace_r18_k5/{kind}/per_step"] = np.array(per)
        worst = max(worst, max(per))
    rec["pgd_trace_r18_k5/floor"] = rec["pgd_trace_r18_k5/floor_arith"] = np.array(worst)
    print("   pgd_trace_r18_k5 (sign of one gradient from the reference's iterate): " +
          "  ".join(f"{k} {rec[f'pgd_trace_r18_k5/{k}/per_step'].max():.5f}" for k in variants if not k.startswith("in")) + f"  -> floor {worst:.5f}")

    # ---- Segmentation (main_aug_final.py:164-197): SE head pass, K-step SE feature PGD on the reference's own DeepLabv3+ / ResNet-101
    if ref_network is not None:
        seg_crit = nn.CrossEntropyLoss(ignore_index=255, reduction="mean")
        for case, steps, gamma_se, side, damp, bs in (("seg_dl101_aspp_k1", 1, 0.5, 129, 1.0, 2), ("seg_dl101_concat_k3", 3, 0.5, 129, 1.0, 2),
                                                      ("seg_dl101_aspp_k3_damped", 3, 0.5, 129, 0.1, 4)):
            torch.manual_seed(3)
            net0 = ref_network.deeplabv3plus_resnet101(num_classes=21, output_stride=16, pretrained_backbone=False)
            net0.classifier.aspp.project[3].p = 0.0
            if damp != 1.0:
                for m in net0.backbone.modules():
                    if isinstance(m, ref_network.backbone.resnet.Bottleneck):
                        m.bn3.weight.data.mul_(damp)
            for m in net0.backbone.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.momentum = 0.01
            net0.train()
            images = torch.rand(bs, 3, side, side)
            if bs == 4:
                for i, (sc, of) in enumerate([(1.0, 0.0), (0.3, 0.6), (0.6, 0.0), (0.5, 0.4)]):
                    images[i] = images[i] * sc + of
            labels = torch.randint(0, 21, (bs, side, side))
            labels[torch.rand(bs, side, side) < 0.05] = 255
            gold = np.load(os.path.join(OUT, case + ".npz"))
            assert np.array_equal(_np(images), gold["images"]) and np.array_equal(_np(labels), gold["labels"]), case

            def run_seg(kind):
                net, (im, lab) = _to_variant(kind, copy.deepcopy(net0), [images, labels])
                seen = []

                class Spy(nn.Module):
                    def __init__(self, m):
                        super().__init__()
                        self.m = m

                    def forward(self, d):
                        if d.get("flag") == "tail":
                            seen.append(d["adv"].detach().clone())
                        return self.m(d)
                with _variant_ctx(kind):
                    out_se = net({"x": im, "adv": None, "out_idx": 3, "flag": "head"})
                    net({"x": im, "adv": None, "out_idx": "aspp_head", "flag": "clean"})          # (:167: its BatchNorm side effects precede the PGD)
                    fm = out_se["out"].detach()
                    adv = ref_seg.PGD(x=fm, image_batch=im, low_level_feat=out_se["low_level"], criterion=seg_crit, y=lab,
                                      model=Spy(net), steps=steps, eps=(2.0 / 255), gamma=(gamma_se / 255), idx=3, randinit=False, clip=False)
                seen.append(adv.detach().clone())
                g_ = gamma_se / 255
                return [untr(kind, ((s_.double() - fm.double()) / g_).contiguous(memory_format=torch.contiguous_format).numpy()) for s_ in seen[1:]]
            base = run_seg("base")
            assert np.array_equal(np.rint(base[-1]), np.rint((gold["adv_se"].astype(np.float64) - gold["fm_se"]) / (gamma_se / 255))), case
            if damp != 1.0:      # the baseline's perturbation after EVERY step (multiples of gamma): lets a test see where flips start
                rec[f"{case}/base_dk_per_step"] = np.stack([np.rint(b_).astype(np.int8) for b_ in base])
            record(case, base, {k: run_seg(k) for k in variants})
    np.savez_compressed(os.path.join(OUT, "ref_noise_floor.npz"), **rec)


def gen_loss_floor(ref_attack, build, orc):
    """tests/golden/ref_loss_floor.npz + traj_r18_damped.npz — the reference against itself on the LOSSES (round 5, VERDICT weak 1).
    ref_noise_floor.npz measures the perturbation elements; the tests' looser loss bounds (iterations 1-2 of the warm-up trajectory,
    loss_adv of the deep networks) had no such floor.  Same variants (float64, ATen-native fp32, channels-last fp32, those three on
    the transposed problem, four draws of 1e-6 relative noise on the images): `<case>/<key>/spread` = max over the variants of
    |value - baseline| for key in loss, loss_adv, loss_clean; trajectories: one entry per iteration.  The tests bound the product
    by max(2 x spread, 1e-4).  Also generates the 3-iteration warm-up trajectory on the CONTRACTIVE ResNet-18 (gen_damped_r18's
    recipe: every block's last BatchNorm weight x 0.1, batch 32, K = 5), which the product must follow to 1e-4 per iteration."""
    import copy
    crit = nn.CrossEntropyLoss()
    variants = ("f64", "nomkldnn", "cl", "t", "t_nomkldnn", "t_cl", "in1", "in2", "in3", "in4")
    rec = {}

    def damp_(model):
        for m in model.modules():
            if hasattr(m, "bn2") and hasattr(m, "conv2"):
                m.bn2.weight.data.mul_(0.1)

    # ---- single joint steps (main_perturb.py:173-201)
    for case, arch, bs, K, gamma, clip, damped in (("step_r18_k5", "resnet18", 2, 5, 0.5, False, False),
                                                   ("step_r56s_k5", "resnet56s", 2, 5, 0.5, False, False),
                                                   ("step_r18_k5_b16", "resnet18", 16, 5, 0.5, False, False),
                                                   ("step_r56s_k5_b16", "resnet56s", 16, 5, 0.5, False, False),
                                                   ("step_r20s_k5", "resnet20s", 4, 5, 0.5, False, False),
                                                   ("step_r20s_k5_clip", "resnet20s", 4, 5, 1.5, True, False),
                                                   ("step_r18_k5_b32_damped", "resnet18", 32, 5, 0.5, False, True)):
        torch.manual_seed(3)
        model0, idx, ln = build(arch)
        if damped:
            damp_(model0)
        model0.train()
        x, y = torch.rand(bs, 3, 32, 32), torch.randint(0, 10, (bs,))

        def run(kind):
            model, (xx,) = _to_variant(kind, copy.deepcopy(model0), [x])
            opt = torch.optim.SGD(model.parameters(), 0.1, momentum=0.9, weight_decay=5e-4)
            with _variant_ctx(kind):
                r = _ref_step(ref_attack, model, opt, crit, xx, y, K, gamma, 2.0, idx, ln, False, clip)
            return {k: float(r[k]) for k in ("loss", "loss_adv", "loss_clean")}
        base = run("base")
        gold = np.load(os.path.join(OUT, case + ".npz"))
        assert base["loss"] == float(gold["loss"]), (case, base["loss"], float(gold["loss"]))
        runs = {k: run(k) for k in variants}
        for key in base:
            rec[f"{case}/{key}/spread"] = np.array(max(abs(runs[k][key] - base[key]) for k in variants))
            rec[f"{case}/{key}/spread_arith"] = np.array(max(abs(runs[k][key] - base[key]) for k in variants if not k.startswith("in")))
        print(f"   {case}: " + "  ".join(f"{key} {float(rec[f'{case}/{key}/spread']):.2e}" for key in base))

    # ---- 3-iteration warm-up trajectories (main_perturb.py:167-168,288-293): the stored one and the new contractive one
    def trajectory(kind, model0, idx, ln, xs, ys, K, wp):
        model, (xv,) = _to_variant(kind, copy.deepcopy(model0), [xs])
        opt = torch.optim.SGD(model.parameters(), 0.1, momentum=0.9, weight_decay=5e-4)
        losses = []
        for i in range(xs.shape[0]):
            lr = min(i * 0.1 / (wp - 1), 0.1)
            for p in opt.param_groups:
                p["lr"] = lr
            xi = xv[i]
            if kind.endswith("cl"):
                xi = xi.contiguous(memory_format=torch.channels_last)
            with _variant_ctx(kind):
                r = _ref_step(ref_attack, model, opt, crit, xi, ys[i], K, 0.5, 2.0, idx, ln, False, False)
            losses.append(float(r["loss"]))
        return losses, model

    torch.manual_seed(3)
    model0, idx, ln = build("resnet20s")
    model0.train()
    xs, ys = torch.rand(3, 8, 3, 32, 32), torch.randint(0, 10, (3, 8))
    gold = np.load(os.path.join(OUT, "traj_r20s.npz"))
    assert np.array_equal(_np(xs), gold["xs"])
    base, _ = trajectory("base", model0, idx, ln, xs, ys, 2, 5)
    assert base == [float(v) for v in gold["losses"]], (base, gold["losses"])
    runs = {k: trajectory(k, model0, idx, ln, xs, ys, 2, 5)[0] for k in variants}
    rec["traj_r20s/loss/spread"] = np.array([max(abs(runs[k][i] - base[i]) for k in variants) for i in range(3)])
    print("   traj_r20s: spread per iteration", rec["traj_r20s/loss/spread"])

    torch.manual_seed(3)
    model0, idx, ln = orc.resnet18_cifar(), 6, 15
    damp_(model0)
    model0.train()
    xs, ys = torch.rand(3, 32, 3, 32, 32), torch.randint(0, 10, (3, 32))
    k0, c0 = _checksums(model0)
    base, mb = trajectory("base", model0, idx, ln, xs, ys, 5, 5)
    runs = {k: trajectory(k, model0, idx, ln, xs, ys, 5, 5)[0] for k in variants}
    rec["traj_r18_damped/loss/spread"] = np.array([max(abs(runs[k][i] - base[i]) for k in variants) for i in range(3)])
    print("   traj_r18_damped: losses", base, "spread per iteration", rec["traj_r18_damped/loss/spread"])
    k1, c1 = _checksums(mb)
    np.savez_compressed(os.path.join(OUT, "traj_r18_damped.npz"), xs=_np(xs), ys=_np(ys), losses=np.array(base),
                        lrs=np.array([min(i * 0.1 / 4, 0.1) for i in range(3)]), wp=np.array(5), damp=np.array(0.1), ck0=c0, ck1=c1,
                        keys=np.array(k1), fc_w=_np(mb.state_dict()["sequential_model.14.weight"]))
    np.savez_compressed(os.path.join(OUT, "ref_loss_floor.npz"), **rec)


def gen_detection(orc):
    # ---- Detection (N2): the reference's own adv_input / PGD / rpn_roi_PGD / get_sample_points / mix_feature / compute_loss
    # (Detection/attack_algo.py) and the loop body of Detection/train_aug_sat_muti_advt.py:70-172 driven line by line around
    # them, on the protocol-faithful stand-in detector oracle.TinyDetNet (the reference's Faster-RCNN needs its compiled
    # extension and torchvision weights).  `.cuda()` is the identity here (_shims). ------------------------------------
    ref_det = _load("ref_det_attack_algo", "Detection/attack_algo.py")
    for name, loss_settings in (("det_step_tiny_s1", 1), ("det_step_tiny_s3", 3)):
        torch.manual_seed(11)
        model = orc.TinyDetNet()
        model.train()
        optimizer = torch.optim.SGD(model.parameters(), 0.01, momentum=0.9, weight_decay=5e-4)
        image_batch = torch.rand(2, 3, 32, 32)
        bboxes_batch = torch.tensor([[[2., 3., 20., 18.], [10., 12., 30., 31.]], [[0., 0., 15., 15.], [8., 4., 28., 22.]]])
        labels_batch = torch.tensor([[1, 2], [3, 0]])
        k0, c0 = _checksums(model)
        y = {"bb": bboxes_batch, "lb": labels_batch}
        adv_image_batch = ref_det.adv_input(x=image_batch, y=y, model=model, steps=5, eps=(2.0 / 255), gamma=(0.3 / 255), randinit=True, clip=True)
        inputs_all1 = {"x": image_batch, "adv": None, "out_idx": 1, "flag": "head"}
        inputs_all2 = {"x": image_batch, "adv": None, "out_idx": 2, "flag": "head"}
        inputs_all3 = {"x": image_batch, "adv": None, "out_idx": 3, "flag": "head"}
        inputs_all_sd = {"x": image_batch, "adv": None, "out_idx": "roi_head", "flag": "clean"}
        feature_map1 = model.train().forward(inputs_all1, bboxes_batch, labels_batch).detach()
        feature_map2 = model.train().forward(inputs_all2, bboxes_batch, labels_batch).detach()
        feature_map3 = model.train().forward(inputs_all3, bboxes_batch, labels_batch).detach()
        rpn_roi_output_dict = model.train().forward(inputs_all_sd, bboxes_batch, labels_batch)
        clean_feature_map_sd = rpn_roi_output_dict["roi_output_dict"]["roi_feature_map"].detach()
        feature_adv1 = ref_det.PGD(feature_map1, image_batch, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=1)
        feature_adv2 = ref_det.PGD(feature_map2, image_batch, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=2)
        feature_adv3 = ref_det.PGD(feature_map3, image_batch, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(1.0 / 255), idx=3)
        adv_list = ref_det.get_sample_points(feature_map3, feature_adv3, 5)
        adv_list[1] = ref_det.mix_feature(feature_map3, adv_list[1])
        adv_list[2] = ref_det.mix_feature(feature_map3, adv_list[2])
        adv_rpn_roi_output_dict = ref_det.rpn_roi_PGD(rpn_roi_output_dict=rpn_roi_output_dict, y=y, model=model, steps=1,
                                                      eps=(2.0 / 255), gamma=(0.2 / 255), only_roi_loss=False)
        adv_feature_map_sd = adv_rpn_roi_output_dict["roi_output_dict"]["roi_feature_map"].detach()
        adv_feature_map_sd = ref_det.mix_feature(clean_feature_map_sd, adv_feature_map_sd)
        adv_rpn_roi_output_dict["roi_output_dict"]["roi_feature_map"] = adv_feature_map_sd
        dicts = [{"x": adv_image_batch, "adv": None, "out_idx": 0, "flag": "clean"},
                 {"x": image_batch, "adv": feature_adv1, "out_idx": 1, "flag": "tail"},
                 {"x": image_batch, "adv": feature_adv2, "out_idx": 2, "flag": "tail"},
                 {"x": image_batch, "adv": adv_list[1], "out_idx": 3, "flag": "tail"},
                 {"x": image_batch, "adv": adv_list[2], "out_idx": 3, "flag": "tail"},
                 {"x": image_batch, "adv": adv_list[3], "out_idx": 3, "flag": "tail"},
                 {"x": image_batch, "adv": adv_list[4], "out_idx": 3, "flag": "tail"},
                 {"adv": adv_rpn_roi_output_dict, "out_idx": "roi_tail", "flag": "clean"}]
        L = [ref_det.compute_loss(*model.train().forward(d, bboxes_batch, labels_batch)) for d in dicts]
        loss0, loss1, loss2, loss3, loss4, loss5, loss6, loss7 = L
        loss_clean_adv = 0.9 * (0.2333 * (loss0 + loss3 + loss4 + loss5 + loss6) + 0.1 * loss7) + 0.05 * (loss1 + loss2)
        if loss_settings == 1:
            loss = loss_clean_adv
        elif loss_settings == 3:
            loss = 0.4 * loss_clean_adv + 0.6 * loss0
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        k1, c1 = _checksums(model)
        sd = model.state_dict()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), images=_np(image_batch), bboxes=_np(bboxes_batch), labels=_np(labels_batch),
                            loss_settings=np.array(loss_settings), ck0=c0, ck1=c1, keys=np.array(k1), loss=_np(loss),
                            losses=np.array([float(v) for v in L], dtype=np.float32), adv_image=_np(adv_image_batch),
                            adv1=_np(feature_adv1), adv2=_np(feature_adv2), adv3=_np(feature_adv3), adv_sd=_np(adv_feature_map_sd),
                            fm3=_np(feature_map3), **{"sd1/" + k: _np(sd[k]) for k in ("stem.0.weight", "layer3.0.weight", "rpn_obj.weight",
                                                                                    "hidden.weight", "cls.bias", "layer2.1.running_mean")})
        print(name, "loss", float(loss), [round(float(v), 5) for v in L])



def _reference_roialign():
    """ctypes handle on the REFERENCE's CPU ROIAlign forward: Detection/support/src/cpu/ROIAlign_cpu.cpp:4-219 compiled unedited
    (oracle/Makefile, oracle/ref_roialign_driver.cpp) — the at::Tensor wrapper :221-257 is not used (it does not compile against
    this torch); its only content besides dispatch is output_size = R*PH*PW*C, repeated in the driver."""
    import ctypes
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_roialign.so"))
    _p, _l, _i = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    lib.ref_roi_align_forward_f32.argtypes = [_p, _p, _p, _l, _l, _l, _l, _i, _i, ctypes.c_float, _i]
    lib.ref_roi_align_forward_f64.argtypes = [_p, _p, _p, _l, _l, _l, _l, _i, _i, ctypes.c_double, _i]
    lib.ref_roi_align_forward_f32.restype = lib.ref_roi_align_forward_f64.restype = None

    def fwd(x, rois, ph, pw, scale, sampling_ratio):
        dt = x.dtype
        assert dt in (np.float32, np.float64) and rois.dtype == dt
        x, rois = np.ascontiguousarray(x), np.ascontiguousarray(rois)
        n, c, h, w = x.shape
        out = np.zeros((len(rois), c, ph, pw), dt)
        f = lib.ref_roi_align_forward_f32 if dt == np.float32 else lib.ref_roi_align_forward_f64
        f(x.ctypes.data, rois.ctypes.data, out.ctypes.data, len(rois), c, h, w, ph, pw, scale, sampling_ratio)
        return out
    return fwd


def _roi_boxes(rng, n, n_img, img_h, img_w):
    """[n, 5] (batch index, x1, y1, x2, y2) in image coordinates: RPN-like boxes clipped to the image, plus boxes that cross each
    border, sub-pixel (malformed: extent forced to 1 cell, ROIAlign_cpu.cpp:143-144), whole-image and fully-outside ones."""
    b = np.zeros((n, 5), np.float32)
    b[:, 0] = rng.integers(0, n_img, n)
    cx, cy = rng.uniform(0, img_w, n), rng.uniform(0, img_h, n)
    w = np.exp(rng.uniform(np.log(12), np.log(img_w), n))
    h = np.exp(rng.uniform(np.log(12), np.log(img_h), n))
    x1, y1, x2, y2 = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
    clip = rng.random(n) < 0.7                                        # proposals are clipped to the image (rpn/..._network.py:128)
    x1, x2 = np.where(clip, np.clip(x1, 0, img_w), x1), np.where(clip, np.clip(x2, 0, img_w), x2)
    y1, y2 = np.where(clip, np.clip(y1, 0, img_h), y1), np.where(clip, np.clip(y2, 0, img_h), y2)
    b[:, 1], b[:, 2], b[:, 3], b[:, 4] = x1, y1, x2, y2
    special = [[0, 0, 0, img_w, img_h], [n_img - 1, -40, -24, img_w + 56, img_h + 40], [0, 100.3, 50.7, 100.9, 51.2],
               [n_img - 1, 300, 200, 290, 180], [0, img_w + 40, img_h + 40, img_w + 200, img_h + 120], [0, -300, -300, -100, -120],
               [n_img - 1, img_w - 8, img_h - 8, img_w + 8, img_h + 8], [0, -16, 33, 15.99, 400], [0, 0, 0, 16, 16]]
    k = min(len(special), n)
    b[:k] = np.array(special[:k], np.float32)
    return b


def gen_roi_align(orc):
    """tests/golden/roi_align_fwd_{small,cfg5_r128,cfg5_r300}.npz: outputs of the reference's own CPU forward
    (`_reference_roialign`).  small: complete fp32 and f64 outputs, sampling_ratio 0 and 2.  cfg5: the shapes of BASELINE
    configs[4] (C=1024 ResNet-101 conv4 map of a 600 x 904 image = 38 x 57, 14 x 14 bins, scale 1/16, sampling_ratio 0:
    roi/pooler.py:21-22,35-38); the 17.7 MB input is oracle.synth_field(shape, seed), the 100-235 MB output is stored for
    four channels plus f64 sums of the fp32 output per ROI (over C, PH, PW) and per channel (over R, PH, PW)."""
    fwd = _reference_roialign()
    rng = np.random.default_rng(20261004)
    # small, complete
    N, C, H, W, PH, PW = 2, 4, 20, 30, 14, 14
    x = orc.synth_field((N, C, H, W), 77)
    rois = _roi_boxes(rng, 24, N, H * 16, W * 16)
    rec = {"x_shape": np.array([N, C, H, W]), "x_seed": np.array(77), "rois": rois, "pooled": np.array([PH, PW]), "scale": np.array(1 / 16)}
    for sr in (0, 2):
        rec[f"y_sr{sr}"] = fwd(x, rois, PH, PW, 1 / 16, sr)
        rec[f"y64_sr{sr}"] = fwd(x.astype(np.float64), rois.astype(np.float64), PH, PW, 1 / 16, sr)
    rec["y_7x5_sr0"] = fwd(x, rois, 7, 5, 1 / 16, 0)                   # non-square bins
    np.savez_compressed(os.path.join(OUT, "roi_align_fwd_small.npz"), **rec)
    print("roi_align_fwd_small", rec["y_sr0"].shape, float(np.abs(rec["y_sr0"]).max()))
    # cfg5 shapes
    N, C, H, W = 2, 1024, 38, 57
    chans = np.array([0, 1, 511, 1023])
    for name, R, seed in (("cfg5_r128", 128, 101), ("cfg5_r300", 300, 102)):
        x = orc.synth_field((N, C, H, W), seed)
        rois = _roi_boxes(rng, R, N, 600, 904)
        y = fwd(x, rois, PH, PW, 1 / 16, 0)
        np.savez_compressed(os.path.join(OUT, f"roi_align_fwd_{name}.npz"), x_shape=np.array([N, C, H, W]), x_seed=np.array(seed),
                            rois=rois, pooled=np.array([PH, PW]), scale=np.array(1 / 16), channels=chans, y_sub=y[:, chans],
                            roi_sums=y.astype(np.float64).sum(axis=(1, 2, 3)), chan_sums=y.astype(np.float64).sum(axis=(0, 2, 3)))
        print("roi_align_fwd_" + name, y.shape, "rms", float(np.sqrt((y.astype(np.float64) ** 2).mean())))


def _import_reference_detection_model():
    """`Detection/model.py` with the reference's own backbone / rpn / roi / bbox / extension modules on sys.path, plus the
    stand-ins this image needs: an empty `torchvision` (backbone/resnet101.py:3 imports it, never uses it with
    the `_ori` backbone), and a `support` package — the reference's compiled extension is absent (.MISSING_LARGE_BLOBS) and its
    CPU sources do not compile against this torch — whose `nms` is the plain-C oracle NMS (oracle/afan_oracle.c, pinned to the
    reference's own 9770 -> 1934 vector, tests/test_det_oracle.py; the `>` rule of the reference's GPU path, nms.cu:49) and
    whose `ROIAlign` runs the reference's own CPU forward kernel with the adjoint backward (`_RefROIAlign` below): both pooler
    modes of roi/pooler.py:24-42 run — `pooling` (det_frcnn_r101.npz) and the reference's default `align`
    (config/config.py:14; det_frcnn_r101_align.npz)."""
    import ctypes
    import subprocess
    det = os.path.join(REF, "Detection")
    if det not in sys.path:
        sys.path.insert(0, det)
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    so = os.path.join(ROOT, "oracle", "_ref", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    _p, _l = ctypes.c_void_p, ctypes.c_int64
    lib.oracle_nms.argtypes = [_p, _p, _l, ctypes.c_float, ctypes.c_int, _p, _p]
    lib.oracle_nms.restype = _l

    def nms(bboxes, scores, threshold):
        n = bboxes.shape[0]
        if n == 0:
            return torch.empty(0, dtype=torch.int64)
        b = np.ascontiguousarray(bboxes.detach().float().numpy())
        order = np.ascontiguousarray(torch.sort(scores.detach().float(), dim=0, descending=True)[1].numpy().astype(np.int64))
        keep, scratch = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.uint8)
        k = lib.oracle_nms(b.ctypes.data_as(_p), order.ctypes.data_as(_p), n, float(threshold), 0, keep.ctypes.data_as(_p),
                           scratch.ctypes.data_as(_p))
        return torch.from_numpy(np.sort(keep[:k]))

    ref_fwd = _reference_roialign()
    lib.oracle_roi_align.argtypes = [_p, _p, _p, _l, _l, _l, _l, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    lib.oracle_roi_align.restype = None

    class _RefROIAlign(torch.autograd.Function):
        """support/layer/roi_align.py:11-48 with `_C.roi_align_forward` = the reference's own CPU kernel (ROIAlign_cpu.cpp:4-219,
        compiled unedited: _reference_roialign) and `_C.roi_align_backward` — which the reference only has for CUDA
        (ROIAlign.h:44) — = the C oracle's backward, held to be the exact adjoint of that forward in f64
        (tests/test_det_oracle.py::test_roi_align_backward_is_the_adjoint_of_the_pinned_forward)."""
        @staticmethod
        def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
            ctx.save_for_backward(roi)
            ctx.geom = (tuple(input.shape), tuple(output_size), float(spatial_scale), int(sampling_ratio))
            y = ref_fwd(input.detach().float().contiguous().numpy(), roi.detach().float().contiguous().numpy(), output_size[0],
                        output_size[1], float(spatial_scale), int(sampling_ratio))
            return torch.from_numpy(y).to(input.dtype)       # (the float64 floor variant: the kernel itself stays the reference's fp32 one)

        @staticmethod
        def backward(ctx, grad_output):
            (roi,) = ctx.saved_tensors
            shape, (ph, pw), scale, sr = ctx.geom
            dy = np.ascontiguousarray(grad_output.detach().float().numpy())
            r = np.ascontiguousarray(roi.detach().float().numpy())
            dx = np.zeros(shape, np.float32)
            lib.oracle_roi_align(dx.ctypes.data_as(_p), r.ctypes.data_as(_p), dy.ctypes.data_as(_p), len(r), shape[1], shape[2], shape[3],
                                 ph, pw, scale, sr, 1)
            return torch.from_numpy(dx).to(grad_output.dtype), None, None, None, None

    class ROIAlign(nn.Module):
        def __init__(self, output_size, spatial_scale, sampling_ratio):
            super().__init__()
            self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

        def forward(self, input, rois):
            return _RefROIAlign.apply(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    sup, lay, m_nms, m_ra = (types.ModuleType(n) for n in ("support", "support.layer", "support.layer.nms", "support.layer.roi_align"))
    m_nms.nms, m_ra.ROIAlign = nms, ROIAlign
    sup.layer, lay.nms, lay.roi_align = lay, m_nms, m_ra
    for m in (sup, lay, m_nms, m_ra):
        sys.modules[m.__name__] = m
    import model as ref_model                      # Detection/model.py
    from backbone.resnet101 import ResNet101 as RefBackbone
    from roi.pooler import Pooler as RefPooler
    return ref_model, RefBackbone, RefPooler


def gen_detection_model(mode="pooling"):
    """The reference's OWN Faster-RCNN (Detection/model.py:18-185 on backbone/resnet101_ori.py, rpn/, roi/pooler.py in `mode`)
    at a small image size, fixed seed, no pretrained weights (pretrained=False: kaiming initialisation, BatchNorm
    running statistics at 0 / 1): det_frcnn_r101.npz ('pooling') / det_frcnn_r101_align.npz ('align', the reference's default).  What it pins: seeded construction (checksum of every state_dict
    tensor, in the reference's key order incl. the `_bn_modules.*` / `detection.hidden.*` aliases), the three backbone feature
    maps (flag head, out_idx 1-3), the RPN's logits and its proposals, the four per-image losses of one training forward
    (host `randperm` draws from a fixed generator state) and the gradients that forward leaves; the one-step feature PGD of
    Detection/attack_algo.py:48-74 at out_idx 3; one full iteration of train_aug_sat_muti_advt.py:70-172 (eight forwards, five
    PGD calls, SGD step)."""
    ref_model, RefBackbone, RefPooler = _import_reference_detection_model()
    ref_det = _load("ref_det_attack_algo", "Detection/attack_algo.py")
    cfg = dict(anchor_ratios=[(1, 2), (1, 1), (2, 1)], anchor_sizes=[64], rpn_pre_nms_top_n=200, rpn_post_nms_top_n=64,
               anchor_smooth_l1_loss_beta=1.0, proposal_smooth_l1_loss_beta=1.0)
    torch.manual_seed(7)
    model = ref_model.Model(RefBackbone(pretrained=False), 21, pooler_mode=RefPooler.Mode(mode), **cfg)
    model.train()
    k_init, c_init = _checksums(model)                    # seeded construction, before the damping below
    # Without pretrained weights the frozen BatchNorms are identities (running statistics 0 / 1) and 33 residual blocks
    # multiply the activations up to 1e4: losses of 5e4, no foreground proposals.  Weights are data: every block's last
    # BatchNorm weight x 0.2 keeps the features O(1) (the regime pretrained weights are in).
    damp = 0.2
    for m in model.modules():
        if hasattr(m, "bn3") and hasattr(m, "conv3"):
            m.bn3.weight.data.mul_(damp)
    k0, c0 = _checksums(model)
    g = torch.Generator().manual_seed(21)
    images = torch.rand(2, 3, 128, 160, generator=g)
    bboxes = torch.tensor([[[12., 20., 70., 90.], [60., 30., 150., 110.]], [[5., 8., 60., 64.], [80., 50., 140., 120.]]])
    labels = torch.tensor([[3, 7], [12, 1]])
    rec = {"images": _np(images), "bboxes": _np(bboxes), "labels": _np(labels), "keys": np.array(k0), "ck0": c0, "ck_init": c_init,
           "damp": np.array(damp), "pooler_mode": np.array(mode),
           "anchor_sizes": np.array(cfg["anchor_sizes"]), "nms_top_n": np.array([cfg["rpn_pre_nms_top_n"], cfg["rpn_post_nms_top_n"]])}
    # (1) head passes
    for i in (1, 2, 3):
        fm = model.train().forward({"x": images, "adv": None, "out_idx": i, "flag": "head"}, bboxes, labels).detach()
        rec[f"fm{i}_sub"] = _np(fm[:, ::8, ::2, ::2])
        rec[f"fm{i}_norm"] = np.array(float(fm.double().norm()))
        print(f"   fm{i}: shape {tuple(fm.shape)} rms {float(fm.double().pow(2).mean().sqrt()):.3f} max {float(fm.abs().max()):.2f}")
    fm3 = fm
    # (2) one training forward with recorded RPN outputs / proposals, and its backward
    seen = {}
    real_gp = model.rpn.generate_proposals

    def spy(anchor_bboxes, objectnesses, transformers, image_width, image_height):
        out = real_gp(anchor_bboxes, objectnesses, transformers, image_width, image_height)
        seen.update(obj=objectnesses.detach().clone(), tr=transformers.detach().clone(), proposals=out.detach().clone())
        return out
    model.rpn.generate_proposals = spy
    torch.manual_seed(100)
    ao, at, pc, pt = model.train().forward({"x": images, "adv": None, "out_idx": 0, "flag": "clean"}, bboxes, labels)
    model.rpn.generate_proposals = real_gp
    rec.update(rpn_obj=_np(seen["obj"]), rpn_tr=_np(seen["tr"]), proposals=_np(seen["proposals"]),
               fwd_losses=np.stack([_np(ao), _np(at), _np(pc), _np(pt)]))
    for p in model.parameters():
        p.grad = None
    (ao.mean() + at.mean() + pc.mean() + pt.mean()).backward()
    named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
    rec["param_names"] = np.array([n for n, _ in named])
    rec["grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in named])
    for n, p in named:
        if n in ("features.layer2.0.conv1.weight", "features.layer3.22.conv3.weight", "rpn._anchor_objectness.weight",
                 "rpn._anchor_objectness.bias", "detection._proposal_class.weight", "detection._proposal_transformer.bias"):
            rec["grad/" + n] = _np(p.grad)
    # (3) one-step feature PGD at the deepest point (train_aug_sat_muti_advt.py:91)
    y = {"bb": bboxes, "lb": labels}
    torch.manual_seed(101)
    adv3 = ref_det.PGD(fm3, images, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(1.0 / 255), idx=3)
    rec["adv3_sign"] = _np(torch.round((adv3.detach() - fm3) / np.float32(1.0 / 255))).astype(np.int8)
    # (4) one full iteration (loss_settings 1), SGD as train_aug_sat_muti_advt.py:46-47 / config/train_config.py
    optimizer = torch.optim.SGD(model.parameters(), lr=0.001, momentum=0.9, weight_decay=0.0005)
    torch.manual_seed(102)
    fwd = lambda d: model.train().forward(d, bboxes, labels)
    adv_image = ref_det.adv_input(x=images, y=y, model=model, steps=5, eps=(2.0 / 255), gamma=(0.3 / 255), randinit=True, clip=True)
    f1 = fwd({"x": images, "adv": None, "out_idx": 1, "flag": "head"}).detach()
    f2 = fwd({"x": images, "adv": None, "out_idx": 2, "flag": "head"}).detach()
    f3 = fwd({"x": images, "adv": None, "out_idx": 3, "flag": "head"}).detach()
    rr = fwd({"x": images, "adv": None, "out_idx": "roi_head", "flag": "clean"})
    clean_sd = rr["roi_output_dict"]["roi_feature_map"].detach()
    a1 = ref_det.PGD(f1, images, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=1)
    a2 = ref_det.PGD(f2, images, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=2)
    a3 = ref_det.PGD(f3, images, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(1.0 / 255), idx=3)
    pts = ref_det.get_sample_points(f3, a3, 5)
    pts[1] = ref_det.mix_feature(f3, pts[1])
    pts[2] = ref_det.mix_feature(f3, pts[2])
    arr = ref_det.rpn_roi_PGD(rpn_roi_output_dict=rr, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(0.2 / 255), only_roi_loss=False)
    adv_sd = ref_det.mix_feature(clean_sd, arr["roi_output_dict"]["roi_feature_map"].detach())
    arr["roi_output_dict"]["roi_feature_map"] = adv_sd
    dicts = [{"x": adv_image, "adv": None, "out_idx": 0, "flag": "clean"}, {"x": images, "adv": a1, "out_idx": 1, "flag": "tail"},
             {"x": images, "adv": a2, "out_idx": 2, "flag": "tail"}] + \
            [{"x": images, "adv": pts[j], "out_idx": 3, "flag": "tail"} for j in (1, 2, 3, 4)] + \
            [{"adv": arr, "out_idx": "roi_tail", "flag": "clean"}]
    L = [ref_det.compute_loss(*fwd(d)) for d in dicts]
    loss = 0.9 * (0.2333 * (L[0] + L[3] + L[4] + L[5] + L[6]) + 0.1 * L[7]) + 0.05 * (L[1] + L[2])
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    k1, c1 = _checksums(model)
    assert k1 == k0
    rec.update(step_loss=_np(loss), step_losses=np.array([float(v) for v in L], dtype=np.float32), ck1=c1,
               adv_image_sub=_np(adv_image.detach()[:, :, ::4, ::4]))
    fname = "det_frcnn_r101" + ("" if mode == "pooling" else "_" + mode)
    np.savez_compressed(os.path.join(OUT, fname + ".npz"), **rec)
    print(fname + ": forward losses", rec["fwd_losses"].tolist(), "iteration loss", float(loss), [round(float(v), 4) for v in L],
          "proposals", tuple(seen["proposals"].shape), "params with grad", len(named), "keys", len(k0))


def gen_detection_floor(mode="pooling"):
    """tests/golden/ref_det_floor.npz (round 6, VERDICT r5 weak 3 / missing 6) — the reference's OWN Detection iteration against itself.
    The full iteration of train_aug_sat_muti_advt.py:70-172 on det_frcnn_r101[_align].npz's model and inputs is run again in other
    arithmetics — float64, ATen-native fp32 (oneDNN off), channels-last fp32 — and on four draws of 1e-6 relative noise on the images
    (the rounding noise ~100 fp32 layers accumulate in ANY implementation), same seeds, same host-RNG history.  Stored per variant:
    the eight losses' relative distance from the fp32 baseline's, the iteration loss's, the fraction of adversarial-image pixels
    (every 4th, as in the golden) that differ, the post-SGD checksums' relative distance; `<mode>/spread_*` = the maximum over the
    variants.  tests/test_det_model_gpu.py::test_faster_rcnn_iteration_on_reference_golden bounds the product by
    max(2 x spread, 1e-4) instead of the hand-set 2 %: one flipped sign() of the five-step image PGD reorders proposals, the positional
    `randperm` sampling (Detection/model.py:274-277, rpn/region_proposal_network.py:87-90) then picks other boxes — in the
    reference against itself exactly as in any port."""
    import copy
    ref_model, RefBackbone, RefPooler = _import_reference_detection_model()
    ref_det = _load("ref_det_attack_algo", "Detection/attack_algo.py")
    cfg = dict(anchor_ratios=[(1, 2), (1, 1), (2, 1)], anchor_sizes=[64], rpn_pre_nms_top_n=200, rpn_post_nms_top_n=64,
               anchor_smooth_l1_loss_beta=1.0, proposal_smooth_l1_loss_beta=1.0)
    torch.manual_seed(7)
    model0 = ref_model.Model(RefBackbone(pretrained=False), 21, pooler_mode=RefPooler.Mode(mode), **cfg)
    for m in model0.modules():
        if hasattr(m, "bn3") and hasattr(m, "conv3"):
            m.bn3.weight.data.mul_(0.2)
    fname = "det_frcnn_r101" + ("" if mode == "pooling" else "_" + mode)
    gold = np.load(os.path.join(OUT, fname + ".npz"))
    images0, bboxes, labels = (torch.from_numpy(gold[k]) for k in ("images", "bboxes", "labels"))

    def iteration(kind):
        model, (images,) = _to_variant(kind, copy.deepcopy(model0), [images0])
        bb = bboxes.to(images.dtype)
        model.train()
        y = {"bb": bb, "lb": labels}
        optimizer = torch.optim.SGD(model.parameters(), lr=0.001, momentum=0.9, weight_decay=0.0005)
        torch.manual_seed(102)
        fwd = lambda d: model.train().forward(d, bb, labels)
        with _variant_ctx(kind):
            adv_image = ref_det.adv_input(x=images, y=y, model=model, steps=5, eps=(2.0 / 255), gamma=(0.3 / 255), randinit=True, clip=True)
            f1 = fwd({"x": images, "adv": None, "out_idx": 1, "flag": "head"}).detach()
            f2 = fwd({"x": images, "adv": None, "out_idx": 2, "flag": "head"}).detach()
            f3 = fwd({"x": images, "adv": None, "out_idx": 3, "flag": "head"}).detach()
            rr = fwd({"x": images, "adv": None, "out_idx": "roi_head", "flag": "clean"})
            clean_sd = rr["roi_output_dict"]["roi_feature_map"].detach()
            a1 = ref_det.PGD(f1, images, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=1)
            a2 = ref_det.PGD(f2, images, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=2)
            a3 = ref_det.PGD(f3, images, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(1.0 / 255), idx=3)
            pts = ref_det.get_sample_points(f3, a3, 5)
            pts[1] = ref_det.mix_feature(f3, pts[1])
            pts[2] = ref_det.mix_feature(f3, pts[2])
            arr = ref_det.rpn_roi_PGD(rpn_roi_output_dict=rr, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(0.2 / 255), only_roi_loss=False)
            adv_sd = ref_det.mix_feature(clean_sd, arr["roi_output_dict"]["roi_feature_map"].detach())
            arr["roi_output_dict"]["roi_feature_map"] = adv_sd
            dicts = [{"x": adv_image, "adv": None, "out_idx": 0, "flag": "clean"}, {"x": images, "adv": a1, "out_idx": 1, "flag": "tail"},
                     {"x": images, "adv": a2, "out_idx": 2, "flag": "tail"}] + \
                    [{"x": images, "adv": pts[j], "out_idx": 3, "flag": "tail"} for j in (1, 2, 3, 4)] + \
                    [{"adv": arr, "out_idx": "roi_tail", "flag": "clean"}]
            L = [ref_det.compute_loss(*fwd(d)) for d in dicts]
            loss = 0.9 * (0.2333 * (L[0] + L[3] + L[4] + L[5] + L[6]) + 0.1 * L[7]) + 0.05 * (L[1] + L[2])
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        ck = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in model.state_dict().values()])
        adv = adv_image.detach().contiguous(memory_format=torch.contiguous_format).double().numpy()[:, :, ::4, ::4]
        return float(loss), np.array([float(v) for v in L], dtype=np.float64), adv, ck

    b_loss, b_L, b_adv, b_ck = iteration("base")
    assert np.float32(b_loss) == gold["step_loss"] and np.array_equal(b_L.astype(np.float32), gold["step_losses"]), (b_loss, gold["step_loss"])
    rec = {}
    variants = ("f64", "nomkldnn", "cl", "in1", "in2", "in3", "in4")
    done = []
    for kind in variants:
        try:
            v_loss, v_L, v_adv, v_ck = iteration(kind)
        except Exception as e:      # noqa: BLE001 — a variant the reference's own code does not run in (dtype assumptions): recorded, skipped
            print(f"   {fname}/{kind}: not runnable ({type(e).__name__}: {str(e)[:120]})")
            continue
        done.append(kind)
        rec[f"{fname}/{kind}/losses_rel"] = np.abs(v_L - b_L) / np.abs(b_L)
        rec[f"{fname}/{kind}/loss_rel"] = np.array(abs(v_loss - b_loss) / abs(b_loss))
        rec[f"{fname}/{kind}/adv_pixels_off"] = np.array(float((np.abs(v_adv - b_adv) > 1e-6).mean()))
        rec[f"{fname}/{kind}/ck_abs_rel"] = np.array(float((np.abs(v_ck[:, 1] - b_ck[:, 1]) / (np.abs(b_ck[:, 1]) + 1e-4)).max()))
        print(f"   {fname}/{kind}: losses rel {rec[f'{fname}/{kind}/losses_rel'].max():.3e}  loss rel {float(rec[f'{fname}/{kind}/loss_rel']):.3e}  "
              f"adv pixels off {float(rec[f'{fname}/{kind}/adv_pixels_off']):.4f}  checksums rel {float(rec[f'{fname}/{kind}/ck_abs_rel']):.3e}")
    rec[f"{fname}/variants"] = np.array(done)
    rec[f"{fname}/spread_losses_rel"] = np.array(max(float(rec[f"{fname}/{k}/losses_rel"].max()) for k in done))
    rec[f"{fname}/spread_loss_rel"] = np.array(max(float(rec[f"{fname}/{k}/loss_rel"]) for k in done))
    rec[f"{fname}/spread_adv_pixels_off"] = np.array(max(float(rec[f"{fname}/{k}/adv_pixels_off"]) for k in done))
    rec[f"{fname}/spread_ck_abs_rel"] = np.array(max(float(rec[f"{fname}/{k}/ck_abs_rel"]) for k in done))
    print(f"   {fname}: spread over {done}: losses {float(rec[f'{fname}/spread_losses_rel']):.3e}  loss {float(rec[f'{fname}/spread_loss_rel']):.3e}  "
          f"adv pixels {float(rec[f'{fname}/spread_adv_pixels_off']):.4f}  checksums {float(rec[f'{fname}/spread_ck_abs_rel']):.3e}")
    path = os.path.join(OUT, "ref_det_floor.npz")
    old = dict(np.load(path)) if os.path.exists(path) else {}
    old.update(rec)
    np.savez_compressed(path, **old)


def gen_bf16_floor(ref_attack, orc):
    """tests/golden/ref_bf16_floor.npz (round 6) — what bf16 arithmetic does to the REFERENCE ITSELF.  The benched
    configuration computes its convolutions in bf16 (BASELINE configs[1]); its perturbation differs from the reference's fp32 one on
    14 % of the elements of the contractive golden, and the tests held that with a hand-set 0.20.  Here the reference's own code
    (Classification/attack_algo.PGD + the loop body of main_perturb.py:173-201 on the step_r18_k5_b32_damped network and inputs) is
    run under torch.autocast(bfloat16) — convolutions and the linear layer in bf16, BatchNorm statistics and the loss in fp32: the
    mixed precision of the product — and compared with its own fp32 run: fraction of perturbation elements that differ (after every
    step), the three losses, every parameter's gradient norm, the running statistics.  A second variant keeps EVERYTHING in bf16
    (model.bfloat16(), bf16 inputs): the upper end of what "a bf16 run of the reference" can mean.  The product's bf16 path is then
    bounded by max(2 x the autocast variant's distance, the fp32 bound)."""
    import copy
    crit = nn.CrossEntropyLoss()
    gold = np.load(os.path.join(OUT, "step_r18_k5_b32_damped.npz"))
    torch.manual_seed(3)
    model0, idx, ln = orc.resnet18_cifar(), 6, 15
    for m in model0.modules():
        if isinstance(m, orc.Block):
            m.bn2.weight.data.mul_(float(gold["damp"]))
    model0.train()
    x, y = torch.from_numpy(gold["x"]), torch.from_numpy(gold["y"])
    K, gamma, eps = 5, 0.5, 2.0

    def run(kind):
        model = copy.deepcopy(model0)
        xx = x
        ctx = contextlib.nullcontext()
        if kind == "autocast":
            ctx = torch.autocast("cpu", dtype=torch.bfloat16)
        elif kind == "allbf16":
            model, xx = model.bfloat16(), x.bfloat16()
        opt = torch.optim.SGD(model.parameters(), 0.1, momentum=0.9, weight_decay=5e-4)
        with ctx:
            fm = model(xx, end_point=idx, start_point=0).detach()
            snaps = _pgd_trace(ref_attack, model, crit, fm, y, K, gamma, eps, idx, ln, False)
            x_adv = snaps[-1]
            out_adv = model(x_adv, end_point=ln, start_point=idx)
            out_clean = model(xx, end_point=ln, start_point=0)
            la, lc = crit(out_adv.float(), y), crit(out_clean.float(), y)
            loss = (la + lc) / 2
            opt.zero_grad()
            loss.backward()
        dks = [np.rint(((s_.double() - fm.double()) / (gamma / 255)).numpy()) for s_ in snaps[1:]]
        gn = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters() if p.grad is not None])
        rs = {k: v.double().numpy() for k, v in model.state_dict().items() if "running_" in k}
        return dict(dks=dks, loss=float(loss), loss_adv=float(la), loss_clean=float(lc), gn=gn, rs=rs, fm=fm.double().numpy())

    import contextlib
    base = run("base")
    assert np.array_equal(base["dks"][-1].astype(np.int8), gold["dk"]) and np.float32(base["loss"]) == gold["loss"], "baseline is not the golden"
    rec = {}
    for kind in ("autocast", "allbf16"):
        v = run(kind)
        pre = f"step_r18_k5_b32_damped/{kind}/"
        rec[pre + "flips_per_step"] = np.array([float((a != b).mean()) for a, b in zip(v["dks"], base["dks"])])
        for k in ("loss", "loss_adv", "loss_clean"):
            rec[pre + k + "_rel"] = np.array(abs(v[k] - base[k]) / max(1.0, abs(base[k])))
        rec[pre + "grad_norm_rel_max"] = np.array(float((np.abs(v["gn"] - base["gn"]) / (base["gn"] + 1e-6 * base["gn"].max())).max()))
        rec[pre + "running_stats_max"] = np.array(max(float(np.max(np.abs(v["rs"][k] - base["rs"][k]) / (np.abs(base["rs"][k]) + 1.0))) for k in base["rs"]))
        rec[pre + "feature_map_l2_rel"] = np.array(float(np.linalg.norm((v["fm"] - base["fm"]).ravel()) / np.linalg.norm(base["fm"].ravel())))
        print(f"   {pre}: flips per step {np.round(rec[pre + 'flips_per_step'], 4).tolist()}  losses rel "
              f"{float(rec[pre + 'loss_rel']):.2e} / {float(rec[pre + 'loss_adv_rel']):.2e} / {float(rec[pre + 'loss_clean_rel']):.2e}  grad norms "
              f"{float(rec[pre + 'grad_norm_rel_max']):.3f}  running stats {float(rec[pre + 'running_stats_max']):.2e}  feature map "
              f"{float(rec[pre + 'feature_map_l2_rel']):.2e}")
    np.savez_compressed(os.path.join(OUT, "ref_bf16_floor.npz"), **rec)


def main():
    assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
    _shims()
    os.makedirs(OUT, exist_ok=True)
    ref_attack = _load("ref_cls_attack_algo", "Classification/attack_algo.py")
    ref_resnet = _load("ref_cls_resnet_s", "Classification/resnet_s.py")
    ref_seg = _load("ref_seg_attack_algo", "Segmentation/attack_algo.py")
    from oracle import afan_oracle as orc  # only for the build-defined ResNet-18 module (no reference class exists)

    crit = nn.CrossEntropyLoss()

    def build(arch):
        if arch == "resnet20s":
            return ref_resnet.ResNet(ref_resnet.BasicBlock, [3, 3, 3]), 7, 16
        if arch == "resnet56s":
            return ref_resnet.resnet56(), 13, 34
        if arch == "resnet18":
            return orc.resnet18_cifar(), 6, 15
        raise KeyError(arch)

    # ---- single-step cases: reference PGD + joint step -------------------------------------------
    cases = [
        # name, arch, batch, K, gamma, eps, randinit, clip, store_weights
        ("step_r20s_k1", "resnet20s", 4, 1, 0.5, 2.0, False, False, True),
        ("step_r20s_k5", "resnet20s", 4, 5, 0.5, 2.0, False, False, False),
        ("step_r20s_k5_clip", "resnet20s", 4, 5, 1.5, 2.0, False, True, False),
        ("step_r20s_k3_clip_rand", "resnet20s", 4, 3, 1.5, 2.0, True, True, False),
        ("step_r56s_k5", "resnet56s", 2, 5, 0.5, 2.0, False, False, False),
        ("step_r18_k5", "resnet18", 2, 5, 0.5, 2.0, False, False, False),
        # batch 16: BatchNorm statistics over >= 16k samples — sign() flips no longer compound through batch-2 moments, so
        # the perturbation is held to <= 1e-2 of the elements.  Stored compactly: the perturbation as int8 multiples of
        # gamma (exact: x_adv - x is a sum of K terms +-gamma), the feature map sub-sampled.
        ("step_r56s_k5_b16", "resnet56s", 16, 5, 0.5, 2.0, False, False, False),
        ("step_r18_k5_b16", "resnet18", 16, 5, 0.5, 2.0, False, False, False),
    ]
    for name, arch, bs, K, gamma, eps, randinit, clip, store_w in cases:
        torch.manual_seed(3)
        model, idx, ln = build(arch)
        model.train()
        opt = torch.optim.SGD(model.parameters(), 0.1, momentum=0.9, weight_decay=5e-4)
        x = torch.rand(bs, 3, 32, 32)
        y = torch.randint(0, 10, (bs,))
        rec = {"x": _np(x), "y": _np(y), "meta": np.array([K, idx, ln, int(randinit), int(clip)]),
               "gamma_eps": np.array([gamma, eps], dtype=np.float64)}
        if store_w:
            rec.update(_state_np(model, "sd0/"))
        k0, c0 = _checksums(model)
        rec["ck0"] = c0
        if randinit:
            # the noise PGD draws is the next torch.rand(feature shape) of the CPU generator (attack_algo.py:44):
            # peek it, then rewind the generator so the reference draws the same numbers.
            model.eval()  # eval-mode probe: BN buffers untouched
            with torch.no_grad():
                fm_shape = model(x, end_point=idx, start_point=0).shape
            model.train()
            gstate = torch.get_rng_state()
            rec["u"] = _np(torch.rand(fm_shape))
            torch.set_rng_state(gstate)
        r = _ref_step(ref_attack, model, opt, crit, x, y, K, gamma, eps, idx, ln, randinit, clip)
        if name.endswith("_b16"):
            for k in ("l2", "linf", "loss", "loss_adv", "loss_clean", "out_clean"):
                rec[k] = _np(r[k])
            dk = torch.round((r["x_adv"] - r["feature_map"]) / np.float32(gamma / 255))
            assert float(dk.abs().max()) <= K
            rec["dk"] = _np(dk).astype(np.int8)
            rec["feature_map_sub"] = _np(r["feature_map"][:, ::4, ::2, ::2])
        else:
            for k in ("feature_map", "x_adv", "l2", "linf", "loss", "loss_adv", "loss_clean", "out_clean"):
                rec[k] = _np(r[k])
        k1, c1 = _checksums(model)
        assert k0 == k1
        rec["keys"] = np.array(k1)
        rec["ck1"] = c1
        sd = model.state_dict()
        for k in (f"sequential_model.2.running_mean", f"sequential_model.2.running_var",
                  f"sequential_model.2.num_batches_tracked", f"sequential_model.{idx}.bn1.running_mean",
                  f"sequential_model.{idx}.bn1.running_var", f"sequential_model.{idx}.bn1.num_batches_tracked",
                  "sequential_model.1.weight", f"sequential_model.{ln - 1}.weight", f"sequential_model.{ln - 1}.bias"):
            rec["sd1/" + k] = _np(sd[k])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, "loss", float(r["loss"]), "l2", r["l2"].tolist())

    gen_damped_r18(ref_attack, orc)

    # ---- per-step PGD trace (kernel-level golden: x_adv before/after every step, with its gradient) ----
    gen_pgd_traces(ref_attack, build, crit)

    # ---- 3-iteration trajectory with warm-up lr (main_perturb.py:167-168,288-293) ----------------------
    torch.manual_seed(3)
    model, idx, ln = build("resnet20s")
    model.train()
    opt = torch.optim.SGD(model.parameters(), 0.1, momentum=0.9, weight_decay=5e-4)
    xs = torch.rand(3, 8, 3, 32, 32)
    ys = torch.randint(0, 10, (3, 8))
    wp = 5
    losses, lrs = [], []
    for i in range(3):
        lr = min(i * 0.1 / (wp - 1), 0.1)  # main_perturb.py:288-293
        for p in opt.param_groups:
            p["lr"] = lr
        lrs.append(lr)
        r = _ref_step(ref_attack, model, opt, crit, xs[i], ys[i], 2, 0.5, 2.0, idx, ln, False, False)
        losses.append(float(r["loss"]))
    keys, ck = _checksums(model)
    np.savez_compressed(os.path.join(OUT, "traj_r20s.npz"), xs=_np(xs), ys=_np(ys), losses=np.array(losses),
                        lrs=np.array(lrs), ck=ck, keys=np.array(keys), wp=np.array(wp),
                        fc_w=_np(model.state_dict()["sequential_model.15.weight"]))
    print("traj", losses)

    # ---- learnable multi-layer A-FAN: the reference's own train() (main_learnable.py:175-277), one batch ------------
    # The module's top-level imports of torchvision / matplotlib / PIL / its CIFAR loader are absent here and unused by
    # train(): empty stand-in modules let the file import; every line of arithmetic that runs is the reference's.
    import argparse
    for name in ("matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image", "torchvision", "torchvision.models",
                 "torchvision.transforms", "torchvision.datasets", "dataset"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["dataset"].cifar10_dataloaders = None
    sys.modules["resnet_s"], sys.modules["attack_algo"] = ref_resnet, ref_attack
    ref_learn = _load("ref_cls_main_learnable", "Classification/main_learnable.py")
    for name, bs, K, gamma, clip in (("learn_r56s_k1", 4, 1, 0.5, False), ("learn_r56s_k2_clip", 2, 2, 1.5, True)):
        torch.manual_seed(3)
        model = ref_resnet.resnet56(init_weight_eta=1 / 9)
        model.train()
        opt = torch.optim.SGD(model.sequential_model.parameters(), 0.1, momentum=0.9, weight_decay=5e-4)
        opt_w = torch.optim.SGD([{"params": model.w, "lr": 0.01, "weight_decay": 0}], 0.01, momentum=0.9, weight_decay=0)
        x = torch.rand(bs, 3, 32, 32)
        y = torch.randint(0, 10, (bs,))
        ref_learn.args = argparse.Namespace(steps=K, gamma=gamma, eps=2.0, randinit=False, clip=clip, print_freq=1000,
                                            l1_coef=1.0, lr=0.1)
        k0, c0 = _checksums(model)
        acc, loss, l2m, linfm = ref_learn.train([(x, y)], model, crit, opt, 1, opt_w)     # epoch 1: no warm-up branch
        k1, c1 = _checksums(model)
        sd = model.state_dict()
        rec = {"x": _np(x), "y": _np(y), "meta": np.array([K, int(clip)]), "gamma_eps": np.array([gamma, 2.0]),
               "idx_list": np.array(ref_learn.perturb_idx_list), "layer_number": np.array(ref_learn.layer_number),
               "ck0": c0, "ck1": c1, "keys": np.array(k1), "loss": np.array(loss), "acc": np.array(acc),
               "l2_mean": np.asarray(l2m), "linf_mean": np.asarray(linfm), "w1": _np(sd["w"]),
               "sd1/fc_w": _np(sd["sequential_model.33.weight"]), "sd1/conv1_w": _np(sd["sequential_model.1.weight"]),
               "sd1/bn1_rm": _np(sd["sequential_model.2.running_mean"]),
               "sd1/bn1_nbt": _np(sd["sequential_model.2.num_batches_tracked"])}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, "loss", loss, "w", sd["w"].tolist())

    # ---- Segmentation operators + iteration (N1, first slice): the reference's own PGD / decoder_PGD / adv_input /
    # get_sample_points / mix_feature (Segmentation/attack_algo.py) and the loop body of main_aug_final.py:158-232 driven line
    # by line around them, on the protocol-faithful stand-in network oracle.TinySegNet (the reference's DeepLab needs
    # torchvision; its restatement is the next slice) --------------------------------------------------------------------
    for name, steps, gamma_se, gamma_sd, sd_idx, clip in (("seg_step_aspp_k1", 1, 0.5, 0.5, "aspp", False),
                                                          ("seg_step_concat_k2", 2, 1.0, 0.5, "concat", False)):
        torch.manual_seed(5)
        net = orc.TinySegNet()
        net.train()
        opt = torch.optim.SGD(net.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
        seg_crit = nn.CrossEntropyLoss(ignore_index=255, reduction="mean")
        images = torch.rand(2, 3, 33, 33)
        labels = torch.randint(0, 5, (2, 33, 33))
        labels[torch.rand(2, 33, 33) < 0.05] = 255
        k0, c0 = _checksums(net)
        se_idx, eps = 3, 2.0
        inputs_all_se = {"x": images, "adv": None, "out_idx": se_idx, "flag": "head"}
        inputs_all_sd = {"x": images, "adv": None, "out_idx": sd_idx + "_head", "flag": "clean"}
        opt.zero_grad()
        output_dict_se = net(inputs_all_se)
        decoder_feature_map_dict = net(inputs_all_sd)
        feature_map_sd = decoder_feature_map_dict["adv"].detach()
        low_level_feat = output_dict_se["low_level"]
        feature_map_se = output_dict_se["out"].detach()
        feature_adv_se = ref_seg.PGD(x=feature_map_se, image_batch=images, low_level_feat=low_level_feat, criterion=seg_crit,
                                     y=labels, model=net, steps=steps, eps=(eps / 255), gamma=(gamma_se / 255), idx=se_idx,
                                     randinit=False, clip=clip)
        feature_adv_sd_dict = ref_seg.decoder_PGD(input_dict=decoder_feature_map_dict, image_batch=images, criterion=seg_crit,
                                                  y=labels, model=net, steps=steps, eps=(eps / 255), gamma=(gamma_sd / 255),
                                                  idx=sd_idx, randinit=False, clip=False)
        adv_feature_map_sd = feature_adv_sd_dict["adv"].detach()
        adv_feature_map_sd = ref_seg.mix_feature(feature_map_sd, adv_feature_map_sd)          # opts.mix_sd
        feature_adv_sd_dict["adv"] = adv_feature_map_sd
        adv_list_se = ref_seg.get_sample_points(feature_map_se, feature_adv_se, 3)
        adv_list_se[1] = ref_seg.mix_feature(feature_map_se, adv_list_se[1])                  # mix_layer "11"
        adv_list_se[2] = ref_seg.mix_feature(feature_map_se, adv_list_se[2])
        clean_input_dict = {"x": images, "adv": None, "out_idx": 0, "flag": "clean"}
        adv_input_se_dict1 = {"x": images, "adv": adv_list_se[1], "out_idx": se_idx, "flag": "tail", "low_level_feat": low_level_feat}
        adv_input_se_dict2 = {"x": images, "adv": adv_list_se[2], "out_idx": se_idx, "flag": "tail", "low_level_feat": low_level_feat}
        adv_input_sd_dict = {"x": images, "adv": feature_adv_sd_dict, "out_idx": sd_idx + "_tail", "flag": "clean"}
        output0, output1 = net(clean_input_dict), net(adv_input_se_dict1)
        output2, output3 = net(adv_input_se_dict2), net(adv_input_sd_dict)
        loss0, loss1 = seg_crit(output0, labels), seg_crit(output1, labels)
        loss2, loss3 = seg_crit(output2, labels), seg_crit(output3, labels)
        loss = 0.7 * loss0 + 0.1 * loss1 + 0.1 * loss2 + 0.1 * loss3
        loss.backward()
        opt.step()
        k1, c1 = _checksums(net)
        # image-space PGD of the same file (adv_input), on the updated network in eval mode (no BN side effects)
        net.eval()
        x_img = ref_seg.adv_input(x=images, criterion=seg_crit, y=labels, model=net, steps=2, eps=(2.0 / 255),
                                  gamma=(1.0 / 255), randinit=False, clip=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), images=_np(images), labels=_np(labels),
                            meta=np.array([steps, se_idx, int(clip)]), gammas=np.array([gamma_se, gamma_sd, eps]),
                            sd_idx=np.array(sd_idx), ck0=c0, ck1=c1, keys=np.array(k1), loss=_np(loss),
                            losses=np.array([float(loss0), float(loss1), float(loss2), float(loss3)], dtype=np.float32),
                            adv_se=_np(feature_adv_se), adv_sd=_np(adv_feature_map_sd), fm_se=_np(feature_map_se),
                            out_clean=_np(output0), x_img=_np(x_img))
        print(name, "loss", float(loss))

    # ---- DeepLabv3+ (ResNet-101, output stride 16): the reference's OWN network (Segmentation/network/, imported with one
    # more shim: torchvision.models.utils.load_state_dict_from_url, unused with pretrained_backbone=False) driven through
    # the loop body of main_aug_final.py:158-232 with the reference's own attack_algo functions.  Dropout(0.1) of the ASPP
    # projection (_deeplab.py:185) is switched off (p = 0: a device-RNG mask cannot be matched); nothing
    # else is touched.  Initial weights are a function of the seed (construction order), so only checksums travel. -----
    tv = sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    tvm = sys.modules.setdefault("torchvision.models", types.ModuleType("torchvision.models"))
    tvu = types.ModuleType("torchvision.models.utils")
    tvu.load_state_dict_from_url = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no network"))
    sys.modules["torchvision.models.utils"] = tvu
    tv.models, tvm.utils = tvm, tvu
    sys.path.insert(0, os.path.join(REF, "Segmentation"))
    for clash in ("network", "utils"):
        sys.modules.pop(clash, None)
    import network as ref_network            # noqa: E402  (Segmentation/network)
    sys.path.pop(0)
    # `damp`: every residual branch's last BatchNorm weight (bn3) is scaled by it before the step.  A freshly initialised
    # 101-layer BatchNorm network amplifies a 0.3 % input perturbation to 58 % at layer3's output (measured, fp32 CPU): an
    # end-to-end comparison of a bf16 run is only meaningful on a contractive network — weights are data, so the damped
    # cases change no code path (the reference's own zero_init_residual option, backbone/resnet.py:170-175, is damp = 0).
    # The damped case also uses 4 images of different brightness / contrast: the ASPP pooling branch's BatchNorm
    # (_deeplab.py:152-157) normalises ONE value per image and channel; with 2 look-alike random images 12 % of its channels
    # have a batch variance within an order of magnitude of eps, where d(out)/d(in) swings between 0.05 and 158 — their
    # gradient share is 86 % and no two arithmetic implementations agree on it (measured).  4 distinct images: none.
    for name, steps, gamma_se, gamma_sd, sd_idx, mix_layer, mix_sd, side, damp, bs in (
            ("seg_dl101_aspp_k1", 1, 0.5, 0.5, "aspp", "11", False, 129, 1.0, 2),
            ("seg_dl101_concat_k3", 3, 0.5, 1.5, "concat", "01", True, 129, 1.0, 2),
            ("seg_dl101_aspp_k3_damped", 3, 0.5, 0.5, "aspp", "11", True, 129, 0.1, 4)):
        torch.manual_seed(3)
        net = ref_network.deeplabv3plus_resnet101(num_classes=21, output_stride=16, pretrained_backbone=False)
        net.classifier.aspp.project[3].p = 0.0
        if damp != 1.0:
            for m in net.backbone.modules():
                if isinstance(m, ref_network.backbone.resnet.Bottleneck):
                    m.bn3.weight.data.mul_(damp)
        for m in net.backbone.modules():                       # utils.set_bn_momentum(model.backbone, 0.01), main_aug_final.py:77
            if isinstance(m, nn.BatchNorm2d):
                m.momentum = 0.01
        net.train()
        lr = 0.01
        opt = torch.optim.SGD(params=[{"params": net.backbone.parameters(), "lr": 0.1 * lr},
                                      {"params": net.classifier.parameters(), "lr": lr}], lr=lr, momentum=0.9, weight_decay=1e-4)
        seg_crit = nn.CrossEntropyLoss(ignore_index=255, reduction="mean")
        images = torch.rand(bs, 3, side, side)
        if bs == 4:
            for i, (sc, of) in enumerate([(1.0, 0.0), (0.3, 0.6), (0.6, 0.0), (0.5, 0.4)]):
                images[i] = images[i] * sc + of
        labels = torch.randint(0, 21, (bs, side, side))
        labels[torch.rand(bs, side, side) < 0.05] = 255
        k0, c0 = _checksums(net)
        se_idx, eps = 3, 2.0
        f0, f1 = int(mix_layer[0]), int(mix_layer[1])
        inputs_all_se = {"x": images, "adv": None, "out_idx": se_idx, "flag": "head"}
        inputs_all_sd = {"x": images, "adv": None, "out_idx": sd_idx + "_head", "flag": "clean"}
        opt.zero_grad()
        output_dict_se = net(inputs_all_se)
        decoder_feature_map_dict = net(inputs_all_sd)
        feature_map_sd = decoder_feature_map_dict["adv"].detach()
        low_level_feat = output_dict_se["low_level"]
        feature_map_se = output_dict_se["out"].detach()
        feature_adv_se = ref_seg.PGD(x=feature_map_se, image_batch=images, low_level_feat=low_level_feat, criterion=seg_crit,
                                     y=labels, model=net, steps=steps, eps=(eps / 255), gamma=(gamma_se / 255), idx=se_idx,
                                     randinit=False, clip=False)
        feature_adv_sd_dict = ref_seg.decoder_PGD(input_dict=decoder_feature_map_dict, image_batch=images, criterion=seg_crit,
                                                  y=labels, model=net, steps=steps, eps=(eps / 255), gamma=(gamma_sd / 255),
                                                  idx=sd_idx, randinit=False, clip=False)
        adv_feature_map_sd = feature_adv_sd_dict["adv"].detach()
        if mix_sd:
            adv_feature_map_sd = ref_seg.mix_feature(feature_map_sd, adv_feature_map_sd)
        feature_adv_sd_dict["adv"] = adv_feature_map_sd
        adv_list_se = ref_seg.get_sample_points(feature_map_se, feature_adv_se, 3)
        if f0:
            adv_list_se[1] = ref_seg.mix_feature(feature_map_se, adv_list_se[1])
        if f1:
            adv_list_se[2] = ref_seg.mix_feature(feature_map_se, adv_list_se[2])
        clean_input_dict = {"x": images, "adv": None, "out_idx": 0, "flag": "clean"}
        adv_input_se_dict1 = {"x": images, "adv": adv_list_se[1], "out_idx": se_idx, "flag": "tail", "low_level_feat": low_level_feat}
        adv_input_se_dict2 = {"x": images, "adv": adv_list_se[2], "out_idx": se_idx, "flag": "tail", "low_level_feat": low_level_feat}
        adv_input_sd_dict = {"x": images, "adv": feature_adv_sd_dict, "out_idx": sd_idx + "_tail", "flag": "clean"}
        output0, output1 = net(clean_input_dict), net(adv_input_se_dict1)
        output2, output3 = net(adv_input_se_dict2), net(adv_input_sd_dict)
        loss0, loss1 = seg_crit(output0, labels), seg_crit(output1, labels)
        loss2, loss3 = seg_crit(output2, labels), seg_crit(output3, labels)
        loss = 0.7 * loss0 + 0.1 * loss1 + 0.1 * loss2 + 0.1 * loss3
        loss.backward()
        grad_norms = np.array([float(p.grad.double().norm()) for p in net.parameters()])
        grad_keep = {n: _np(p.grad) for n, p in net.named_parameters()
                     if n in ("backbone.conv1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer3.10.bn2.weight",
                              "backbone.layer4.2.bn3.bias", "classifier.project.0.weight", "classifier.aspp.convs.1.1.weight",
                              "classifier.classifier.3.weight", "classifier.classifier.3.bias")}
        opt.step()
        k1, c1 = _checksums(net)
        sd = net.state_dict()
        rec = dict(images=_np(images), labels=_np(labels), meta=np.array([steps, se_idx, int(mix_sd)]),
                   damp=np.array(damp), grad_norms=grad_norms, param_names=np.array([n for n, _ in net.named_parameters()]),
                   **{"grad/" + k: v for k, v in grad_keep.items()},
                   gammas=np.array([gamma_se, gamma_sd, eps]), sd_idx=np.array(sd_idx), mix_layer=np.array(mix_layer),
                   seed=np.array(3), lr=np.array(lr), ck0=c0, ck1=c1, keys=np.array(k1), loss=_np(loss),
                   losses=np.array([float(loss0), float(loss1), float(loss2), float(loss3)], dtype=np.float32),
                   adv_se=_np(feature_adv_se), fm_se=_np(feature_map_se), fm_sd_sub=_np(feature_map_sd[:, ::4]),
                   adv_sd_sub=_np(adv_feature_map_sd[:, ::4]),
                   adv_sd_sum=np.array([float(adv_feature_map_sd.double().sum()), float(adv_feature_map_sd.double().abs().sum())]),
                   out_clean_sub=_np(output0[:, :, ::4, ::4]),
                   out_clean_sum=np.array([float(output0.double().sum()), float(output0.double().abs().sum())]))
        for k in ("classifier.classifier.3.weight", "classifier.classifier.3.bias", "backbone.bn1.running_mean",
                  "backbone.bn1.running_var", "backbone.bn1.num_batches_tracked", "backbone.layer4.0.bn1.running_mean",
                  "backbone.layer4.0.bn1.num_batches_tracked", "classifier.aspp.convs.4.2.running_var",
                  "classifier.classifier.1.num_batches_tracked", "classifier.project.1.running_mean",
                  "classifier.project.1.num_batches_tracked", "backbone.conv1.weight"):
            rec["sd1/" + k] = _np(sd[k])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, "loss", float(loss), [float(loss0), float(loss1), float(loss2), float(loss3)])
    for clash in ("network", "utils"):
        sys.modules.pop(clash, None)

    gen_detection(orc)
    gen_roi_align(orc)

    # ---- Segmentation operators: mix_feature, get_sample_points (reference functions, direct) ----------
    torch.manual_seed(7)
    rec = {}
    for tag, shape in (("a", (2, 19, 7, 9)), ("b", (1, 304, 5, 5)), ("c", (3, 64, 1, 33))):
        clean = torch.randn(shape) * 1.7 + 0.3
        adv = clean + torch.randn(shape) * 0.2 + 0.05
        rec[f"mix_{tag}_clean"] = _np(clean)
        rec[f"mix_{tag}_adv"] = _np(adv)
        rec[f"mix_{tag}_out"] = _np(ref_seg.mix_feature(clean, adv))
        for n in (3, 5):
            pts = ref_seg.get_sample_points(clean, adv, n)
            rec[f"lerp_{tag}_{n}"] = np.stack([_np(p) for p in pts])
    np.savez_compressed(os.path.join(OUT, "seg_ops.npz"), **rec)
    print("seg_ops ok")

    # ---- tensor_clamp / linfball_proj edge cases (reference functions, direct) ------------------------
    t = torch.tensor([0.0, 1.0, -1.0, float("nan"), 5.0, -5.0, 0.5, 0.25], dtype=torch.float32)
    c = torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0, -1.0, float("nan"), 0.25], dtype=torch.float32)
    out = ref_attack.linfball_proj(c, 0.5, t.clone(), in_place=True)
    np.savez_compressed(os.path.join(OUT, "clamp_edges.npz"), t=_np(t), c=_np(c), radius=np.array(0.5, dtype=np.float32),
                        out=_np(out))
    print("clamp ok")


if __name__ == "__main__":
    if sys.argv[1:] == ["trace18"]:       # only the ResNet-18 PGD trace
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        _shims()
        os.makedirs(OUT, exist_ok=True)
        from oracle import afan_oracle as _orc
        _ra = _load("ref_cls_attack_algo", "Classification/attack_algo.py")
        gen_pgd_traces(_ra, lambda a: (_orc.resnet18_cifar(), 6, 15), nn.CrossEntropyLoss(), only=("pgd_trace_r18_k5",))
    elif sys.argv[1:] == ["damped18"]:    # only the contractive ResNet-18 step
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        _shims()
        os.makedirs(OUT, exist_ok=True)
        from oracle import afan_oracle as _orc
        gen_damped_r18(_load("ref_cls_attack_algo", "Classification/attack_algo.py"), _orc)
    elif sys.argv[1:] == ["det"]:         # only the Detection fixtures (the full run regenerates every file bit-identically)
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        _shims()
        os.makedirs(OUT, exist_ok=True)
        from oracle import afan_oracle as _orc
        gen_detection(_orc)
    elif sys.argv[1:] == ["floor"]:       # only ref_noise_floor.npz (reads the step goldens it refers to)
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        _shims()
        from oracle import afan_oracle as _orc
        _ra = _load("ref_cls_attack_algo", "Classification/attack_algo.py")
        _rr = _load("ref_cls_resnet_s", "Classification/resnet_s.py")
        _rs = _load("ref_seg_attack_algo", "Segmentation/attack_algo.py")
        _tv = sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
        _tvm = sys.modules.setdefault("torchvision.models", types.ModuleType("torchvision.models"))
        _tvu = types.ModuleType("torchvision.models.utils")
        _tvu.load_state_dict_from_url = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no network"))
        sys.modules["torchvision.models.utils"] = _tvu
        _tv.models, _tvm.utils = _tvm, _tvu
        sys.path.insert(0, os.path.join(REF, "Segmentation"))
        import network as _ref_network        # noqa: E402  (Segmentation/network)
        sys.path.pop(0)

        def _build(arch):
            if arch == "resnet20s":
                return _rr.ResNet(_rr.BasicBlock, [3, 3, 3]), 7, 16
            if arch == "resnet56s":
                return _rr.resnet56(), 13, 34
            return _orc.resnet18_cifar(), 6, 15
        gen_noise_floor(_ra, _build, _rs, _ref_network)
    elif sys.argv[1:] == ["lossfloor"]:   # only ref_loss_floor.npz + traj_r18_damped.npz (reads the step goldens it refers to)
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        _shims()
        from oracle import afan_oracle as _orc
        _ra = _load("ref_cls_attack_algo", "Classification/attack_algo.py")
        _rr = _load("ref_cls_resnet_s", "Classification/resnet_s.py")

        def _build(arch):
            if arch == "resnet20s":
                return _rr.ResNet(_rr.BasicBlock, [3, 3, 3]), 7, 16
            if arch == "resnet56s":
                return _rr.resnet56(), 13, 34
            return _orc.resnet18_cifar(), 6, 15
        gen_loss_floor(_ra, _build, _orc)
    elif sys.argv[1:] == ["roialign"]:    # only the ROIAlign vectors (the reference's own CPU kernel, compiled by oracle/Makefile)
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        os.makedirs(OUT, exist_ok=True)
        from oracle import afan_oracle as _orc
        gen_roi_align(_orc)
    elif sys.argv[1:] == ["frcnn"]:       # only the Faster-RCNN fixture (own process: Detection/ goes on sys.path)
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        _shims()
        os.makedirs(OUT, exist_ok=True)
        gen_detection_model("pooling")
        gen_detection_model("align")
    elif sys.argv[1:] == ["bf16floor"]:   # only ref_bf16_floor.npz (reads step_r18_k5_b32_damped.npz)
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        _shims()
        from oracle import afan_oracle as _orc
        gen_bf16_floor(_load("ref_cls_attack_algo", "Classification/attack_algo.py"), _orc)
    elif sys.argv[1:2] == ["detfloor"]:   # only ref_det_floor.npz (reads det_frcnn_r101[_align].npz); optional mode argument
        assert os.path.isdir(REF), f"{REF} not found: this script only runs in the build container"
        _shims()
        for mode_ in (sys.argv[2:3] or ["pooling", "align"]):
            gen_detection_floor(mode_)
    else:
        main()
