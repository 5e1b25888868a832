"""Segmentation feature-space operators (first slice) — names, signatures and error behaviour of the
reference's `Segmentation/attack_algo.py`:

    PGD(x, image_batch, low_level_feat, criterion, y, model, steps, eps, gamma, idx, randinit, clip)        :40-59
    decoder_PGD(input_dict, image_batch, criterion, y, model, steps, eps, gamma, idx, randinit, clip)       :61-84
    adv_input(x, criterion, y, model, steps, eps, gamma, randinit, clip)                                    :86-105
    get_sample_points / mix_feature / tensor_clamp / linfball_proj                  (shared with attack_algo.py)

`model` is anything that follows the reference's dict-dispatch protocol (`Segmentation/network/utils.py:14-47`):
`model({'x', 'adv', 'out_idx', 'flag', 'low_level_feat'}) -> logits | feature dict`.  The sign-step / projection / noise /
mix / lerp arithmetic runs in libafan_hip.so; the model's own layers are whatever the caller built (the DeepLabv3+
network with the library's kernels is the next slice)."""
import contextlib

import torch

from . import ops
from .attack_algo import get_sample_points, linfball_proj, mix_feature, sample_points_mixed, tensor_clamp  # noqa: F401 (same functions)
from .resnet_s import dgrad_only


def _start(x, eps, randinit):
    if x.device.type != "cuda":
        raise ops.AfanLibraryError("x must live on the MI355X (no CPU path in this build)")
    x = x.detach().float()
    x = x if (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))) else x.contiguous()
    x_adv = x.clone()     # keeps the feature map's own dense layout (channels-last on the bf16 path: no transposes per step)
    if randinit:   # noise from the CPU default generator, like the reference (:44)
        u = torch.rand(x_adv.shape).to(x.device, non_blocking=True)
        if u.stride() != x_adv.stride():
            u = u.contiguous(memory_format=torch.channels_last)
        ops.axpy_noise_(x_adv, u, eps)
    return x, x_adv


def _ascend(x_adv, logits_of, criterion, y, gamma, x, eps, clip):
    xin = x_adv.detach().requires_grad_(True)
    with dgrad_only():          # only_inputs=True (:52): the library's layers must not add into the parameters' .grad here
        loss = criterion(logits_of(xin), y)
        root = ops.one(loss.device) if (getattr(criterion, "fused", False) and loss.dim() == 0) else None
        grad = torch.autograd.grad(loss, xin, grad_outputs=root, only_inputs=True)[0]
    if grad.stride() != x_adv.stride():
        grad = grad.contiguous(memory_format=torch.channels_last) if x_adv.is_contiguous(memory_format=torch.channels_last) and not x_adv.is_contiguous() else grad.contiguous()
    ops.pgd_step_(x_adv, grad, gamma, x, eps, clip)          # one launch: sign step (+ projection)


def _low_res(criterion):
    """True when `criterion` is seg_criterion's fused callable: the model may then hand back its low-resolution logits
    (deeplab.LowResLogits) and the criterion resizes inside its own kernel (afan_ce2d_upsampled)."""
    return bool(getattr(criterion, "low_res", False))


def _f32_logits(o):
    o = getattr(o, "logits", o)
    return o.is_cuda and o.dtype == torch.float32


def _halves(o):
    """The two passes' logits out of one concatenated pass (deeplab.LowResLogits or a plain tensor)."""
    from .deeplab import LowResLogits
    if isinstance(o, LowResLogits):
        n = o.logits.shape[0] // 2
        return LowResLogits(o.logits[:n], o.size), LowResLogits(o.logits[n:], o.size)
    n = o.shape[0] // 2
    return o[:n], o[n:]


def _first_step(x_adv, grad0, gamma, x, eps, clip):
    """The first ascent step from a gradient the caller already has (a positive multiple of d(loss)/d(x) at x itself)."""
    g = grad0.detach()
    if g.stride() != x_adv.stride():
        g = g.contiguous(memory_format=torch.channels_last) if (x_adv.dim() == 4 and not x_adv.is_contiguous()) else g.contiguous()
    ops.pgd_step_(x_adv, g, gamma, x, eps if eps is not None else 0.0, clip)


def PGD(x, image_batch, low_level_feat, criterion, y=None, model=None, steps=3, eps=None, gamma=None, idx=1,
        randinit=False, clip=False, grad0=None):
    """SE-branch feature PGD (:40-59).  Returns a new fp32 leaf with requires_grad=True; `x` is not modified.
    grad0 (an addition; not with randinit): the first step's gradient, computed by the caller's clean pass."""
    if grad0 is not None and (randinit or steps < 1):
        raise ValueError("grad0 is the gradient at x: not with randinit, and only when there is a first step")
    x, x_adv = _start(x, eps, randinit)
    for t in range(steps):
        if t == 0 and grad0 is not None:
            _first_step(x_adv, grad0, gamma, x, eps, clip)
            continue
        _ascend(x_adv, lambda t: model({"x": image_batch, "adv": t, "out_idx": idx, "flag": "tail",
                                        "low_level_feat": low_level_feat, "low_res": _low_res(criterion)}), criterion, y, gamma, x, eps, clip)
    return x_adv.requires_grad_(True)


def decoder_PGD(input_dict, image_batch, criterion, y=None, model=None, steps=3, eps=None, gamma=None, idx=1,
                randinit=False, clip=False, grad0=None):
    """SD-branch (decoder feature) PGD (:61-84): perturbs input_dict['adv'] in the dict and returns the dict.  The
    reference cannot clip here — its projection names an undefined `x` and raises NameError after the first step — and
    neither does this: same exception."""
    if grad0 is not None and (randinit or steps < 1):
        raise ValueError("grad0 is the gradient at the clean feature: not with randinit, and only when there is a first step")
    _, x_adv = _start(input_dict["adv"], eps, randinit)
    input_dict["adv"] = x_adv
    for t in range(steps):
        def logits_of(t):
            d = dict(input_dict)
            d["adv"] = t
            return model({"x": image_batch, "adv": d, "out_idx": idx + "_tail", "flag": "clean", "low_res": _low_res(criterion)})
        if t == 0 and grad0 is not None:
            _first_step(x_adv, grad0, gamma, None, 0.0, False)
        else:
            _ascend(x_adv, logits_of, criterion, y, gamma, None, 0.0, False)
        if clip:
            raise NameError("name 'x' is not defined")
    input_dict["adv"] = x_adv.requires_grad_(True)
    return input_dict


def adv_input(x=None, criterion=None, y=None, model=None, steps=3, eps=None, gamma=None, randinit=False, clip=False):
    """Image-space PGD (:86-105), clamped to [0, 1] at the end."""
    x, x_adv = _start(x, eps, randinit)
    for _ in range(steps):
        _ascend(x_adv, lambda t: model({"x": t, "adv": None, "out_idx": 0, "flag": "clean", "low_level_feat": None}),
                criterion, y, gamma, x, eps, clip)
    lo, hi = torch.zeros_like(x_adv), torch.ones_like(x_adv)
    ops.tensor_clamp_(x_adv, lo, hi)
    return x_adv.requires_grad_(True)


def _dropout_active(model):
    """A training-mode nn.Dropout with p > 0 (and no caller-supplied mask pending) anywhere in `model`."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout) and m.training and m.p > 0 and getattr(m, "mask", None) is None:
            return True
    return False


def seg_train_step(model, optimizer, criterion, images, labels, **kw):
    """One iteration of Segmentation/main_aug_final.py:158-232: SE feature PGD + SD decoder PGD, three SAT sample points
    (`get_sample_points`), `mix_feature` where --mix_layer / --mix_sd say so, four forwards, loss = 0.7*clean +
    0.1*(se1 + se2 + sd), backward, optimizer step.  Flags carry the reference's names (args.py:19-34); see seg_train_phases."""
    out = {}
    for _ in seg_train_phases(model, optimizer, criterion, images, labels, out, **kw):
        pass
    return out


def seg_train_phases(model, optimizer, criterion, images, labels, out, *, steps=1, eps=2.0, gamma_se=0.5, gamma_sd=0.5,
                     pertub_idx_se=3, pertub_idx_sd="aspp", mix_layer="11", mix_sd=True, noise_sd=0.0, randinit=False,
                     clip=False, dual_bn=False, fold_clean=None, defer_step=False, fold_pgd0=None, batch_tails=None):
    """The iteration as a generator (data-parallel callers, seg_trainer.SegTrainer): with the folded clean pass the graph is
    cut at the SE point and at the low-level feature, the joint backward runs in two parts — everything behind the cuts
    (layer4, ASPP, decoder: 53 % of the parameters), then the head — and the generator yields "tail" in between: those
    layers' gradients are final there and their exchange can fly while the head's backward computes.  `out` is filled with
    the step's observables at the end."""
    from .deeplab import seg_criterion
    f0, f1 = int(mix_layer[0]), int(mix_layer[1])
    criterion = seg_criterion(criterion)        # one-pass HIP cross-entropy for a plain nn.CrossEntropyLoss(ignore_index=...)
    if images.is_cuda:
        ops.acc_reset(images.device)            # BatchNorm accumulator arena: one memset per iteration
    optimizer.zero_grad()
    # fold_clean (None: whenever the model offers it): the head pass (:166), the decoder-PGD input pass (:167) and the
    # clean forward (:193) are the same layers on the same images with the same weights — one pass (deeplab.py
    # forward_clean_folded) instead of three, BatchNorm running statistics updated in the reference's order.
    fold = getattr(model, "fold_ok", None) is not None and model.fold_ok(images) and type(pertub_idx_se) == int \
        and pertub_idx_sd in ("aspp", "concat")
    fold = fold if fold_clean is None else (bool(fold_clean) and fold)
    # fold_pgd0 (None: whenever the fold runs, PGD starts at the clean feature and takes at least one step): the first pass
    # of BOTH PGD loops is the clean forward too — same logits, same cross-entropy, same input gradient (sign() ignores
    # the joint loss's 0.7) — so the clean loss is back-propagated through the tail right away and the loops start at
    # their second pass.  Not with dual_bn (the PGD passes then normalise with the other BatchNorm set).
    # ... and not while a Dropout with p > 0 is active in the folded region: the reference draws a fresh mask in every
    # model(...) call (_deeplab.py:185: the decoder-PGD input pass :167, the clean forward :193 and every SE-PGD pass run
    # ASPP's dropout), so those passes are NOT the same function of the same point and sharing one pass would share one
    # mask.  The clean fold alone keeps the reference's process: :167 and :193 each apply their own draw to the one
    # pre-dropout ASPP output (K + 4 draws per iteration either way: tests/test_deeplab_gpu.py).
    pgd0 = fold and steps >= 1 and not randinit and not dual_bn and not clip and not _dropout_active(model)
    pgd0 = pgd0 if fold_pgd0 is None else (bool(fold_pgd0) and pgd0)
    g_se = g_sd = l0 = None
    wts = (0.7, 0.1, 0.1, 0.1)                                       # main_aug_final.py:216
    fused = getattr(criterion, "fused", False)
    if fold:
        fc = model.forward_clean_folded(images, pertub_idx_se, pertub_idx_sd, pgd0, cut=True)
        dec, low = fc.dec, fc.low
        fm_sd, fm_se = dec["adv"].detach().float(), fc.fm_se.detach().float()
        if pgd0:
            if fused and fc.logits.is_cuda and fc.logits.dtype == torch.float32:
                l0 = criterion(fc.logits, labels, grad_scale=wts[0])
                torch.autograd.backward([l0], [ops.one(images.device)])
            else:
                l0 = criterion(fc.logits, labels)
                (wts[0] * l0).backward()
            g_se, g_sd = fc.se_in.grad, fc.sd_t.grad
    else:
        fc = None
        out_se = model({"x": images, "adv": None, "out_idx": pertub_idx_se, "flag": "head"})
        dec = model({"x": images, "adv": None, "out_idx": pertub_idx_sd + "_head", "flag": "clean"})
        fm_sd = dec["adv"].detach().float()       # (bf16 activations on the product path: the A-FAN operators work in fp32)
        low = out_se["low_level"]
        fm_se = out_se["out"].detach().float()
    # dual_bn (option, no reference counterpart): every pass over adversarial features — the two PGD loops and the three
    # perturbed forwards — normalises with the auxiliary BatchNorm set (resnet_s.enable_dual_bn); no-op otherwise
    from .resnet_s import bn_branch, bn_groups
    adv_bn = (lambda: bn_branch(model, "adv")) if dual_bn else contextlib.nullcontext
    with adv_bn():
        adv_se = PGD(x=fm_se, image_batch=images, low_level_feat=low, criterion=criterion, y=labels, model=model, steps=steps,
                     eps=(eps / 255), gamma=(gamma_se / 255), idx=pertub_idx_se, randinit=randinit, clip=clip, grad0=g_se)
        if pgd0:
            fc.replay_sd_pgd0_bn()              # the decoder loop's first pass updates the decoder's BatchNorms here
        adv_sd_dict = decoder_PGD(input_dict=dec, image_batch=images, criterion=criterion, y=labels, model=model, steps=steps,
                                  eps=(eps / 255), gamma=(gamma_sd / 255), idx=pertub_idx_sd, randinit=randinit, clip=clip,
                                  grad0=g_sd)
    adv_sd = adv_sd_dict["adv"].detach()
    if mix_sd:
        adv_sd = mix_feature(fm_sd, adv_sd)
    if noise_sd != 0:
        ops.axpy_noise_(adv_sd, torch.rand(adv_sd.shape).to(adv_sd.device, non_blocking=True), gamma_sd * noise_sd)
    adv_sd_dict["adv"] = adv_sd
    # get_sample_points + mix_feature on the flagged points (main_aug_final.py:186-192) in one launch
    pts = sample_points_mixed(fm_se, adv_se.detach(), 3, (f0, f1)) if (f0 or f1) else get_sample_points(fm_se, adv_se.detach(), 3)
    if fold:
        fc.replay_deferred_bn()                   # :193's BatchNorm side effects, after the PGD loops as in the reference
        o0 = fc.logits
    else:
        o0 = model({"x": images, "adv": None, "out_idx": 0, "flag": "clean"})
    with adv_bn():
        lr = _low_res(criterion)        # the perturbed forwards' full-resolution logits are never looked at: loss only
        # batch_tails (None: whenever the model offers it): the two sample-point forwards (:206-209) run the same layers with the
        # same weights on two feature maps — ONE pass over their concatenation [point 1 | point 2] with BatchNorm statistics,
        # running updates (point 1 first) and dropout draws per half (resnet_s.bn_groups(2)): the reference's two calls, half
        # the launches of a part of the step that is bound by their count (2 images on 256 CUs)
        bt = fold if batch_tails is None else (bool(batch_tails) and getattr(model, "fold_ok", None) is not None and model.fold_ok(images))
        if bt:
            with bn_groups(2):
                o12 = model({"x": images, "adv": torch.cat([pts[1], pts[2]], dim=0), "out_idx": pertub_idx_se, "flag": "tail",
                             "low_level_feat": torch.cat([low, low], dim=0), "low_res": lr})
            o1, o2 = _halves(o12)
        else:
            o1 = model({"x": images, "adv": pts[1], "out_idx": pertub_idx_se, "flag": "tail", "low_level_feat": low, "low_res": lr})
            o2 = model({"x": images, "adv": pts[2], "out_idx": pertub_idx_se, "flag": "tail", "low_level_feat": low, "low_res": lr})
        o3 = model({"x": images, "adv": adv_sd_dict, "out_idx": pertub_idx_sd + "_tail", "flag": "clean", "low_res": lr})
    one = ops.one(images.device) if images.is_cuda else None
    if pgd0:
        # the clean term went through the tail already (what it left at the cuts enters the head graph below)
        if fused and all(_f32_logits(o) for o in (o1, o2, o3)):
            l1, l2, l3 = (criterion(o, labels, grad_scale=w) for o, w in zip((o1, o2, o3), wts[1:]))
            torch.autograd.backward([l1, l2, l3], [one, one, one])
        else:
            l1, l2, l3 = (criterion(o, labels) for o in (o1, o2, o3))
            (0.1 * l1 + 0.1 * l2 + 0.1 * l3).backward()
        with torch.no_grad():
            loss = 0.7 * l0 + 0.1 * l1 + 0.1 * l2 + 0.1 * l3
    elif fused and all(_f32_logits(o) for o in (o0, o1, o2, o3)):
        # every term's stored gradient already carries its weight: four roots, no scaling passes over the logits
        l0, l1, l2, l3 = (criterion(o, labels, grad_scale=w) for o, w in zip((o0, o1, o2, o3), wts))
        with torch.no_grad():
            loss = 0.7 * l0 + 0.1 * l1 + 0.1 * l2 + 0.1 * l3
        torch.autograd.backward([l0, l1, l2, l3], [one, one, one, one])
    else:
        l0, l1, l2, l3 = (criterion(o, labels) for o in (o0, o1, o2, o3))
        loss = 0.7 * l0 + 0.1 * l1 + 0.1 * l2 + 0.1 * l3
        loss.backward()
    if fold:
        # behind the cuts everything is final: layer4 / ASPP / decoder parameter gradients can be exchanged now
        yield "tail"
        cut_t = [t for t, g in ((fc.fm_se, fc.se_in.grad), (fc.low_graph, fc.low_in.grad)) if g is not None]
        cut_g = [g for g in (fc.se_in.grad, fc.low_in.grad) if g is not None]
        if cut_t:
            torch.autograd.backward(cut_t, cut_g)               # the head, traversed once
    if not defer_step:          # (data-parallel callers all-reduce the gradient arena first: seg_trainer.SegTrainer)
        optimizer.step()
    out.update({"loss": loss.detach(), "losses": torch.stack([l0, l1, l2, l3]).detach(), "adv_se": adv_se.detach(),
            "adv_sd": adv_sd.detach(), "fm_se": fm_se, "fm_sd": fm_sd, "out_clean": o0.detach(),
            "fold_clean": bool(fold), "fold_pgd0": bool(pgd0), "batch_tails": bool(bt)})         # (which schedule ran: bench.py reports it)
