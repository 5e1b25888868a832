"""Detection feature-space operators and training iteration — names, signatures and error
behaviour of the reference's `Detection/attack_algo.py`, and the loop body of `Detection/train_aug_sat_muti_advt.py:70-172`:

    compute_loss(l1, l2, l3, l4)                                                                          :21-27
    PGD(x, image_batch, y, model, steps, eps, gamma, idx, randinit, clip)                                 :48-74
    rpn_roi_PGD(layer, rpn_roi_output_dict, y, model, steps, eps, gamma, randinit, clip, only_roi_loss)   :77-150
    adv_input(x, y, model, steps, eps, gamma, randinit, clip)                                             :153-178
    get_sample_points / mix_feature / tensor_clamp / linfball_proj                      (shared with attack_algo.py)
    det_train_step(model, optimizer, image_batch, bboxes_batch, labels_batch, loss_settings)   train_aug_sat_muti_advt.py:70-172
    det_train_phases(...)                         the same iteration as a generator with the backward cut at the backbone's output

`model` is anything that follows the reference's protocol (`Detection/model.py:40-185`):
`model.train().forward({'x', 'adv', 'out_idx', 'flag'}, bboxes, labels)` -> four per-image loss tensors, a feature map
(flag 'head') or the ROI dict ('roi_head').  The sign step / projection / noise / clamp / mix / sample-point arithmetic runs in
libafan_hip.so; NMS and ROIAlign for the model's own layers are in det_ops.py.  The Faster-RCNN model itself is out of
scope; the Faster-RCNN model is det_model.py."""
import os

import torch

from . import ops
from .attack_algo import get_sample_points, linfball_proj, mix_feature, sample_points_mixed, tensor_clamp  # noqa: F401
from .det_ops import PGD, sum_of_means  # noqa: F401  (Detection/attack_algo.py:48-74)
from .resnet_s import dgrad_only


def compute_loss(loss1, loss2, loss3, loss4):
    """:21-27: the sum of the four means."""
    return sum_of_means(loss1, loss2, loss3, loss4)


BATCH_LAYER3 = os.environ.get("AFAN_DET_BATCH_L3", "1") != "0"      # 0: every final pass runs its own layer3 (A/B, tests)
BATCH_ROI_HEAD = os.environ.get("AFAN_DET_BATCH_ROI", "1") != "0"    # 0: every final pass runs its own ROI head (A/B, tests)
SHARE_PROPOSALS = os.environ.get("AFAN_DET_SHARE_PROPOSALS", "1") != "0"   # 0: every tail on the clean conv4 map computes its own labels / proposals (A/B, tests)
BATCH_PGD_TAILS = os.environ.get("AFAN_DET_BATCH_PGD", "1") != "0"   # 0: the three one-step feature PGDs run their tails one by one (A/B, tests)


class NoiseAhead:
    """The image PGD's random start (:159: `torch.rand(x.shape)` on the CPU default generator, 1.6 M numbers for a 600 x 904 image:
    2.5 ms of host time with the device idle behind it) drawn at the END of the previous iteration instead, while the device still
    runs that iteration's backward.  The generator's stream is unchanged as long as the caller draws nothing from it between two
    iterations (a synthetic loop, a loader with its own generator): the iteration's first draw IS this one.  Opt-in
    (det_trainer.DetTrainer(noise_ahead=True)); `take(shape)` hands the stored draw back, or None."""

    def __init__(self):
        self.u = None

    def draw(self, shape):
        self.u = torch.rand(tuple(shape)).pin_memory()

    def take(self, shape):
        u, self.u = self.u, None
        return u if (u is not None and tuple(u.shape) == tuple(shape)) else None


def _start(x, eps, randinit, ahead=None):
    if x.device.type != "cuda":
        raise ops.AfanLibraryError("x must live on the MI355X (no CPU path in this build)")
    x = x.detach().float()
    x = x if (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))) else x.contiguous()
    x_adv = x.clone()
    if randinit:   # noise from the CPU default generator, then host -> device, like the reference (:52, :159)
        u = ahead.take(x_adv.shape) if ahead is not None else None
        u = (u if u is not None else torch.rand(x_adv.shape)).to(x.device, non_blocking=True)
        if u.stride() != x_adv.stride():
            u = u.contiguous(memory_format=torch.channels_last)
        ops.axpy_noise_(x_adv, u, eps)
    return x, x_adv


def _ascend(x_adv, loss_of, gamma, x, eps, clip):
    xin = x_adv.detach().requires_grad_(True)
    with dgrad_only():          # only_inputs=True (:66,:103,:170): no parameter gradient is computed, none is touched
        grad = torch.autograd.grad(loss_of(xin), xin, only_inputs=True)[0]
    if grad.stride() != x_adv.stride():
        grad = grad.contiguous(memory_format=torch.channels_last) if (x_adv.dim() == 4 and not x_adv.is_contiguous()) else grad.contiguous()
    ops.pgd_step_(x_adv, grad, gamma, x, eps if eps is not None else 0.0, clip)     # one launch: sign step (+ projection)


def _pgd1_from_clean(model, col, idx, image_batch, y, eps, gamma):
    """`PGD(fm[idx - 1], ..., steps=1, idx=idx)` without a random start (:84-86) when the clean pass's own activations are at hand:
    its one forward runs the backbone's remaining stages on the CLEAN feature map — the numbers the clean ROI-head pass already
    computed — so only the part behind the backbone is run again (RPN, proposals, ROI head: their sampling draws are this call's
    own, as in the reference), and its gradient is carried back through the clean pass's stored activations with the
    input-gradient-only launches (det_model.stage_input_gradient).  Same values as PGD(); None where the stages did not run in the
    one-node form (fp32, NCHW: PGD() itself then runs)."""
    from . import det_model
    stages = [model.features.layer1, model.features.layer2, model.features.layer3]
    outs = [col.get(("out", i)) for i in (1, 2, 3)]
    if not all(det_model.stage_input_gradient(stages[i - 1], outs[i - 1]) for i in range(idx + 1, 4)):
        return None                  # (decided before anything is drawn from the host generator)
    x = col[idx].detach().float()
    x = x if (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))) else x.contiguous()
    x_adv = x.clone()
    xin = col[3].detach().requires_grad_(True)
    with dgrad_only():
        l1, l2, l3, l4 = model.train().forward({"x": image_batch, "adv": xin, "out_idx": 3, "flag": "tail"}, y["bb"], y["lb"])
        g = torch.autograd.grad(sum_of_means(l1, l2, l3, l4), xin, only_inputs=True)[0]
    for i in range(3, idx, -1):
        g = det_model.stage_input_gradient(stages[i - 1], outs[i - 1], g)
    grad = g.float()
    if grad.stride() != x_adv.stride():
        grad = grad.contiguous(memory_format=torch.channels_last) if (x_adv.dim() == 4 and not x_adv.is_contiguous()) else grad.contiguous()
    ops.pgd_step_(x_adv, grad, gamma, x, eps, False)
    return x_adv.requires_grad_(True)


def _pgd_three_from_clean(model, col, image_batch, y, specs, share=None):
    """The three one-step feature PGDs of :84-88 — `PGD(fm[0], idx=1)`, `PGD(fm[1], idx=2)`, `PGD(fm[2], idx=3)`, no random start — in ONE
    tail: each of them runs RPN + proposals + ROI head on the CLEAN conv4 map (their forwards differ only in the host generator's
    sampling draws) and takes the gradient back to its own layer.  Here the three tails are `Model.forward_heads_many` on three leaves of
    that map (the RPN trunk, ROIAlign, layer4 and the Linear pairs run once on three passes' rows; sampling pass by pass in the
    reference's order), one backward of the three losses' sum gives each leaf its own gradient (a pass's loss depends on its leaf only),
    and the first two are carried back through the clean pass's stored activations as in _pgd1_from_clean.  specs: [(idx, eps, gamma)]
    for idx 1, 2, 3.  None where the one-node stages or forward_heads_many are not there (the caller runs the three calls)."""
    from . import det_model
    if not hasattr(model, "forward_heads_many"):
        return None
    stages = [model.features.layer1, model.features.layer2, model.features.layer3]
    outs = [col.get(("out", i)) for i in (1, 2, 3)]
    if not all(det_model.stage_input_gradient(stages[i - 1], outs[i - 1]) for i in (2, 3)):
        return None                  # (decided before anything is drawn from the host generator)
    xins = [col[3].detach().requires_grad_(True) for _ in specs]
    with dgrad_only():
        res = model.train().forward_heads_many([{"x": image_batch, "adv": xi, "out_idx": 3, "flag": "tail"} for xi in xins], y["bb"], y["lb"],
                                               share=share)
        losses = [sum_of_means(*r) for r in res]
        total = losses[0]
        for l in losses[1:]:
            total = total + l
        grads = torch.autograd.grad(total, xins, only_inputs=True)
    advs = []
    for (idx, eps, gamma), g in zip(specs, grads):
        for i in range(3, idx, -1):
            g = det_model.stage_input_gradient(stages[i - 1], outs[i - 1], g)
        x = col[idx].detach().float()
        x = x if (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))) else x.contiguous()
        x_adv = x.clone()
        grad = g.float()
        if grad.stride() != x_adv.stride():
            grad = grad.contiguous(memory_format=torch.channels_last) if (x_adv.dim() == 4 and not x_adv.is_contiguous()) else grad.contiguous()
        ops.pgd_step_(x_adv, grad, gamma, x, eps, False)
        advs.append(x_adv.requires_grad_(True))
    return advs


def rpn_roi_PGD(layer="roi", rpn_roi_output_dict=None, y=None, model=None, steps=1, eps=None, gamma=None, randinit=False,
                clip=False, only_roi_loss=True):
    """:77-150.  layer 'roi': perturbs rpn_roi_output_dict['roi_output_dict']['roi_feature_map'] in the dict (loss = the two
    proposal losses, or all four).  clip=True names an undefined `rpn_feature1` in the reference (:110) and raises
    NameError after the first step — so does this.  layer 'rpn': the reference runs `steps` forwards and never moves the
    feature (:129-146: the update is commented out) — the dict comes back with an unperturbed requires_grad copy."""
    d = rpn_roi_output_dict
    if layer == "roi":
        _, x_adv = _start(d["roi_output_dict"]["roi_feature_map"], eps, randinit)
        d["roi_output_dict"]["roi_feature_map"] = x_adv
        for _ in range(steps):
            def loss_of(t):
                d["roi_output_dict"]["roi_feature_map"] = t
                ao, at, pc, pt = model.train().forward({"adv": d, "out_idx": "roi_tail", "flag": "clean"}, y["bb"], y["lb"])
                return sum_of_means(pc, pt) if only_roi_loss else sum_of_means(ao, at, pc, pt)
            _ascend(x_adv, loss_of, gamma, None, 0.0, False)
            if clip:
                raise NameError("name 'rpn_feature1' is not defined")
        d["roi_output_dict"]["roi_feature_map"] = x_adv.requires_grad_(True)
        return d
    if layer == "rpn":
        _, x_adv = _start(d["rpn_feature_map_dict"]["rpn_feature"], eps, randinit)
        x_adv.requires_grad_(True)
        d["rpn_feature_map_dict"]["rpn_feature"] = x_adv
        for _ in range(steps):
            model.train().forward({"adv": d, "out_idx": "rpn_tail", "flag": "clean"}, y["bb"], y["lb"])
        return d
    assert False


def adv_input(x=None, y=None, model=None, steps=3, eps=None, gamma=None, randinit=False, clip=False, noise_ahead=None):
    """:153-178: image-space PGD under the sum of the four losses, clamped to [0, 1] at the end."""
    x, x_adv = _start(x, eps, randinit, noise_ahead)
    for _ in range(steps):
        _ascend(x_adv, lambda t: compute_loss(*model.train().forward({"x": t, "adv": None, "out_idx": -1, "flag": "clean"},
                                                                     y["bb"], y["lb"])), gamma, x, eps, clip)
    ops.tensor_clamp_(x_adv, torch.zeros_like(x_adv), torch.ones_like(x_adv))
    return x_adv.requires_grad_(True)


def det_train_step(model, optimizer, image_batch, bboxes_batch, labels_batch, loss_settings=1):
    """One iteration of Detection/train_aug_sat_muti_advt.py:70-172 (see det_train_phases)."""
    out = {}
    for _ in det_train_phases(model, optimizer, image_batch, bboxes_batch, labels_batch, out, loss_settings=loss_settings):
        pass
    return out


def det_train_phases(model, optimizer, image_batch, bboxes_batch, labels_batch, out, *, loss_settings=1, cut=False, defer_step=False,
                     noise_ahead=None):
    """One iteration of Detection/train_aug_sat_muti_advt.py:70-172 as a generator: adversarial image (5 steps, randinit,
    clip), the three backbone feature maps and the ROI dict, three one-step feature PGDs (`multi-layer`), five SAT sample
    points of the deepest one with points 1 and 2 re-normalised by mix_feature (one fused launch), the one-step ROI feature
    PGD + mix_feature, eight forwards, and the weighted loss of :141-153 (loss_settings 1-4).

    cut=True (data-parallel callers, det_trainer.DetTrainer; needs a model with `cut_features`): the eight training forwards'
    graphs are cut at the backbone's output (the conv4 feature map every forward hands to the RPN and the ROI head), the joint
    backward runs in two parts — everything behind the cuts (RPN, ROIAlign, layer4, the two heads: 8 passes), then the
    backbone (layer3 / layer2: the passes that reach it) — and the generator yields "tail" in between: the gradients of
    layer4 / rpn / detection heads (the arena's suffix) are final there and their all-reduce can fly under the backbone's
    backward.  Same arithmetic: what the tails leave at the cuts enters the backbone graph as its output gradient."""
    y = {"bb": bboxes_batch, "lb": labels_batch}
    fwd = lambda d: model.train().forward(d, bboxes_batch, labels_batch)
    if hasattr(model, "begin_iteration"):
        model.begin_iteration()
    adv_image = adv_input(x=image_batch, y=y, model=model, steps=5, eps=(2.0 / 255), gamma=(0.3 / 255), randinit=True, clip=True,
                          noise_ahead=noise_ahead)
    col = None
    if getattr(model, "collects_head_features", False):
        # the three head passes (:78-80) and the clean ROI-head pass (:81) run the same backbone on the same images (frozen
        # BatchNorm, no dropout, no random draw before the RPN): the head passes' values are that pass's stage outputs
        col = {}
        rpn_share = {} if SHARE_PROPOSALS else None      # the clean pass's anchor labels / proposals / candidate lists: the feature PGDs' tails see the same map
        rr = fwd({"x": image_batch, "adv": None, "out_idx": "roi_head", "flag": "clean", "collect": col, "rpn_share": rpn_share})
        fm = [col[i] for i in (1, 2, 3)]
    else:
        if hasattr(model, "head_features"):      # prefixes of one another: one pass, no graph
            fm = model.train().head_features(image_batch, (1, 2, 3))
        else:
            fm = [fwd({"x": image_batch, "adv": None, "out_idx": i, "flag": "head"}).detach() for i in (1, 2, 3)]
        rr = fwd({"x": image_batch, "adv": None, "out_idx": "roi_head", "flag": "clean"})
    clean_sd = rr["roi_output_dict"]["roi_feature_map"].detach()
    fold = col is not None and os.environ.get("AFAN_DET_FOLD_PGD", "1") != "0"      # 0: PGD() as written (A/B, tests)
    three = (_pgd_three_from_clean(model, col, image_batch, y, [(1, 0.1 / 255, 0.001 / 255), (2, 0.1 / 255, 0.001 / 255), (3, 2.0 / 255, 1.0 / 255)],
                                   share=rpn_share if SHARE_PROPOSALS else None)
             if (fold and BATCH_PGD_TAILS) else None)
    if three is not None:
        adv1, adv2, adv3 = three
    else:
        adv1 = _pgd1_from_clean(model, col, 1, image_batch, y, 0.1 / 255, 0.001 / 255) if fold else None
        adv2 = _pgd1_from_clean(model, col, 2, image_batch, y, 0.1 / 255, 0.001 / 255) if fold else None
        if adv1 is None:
            adv1 = PGD(fm[0], image_batch, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=1)
        if adv2 is None:
            adv2 = PGD(fm[1], image_batch, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=2)
        adv3 = PGD(fm[2], image_batch, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(1.0 / 255), idx=3)
    pts = sample_points_mixed(fm[2].float(), adv3.detach(), 5, (True, True, False, False))       # :95-97 in one launch
    adv_rr = rpn_roi_PGD(rpn_roi_output_dict=rr, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(0.2 / 255), only_roi_loss=False)
    adv_sd = mix_feature(clean_sd.float(), adv_rr["roi_output_dict"]["roi_feature_map"].detach())
    adv_rr["roi_output_dict"]["roi_feature_map"] = adv_sd
    dicts = [{"x": adv_image.detach(), "adv": None, "out_idx": 0, "flag": "clean"},
             {"x": image_batch, "adv": adv1, "out_idx": 1, "flag": "tail"},
             {"x": image_batch, "adv": adv2, "out_idx": 2, "flag": "tail"}] + \
            [{"x": image_batch, "adv": pts[j], "out_idx": 3, "flag": "tail"} for j in (1, 2, 3, 4)] + \
            [{"adv": adv_rr, "out_idx": "roi_tail", "flag": "clean"}]
    if BATCH_LAYER3 and hasattr(getattr(model, "features", None), "forward_many"):
        # (round 6) the three passes that reach the backbone are independent of one another until their RPN: their layer3 runs once on the
        # three feature maps' batch (det_model backbone.forward_many), each pass then enters behind it like the sample-point passes do
        f3 = model.train().features.forward_many(dicts[:3])
        if f3 is not None:
            dicts[:3] = [{"x": image_batch, "adv": f, "out_idx": 3, "flag": "tail"} for f in f3]
    cuts = []
    def run_all():
        # (round 6) the seven passes in front of the ROI-tail pass are independent of one another: their ROI heads run as one
        if BATCH_ROI_HEAD and hasattr(model, "forward_heads_many"):
            outs = model.train().forward_heads_many(dicts[:7], bboxes_batch, labels_batch)
            return [compute_loss(*o) for o in outs] + [compute_loss(*fwd(d)) for d in dicts[7:]]
        return [compute_loss(*fwd(d)) for d in dicts]

    if cut:
        if not hasattr(model, "cut_features"):
            raise ops.AfanLibraryError("det_train_phases(cut=True) needs a model with cut_features() (det_model.Model)")
        with model.cut_features(cuts):
            L = run_all()
    else:
        L = run_all()
    loss_clean_adv = 0.9 * (0.2333 * (L[0] + L[3] + L[4] + L[5] + L[6]) + 0.1 * L[7]) + 0.05 * (L[1] + L[2])
    if loss_settings == 1:
        loss = loss_clean_adv
    elif loss_settings in (2, 3, 4):
        a, b = {2: (0.5, 0.5), 3: (0.4, 0.6), 4: (0.3, 0.7)}[loss_settings]
        loss = a * loss_clean_adv + b * L[0]
    else:
        assert False
    optimizer.zero_grad()
    loss.backward()
    if cut:
        yield "tail"            # behind the cuts everything is final
        live = [(f, c.grad) for f, c in cuts if c.grad is not None]
        if live:
            torch.autograd.backward([f for f, _ in live], [g for _, g in live])         # the backbone, every pass at once
    if noise_ahead is not None:      # every draw of this iteration is behind us; the device is busy with the backward just issued
        noise_ahead.draw(image_batch.shape)
    if not defer_step:
        optimizer.step()
    out.update({"loss": loss.detach(), "losses": torch.stack(L).detach(), "adv_image": adv_image.detach(), "adv1": adv1.detach(),
                "adv2": adv2.detach(), "adv3": adv3.detach(), "adv_sd": adv_sd.detach(), "fm3": fm[2], "cuts": len(cuts)})
