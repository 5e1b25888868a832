"""Faster-RCNN / frozen-BatchNorm ResNet-101 on the library's kernels (cv_a-fan_amd/det_model.py) against the reference's OWN
`Detection/model.py` (+ backbone/resnet101_ori.py, rpn/, roi/pooler.py, bbox.py), run by oracle/gen_golden.py `frcnn` at
2 x 3 x 128 x 160 in BOTH pooler modes — 'pooling' (tests/golden/det_frcnn_r101.npz) and the reference's default 'align'
(config/config.py:14; det_frcnn_r101_align.npz) — with `support.layer.nms` supplied by the plain-C oracle NMS (pinned to the
reference's own nms vector) and `support.layer.roi_align` by the reference's own CPU forward kernel (ROIAlign_cpu.cpp:4-219
compiled unedited) with the adjoint backward (pinned: tests/test_det_oracle.py).

Held tightly: seeded construction (tests/test_host_logic.py, CPU), the three backbone feature maps, the RPN logits, the
proposals, the four per-image losses of a training forward with the reference's host `randperm` draws, the gradients of
that forward, the signs of the one-step feature PGD — and, since round 6, the full iteration of
train_aug_sat_muti_advt.py:70-172 too.  Rounds 3-5 held its eight losses to 2 % with an argument (the adversarial image is five
sign() steps on 61 440 pixels; a flipped pixel reorders proposals of nearly equal score; `randperm` samples by position) and no
measurement.  tests/golden/ref_det_floor.npz (oracle/gen_golden.py detfloor) is the measurement: the reference's own iteration run
again in float64 / ATen-native fp32 / channels-last fp32 and under four draws of 1e-6 image noise moves 6.6-16.8 % of the adversarial
image's pixels (arithmetic variants alone: 6.6-7.9 %) and the eight losses by at most 6.1e-5 relative — the argument was wrong about
the losses — and the product sits inside that: 4.1e-5 / 1.7e-5 (pooling / align; profiles/r06_parity_measurements.txt).  The bounds
are max(2 x the reference's own spread, 1e-4), as for the other two trainers."""
import os

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _build(pkg, g, gpu, dtype, nhwc, mode):
    torch.manual_seed(7)
    m = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode=mode, anchor_sizes=tuple(int(v) for v in g["anchor_sizes"]),
                                           rpn_pre_nms_top_n=int(g["nms_top_n"][0]), rpn_post_nms_top_n=int(g["nms_top_n"][1]))
    for b in m.modules():
        if isinstance(b, pkg.det_model.Bottleneck):
            b.bn3.weight.data.mul_(float(g["damp"]))
    ck = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in m.state_dict().values()])
    np.testing.assert_allclose(ck, g["ck0"], rtol=1e-12, atol=1e-9)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    return m.set_compute_dtype(dtype).set_channels_last(nhwc).to(gpu).train()


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def _golden_for(mode):
    g = golden("det_frcnn_r101" + ("" if mode == "pooling" else "_" + mode))
    assert str(g["pooler_mode"]) == mode
    return g


@pytest.mark.parametrize("mode", ["pooling", "align"])
@pytest.mark.parametrize("nhwc", [False, True])
def test_faster_rcnn_fp32_matches_reference_model(pkg, gpu, nhwc, mode):
    g = _golden_for(mode)
    m = _build(pkg, g, gpu, torch.float32, nhwc, mode)
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    before = dict(pkg.ops.CALLS)
    # (1) head passes: model.py:42-50 -> resnet101_ori.py:203-237
    for i in (1, 2, 3):
        fm = m.train().forward({"x": images, "adv": None, "out_idx": i, "flag": "head"}, bboxes, labels).detach()
        assert _rel(fm[:, ::8, ::2, ::2].float().cpu().numpy(), g[f"fm{i}_sub"]) <= 2e-5
        assert abs(float(fm.double().norm()) - float(g[f"fm{i}_norm"])) <= 2e-5 * float(g[f"fm{i}_norm"])
    fm3 = fm
    # (2) one training forward: RPN logits, proposals, the four per-image losses (model.py:54-72)
    seen = {}
    real = m.rpn.forward_and_propose             # (the training forward's RPN pass + proposal layer: one host read for both)

    def spy(*a, **k):
        out = real(*a, **k)
        seen.update(obj=out[0].detach().float().clone(), tr=out[1].detach().float().clone(), proposals=out[4].detach().clone())
        return out
    m.rpn.forward_and_propose = spy
    torch.manual_seed(100)
    losses = m.train().forward({"x": images, "adv": None, "out_idx": 0, "flag": "clean"}, bboxes, labels)
    m.rpn.forward_and_propose = real
    np.testing.assert_allclose(seen["obj"].cpu().numpy(), g["rpn_obj"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(seen["tr"].cpu().numpy(), g["rpn_tr"], rtol=1e-4, atol=2e-5)
    assert seen["proposals"].shape == g["proposals"].shape
    np.testing.assert_allclose(seen["proposals"].cpu().numpy(), g["proposals"], rtol=1e-4, atol=1e-2)      # pixels
    got = np.stack([t.detach().float().cpu().numpy() for t in losses])
    np.testing.assert_allclose(got, g["fwd_losses"], rtol=1e-4, atol=1e-5)            # north_star's 1e-4, per image and per term
    for p in m.parameters():
        p.grad = None
    sum(t.mean() for t in losses).backward()
    named = {n: p for n, p in m.named_parameters() if p.grad is not None}
    names = [str(k) for k in g["param_names"]]
    assert sorted(named) == sorted(names)              # the same parameters receive a gradient (frozen stem / layer1 / BatchNorms do not)
    gn = np.array([float(named[n].grad.double().norm()) for n in names])
    rel = np.abs(gn - g["grad_norms"]) / (g["grad_norms"] + 1e-6 * g["grad_norms"].max())
    assert rel.max() <= 1e-3, [(names[i], gn[i], g["grad_norms"][i]) for i in np.argsort(-rel)[:4]]
    for k in g.files:
        if k.startswith("grad/"):
            assert _rel(named[k[5:]].grad.float().cpu().numpy(), g[k]) <= 1e-2, k      # (measured 5.5e-3 at layer2.0: ReLU / max-pool ties of 30 blocks)
    # (3) one-step feature PGD at out_idx 3 (attack_algo.py:48-74): the sign pattern
    torch.manual_seed(101)
    adv3 = pkg.det_attack_algo.PGD(fm3.float(), images, y={"bb": bboxes, "lb": labels}, model=m, steps=1, eps=(2.0 / 255),
                                   gamma=(1.0 / 255), idx=3)
    sg = torch.round((adv3.detach() - fm3.float()) / np.float32(1.0 / 255)).cpu().numpy().astype(np.int8)
    agree = float((sg == g["adv3_sign"]).mean())
    assert agree >= 0.995, agree
    assert pkg.ops.CALLS["vendor_conv"] == 0 and pkg.ops.CALLS["conv_general"] > before["conv_general"]


_DFLOOR = golden("ref_det_floor")


def _det_spread(mode, key, arith_only=False):
    pre = "det_frcnn_r101" + ("" if mode == "pooling" else "_" + mode)
    if arith_only:      # the arithmetic variants alone (float64, ATen-native fp32, channels-last fp32), without the input-noise draws
        return max(float(np.max(_DFLOOR[f"{pre}/{k}/{key}"])) for k in ("f64", "nomkldnn", "cl"))
    return float(_DFLOOR[pre + "/spread_" + key])


@pytest.mark.parametrize("mode", ["pooling", "align"])
def test_faster_rcnn_iteration_on_reference_golden(pkg, gpu, mode):
    """One iteration of train_aug_sat_muti_advt.py:70-172 (det_attack_algo.det_train_step) on the real model: the eight losses and
    the iteration loss within max(2 x the reference's own spread, 1e-4) of the reference's (ref_det_floor.npz: the reference's
    iteration against itself in three other arithmetics and under four draws of 1e-6 image noise), the adversarial image's pixels
    within twice the reference's own flip fraction, the weights after the SGD step by checksum."""
    g = _golden_for(mode)
    m = _build(pkg, g, gpu, torch.float32, True, mode)
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    arena = pkg.arena.ParamArena(m, skip=())
    opt = pkg.arena.ArenaSGD(arena, lr=0.001, momentum=0.9, weight_decay=0.0005)
    # replay the reference generator's host-RNG history up to the iteration (gen_detection_model: the draws of steps (2), (3))
    torch.manual_seed(102)
    r = pkg.det_attack_algo.det_train_step(m, opt, images, bboxes, labels, loss_settings=1)
    L = r["losses"].float().cpu().numpy()
    rel_L = float((np.abs(L - g["step_losses"]) / np.abs(g["step_losses"])).max())
    rel_loss = abs(float(r["loss"]) - float(g["step_loss"])) / float(g["step_loss"])
    d = (r["adv_image"][:, :, ::4, ::4].float().cpu().numpy() - g["adv_image_sub"])
    off = float((np.abs(d) > 1e-6).mean())
    b_L, b_loss = max(2 * _det_spread(mode, "losses_rel"), 1e-4), max(2 * _det_spread(mode, "loss_rel"), 1e-4)
    b_off = 2 * _det_spread(mode, "adv_pixels_off", arith_only=True)
    print(f"PARITY det_frcnn_r101 [{mode}, fp32 NHWC]: eight losses rel {rel_L:.2e} (reference-vs-reference spread {_det_spread(mode, 'losses_rel'):.2e}, "
          f"bound {b_L:.2e})   iteration loss rel {rel_loss:.2e} (spread {_det_spread(mode, 'loss_rel'):.2e}, bound {b_loss:.2e})   adversarial-image "
          f"pixels off {off:.4f} (reference-vs-reference {_det_spread(mode, 'adv_pixels_off'):.4f}, arithmetic variants only "
          f"{_det_spread(mode, 'adv_pixels_off', True):.4f}, bound {b_off:.4f})")
    assert rel_L <= b_L and rel_loss <= b_loss, (rel_L, b_L, rel_loss, b_loss)
    assert off <= b_off, (off, b_off)          # five sign() steps from a random start: the reference moves as many against itself
    ck1 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in m.state_dict().values()])
    np.testing.assert_allclose(ck1[:, 1], g["ck1"][:, 1], rtol=1e-5, atol=1e-6)      # (measured 5e-7; rounds 3-5 allowed 1e-3)
    assert pkg.ops.CALLS["vendor_conv"] == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_det_trainer_cut_backward_equals_one_backward(pkg, gpu, dtype):
    """det_trainer.DetTrainer's two-part backward (graphs cut at the backbone's output, "tail" announced in between: what the
    data-parallel exchange hangs on) against the one-backward iteration: the same forwards (identical losses), the same
    gradients up to summation order (ROIAlign's atomics), the same weights after the step; three cuts (the passes that reach
    the backbone: adversarial image, out_idx 1, out_idx 2)."""
    g = _golden_for("align")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    res = {}
    for seg in (False, True):
        m = _build(pkg, g, gpu, dtype, True, "align")
        tr = pkg.det_trainer.DetTrainer(m, segmented=seg)
        assert tr.tail_range() == (83, 103)        # layer4 + RPN + heads (the never-used features.fc stays outside the arena)
        torch.manual_seed(102)
        r = tr.step(images, bboxes, labels)
        torch.cuda.synchronize()
        res[seg] = (r, tr.arena.grad.clone(), tr.arena.param.clone())
    assert res[False][0]["cuts"] == 0 and res[True][0]["cuts"] == 3
    assert torch.equal(res[False][0]["losses"], res[True][0]["losses"]) and torch.equal(res[False][0]["adv_image"], res[True][0]["adv_image"])
    tol = 1e-5 if dtype == torch.float32 else 2e-2          # (bf16: the gradient at the cut is stored once in bf16 instead of being summed in bf16 by autograd)
    assert _rel(res[True][1].cpu().numpy(), res[False][1].cpu().numpy()) <= tol
    lo = tr.arena.offsets[83]
    assert float(res[True][1][lo:].abs().sum()) > 0 and float(res[True][1][:lo].abs().sum()) > 0
    assert _rel(res[True][2].cpu().numpy(), res[False][2].cpu().numpy()) <= 1e-6


def test_faster_rcnn_bf16_align_mode_runs_on_library_kernels(pkg, gpu):
    """The product configuration: bf16 channels-last backbone on the tuned MFMA kernels, ROIAlign + NMS + pooling kernels of
    the library, fp32 RPN / detection heads: a full iteration at a VOC-like image size; finite, protocol intact."""
    g = golden("det_frcnn_r101")
    torch.manual_seed(7)
    m = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode="align", rpn_pre_nms_top_n=2000, rpn_post_nms_top_n=300)
    for b in m.modules():
        if isinstance(b, pkg.det_model.Bottleneck):
            b.bn3.weight.data.mul_(0.2)
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    arena = pkg.arena.ParamArena(m, skip=())
    # every backbone / RPN-trunk convolution on the tuned bf16 kernels; the two 1x1 RPN heads (18 / 36 fp32 outputs) and the 7x7
    # stem (3 input channels: not listed) are the general f32-MFMA kernel's
    assert set(pkg.resnet_s.general_convs(m)) <= {"rpn._anchor_objectness", "rpn._anchor_transformer"}
    opt = pkg.arena.ArenaSGD(arena, lr=0.001, momentum=0.9, weight_decay=0.0005)
    gen = torch.Generator().manual_seed(3)
    images = torch.rand(1, 3, 384, 512, generator=gen).to(gpu)
    bboxes = torch.tensor([[[30., 40., 200., 260.], [220., 100., 480., 330.], [100., 200., 260., 370.]]], device=gpu)
    labels = torch.tensor([[5, 11, 2]], device=gpu)
    before = dict(pkg.ops.CALLS)
    torch.manual_seed(1)
    r = pkg.det_attack_algo.det_train_step(m, opt, images, bboxes, labels, loss_settings=1)
    assert np.isfinite(float(r["loss"])) and torch.isfinite(r["losses"]).all()
    assert float((r["adv_image"] - images).abs().max()) <= 2.0 / 255 + 1e-6          # projected onto the eps-ball, clamped to [0, 1]
    assert r["fm3"].shape == (1, 1024, 24, 32)
    assert pkg.ops.CALLS["vendor_conv"] == 0 and pkg.ops.CALLS["conv_fwd"] > before["conv_fwd"] and pkg.ops.CALLS["conv_dgrad"] > before["conv_dgrad"]
    assert torch.isfinite(arena.param).all() and float(arena.momentum_buf.abs().max()) > 0
    m.eval()
    with torch.no_grad():
        boxes, classes, probs, idx = m({"x": images, "adv": None, "out_idx": 0, "flag": "clean"})
    assert boxes.shape[1] == 4 and len(boxes) == len(classes) == len(probs) == len(idx)


def test_faster_rcnn_full_size_iteration_properties(pkg, gpu):
    """BASELINE configs[4] at the size bench.py times it (one 600 x 904 image per GPU, bf16 channels-last, the reference's default
    `align` pooler, 12 000 -> 2 000 proposals): one DetTrainer iteration, held to what does not depend on the size — every loss
    finite; the adversarial image inside the eps-ball and [0, 1]; each one-step feature perturbation exactly +-gamma (or 0) off its
    feature map (Detection/attack_algo.py:67: x + gamma * sign(g)); the 38 x 57 conv4 map; both halves of the gradient arena
    written; and the whole iteration REPRODUCIBLE bit for bit from the same state and host seed (no atomics left on the path: the
    channels-last ROIAlign backward is a gather)."""
    def run():
        torch.manual_seed(7)
        m = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode="align")
        for b in m.modules():
            if isinstance(b, pkg.det_model.Bottleneck):
                b.bn3.weight.data.mul_(0.2)
        m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
        tr = pkg.det_trainer.DetTrainer(m, lr=0.001, momentum=0.9, weight_decay=0.0005)
        gen = torch.Generator().manual_seed(3)
        images = torch.rand(1, 3, 600, 904, generator=gen).to(gpu)
        x0, y0 = torch.rand(1, 6, 1, generator=gen) * 640, torch.rand(1, 6, 1, generator=gen) * 340
        wh = 60 + torch.rand(1, 6, 2, generator=gen) * 200
        bboxes = torch.cat([x0, y0, x0 + wh[..., :1], y0 + wh[..., 1:]], dim=-1).to(gpu)
        labels = torch.randint(1, 21, (1, 6), generator=gen).to(gpu)
        fc0 = m.features.fc.weight.detach().clone()
        torch.manual_seed(1)
        r = tr.step(images, bboxes, labels)
        torch.cuda.synchronize()
        # the backbone's unused ImageNet classifier never receives a gradient: the reference's optim.SGD leaves it alone
        assert torch.equal(m.features.fc.weight.detach(), fc0) and not any(n.startswith("features.fc.") for n in tr.arena.names)
        return tr, images, r
    tr, images, r = run()
    assert torch.isfinite(r["losses"]).all() and np.isfinite(float(r["loss"])) and r["losses"].shape == (8,)
    assert float((r["adv_image"] - images).abs().max()) <= 2.0 / 255 + 1e-6 and 0.0 <= float(r["adv_image"].min()) and float(r["adv_image"].max()) <= 1.0
    assert tuple(r["fm3"].shape) == (1, 1024, 38, 57) and tuple(r["adv3"].shape) == (1, 1024, 38, 57)
    for key, gamma in (("adv3", 1.0 / 255),):
        d = (r[key].float() - r["fm3"].float()) / gamma
        assert float((d - d.round()).abs().max()) <= 2e-3 and float(d.abs().max()) <= 1.0 + 2e-3          # one sign() step
        assert float((d.round() != 0).float().mean()) > 0.5
    lo = tr.arena.offsets[tr.tail_range()[0]]
    assert float(tr.arena.grad[:lo].abs().sum()) > 0 and float(tr.arena.grad[lo:].abs().sum()) > 0 and torch.isfinite(tr.arena.param).all()
    assert pkg.ops.CALLS["vendor_conv"] == 0
    tr2, _, r2 = run()
    assert torch.equal(r["losses"], r2["losses"]) and torch.equal(tr.arena.grad, tr2.arena.grad) and torch.equal(tr.arena.param, tr2.arena.param)


def test_batched_layer3_of_the_final_passes_equals_the_per_pass_runs(pkg, gpu):
    """det_attack_algo.BATCH_LAYER3: the three final passes that reach the backbone run layer3 once on their concatenated feature maps
    (frozen BatchNorm: rows do not know their batch).  Against per-pass layer3 from the same state: the iteration's eight losses, the
    adversarial tensors and every input-side quantity bit-equal (one K order in every tiled variant); the parameter gradients of layer3
    are summed inside one reduction instead of three accumulations: fp32 order only."""
    g = _golden_for("align")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    da = pkg.det_attack_algo
    old, res = da.BATCH_LAYER3, {}
    try:
        for on in (True, False):
            da.BATCH_LAYER3 = on
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            tr = pkg.det_trainer.DetTrainer(m)
            torch.manual_seed(5)
            r = tr.step(images, bboxes, labels)
            torch.cuda.synchronize()
            res[on] = (r["losses"].clone(), r["adv_image"].clone(), r["adv3"].clone(), tr.arena.grad.clone())
    finally:
        da.BATCH_LAYER3 = old
    for k in range(3):
        assert torch.equal(res[True][k], res[False][k]), k
    ga, gb = res[True][3], res[False][3]
    assert float((ga - gb).norm() / gb.norm()) < 1e-5


def test_anchor_label_cache_follows_the_ground_truth(pkg, gpu):
    """det_model.ANCHOR_LABEL_CACHE: the anchors' labels / candidate lists are remembered per (anchor grid, ground-truth tensor address,
    version, shape) and the entry keeps that tensor alive: the same tensor again is a hit (the same objects), an in-place edit or another
    tensor is a miss with the labels of ITS boxes."""
    g = _golden_for("align")
    m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
    rpn = m.rpn
    images, bboxes = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["bboxes"]).to(gpu)
    feats = torch.zeros(1, 1024, 38, 57, device=gpu)
    anchors, iw, ih = m._anchors(feats, images.shape)
    obj, tr = torch.zeros(1, anchors.shape[1], 2, device=gpu), torch.zeros(1, anchors.shape[1], 4, device=gpu)
    a = rpn._losses(obj, tr, anchors, bboxes, iw, ih, pending=True)
    b = rpn._losses(obj, tr, anchors, bboxes, iw, ih, pending=True)
    assert a[0] is b[0] and a[1] is b[1]                                  # hit: the same list / label tensors
    moved = bboxes.clone()
    moved[..., [0, 2]] += 200.0
    c = rpn._losses(obj, tr, anchors, moved, iw, ih, pending=True)
    assert c[1] is not a[1] and not torch.equal(c[1], a[1])              # another tensor: its own labels
    want, _ = pkg.det_ops.box_assign(a[3], moved, "anchor", 0.3, 0.7)
    assert torch.equal(c[1], want)
    moved[..., [0, 2]] -= 200.0                                          # in-place edit: the version moves, the entry does not answer
    d = rpn._losses(obj, tr, anchors, moved, iw, ih, pending=True)
    assert d[1] is not c[1] and torch.equal(d[1], a[1])


def test_batched_feature_pgd_tails_equal_the_one_by_one_runs(pkg, gpu):
    """det_attack_algo.BATCH_PGD_TAILS: the three one-step feature PGDs (:84-88) run their RPN + ROI-head tails on the clean conv4 map as
    ONE forward_heads_many call and one backward of the losses' sum.  Every kernel on the way computes a row (a region, a pixel) from
    that row alone, the sampling draws keep their order: the three adversarial feature maps, everything made from them and the
    iteration's parameters are the same bits as with one call per PGD."""
    g = _golden_for("align")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    da = pkg.det_attack_algo
    old, res = da.BATCH_PGD_TAILS, {}
    try:
        for on in (True, False):
            da.BATCH_PGD_TAILS = on
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            tr = pkg.det_trainer.DetTrainer(m)
            torch.manual_seed(5)
            r = tr.step(images, bboxes, labels)
            torch.cuda.synchronize()
            res[on] = (r["adv1"].clone(), r["adv2"].clone(), r["adv3"].clone(), r["losses"].clone(), tr.arena.param.clone(), torch.get_rng_state().clone())
    finally:
        da.BATCH_PGD_TAILS = old
    for k in range(6):
        assert torch.equal(res[True][k], res[False][k]), k
    assert float((res[True][2] - images.new_zeros(1)).abs().sum()) > 0


def test_batched_roi_head_of_the_final_passes_equals_the_per_pass_runs(pkg, gpu):
    """det_attack_algo.BATCH_ROI_HEAD: the seven final passes in front of the ROI-tail pass run their ROI heads (ROIAlign, layer4, the
    two Linear layers) once on all passes' sampled regions (Model.forward_heads_many).  Same state, same host draws in the same order:
    the adversarial tensors bit-equal (they are made before), the sampled regions the same, the eight losses bit-equal (every forward
    kernel computes a row from that row alone — the small Linear kernels split their reduction per chunk for every row count), the
    parameter gradients to fp32 summation order (they sum over the passes inside one reduction)."""
    g = _golden_for("align")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    da = pkg.det_attack_algo
    old, res = da.BATCH_ROI_HEAD, {}
    try:
        for on in (True, False):
            da.BATCH_ROI_HEAD = on
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            tr = pkg.det_trainer.DetTrainer(m, segmented=True)          # (the two-part backward of the data-parallel path as well)
            torch.manual_seed(5)
            r = tr.step(images, bboxes, labels)
            torch.cuda.synchronize()
            res[on] = (r["losses"].clone(), r["adv_image"].clone(), r["adv3"].clone(), tr.arena.grad.clone(), torch.get_rng_state().clone())
    finally:
        da.BATCH_ROI_HEAD = old
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])
    assert torch.equal(res[True][4], res[False][4])                      # the host generator has made the same draws
    assert torch.equal(res[True][0], res[False][0])                      # the eight losses: the same bits (every forward kernel is row-wise)
    ga, gb = res[True][3], res[False][3]
    assert float((ga - gb).norm() / gb.norm()) < 2e-3                    # (bf16 activations' gradients through layer4: rounding of sums in another order)


def test_backbone_stage_graphs_equal_eager_launches(pkg, gpu):
    """det_model._StageGraphs (AFAN_DET_GRAPHS=1: the backbone's stages replayed from hipGraphs forward and backward, instances with
    private pools) against the eager stage nodes over four iterations from the same state: the same launches on the same data —
    identical losses and parameters, iteration by iteration (the whole iteration is deterministic)."""
    g = _golden_for("align")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    sg = pkg.det_model._StageGraphs
    old, res = sg.ON, {}
    da = pkg.det_attack_algo
    old_batch, da.BATCH_LAYER3 = da.BATCH_LAYER3, False      # (the graph instances are captured per pass: the eager side runs per pass too)
    try:
        for on in (False, True):
            sg.ON = on
            sg.cache.clear(), sg.warm.clear()
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            tr = pkg.det_trainer.DetTrainer(m)
            torch.manual_seed(5)
            losses = []
            for _ in range(4):
                losses.append(tr.step(images, bboxes, labels)["losses"].clone())
            torch.cuda.synchronize()
            res[on] = (torch.stack(losses), tr.arena.param.clone(), len(sg.cache))
    finally:
        sg.ON = old
        da.BATCH_LAYER3 = old_batch
        sg.cache.clear(), sg.warm.clear()
    assert res[True][2] > 0 and res[False][2] == 0                       # stages WERE captured and replayed
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def test_head_features_equal_the_three_head_forwards(pkg, gpu):
    """Model.head_features: the three `flag: 'head'` forwards of train_aug_sat_muti_advt.py:78-80 as one pass without an autograd
    graph — bit-identical feature maps (frozen BatchNorm, deterministic kernels)."""
    g = golden("det_frcnn_r101")
    m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
    x = torch.from_numpy(g["images"]).to(gpu)
    one = m.head_features(x, (1, 2, 3))
    for i, f in zip((1, 2, 3), one):
        ref = m.train().forward({"x": x, "adv": None, "out_idx": i, "flag": "head"}).detach()
        assert not f.requires_grad and torch.equal(f, ref), i
    # ... and as a by-product of the clean ROI-head pass (:81), which runs the same backbone on the same images WITH a graph:
    # det_train_phases takes them from there (dict key "collect")
    bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("bboxes", "labels"))
    col = {}
    torch.manual_seed(3)
    rr = m.train().forward({"x": x, "adv": None, "out_idx": "roi_head", "flag": "clean", "collect": col}, bboxes, labels)
    assert sorted(k for k in col if isinstance(k, int)) == [1, 2, 3] and "roi_output_dict" in rr
    for i, f in zip((1, 2, 3), one):
        assert not col[i].requires_grad and torch.equal(col[i], f), i


def test_frozen_bottleneck_node_equals_layer_by_layer_path(pkg, gpu):
    """det_model._FrozenBlockFn (a frozen-BatchNorm bottleneck as one autograd node: same launches, one Function.apply) against
    the layer-by-layer modules on the same bf16 kernels: a training forward's four losses and every parameter gradient.  The
    only arithmetic difference is the identity shortcut's gradient, added in the dgrad epilogue instead of by a bf16 add."""
    g = golden("det_frcnn_r101")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    fn = pkg.det_model._FrozenBlockFn
    res, old = {}, fn.ON
    try:
        for on in (True, False):
            fn.ON = on
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            arena = pkg.arena.ParamArena(m, skip=())
            torch.manual_seed(11)                                   # the host randperm draws of the RPN / ROI sampling
            x = images.clone().requires_grad_(True)
            before = pkg.ops.CALLS["conv_fwd"]
            losses = m.train().forward({"x": x, "adv": None, "out_idx": 0, "flag": "clean"}, bboxes, labels)
            convs = pkg.ops.CALLS["conv_fwd"] - before
            arena.zero_grad()
            sum(l.mean() for l in losses).backward()
            torch.cuda.synchronize()
            res[on] = ([float(l.detach().mean()) for l in losses], {n: p.grad.float().clone() for n, p in zip(arena.names, arena.params)},
                       x.grad.float().clone(), convs)
    finally:
        fn.ON = old
    a, b = res[True], res[False]
    assert a[3] == b[3] > 100                                       # the same convolution launches either way
    np.testing.assert_allclose(a[0], b[0], rtol=2e-3, atol=2e-4)
    assert _rel(a[2].cpu().numpy(), b[2].cpu().numpy()) < 3e-2      # image gradient through 33 bf16 blocks
    worst = max((_rel(a[1][n].cpu().numpy(), b[1][n].cpu().numpy()), n) for n in b[1] if float(b[1][n].abs().max()) > 0)
    assert worst[0] < 3e-2, worst


def test_stage_backward_chain_equals_block_by_block(pkg, gpu):
    """det_model._stage_backward hands block i's first backward step (the gradient masked by its output, once plain and once times
    the last BatchNorm's alpha) to the epilogue of block i + 1's last input-gradient launch (afan_frozen_bottleneck_bwd_chain);
    with one autograd node per block (`_FrozenStageFn.ON = False`) every block issues its own launch for it.  Same bits: the four
    losses of a training forward, the image gradient and every parameter gradient."""
    g = golden("det_frcnn_r101")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    fn = pkg.det_model._FrozenStageFn
    res, old = {}, fn.ON
    try:
        for on in (True, False):
            fn.ON = on
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            arena = pkg.arena.ParamArena(m, skip=())
            torch.manual_seed(11)                                   # the host randperm draws of the RPN / ROI sampling
            x = images.clone().requires_grad_(True)
            losses = m.train().forward({"x": x, "adv": None, "out_idx": 0, "flag": "clean"}, bboxes, labels)
            arena.zero_grad()
            sum(l.mean() for l in losses).backward()
            torch.cuda.synchronize()
            res[on] = ([l.detach().clone() for l in losses], {n: p.grad.clone() for n, p in zip(arena.names, arena.params)}, x.grad.clone())
    finally:
        fn.ON = old
    a, b = res[True], res[False]
    assert all(torch.equal(u, v) for u, v in zip(a[0], b[0]))
    assert torch.equal(a[2], b[2])
    assert a[1].keys() == b[1].keys()
    for n in a[1]:
        assert torch.equal(a[1][n], b[1][n]), n


def test_feature_pgd_folded_into_the_clean_pass_changes_no_bit(pkg, gpu):
    """det_train_phases takes the two one-step feature PGDs at out_idx 1 / 2 (train_aug_sat_muti_advt.py:84-85: no random start,
    so their one forward runs the rest of the backbone on the CLEAN feature map) from the clean ROI-head pass's own activations:
    only the part behind the backbone runs again, the gradient goes back through the stored activations with the
    input-gradient-only launches (det_attack_algo._pgd1_from_clean, det_model.stage_input_gradient).  Against PGD() as written
    (AFAN_DET_FOLD_PGD=0), same seed: the two adversarial feature maps, every loss, the gradient arena — the same bits; and the
    fold really ran (fewer convolution launches)."""
    g = _golden_for("align")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    res, old = {}, os.environ.get("AFAN_DET_FOLD_PGD")
    try:
        for fold in ("1", "0"):
            os.environ["AFAN_DET_FOLD_PGD"] = fold
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            tr = pkg.det_trainer.DetTrainer(m)
            torch.manual_seed(102)
            before = pkg.ops.CALLS["conv_fwd"]
            r = tr.step(images, bboxes, labels)
            torch.cuda.synchronize()
            res[fold] = (r, tr.arena.grad.clone(), pkg.ops.CALLS["conv_fwd"] - before)
    finally:
        if old is None:
            os.environ.pop("AFAN_DET_FOLD_PGD", None)
        else:
            os.environ["AFAN_DET_FOLD_PGD"] = old
    a, b = res["1"], res["0"]
    assert a[2] < b[2] - 100                              # stages 2 + 3 and stage 3 once less: ~150 convolution launches
    for k in ("adv1", "adv2", "adv3", "adv_image", "losses", "loss"):
        assert torch.equal(a[0][k], b[0][k]), k
    assert torch.equal(a[1], b[1])


def test_stage_activations_are_freed_with_their_graph(pkg, gpu):
    """A one-node stage leaves references to its activations on its output tensor (det_model.stage_input_gradient reuses them).  The
    output itself must not be among them: `x.__dict__ -> tuple -> x` is a cycle only the cyclic collector frees, and the caching
    allocator does not run it when memory is short.  With the collector OFF: allocated memory is back at its baseline as soon as
    the graph and the output are dropped; and the stored tuple + the output still give the input gradient the backward gives."""
    import gc
    g = golden("det_frcnn_r101")
    m = _build(pkg, g, gpu, torch.bfloat16, True, "pooling")
    arena = pkg.arena.ParamArena(m, skip=())                 # (the one-call block forms read the arena's transposed weights)
    assert arena.numel > 0
    stage = m.features.layer2
    x = (torch.randn(1, 256, 40, 56, device=gpu) * 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    for _ in range(2):                                   # workspaces and launch plans exist after the first pass
        o = pkg.det_model._run_stage(stage, x.clone().requires_grad_(True))
        o.backward(torch.ones_like(o))
        del o
    gc.collect()
    torch.cuda.synchronize()
    was = gc.isenabled()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated(gpu)
        xin = x.clone().requires_grad_(True)
        out = pkg.det_model._run_stage(stage, xin)
        assert hasattr(out, "_afan_stage_saved") and not any(t is out for t in out._afan_stage_saved)
        go = torch.ones_like(out)
        dx2 = pkg.det_model.stage_input_gradient(stage, out, go)
        out.backward(go)
        assert dx2 is not None and torch.equal(dx2, xin.grad)
        held = torch.cuda.memory_allocated(gpu)
        del out, go, dx2, xin
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated(gpu)
    finally:
        if was:
            gc.enable()
    assert held > base + (1 << 20) and after <= base, (base, held, after)


def test_merged_sampling_reads_and_noise_ahead_change_no_bit(pkg, gpu):
    """Round 5, host side of the Detection iteration: (a) the ROI head's sampling (model.py:256-282) queues its label / list launches
    on the padded proposals before the RPN's one host read and takes its two list lengths from it (Model.MERGE_READS: 16 reads per
    iteration fewer); (b) the image PGD's noise for iteration i + 1 is drawn behind iteration i's backward (DetTrainer(noise_ahead)).
    Both leave the host generator's stream and every number alone: two iterations from the same state and seed, each form against the
    plain one — losses, adversarial image and parameters identical bit for bit."""
    g = _golden_for("align")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    M = pkg.det_model.Model
    old = M.MERGE_READS
    res = {}
    try:
        for merge, ahead in ((False, False), (True, False), (True, True)):
            M.MERGE_READS = merge
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            tr = pkg.det_trainer.DetTrainer(m, noise_ahead=ahead)
            torch.manual_seed(77)
            outs = []
            for _ in range(2):
                r = tr.step(images, bboxes, labels)
                outs.append((r["losses"].clone(), r["adv_image"].clone()))
            torch.cuda.synchronize()
            res[(merge, ahead)] = (outs, tr.arena.param.clone())
    finally:
        M.MERGE_READS = old
    base = res[(False, False)]
    for key in ((True, False), (True, True)):
        for (la, ia), (lb, ib) in zip(res[key][0], base[0]):
            assert torch.equal(la, lb) and torch.equal(ia, ib), key
        assert torch.equal(res[key][1], base[1]), key


def test_fused_loss_sums_change_no_bit(pkg, gpu):
    """`loss1.mean() + loss2.mean() + loss3.mean() + loss4.mean()` (train_aug_sat_muti_advt.py:21-27, attack_algo.py:62) as ONE launch
    for one-image loss vectors (det_ops.sum_of_means -> afan_sum_scalars_f32; backward: no launch) against the plain expression: two
    iterations from the same state and seed — losses, adversarial image and parameters identical bit for bit; and the values of the
    function itself on two-image vectors (the plain path) and on one-image ones (both paths)."""
    dops = pkg.det_ops
    a, b, c, d = (torch.randn(1, device=gpu, requires_grad=True) for _ in range(4))
    fused = dops.sum_of_means(a, b, c, d)
    plain = a.mean() + b.mean() + c.mean() + d.mean()
    assert torch.equal(fused.detach(), plain.detach()) and fused.dim() == 0
    ga = torch.autograd.grad(fused * 3.0, (a, b, c, d))
    assert all(torch.equal(x_, torch.full((1,), 3.0, device=gpu)) for x_ in ga)
    assert torch.equal(dops.sum_of_means(a, b).detach(), (a.mean() + b.mean()).detach())
    v = [torch.randn(2, device=gpu) for _ in range(4)]
    assert torch.equal(dops.sum_of_means(*v), v[0].mean() + v[1].mean() + v[2].mean() + v[3].mean())
    g = _golden_for("align")
    images, bboxes, labels = (torch.from_numpy(g[k]).to(gpu) for k in ("images", "bboxes", "labels"))
    old, res = dops.FUSED_LOSS_SUM, {}
    try:
        for on in (False, True):
            dops.FUSED_LOSS_SUM = on
            m = _build(pkg, g, gpu, torch.bfloat16, True, "align")
            tr = pkg.det_trainer.DetTrainer(m)
            torch.manual_seed(77)
            outs = []
            for _ in range(2):
                r = tr.step(images, bboxes, labels)
                outs.append((r["losses"].clone(), r["adv_image"].clone(), r["loss"].clone()))
            torch.cuda.synchronize()
            res[on] = (outs, tr.arena.param.clone())
    finally:
        dops.FUSED_LOSS_SUM = old
    for x_, y_ in zip(res[True][0], res[False][0]):
        assert all(torch.equal(p_, q_) for p_, q_ in zip(x_, y_))
    assert torch.equal(res[True][1], res[False][1])
