"""DeepLabv3+ (ResNet-101, output stride 16) through one Segmentation A-FAN iteration on the GPU — BASELINE config 4's
network — against goldens produced by the reference's OWN network and attack_algo functions (oracle/gen_golden.py:
Segmentation/network/ + main_aug_final.py:158-232, dropout off), fp32 in both layouts (loss <= 1e-4 relative) and the bf16
channels-last product configuration (every convolution on the library's kernels)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import assert_close_frac, golden

pytestmark = pytest.mark.gpu


def _build(pkg, g, dtype, nhwc, gpu, **kw):
    dl = pkg.deeplab
    torch.manual_seed(int(g["seed"]))
    model = dl.deeplabv3plus_resnet101(num_classes=21, output_stride=16)
    model.classifier.aspp.project[3].p = 0.0            # as in the golden run
    if float(g["damp"]) != 1.0:                         # the contractive variant: bn3.weight scaled (weights are data)
        for m in model.backbone.modules():
            if isinstance(m, dl.Bottleneck):
                m.bn3.weight.data.mul_(float(g["damp"]))
    ck0 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in model.state_dict().values()])
    # seeded construction == the reference's own (keys and values; f64 checksums: summation order differs between hosts)
    np.testing.assert_allclose(ck0, g["ck0"], rtol=1e-12, atol=1e-9)
    assert list(model.state_dict().keys()) == [str(k) for k in g["keys"]]
    model.set_compute_dtype(dtype).set_channels_last(nhwc).to(gpu).train()
    steps, se_idx, mix_sd = [int(v) for v in g["meta"]]
    gamma_se, gamma_sd, eps = [float(v) for v in g["gammas"]]
    tr = pkg.seg_trainer.SegTrainer(model, nn.CrossEntropyLoss(ignore_index=255, reduction="mean"), steps=steps, eps=eps,
                                    gamma_se=gamma_se, gamma_sd=gamma_sd, pertub_idx_se=se_idx, pertub_idx_sd=str(g["sd_idx"]),
                                    mix_layer=str(g["mix_layer"]), mix_sd=bool(mix_sd), lr=float(g["lr"]), weight_decay=1e-4,
                                    **kw)
    return model, tr


def _cks(model):
    return np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in model.state_dict().values()])


def _grad_check(pkg, tr, g, rel_norm, rel_tensor, rel_tensor_backbone=None):
    """parameter gradients left in the arena by the step against the reference's: every tensor's norm, a few tensors whole
    (bf16: ReLU masks of activations within bf16 rounding of zero flip — ~5 % element-wise error per block in the input
    gradient, a random walk over the 33 blocks a backbone gradient has crossed; norms are unaffected)"""
    names = [str(k) for k in g["param_names"]]
    assert tr.arena.names == names
    got = np.array([float(tr.arena.view(tr.arena.grad, i).double().norm()) for i in range(len(names))])
    ref = g["grad_norms"]
    bad = np.abs(got - ref) > rel_norm * ref + 1e-7 * ref.max()
    assert bad.mean() <= 0.02, [(names[i], got[i], ref[i]) for i in np.nonzero(bad)[0][:8]]
    for k in g.files:
        if k.startswith("grad/"):
            i = names.index(k[5:])
            a = tr.arena.view(tr.arena.grad, i).float().cpu().numpy()
            e = np.linalg.norm((a - g[k]).ravel()) / max(np.linalg.norm(g[k].ravel()), 1e-12)
            lim = rel_tensor_backbone if (rel_tensor_backbone is not None and k.startswith("grad/backbone.")) else rel_tensor
            assert e <= lim, (k, e)


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("case", ["seg_dl101_aspp_k1", "seg_dl101_concat_k3", "seg_dl101_aspp_k3_damped"])
def test_deeplab_step_fp32_matches_reference(pkg, gpu, case, nhwc):
    torch.backends.cudnn.deterministic = True
    g = golden(case)
    model, tr = _build(pkg, g, torch.float32, nhwc, gpu, use_graph=False)
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    before = dict(pkg.ops.CALLS)
    r = tr.step(images, labels)
    torch.cuda.synchronize()
    assert pkg.ops.CALLS["vendor_conv"] == 0 and pkg.ops.CALLS["conv_general"] - before["conv_general"] > 300   # the library's own fp32 convolutions
    loss = float(g["loss"])
    assert abs(float(r["loss"]) - loss) <= 1e-4 * max(1.0, abs(loss)), (float(r["loss"]), loss)
    np.testing.assert_allclose(r["losses"].cpu().numpy(), g["losses"], rtol=1e-4, atol=1e-4)
    # (1024 x 9 x 9 map after 91 fp32 convolutions — sequential fmaf chains of up to 4608 terms — and as many batch-of-162 BatchNorms)
    np.testing.assert_allclose(r["fm_se"].float().cpu().numpy(), g["fm_se"], rtol=2e-3, atol=6e-3)
    np.testing.assert_allclose(r["out_clean"][:, :, ::4, ::4].cpu().numpy(), g["out_clean_sub"], rtol=2e-3, atol=1e-2)
    if float(g["damp"]) != 1.0:        # contractive network: tight bounds on features, logits and every parameter gradient
        np.testing.assert_allclose(r["fm_se"].float().cpu().numpy(), g["fm_se"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(r["out_clean"][:, :, ::4, ::4].cpu().numpy(), g["out_clean_sub"], rtol=1e-3, atol=2e-4)
        # Backbone tensors in channels-last: 3e-2 (measured 2.1e-2), NCHW 1e-2 (measured 1.7e-3).  Every convolution is the
        # library's f32-MFMA kernel, bit-identical between the two layouts (one k-ordered fmaf chain), so the difference is
        # BatchNorm rounding alone: ~3e-6 of forward noise flips ~6e-6 of a layer's ReLU-mask bits, which moves that layer's
        # gradient by sqrt(6e-6) = 2.6e-3 in l2 (tools/diag_bn_trace_bwd.py), and sign() turns that into 2.5 % (NHWC) / 0.7 %
        # (NCHW) of the K = 3 perturbation elements off the reference's; the reference's own fp32 run is 0.5 % off a float64
        # run of the same iteration (tools/diag_seg_f64.py).  The head's tensors (upstream of no sign()) stay at 1e-2.
        _grad_check(pkg, tr, g, 1e-2, 1e-2, rel_tensor_backbone=3e-2 if nhwc else None)
    steps = int(g["meta"][0])
    gam = float(g["gammas"][0]) / 255
    adv = r["adv_se"].float().cpu().numpy()
    d = adv - r["fm_se"].float().cpu().numpy()
    assert np.abs(d).max() <= steps * gam * (1 + 1e-5) + 4e-6            # on the sign grid (+ an ulp of a feature of ~30), at most K steps away
    # The perturbation itself, in units of gamma (integers -K..K), against the reference's.  sign() flips an element by
    # 2 where a gradient sits within rounding distance of zero; on the freshly initialised network the
    # gradient reaching layer3's output has crossed the chaotic layer4 + ASPP in vendor-fp32 vs CPU-fp32 arithmetic.
    k_got, k_ref = np.rint(d / gam), np.rint((g["adv_se"] - g["fm_se"]) / gam)
    # bound: the reference against itself (tests/golden/ref_noise_floor.npz: its own SE PGD re-run in float64, ATen-native fp32,
    # channels-last fp32, on the transposed problem, and on images carrying 1e-6 relative noise — the size of the rounding noise the
    # ~90 fp32 layers of the head accumulate in any implementation — differs from its baseline on `floor` of the elements), times two
    fl = golden("ref_noise_floor")
    floor, floor_a = float(fl[case + "/floor"]), float(fl[case + "/floor_arith"])
    flips = 1.0 - float((k_got == k_ref).mean())
    print(f"PARITY {case} [fp32 {'NHWC' if nhwc else 'NCHW'}]: perturbation elements off the reference's {flips:.5f}   "
          f"reference-vs-reference floor {floor:.5f} (arithmetic variants only {floor_a:.5f})   bound {max(2 * floor, 1e-4):.5f}")
    assert flips <= max(2.0 * floor, 1e-4), (flips, floor)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("sd1/") and k.endswith("num_batches_tracked"):
            assert int(sd[k[4:]]) == int(g[k]), k                        # BatchNorm side effects: same number of updates
    np.testing.assert_allclose(sd["backbone.bn1.running_mean"].cpu().numpy(), g["sd1/backbone.bn1.running_mean"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["backbone.bn1.running_var"].cpu().numpy(), g["sd1/backbone.bn1.running_var"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["backbone.layer4.0.bn1.running_mean"].cpu().numpy(),
                               g["sd1/backbone.layer4.0.bn1.running_mean"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd["classifier.classifier.3.bias"].cpu().numpy(), g["sd1/classifier.classifier.3.bias"],
                               rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(sd["classifier.classifier.3.weight"].cpu().numpy(), g["sd1/classifier.classifier.3.weight"],
                               rtol=1e-3, atol=2e-6)
    # whole state_dict after the SGD step (two learning-rate groups): per-tensor checksums.  On the freshly initialised
    # network the backbone's gradients are chaotic (a 3 % difference in conv1's update between vendor-fp32 and CPU-fp32
    # arithmetic): there the head's tensors are held tight and the backbone's loosely; the contractive case holds all tight.
    ck, keys = _cks(model), [str(k) for k in g["keys"]]
    head = np.array([k.startswith("classifier.") for k in keys])
    np.testing.assert_allclose(ck[head], g["ck1"][head], rtol=2e-4, atol=5e-3 if float(g["damp"]) != 1.0 else 3e-2)
    if float(g["damp"]) != 1.0:
        np.testing.assert_allclose(ck[~head], g["ck1"][~head], rtol=2e-4, atol=5e-3)
    else:
        np.testing.assert_allclose(ck[~head][:, 1], g["ck1"][~head][:, 1], rtol=5e-2, atol=5e-3)


@pytest.mark.parametrize("nhwc", [False, True])
def test_deeplab_contractive_flip_onset(pkg, gpu, nhwc):
    """Where the product's perturbation starts to differ from the reference's on the contractive DeepLab golden: the SE feature PGD
    run with K = 1, 2, 3 against the reference baseline's perturbation after 1, 2, 3 steps (ref_noise_floor.npz
    `base_dk_per_step`), next to the reference-vs-reference flip fraction after the same number of steps.  Step 1 is arithmetic
    noise at its source (one gradient from the clean feature map): held to max(2 x floor, 1e-4) in BOTH layouts; the later steps
    compound and are reported.  The function is BISTABLE at this point: under 1e-6 relative input noise the reference itself lands
    on a 0.05 % or a 0.55 % branch after one step (ref_noise_floor.npz in1..in4), and so do both layouts of the product, draw by
    draw on the same branch as each other (tools/diag_dl_chaos.py, profiles/r04_dl_flip_bistability.txt)."""
    case = "seg_dl101_aspp_k3_damped"
    g, fl = golden(case), golden("ref_noise_floor")
    base = fl[case + "/base_dk_per_step"]
    arith = ("f64", "nomkldnn", "cl", "t", "t_nomkldnn", "t_cl")
    noise = ("in1", "in2", "in3", "in4")
    floors_a = np.max(np.stack([fl[f"{case}/{k}/per_step"] for k in arith]), axis=0)
    floors = np.max(np.stack([fl[f"{case}/{k}/per_step"] for k in arith + noise]), axis=0)
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    gam = float(g["gammas"][0]) / 255
    got = []
    for K in (1, 2, 3):
        model, tr = _build(pkg, g, torch.float32, nhwc, gpu, use_graph=False)
        tr.kw["steps"] = K
        r = tr.step(images, labels)
        k_got = np.rint((r["adv_se"].float() - r["fm_se"].float()).cpu().numpy() / gam).astype(np.int8)
        got.append(float((k_got != base[K - 1]).mean()))
        print(f"PARITY-STEP {case} [fp32 {'NHWC' if nhwc else 'NCHW'}] after {K} step(s): product {got[-1]:.5f}   "
              f"reference-vs-reference {floors[K - 1]:.5f} (arithmetic variants only {floors_a[K - 1]:.5f})")
    assert got[0] <= max(2.0 * floors[0], 1e-4), (got, floors.tolist())


@pytest.mark.parametrize("case", ["seg_dl101_aspp_k1", "seg_dl101_concat_k3", "seg_dl101_aspp_k3_damped"])
def test_deeplab_step_bf16_runs_on_the_library_kernels(pkg, gpu, case):
    g = golden(case)
    model, tr = _build(pkg, g, torch.bfloat16, True, gpu, use_graph=False)
    assert pkg.resnet_s.vendor_convs(model) == [], pkg.resnet_s.vendor_convs(model)
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    before = dict(pkg.ops.CALLS)
    r = tr.step(images, labels)
    torch.cuda.synchronize()
    ran = {k: pkg.ops.CALLS[k] - before[k] for k in before}
    assert ran["vendor_conv"] == 0 and ran["conv_fwd"] > 100 and ran["conv_dgrad"] > 100 and ran["conv_wgrad"] > 100, ran
    loss = float(g["loss"])
    # The freshly initialised network is chaotic: + 1e-3 of noise on the images moves the bf16 loss by up to 0.036 from the
    # fp32 golden (six draws, two builds of the library: profiles/r03j_deeplab_loss_spread.txt), so its bound only says
    # "same regime"; on the contractive network the same draws stay within 1.2e-3 and the bound means something.
    tol = 5e-3 if float(g["damp"]) != 1.0 else 6e-2
    assert abs(float(r["loss"]) - loss) <= tol, (float(r["loss"]), loss)
    np.testing.assert_allclose(r["losses"].cpu().numpy(), g["losses"], rtol=0, atol=tol)
    if float(g["damp"]) != 1.0:
        # On the contractive network the bf16 run tracks the fp32 reference end to end (a freshly initialised 101-layer
        # BatchNorm network amplifies ANY 0.3 % perturbation to ~60 % by layer3 — tools/diag_deeplab_layers.py shows
        # 0.5 % per block when every block is fed the reference's input): features, loss, every parameter gradient.
        fm = r["fm_se"].float().cpu().numpy()
        assert np.linalg.norm((fm - g["fm_se"]).ravel()) <= 0.04 * np.linalg.norm(g["fm_se"].ravel())
        assert abs(float(r["loss"]) - loss) <= 5e-3, (float(r["loss"]), loss)
        _grad_check(pkg, tr, g, 0.12, 0.2, 0.5)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("sd1/") and k.endswith("num_batches_tracked"):
            assert int(sd[k[4:]]) == int(g[k]), k
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


def test_deeplab_graph_replay_equals_eager(pkg, gpu):
    """The captured iteration (hipGraph) reproduces the eager one: two trainers from the same seed, three iterations."""
    g = golden("seg_dl101_aspp_k1")
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    losses = {}
    for mode in ("eager", "graph"):
        model, tr = _build(pkg, g, torch.bfloat16, True, gpu, use_graph=(mode == "graph"), graph_warmup=1)
        out = []
        for _ in range(3):
            out.append(float(tr.step(images, labels)["loss"]))
            tr.scheduler.step()
        losses[mode] = out
        if mode == "graph":
            assert tr._graph is not None, tr._graph_failed
    np.testing.assert_allclose(losses["graph"], losses["eager"], rtol=0, atol=2e-3)


@pytest.mark.parametrize("case", ["seg_dl101_aspp_k1", "seg_dl101_concat_k3"])
def test_deeplab_dropout_draws_per_iteration_equal_the_reference(pkg, gpu, case):
    """The reference draws a fresh ASPP-dropout mask in every model(...) call that runs ASPP (_deeplab.py:185): the decoder-PGD
    input pass (main_aug_final.py:167), each of the K SE-PGD passes (attack_algo.py:50), the clean forward (:193) and the two
    perturbed SE forwards (:197-203) — K + 4 draws per iteration; the SD passes start behind ASPP and draw nothing.  With
    p > 0 the step must keep that count under every schedule (the first PGD passes may NOT be folded into the clean pass:
    they would share its mask); with p = 0 the folded schedule is value-identical and is taken."""
    g = golden(case)
    K = int(g["meta"][0])
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    real = pkg.ops.dropout
    for p_drop, fold in ((0.1, None), (0.1, False), (0.0, None)):
        model, tr = _build(pkg, g, torch.bfloat16, True, gpu, use_graph=False, fold_clean=fold)
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = p_drop
        draws = [0]

        def counting(x, p, mask=None, used=None):
            if used is None:
                draws[0] += 1
            return real(x, p, mask, used)
        pkg.ops.dropout = counting
        try:
            r = tr.step(images, labels)
        finally:
            pkg.ops.dropout = real
        if p_drop > 0:
            assert draws[0] == K + 4, (p_drop, fold, draws[0])
            assert r["fold_pgd0"] is False and r["fold_clean"] is (fold is None)
        else:
            assert draws[0] == 0 and r["fold_pgd0"] is True
        assert np.isfinite(float(r["loss"]))


def test_deeplab_full_size_properties(pkg, gpu):
    """BASELINE configs[3]'s per-GPU share at FULL size (DeepLabv3+ ResNet-101, 2 x 3 x 513 x 513, output stride 16, SE point
    layer3 = 1024 x 33 x 33, SD point aspp, K = 3, the reference's Dropout(0.1)) on the product path: size-independent
    properties of the iteration, and the hipGraph replay against eager launches from the same seed."""
    K, eps = 3, 2.0
    g = torch.Generator().manual_seed(5)
    images = torch.rand(2, 3, 513, 513, generator=g).to(gpu)
    labels = torch.randint(0, 21, (2, 513, 513), generator=g)
    labels[torch.rand(2, 513, 513, generator=g) < 0.05] = 255
    labels = labels.to(gpu)
    out = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(3)
        model = pkg.deeplab.deeplabv3plus_resnet101(num_classes=21, output_stride=16)
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0 if mode != "dropout" else 0.1
        model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
        tr = pkg.seg_trainer.SegTrainer(model, steps=K, eps=eps, gamma_se=0.5, gamma_sd=0.5, pertub_idx_se=3, pertub_idx_sd="aspp",
                                        mix_layer="11", mix_sd=True, lr=0.01, use_graph=(mode == "graph"), graph_warmup=1)
        assert not pkg.resnet_s.general_convs(model)
        before = dict(pkg.ops.CALLS)
        losses = []
        for _ in range(3):
            r = tr.step(images, labels)
            tr.scheduler.step()
            losses.append(float(r["loss"]))
        if mode == "graph":
            assert tr._graph is not None, tr._graph_failed
        assert pkg.ops.CALLS["vendor_conv"] == 0 and pkg.ops.CALLS["conv_general"] == before["conv_general"]
        assert all(np.isfinite(v) for v in losses)
        fm, adv = r["fm_se"].float(), r["adv_se"].float()
        assert fm.shape == (2, 1024, 33, 33) and r["fm_sd"].shape == (2, 256, 33, 33)
        gam = np.float32(0.5 / 255)
        k = ((adv - fm) / gam).round()
        assert float(((adv - fm) / gam - k).abs().max()) < 2e-2 and float(k.abs().max()) <= K          # SE perturbation on the sign grid
        lw = r["losses"].float().cpu().numpy()
        assert abs(float(r["loss"]) - float(0.7 * lw[0] + 0.1 * (lw[1] + lw[2] + lw[3]))) < 1e-4        # main_aug_final.py:216
        # BatchNorm updates per iteration, in the reference's count: stem .. SE point 3 (:166, :167, :193); layer4 / ASPP
        # 2 + K + 2 (:167, :193, K SE-PGD passes, SE1, SE2); the decoder's 3x3 1 + 2K + 3 (:193, both PGD loops, three forwards)
        n_it = 3
        assert int(model.backbone.bn1.num_batches_tracked) == 3 * n_it
        assert int(model.backbone.layer3[22].bn3.num_batches_tracked) == 3 * n_it
        assert int(model.backbone.layer4[0].bn1.num_batches_tracked) == (K + 4) * n_it
        assert int(model.classifier.aspp.project[1].num_batches_tracked) == (K + 4) * n_it
        assert int(model.classifier.classifier[1].num_batches_tracked) == (2 * K + 4) * n_it
        for name, buf in model.named_buffers():
            assert torch.isfinite(buf.float()).all(), name
        out[mode] = losses
    np.testing.assert_allclose(out["graph"], out["eager"], rtol=0, atol=5e-3)


def test_deeplab_with_side_stream_weight_gradients_gives_up_no_barrier_and_changes_no_bit(pkg, gpu):
    """Round 5 regression at the iteration's level (profiles/r05j_dl101b8_barrier_timeout_trace.txt): 8 images, the weight gradients
    on the side stream (SegTrainer's own choice from 8 x 513^2 on: that size, because at 257^2 the weight-gradient workgroups are too
    short-lived to pin a CU's LDS and the fault does not show — tools/probe/side_stream_sensitivity.py with the fix switched off:
    no timeout at 257^2, a timeout at 513^2) — layer4's 276-workgroup launches are on the two-per-CU tile, the form that cannot share
    CUs with another kernel's long-lived workgroups.  Beside side-stream kernels the in-launch BatchNorm takes one workgroup per CU only (the others: two launches, the
    same bits): no grid barrier gives up, and losses, perturbations and parameters equal the iteration without the side stream."""
    ops = pkg.ops
    g = torch.Generator().manual_seed(9)
    images = torch.rand(8, 3, 513, 513, generator=g).to(gpu)
    labels = torch.randint(0, 21, (8, 513, 513), generator=g).to(gpu)
    res = {}
    for side in (False, True):
        torch.manual_seed(3)
        model = pkg.deeplab.deeplabv3plus_resnet101(num_classes=21, output_stride=16)
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
        tr = pkg.seg_trainer.SegTrainer(model, steps=2, eps=2.0, gamma_se=0.5, gamma_sd=0.5, pertub_idx_se=3, pertub_idx_sd="aspp",
                                        mix_layer="11", mix_sd=True, lr=0.01, use_graph=False, wgrad_stream=side)
        before = ops.CALLS["conv_bn_fused"]
        outs = []
        for _ in range(2):
            r = tr.step(images, labels)
            outs.append((r["loss"].clone(), r["adv_se"].clone(), r["adv_sd"].clone()))
        torch.cuda.synchronize()
        assert not ops.grid_barrier_error(gpu)
        assert side or ops.CALLS["conv_bn_fused"] > before      # (beside the side stream every launch of this size is beyond one workgroup per CU)
        res[side] = (outs, tr.arena.param.clone())
        del tr, model
    for a, b in zip(res[False][0], res[True][0]):
        for p_, q_ in zip(a, b):
            assert torch.equal(p_, q_)
    assert torch.equal(res[False][1], res[True][1])


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_phased_iteration_equals_the_plain_one(pkg, gpu, dropout):
    """The data-parallel schedule on ONE GPU (SegTrainer(segmented=True)): the joint backward in two parts with a yield in
    between (seg_attack_algo.seg_train_phases), captured as one hipGraph per part, the optimizer step after the last.
    Same kernels in the same order as the one-graph iteration: parameters after three iterations are bit-identical, and the
    tail's parameter range — what the trainer hands to the all-reduce at the yield — is final at the yield."""
    g = torch.Generator().manual_seed(5)
    images = torch.rand(2, 3, 129, 129, generator=g).to(gpu)
    labels = torch.randint(0, 21, (2, 129, 129), generator=g).to(gpu)
    res = {}
    for mode in ("plain", "phased_eager", "phased_graph"):
        torch.manual_seed(3)
        model = pkg.deeplab.deeplabv3plus_resnet50(num_classes=21, output_stride=16)
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = dropout
        model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
        tr = pkg.seg_trainer.SegTrainer(model, steps=2, lr=0.01, use_graph=(mode != "phased_eager"), graph_warmup=1,
                                        segmented=(mode != "plain"))
        torch.manual_seed(9)
        pkg.ops._dropout_state.clear()              # the device-side dropout generator is re-seeded from torch's CPU generator
        for _ in range(3):
            r = tr.step(images, labels)
            tr.scheduler.step()
        if mode != "phased_eager":
            assert tr._graph is not None, tr._graph_failed
            assert len(tr._pieces) == (2 if mode == "phased_graph" else 1)
            if mode == "phased_graph":
                assert [ph for _, ph in tr._pieces] == ["tail", None]
        res[mode] = (tr.arena.param.clone(), float(r["loss"]))
        if mode == "phased_eager":
            # at the yield the tail's gradient range is final: the rest of the backward leaves it alone
            lo, hi = tr._tail_range()
            bounds = tr.arena.offsets + [tr.arena.numel]
            assert hi == len(tr.arena.params) and 0.3 < (bounds[hi] - bounds[lo]) / tr.arena.numel < 0.8
            out, snap = {}, None
            for ph in pkg.seg_attack_algo.seg_train_phases(model, tr.optimizer, tr.criterion, images, labels, out, **tr.kw):
                assert ph == "tail"
                snap = tr.arena.grad[bounds[lo]:bounds[hi]].clone()
                head_before = tr.arena.grad[:bounds[lo]].clone()
            assert snap is not None and torch.equal(snap, tr.arena.grad[bounds[lo]:bounds[hi]])
            assert float(snap.abs().sum()) > 0
            assert not torch.equal(head_before, tr.arena.grad[:bounds[lo]])
    assert torch.equal(res["plain"][0], res["phased_eager"][0])
    assert torch.equal(res["plain"][0], res["phased_graph"][0])


def test_deeplab_checkpoint_interchange(pkg, orc, gpu):
    """state_dict round trip with the reference layout (oracle.SegDeepLabV3Plus has the reference's keys, verified against
    the reference's own network by tests/test_oracle_golden.py) and the two-group optimizer state_dict layout."""
    dl = pkg.deeplab
    torch.manual_seed(11)
    ref = orc.deeplabv3plus_resnet50(num_classes=21, output_stride=16)
    model = dl.deeplabv3plus_resnet50(num_classes=21, output_stride=16)
    model.load_state_dict(ref.state_dict())
    ref.load_state_dict(model.state_dict())
    model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    tr = pkg.seg_trainer.SegTrainer(model, lr=0.01, use_graph=False)
    osd = tr.optimizer.state_dict()
    tsd = orc.seg_make_optimizer(ref, lr=0.01).state_dict()
    assert [g["params"] for g in osd["param_groups"]] == [g["params"] for g in tsd["param_groups"]]
    assert [g["lr"] for g in osd["param_groups"]] == [g["lr"] for g in tsd["param_groups"]]
    tr.optimizer.load_state_dict(osd)
    sd = model.state_dict()
    for k, v in ref.state_dict().items():
        assert tuple(sd[k].shape) == tuple(v.shape) and sd[k].dtype == v.dtype, k


def test_deeplab_dual_bn_option(pkg, gpu):
    """BASELINE configs[3] names "dual-BN"; the reference has none, so it is an option (off by default) without a golden:
    with the auxiliary set a copy of the main one, the first fp32 iteration reproduces the shared-BN losses, the
    auxiliary running statistics are the ones the adversarial passes updated, and the bf16 iteration is graph-captured."""
    g = golden("seg_dl101_aspp_k1")
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    steps = int(g["meta"][0])
    res = {}
    for dual in (False, True):
        model, tr = _build(pkg, g, torch.float32, True, gpu, use_graph=False, dual_bn=dual)
        res[dual] = (tr.step(images, labels), model)
    np.testing.assert_allclose(res[True][0]["losses"].cpu().numpy(), res[False][0]["losses"].cpu().numpy(), rtol=0, atol=2e-5)
    sd0, sd1 = res[False][1].state_dict(), res[True][1].state_dict()
    assert [k for k in sd1 if ".adv." not in k] == list(sd0.keys())
    # a layer4 BatchNorm (behind both perturbation points' tails): main set = the clean head pass + the clean forward of
    # the decoder-PGD input + o0; auxiliary = SE-PGD passes + the two SE tails (the SD tail starts after the backbone)
    k = "backbone.layer4.0.bn1"
    shared, main, aux = (int(sd0[k + ".num_batches_tracked"]), int(sd1[k + ".num_batches_tracked"]),
                         int(sd1[k + ".adv.num_batches_tracked"]))
    assert shared == main + aux and aux == steps + 2 and main >= 2, (shared, main, aux)
    assert not torch.equal(sd1[k + ".adv.running_mean"], sd1[k + ".running_mean"])
    # a stem BatchNorm never sees adversarial features
    assert int(sd1["backbone.bn1.adv.num_batches_tracked"]) == 0
    model, tr = _build(pkg, g, torch.bfloat16, True, gpu, use_graph=True, graph_warmup=1, dual_bn=True)
    losses = []
    for _ in range(3):
        losses.append(float(tr.step(images, labels)["loss"]))
        tr.scheduler.step()
    assert tr._graph is not None, tr._graph_failed
    assert all(np.isfinite(losses))


@pytest.mark.parametrize("case,dtype", [("seg_dl101_aspp_k3_damped", torch.float32), ("seg_dl101_concat_k3", torch.float32),
                                        ("seg_dl101_aspp_k3_damped", torch.bfloat16)])
def test_deeplab_folded_clean_forward_equals_three_passes(pkg, gpu, case, dtype):
    """forward_clean_folded (the head pass, the decoder-PGD input pass and the clean forward as ONE pass, BatchNorm
    running statistics updated in the reference's order with the clean forward's update deferred behind the PGD loops)
    against the three separate passes: two iterations each from the same seed."""
    g = golden(case)
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    res = {}
    # (the first fp32 iteration of a process differs from later ones by ~2 % in the backbone gradients — the vendor's fp32
    # convolutions settle on their algorithms during it, tools/probe/fold_grads.py — so a discarded iteration comes first)
    for fold in ((None, False, True) if dtype == torch.float32 else (False, True)):
        model, tr = _build(pkg, g, dtype, True, gpu, use_graph=False, fold_clean=bool(fold))
        before = dict(pkg.ops.CALLS)
        out = [tr.step(images, labels)]
        if fold is None:
            continue
        n_fwd = pkg.ops.CALLS["conv_fwd"] - before["conv_fwd"]
        if dtype == torch.float32:             # fp32: the general kernels (forward + dgrad + wgrad passes counted together)
            n_fwd = pkg.ops.CALLS["conv_general"] - before["conv_general"]
        assert pkg.ops.CALLS["vendor_conv"] == 0
        grad1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        sd1_ = {k: v.clone() for k, v in model.state_dict().items()}       # after the FIRST iteration: same weights in both runs
        tr.scheduler.step()
        out.append(tr.step(images, labels))
        res[fold] = (out, grad1, sd1_, n_fwd)
    (o0, g0, sd0, n0), (o1, g1, sd1, n1) = res[False], res[True]
    assert n1 < n0 - 150, (n0, n1)               # ResNet-101: two backbone passes to the SE point + one layer4/ASPP pass fewer
    exact = dtype == torch.float32
    # (undamped network: sign() flips of the K = 3 PGD steps on gradients within rounding of zero move a loss term by ~1e-4)
    ltol = (2e-5 if "damped" in case else 1e-3) if exact else 2e-2
    np.testing.assert_allclose(o1[0]["losses"].cpu().numpy(), o0[0]["losses"].cpu().numpy(), rtol=0, atol=ltol)
    # second iteration: the freshly initialised 101-layer network amplifies last-bit differences of the first update
    # (section 9.3 of NOTES.md: 0.3 % in, 58 % out at layer3) — a sanity bound only
    np.testing.assert_allclose(o1[1]["losses"].cpu().numpy(), o0[1]["losses"].cpu().numpy(), rtol=5e-2)
    # gradients: the contractive ("damped") network in fp32 pins the fold tightly; the freshly initialised one amplifies
    # summation-order noise (one graph receives what three graphs received) through its 33 blocks
    gtol = (5e-3 if "damped" in case else 0.1) if exact else 0.35
    for n, a in g0.items():
        rel = float((g1[n] - a).norm() / a.norm().clamp_min(1e-12))
        assert rel <= gtol, (n, rel)
    for k, v in sd0.items():
        if "num_batches_tracked" in k:
            assert int(sd1[k]) == int(v), k       # every BatchNorm saw the reference's number of updates
        elif "running_" in k:
            np.testing.assert_allclose(sd1[k].float().cpu().numpy(), v.float().cpu().numpy(), err_msg=k,
                                       # (the undamped fp32 cases: PGD sign flips between the two schedules move the perturbed
                                       #  passes' batch moments — at momentum 0.01 that is ~1e-4 of a running mean)
                                       **((dict(rtol=1e-4, atol=1e-5) if "damped" in case else dict(rtol=1e-3, atol=2e-4)) if exact
                                          else dict(rtol=5e-2, atol=5e-3)))


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_batched_sample_point_tails_equal_two_passes(pkg, gpu, dropout):
    """seg_train_phases(batch_tails=True): the two sample-point forwards (main_aug_final.py:197-203) as ONE pass over their
    concatenation, BatchNorm statistics / running updates / dropout draws per half (resnet_s.bn_groups(2)) — against the two
    passes, on ONE iteration's losses, gradients and BatchNorm buffers.  The first half is the same arithmetic launch for
    launch; the second half's BatchNorm moments are summed around the running mean BEFORE the first half's update instead of
    after it (a rounding-level difference in the statistics, a bf16 ulp here and there in the activations), and the halves'
    weight-gradient sums are added in another order.  Scale of that against the step's own sensitivity (tools/
    diag_dl_batch_tails.py, same model): gradient cosine 0.99997 between the two schedules, 0.17 between two runs of ONE schedule
    whose images differ by 1e-6 (the sign steps of the PGD loops are discontinuous) — hence one iteration, not several."""
    g = torch.Generator().manual_seed(5)
    images = torch.rand(8, 3, 65, 65, generator=g).to(gpu)
    labels = torch.randint(0, 21, (8, 65, 65), generator=g).to(gpu)
    res = {}
    for batched in (False, True):
        torch.manual_seed(3)
        model = pkg.deeplab.deeplabv3plus_resnet50(num_classes=21, output_stride=16)
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = dropout
        model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
        tr = pkg.seg_trainer.SegTrainer(model, steps=2, lr=0.0, momentum=0.0, weight_decay=0.0, use_graph=False, batch_tails=batched)
        torch.manual_seed(9)
        pkg.ops._dropout_state.clear()              # the device-side dropout generator is re-seeded from torch's CPU generator
        r = tr.step(images, labels)
        torch.cuda.synchronize()
        assert r["batch_tails"] is batched
        res[batched] = (r["losses"].float().cpu().numpy(), torch.cat([p.grad.float().flatten() for p in tr.arena.params]),
                        {n: b.float().clone() for n, b in model.named_buffers()})
    a, b = res[True], res[False]
    assert a[0][0] == b[0][0] and a[0][3] == b[0][3]                              # the clean and the decoder-point losses: the same numbers
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=2e-3)
    cos = float(torch.dot(a[1], b[1]) / (a[1].norm() * b[1].norm()))
    rel = float((a[1] - b[1]).norm() / b[1].norm())
    assert cos > 0.9995 and rel < 0.03, (cos, rel)
    for n in b[2]:
        if "num_batches_tracked" in n:
            assert torch.equal(a[2][n], b[2][n]), n         # the same number of BatchNorm updates, layer by layer
        else:
            assert torch.allclose(a[2][n], b[2][n], rtol=5e-3, atol=5e-3), n
