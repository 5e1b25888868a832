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
    ck0 = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in model.state_dict().values()])
    np.testing.assert_array_equal(ck0, g["ck0"])        # seeded construction == the reference's own (keys and values)
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


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("case", ["seg_dl101_aspp_k1", "seg_dl101_concat_k3"])
def test_deeplab_step_fp32_matches_reference(pkg, gpu, case, nhwc):
    torch.backends.cudnn.deterministic = True
    g = golden(case)
    model, tr = _build(pkg, g, torch.float32, nhwc, gpu, use_graph=False)
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    r = tr.step(images, labels)
    torch.cuda.synchronize()
    loss = float(g["loss"])
    assert abs(float(r["loss"]) - loss) <= 1e-4 * max(1.0, abs(loss)), (float(r["loss"]), loss)
    np.testing.assert_allclose(r["losses"].cpu().numpy(), g["losses"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(r["fm_se"].float().cpu().numpy(), g["fm_se"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(r["out_clean"][:, :, ::4, ::4].cpu().numpy(), g["out_clean_sub"], rtol=1e-3, atol=2e-4)
    steps = int(g["meta"][0])
    gam = float(g["gammas"][0]) / 255
    adv = r["adv_se"].float().cpu().numpy()
    d = np.abs(adv - r["fm_se"].float().cpu().numpy())
    assert d.max() <= steps * gam * (1 + 1e-5) + 1e-7                    # on the sign grid, at most K steps away
    # sign(): a gradient within rounding distance of zero flips an element by 2*gamma
    assert_close_frac(adv, g["adv_se"], 1e-5, 2e-5, 2e-2 if steps == 1 else 0.15, "adv_se")
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
    # whole state_dict after the SGD step (two learning-rate groups): per-tensor checksums
    np.testing.assert_allclose(_cks(model), g["ck1"], rtol=2e-4, atol=5e-3)


@pytest.mark.parametrize("case", ["seg_dl101_aspp_k1", "seg_dl101_concat_k3"])
def test_deeplab_step_bf16_runs_on_the_library_kernels(pkg, gpu, case):
    g = golden(case)
    model, tr = _build(pkg, g, torch.bfloat16, True, gpu, use_graph=False)
    assert pkg.resnet_s.vendor_convs(model) == [], pkg.resnet_s.vendor_convs(model)
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    before = dict(pkg.ops.CALLS)
    r = tr.step(images, labels)
    torch.cuda.synchronize()
    ran = {k: pkg.ops.CALLS[k] - before[k] for k in before}
    assert ran["vendor_conv"] == 0 and ran["conv_fwd"] > 400 and ran["conv_dgrad"] > 100 and ran["conv_wgrad"] > 100, ran
    loss = float(g["loss"])
    assert abs(float(r["loss"]) - loss) <= 3e-2, (float(r["loss"]), loss)
    np.testing.assert_allclose(r["losses"].cpu().numpy(), g["losses"], rtol=0, atol=3e-2)
    fm = r["fm_se"].float().cpu().numpy()
    assert np.abs(fm - g["fm_se"]).max() <= 0.08 * np.abs(g["fm_se"]).max()
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
