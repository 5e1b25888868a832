"""Segmentation operators (Segmentation/attack_algo.py) and the main_aug_final.py iteration on the GPU, against the golden
vectors produced by the reference's own functions (on the protocol-faithful stand-in network) and the oracle beside them."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import assert_close_frac, golden

pytestmark = pytest.mark.gpu


def _net(orc, gpu):
    torch.manual_seed(5)
    net = orc.TinySegNet()
    net.train()
    return net.to(gpu)


@pytest.mark.parametrize("case", ["seg_step_aspp_k1", "seg_step_concat_k2"])
def test_seg_step_matches_reference_functions(pkg, orc, gpu, case):
    torch.backends.cudnn.deterministic = True
    g = golden(case)
    steps, se_idx, clip = [int(v) for v in g["meta"]]
    gamma_se, gamma_sd, eps = [float(v) for v in g["gammas"]]
    net = _net(orc, gpu)
    opt = torch.optim.SGD(net.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    crit = nn.CrossEntropyLoss(ignore_index=255, reduction="mean")
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    r = pkg.seg_attack_algo.seg_train_step(net, opt, crit, images, labels, steps=steps, eps=eps, gamma_se=gamma_se,
                                           gamma_sd=gamma_sd, pertub_idx_se=se_idx, pertub_idx_sd=str(g["sd_idx"]),
                                           mix_layer="11", mix_sd=True, clip=bool(clip))
    loss = float(g["loss"])
    assert abs(float(r["loss"]) - loss) <= 1e-4 * max(1.0, abs(loss))
    np.testing.assert_allclose(r["losses"].cpu().numpy(), g["losses"], rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(r["fm_se"].cpu().numpy(), g["fm_se"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r["out_clean"].cpu().numpy(), g["out_clean"], rtol=1e-3, atol=1e-4)
    # perturbations: sign steps of gamma/255 — a gradient within rounding distance of zero may flip a few elements
    gam = gamma_se / 255
    # (the feature map itself carries ~1e-6 of fp32 convolution noise: MIOpen on the GPU vs the CPU kernels of the golden run)
    assert_close_frac(r["adv_se"].cpu().numpy(), g["adv_se"], 1e-5, 1e-5 + (0 if steps == 1 else 2 * gam), 5e-3, "adv_se")
    d = (r["adv_se"] - r["fm_se"]).abs().cpu().numpy()
    assert d.max() <= steps * gam * (1 + 1e-5) + 1e-7
    assert_close_frac(r["adv_sd"].cpu().numpy(), g["adv_sd"], 1e-3, 2e-3, 1e-2, "adv_sd (mixed)")
    # weights after the SGD step: per-tensor checksums of the whole state_dict
    ck = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in net.state_dict().values()])
    np.testing.assert_allclose(ck, g["ck1"], rtol=2e-4, atol=2e-3)


def test_seg_adv_input_and_decoder_clip_error(pkg, orc, gpu):
    g = golden("seg_step_aspp_k1")
    net = _net(orc, gpu)
    crit = nn.CrossEntropyLoss(ignore_index=255, reduction="mean")
    images, labels = torch.from_numpy(g["images"]).to(gpu), torch.from_numpy(g["labels"]).to(gpu)
    net.eval()
    ref = orc.seg_adv_input(x=images.cpu(), criterion=crit, y=labels.cpu(), model=_net(orc, torch.device("cpu")).eval(),
                            steps=2, eps=2.0 / 255, gamma=1.0 / 255, clip=True)
    got = pkg.seg_attack_algo.adv_input(x=images, criterion=crit, y=labels, model=net, steps=2, eps=2.0 / 255,
                                        gamma=1.0 / 255, clip=True)
    assert got.requires_grad and got.is_leaf
    assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    assert float((got.detach() - images).abs().max()) <= 2.0 / 255 + 1e-7                   # projected onto the eps ball
    assert_close_frac(got.detach().cpu().numpy(), ref.detach().numpy(), 0, 1e-6, 2e-2, "adv_input")
    net.train()
    with pytest.raises(NameError):
        d = net({"x": images, "adv": None, "out_idx": "aspp_head", "flag": "clean"})
        pkg.seg_attack_algo.decoder_PGD(d, images, crit, y=labels, model=net, steps=1, eps=2 / 255, gamma=0.5 / 255,
                                        idx="aspp", clip=True)
    with pytest.raises(pkg.AfanLibraryError):
        pkg.seg_attack_algo.PGD(torch.zeros(1, 2, 2, 2), images, None, crit, y=labels, model=net, steps=1, eps=0.1, gamma=0.1)
