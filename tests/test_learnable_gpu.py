"""Learnable multi-layer A-FAN (main_learnable.py) on the GPU: the blend kernels against their formula, and the whole
step against the oracle / the golden vectors produced by the reference's own train()."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,cl", [((4, 16, 32, 32), False), ((3, 64, 8, 8), True), ((2, 7, 5, 3), False)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_mix_w_forward_backward(pkg, gpu, shape, cl, dt):
    torch.manual_seed(sum(shape))
    conv = (lambda t: t.contiguous(memory_format=torch.channels_last)) if cl else (lambda t: t)
    clean = conv(torch.randn(shape, device=gpu))
    adv = conv(clean + 0.01 * torch.randn(shape, device=gpu))
    w = torch.tensor([0.3, 0.11, -0.2], device=gpu)
    out = pkg.ops.mix_w(clean, adv, w[1:2], dt)
    ref = clean + w[1] * (adv - clean)                     # the reference's eager expression, op by op
    assert out.stride() == clean.stride() and out.dtype == dt
    np.testing.assert_array_equal(out.float().cpu().numpy(), ref.to(dt).float().cpu().numpy())   # bit-exact (+ one RNE)
    g = conv(torch.randn(shape, device=gpu)).to(dt)
    dw = torch.full((3,), 7.0, device=gpu)
    pkg.ops.mix_w_backward(g, clean, adv, dw[2:3])
    want = float((g.double() * (adv.double() - clean.double())).sum())
    assert float(dw[0]) == 7.0 and float(dw[1]) == 7.0
    np.testing.assert_allclose(float(dw[2]), want, rtol=2e-5, atol=1e-6)
    pkg.ops.mix_w_backward(g, clean, adv, dw[2:3], accumulate=True)
    np.testing.assert_allclose(float(dw[2]), 2 * want, rtol=2e-5, atol=2e-6)
    dw2 = torch.zeros(1, device=gpu)
    pkg.ops.mix_w_backward(g, clean, adv, dw2)
    assert float(dw2) * 2 == pytest.approx(float(dw[2]), rel=1e-6)        # deterministic fold


@pytest.mark.parametrize("case", ["learn_r56s_k1", "learn_r56s_k2_clip"])
@pytest.mark.parametrize("channels_last", [False, True])
def test_learnable_step_fp32_matches_reference(pkg, orc, gpu, case, channels_last):
    """One iteration in fp32 against the golden vectors of the reference's train() (and the oracle run beside it)."""
    torch.backends.cudnn.deterministic = True
    g = golden(case)
    K, clip = [int(v) for v in g["meta"]]
    gamma, eps = [float(v) for v in g["gamma_eps"]]
    torch.manual_seed(3)
    ref = orc.resnet56s(init_weight_eta=1 / 9)
    ref.train()
    model = pkg.resnet_s.resnet56(init_weight_eta=1 / 9)
    model.load_state_dict(ref.state_dict())
    model.set_compute_dtype(torch.float32).to(gpu)
    model.set_channels_last(channels_last)
    model.train()
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    tr = pkg.learnable.LearnableTrainer(model, nn.CrossEntropyLoss(), steps=K, gamma=gamma, eps=eps, clip=bool(clip),
                                        layer_number=34)
    r = tr.step(x.to(gpu), y.to(gpu))
    loss = float(g["loss"])
    assert abs(float(r["loss"]) - loss) <= 1e-4 * max(1.0, abs(loss))          # north-star tolerance
    # K > 1: a gradient within rounding distance of zero flips sign(g) for a few elements, which moves
    # a per-depth L2 norm by a fraction of a percent; K = 1 has no second step to be affected
    np.testing.assert_allclose(r["l2"].mean(dim=1).cpu().numpy(), g["l2_mean"], rtol=2e-5 if K == 1 else 1e-2, atol=1e-5)
    np.testing.assert_allclose(r["linf"].mean(dim=1).cpu().numpy(), g["linf_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(model.w.detach().cpu().numpy(), g["w1"], rtol=0, atol=2e-6 if K == 1 else 2e-5)
    assert abs(float(model.w.detach().sum()) - 1.0) < 1e-6
    sd = model.state_dict()
    np.testing.assert_allclose(sd["sequential_model.33.weight"].cpu().numpy(), g["sd1/fc_w"], rtol=0, atol=2e-5 if K == 1 else 2e-4)
    np.testing.assert_allclose(sd["sequential_model.2.running_mean"].cpu().numpy(), g["sd1/bn1_rm"], rtol=1e-5, atol=1e-6)
    assert int(sd["sequential_model.2.num_batches_tracked"]) == 10


def test_learnable_step_bf16_runs_and_tracks_fp32(pkg, orc, gpu):
    g = golden("learn_r56s_k1")
    torch.manual_seed(3)
    ref = orc.resnet56s(init_weight_eta=1 / 9)
    model = pkg.resnet_s.resnet56(init_weight_eta=1 / 9)
    model.load_state_dict(ref.state_dict())
    model.set_compute_dtype(torch.bfloat16).to(gpu)
    model.set_channels_last(True)
    model.train()
    tr = pkg.learnable.LearnableTrainer(model, nn.CrossEntropyLoss(), steps=1, gamma=0.5, eps=2.0, layer_number=34)
    r = tr.step(torch.from_numpy(g["x"]).to(gpu), torch.from_numpy(g["y"]).to(gpu))
    assert abs(float(r["loss"]) - float(g["loss"])) < 0.15                      # bf16 backbone on a batch of 4
    np.testing.assert_allclose(model.w.detach().cpu().numpy(), g["w1"], atol=2e-4)
    assert torch.isfinite(r["l2"]).all() and abs(float(model.w.detach().sum()) - 1.0) < 1e-6


def test_learnable_step_graph_replay_tracks_eager(pkg, orc, gpu):
    """After the warm-up iterations the whole learnable step is replayed from a hipGraph: same trajectory as eager
    launches (bf16, loose), `w` still on the simplex, BatchNorm side effects counted."""
    g = golden("learn_r56s_k1")
    runs = {}
    for use_graph in (False, True):
        torch.manual_seed(3)
        ref = orc.resnet56s(init_weight_eta=1 / 9)
        model = pkg.resnet_s.resnet56(init_weight_eta=1 / 9)
        model.load_state_dict(ref.state_dict())
        model.set_compute_dtype(torch.bfloat16).to(gpu)
        model.set_channels_last(True)
        model.train()
        tr = pkg.learnable.LearnableTrainer(model, nn.CrossEntropyLoss(), steps=1, gamma=0.5, eps=2.0, layer_number=34,
                                            lr=0.01, use_graph=use_graph, graph_warmup=2)
        torch.manual_seed(0)
        x, y = torch.rand(32, 3, 32, 32, device=gpu), torch.randint(0, 10, (32,), device=gpu)
        losses = [float(tr.step(x, y)["loss"]) for _ in range(5)]
        assert (tr._graph is not None) == use_graph and tr._graph_failed is None
        runs[use_graph] = (losses, model.w.detach().cpu().numpy().copy(),
                           int(model.state_dict()["sequential_model.2.num_batches_tracked"]))
    a, b = runs[False], runs[True]
    np.testing.assert_allclose(b[0], a[0], rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(b[1], a[1], atol=2e-3)
    assert abs(float(b[1].sum()) - 1.0) < 1e-5 and a[2] == b[2] == 50
