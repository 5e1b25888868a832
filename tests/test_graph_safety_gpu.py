"""The captured training step must be self-contained: after capture, every byte the caching allocator can hand out is
filled with NaN (what later eager work — validation, checkpointing — may do to freed blocks); the next replays must be
unaffected.  Regression for NaN losses at the first iteration after the first validation pass."""
import math

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _poison(dev):
    keep, sz = [], 1 << 28
    base = torch.cuda.memory_reserved()
    while sz >= 512:
        t = torch.empty(sz // 4, device=dev)
        if torch.cuda.memory_reserved() > base:      # a new segment: keep it cached, fill it with the smaller sizes
            del t
            base = torch.cuda.memory_reserved()
            sz //= 2
            continue
        t.fill_(float("nan"))
        keep.append(t)
    torch.cuda.synchronize()
    return keep


@pytest.mark.parametrize("arch,idx,batch,side,classes", [("resnet18", 6, 64, 32, 10), ("resnet20s", 7, 64, 32, 10),
                                                         ("resnet56s", 13, 32, 32, 10), ("resnet50", 8, 8, 224, 1000)])
@pytest.mark.parametrize("fold", [False, True])
def test_captured_step_owns_its_memory(pkg, gpu, arch, idx, batch, side, classes, fold):
    torch.manual_seed(0)
    ctor, _ = pkg.resnet_s.ARCHS[arch]
    model = ctor()
    model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    tr = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=2, gamma=0.5, eps=2.0, perturb_idx=idx, lr=0.01,
                                    fold_clean=fold)      # both schedules: the reference's, and the one clean tail pass
    assert not pkg.resnet_s.vendor_convs(model)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(batch, 3, side, side, generator=g).to(gpu)
    y = torch.randint(0, classes, (batch,), generator=g).to(gpu)
    for _ in range(5):
        r = tr.step(x, y)
    assert tr._graph is not None, tr._graph_failed
    keep = _poison(gpu)
    del keep
    for _ in range(3):
        r = tr.step(x, y)
        # eager work in between, as validation does: new allocations land in the poisoned blocks
        model.eval()
        with torch.no_grad():
            out = model(x, end_point=model.layer_number, start_point=0)
        model.train()
        assert torch.isfinite(out.float()).all()
    assert math.isfinite(float(r["loss"])) and math.isfinite(float(r["loss_adv"]))
    assert torch.isfinite(r["l2"]).all() and torch.isfinite(tr.arena.param).all()


def test_general_kernel_configurations_are_captured_too(pkg, gpu):
    """NCHW weights (bf16) and fp32 parity mode run their convolutions on the library's general fp32-arithmetic kernels
    (afan_conv_f32.hip) — no vendor library — and the step is replayed from a hipGraph like the tuned configuration."""
    for dtype, cl in ((torch.bfloat16, False), (torch.float32, True), (torch.float32, False)):
        torch.manual_seed(0)
        model = pkg.resnet_s.resnet20()
        model.set_compute_dtype(dtype).set_channels_last(cl).to(gpu).train()
        assert pkg.resnet_s.general_convs(model) and not pkg.resnet_s.vendor_convs(model)
        tr = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=1, gamma=0.5, eps=2.0, perturb_idx=7, lr=0.01)
        x = torch.rand(32, 3, 32, 32, device=gpu)
        y = torch.randint(0, 10, (32,), device=gpu)
        before = dict(pkg.ops.CALLS)
        losses = []
        for _ in range(6):
            r = tr.step(x, y)
            losses.append(float(r["loss"]))
        assert tr._graph is not None, tr._graph_failed
        assert pkg.ops.CALLS["vendor_conv"] == 0 and pkg.ops.CALLS["conv_general"] > before["conv_general"]
        assert all(math.isfinite(v) for v in losses)
        keep = _poison(gpu)
        del keep
        r = tr.step(x, y)
        assert math.isfinite(float(r["loss"])) and torch.isfinite(tr.arena.param).all()


def test_learnable_captured_step_owns_its_memory(pkg, gpu):
    """The learnable multi-layer step (main_learnable.py) is replayed from a hipGraph too: same requirement."""
    torch.manual_seed(0)
    model = pkg.resnet_s.resnet56(init_weight_eta=1 / 9)
    model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    tr = pkg.learnable.LearnableTrainer(model, nn.CrossEntropyLoss(), steps=1, gamma=0.5, eps=2.0, lr=0.01, w_lr=0.01,
                                        l1_coef=0.1)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(32, 3, 32, 32, generator=g).to(gpu)
    y = torch.randint(0, 10, (32,), generator=g).to(gpu)
    for _ in range(5):
        r = tr.step(x, y)
    assert tr._graph is not None, tr._graph_failed
    keep = _poison(gpu)
    del keep
    for _ in range(3):
        r = tr.step(x, y)
        junk = torch.full((1 << 22,), float("nan"), device=gpu)       # eager allocations between replays
        del junk
    assert math.isfinite(float(r["loss"])) and torch.isfinite(r["w"]).all() and torch.isfinite(r["l2"]).all()


def test_ragged_last_batch_runs_eagerly_beside_the_graph(pkg, gpu, orc):
    """The reference's loaders keep the last, smaller batch (drop_last=False): a captured step serves its own shape only;
    another batch size takes eager launches on the same weights, and the graph keeps working afterwards.  The smaller
    batch is not a multiple of any tile (odd image count): it also checks the kernels' row masks end to end against
    the oracle."""
    torch.manual_seed(0)
    model = pkg.resnet_s.resnet20()
    ref = orc.resnet20s()
    ref.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    tr = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=2, gamma=0.5, eps=2.0, perturb_idx=7, lr=0.0)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(64, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (64,), generator=g)
    for _ in range(5):
        tr.step(x.to(gpu), y.to(gpu))
    assert tr._graph is not None
    xs, ys = x[:37], y[:37]
    r = tr.step(xs.to(gpu), ys.to(gpu))                     # eager: different shape
    assert r["l2"].shape == (37,) and math.isfinite(float(r["loss"]))
    # lr = 0: weights unchanged (running statistics move, training-mode passes do not read them) -> comparable to the oracle
    ref.train()
    r_ref = orc.afan_train_step(ref, orc.make_optimizer(ref, lr=0.0), nn.CrossEntropyLoss(), xs, ys, steps=2, gamma=0.5,
                                eps=2.0, perturb_idx=7, layer_number=16)
    assert abs(float(r["loss_clean"]) - float(r_ref["loss_clean"])) < 5e-2       # bf16 backbone vs fp32 oracle
    assert abs(float(r["loss"]) - float(r_ref["loss"])) < 8e-2
    r2 = tr.step(x.to(gpu), y.to(gpu))                      # the captured shape again: replay
    assert tr._graph is not None and math.isfinite(float(r2["loss"])) and r2["l2"].shape == (64,)


@pytest.mark.parametrize("what", ["resnet18", "deeplab"])
@pytest.mark.parametrize("graph", [False, True])
def test_weight_gradient_stream_changes_no_bit(pkg, gpu, what, graph):
    """Weight-gradient launches on a side stream during the backward (resnet_s._WgradStream: a parallel branch of the
    captured graph; switched on by workload size), joined by the autograd engine's end-of-backward callback.  They stay in
    program order among themselves, so the fp32 sums into the gradient arena are the same sums: parameters after three iterations equal, bit for bit, those
    of a run with everything on one stream — which also shows that no launch reads a buffer another stream has recycled."""
    ws = pkg.resnet_s._WgradStream
    g = torch.Generator().manual_seed(1)
    if what == "resnet18":
        x, y = torch.rand(64, 3, 32, 32, generator=g).to(gpu), torch.randint(0, 10, (64,), generator=g).to(gpu)
    else:
        x, y = torch.rand(2, 3, 129, 129, generator=g).to(gpu), torch.randint(0, 21, (2, 129, 129), generator=g).to(gpu)
    res, old, force = {}, ws.ON, ws.FORCE
    ws.FORCE = None
    try:
        for on in (True, False):
            torch.manual_seed(0)
            if what == "resnet18":
                model = pkg.resnet_s.ARCHS["resnet18"][0]()
                model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
                tr = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=2, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.01,
                                                use_graph=graph, graph_warmup=1)
            else:
                model = pkg.deeplab.deeplabv3plus_resnet50(num_classes=21, output_stride=16)
                for m in model.modules():
                    if isinstance(m, nn.Dropout):
                        m.p = 0.0
                model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
                tr = pkg.seg_trainer.SegTrainer(model, steps=2, lr=0.01, use_graph=graph, graph_warmup=1, wgrad_stream=on)
                assert tr._wgrad_side(x) == on
                assert pkg.seg_trainer.SegTrainer.WGRAD_STREAM_MIN_PIXELS <= 8 * 513 * 513      # the batch-8 configuration uses it
            for _ in range(3):
                with pkg.resnet_s.wgrad_stream(on):
                    r = tr.step(x, y)
            if graph:
                assert tr._graph is not None, getattr(tr, "_graph_failed", None)
            torch.cuda.synchronize()
            assert not ws.held and not ws.mains          # joined at the end of every backward
            res[on] = (tr.arena.param.clone(), float(r["loss"]))
    finally:
        ws.ON, ws.FORCE = old, force
    assert math.isfinite(res[True][1])
    assert torch.equal(res[True][0], res[False][0])
