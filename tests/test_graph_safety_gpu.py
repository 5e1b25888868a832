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
def test_captured_step_owns_its_memory(pkg, gpu, arch, idx, batch, side, classes):
    torch.manual_seed(0)
    ctor, _ = pkg.resnet_s.ARCHS[arch]
    model = ctor()
    model.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    tr = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=2, gamma=0.5, eps=2.0, perturb_idx=idx, lr=0.01)
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


def test_vendor_configurations_are_not_captured(pkg, gpu):
    """fp32 parity mode / NCHW weights run their convolutions in the vendor library: eager launches, no hipGraph."""
    torch.manual_seed(0)
    model = pkg.resnet_s.resnet20()
    model.set_compute_dtype(torch.bfloat16).to(gpu).train()          # bf16 but NCHW weights
    assert pkg.resnet_s.vendor_convs(model)
    tr = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=1, gamma=0.5, eps=2.0, perturb_idx=7, lr=0.01)
    x = torch.rand(32, 3, 32, 32, device=gpu)
    y = torch.randint(0, 10, (32,), device=gpu)
    before = dict(pkg.ops.CALLS)
    for _ in range(5):
        r = tr.step(x, y)
    assert tr._graph is None and tr.use_graph is False
    assert pkg.ops.CALLS["vendor_conv"] > before["vendor_conv"]
    assert math.isfinite(float(r["loss"]))


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
