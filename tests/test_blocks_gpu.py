"""Block by block: every block type of the benched bf16 channels-last networks, fed the ORACLE's input, against the oracle's
block under bf16 emulation (values and gradients rounded to bf16 exactly where the product stores bf16, fp32 elsewhere) —
output, input gradient and every parameter gradient of the block.  A wrong kernel (a mis-scaled gradient, a dropped tap, a
fusion reading the wrong statistics) shows here at once; end to end a freshly initialised network hides it behind the
chaos of its own depth.  Shapes: BASELINE configs[1] (ResNet-18 CIFAR: 64..512 channels at 32..4 pixels)
and configs[3] (DeepLabv3+ ResNet-101 at 513 x 513, output stride 16: 129 / 65 / 33 pixel maps, atrous layer4, ASPP, decoder).
Reference blocks: oracle/afan_oracle.py (Classification/resnet_s.py:48-77; Segmentation/network/backbone/resnet.py:76-119,
_deeplab.py:28-80,143-193)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# measured on MI355X (see the assertion messages of a failing run for the values): forward 1e-3..4e-3, input gradient and
# parameter gradients 3e-3..1.2e-2; bounds ~2x
FWD_TOL, DX_TOL, DW_TOL = 1e-2, 2.5e-2, 2.5e-2


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _bf(t):
    return t.bfloat16().float()


def _to_dev(t, gpu):
    return t.to(gpu).bfloat16().contiguous(memory_format=torch.channels_last)


def _check(name, y, y_ref, dx, dx_ref, grads, grads_ref, fwd=FWD_TOL, dxt=DX_TOL, dwt=DW_TOL):
    e = _rel(y, y_ref)
    assert e <= fwd, f"{name}: output off by {e:.3e}"
    if dx is not None:
        e = _rel(dx, dx_ref)
        assert e <= dxt, f"{name}: input gradient off by {e:.3e}"
    for k, gr in grads_ref.items():
        e = _rel(grads[k], gr)
        assert e <= dwt, f"{name}: gradient of {k} off by {e:.3e}"


# ------------------------------------------------------------------------------------------ ResNet-18, configs[1]
def test_resnet18_blocks_match_bf16_oracle_blocks(pkg, orc, gpu):
    torch.manual_seed(3)
    ref = orc.ARCHS["resnet18"][0]()
    ref.train()
    m = pkg.resnet_s.ARCHS["resnet18"][0]()
    m.load_state_dict(ref.state_dict())
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    arena = pkg.arena.ParamArena(m)
    assert pkg.resnet_s.general_convs(m) == []
    torch.manual_seed(5)
    x = torch.rand(64, 3, 32, 32)
    n = len(ref.sequential_model)
    pnames = {id(p): k for k, p in m.named_parameters()}
    # segments: the stem (normalise, conv, BN, ReLU), each residual block, the classifier head
    segs = [(0, 4)] + [(i, i + 1) for i in range(4, n - 3)] + [(n - 3, n)]
    g = torch.Generator().manual_seed(11)
    for (a, b) in segs:
        with orc.emulate_bf16():
            with torch.no_grad():
                t_in = ref(x, end_point=a, start_point=0) if a > 0 else x
            t_in = t_in.detach().clone().requires_grad_(a > 0)
            for p in ref.parameters():
                p.grad = None
            y_ref = ref(t_in, end_point=b, start_point=a)
            gy = torch.randn(y_ref.shape, generator=g) * 1e-2
            gy = _bf(gy) if y_ref.dim() == 4 else gy
            y_ref.backward(gy)
        arena.zero_grad()
        pkg.ops.acc_reset(gpu)
        xin = (x.to(gpu) if a == 0 else _to_dev(t_in.detach(), gpu)).requires_grad_(a > 0)
        y = m(xin, end_point=b, start_point=a)
        gyd = gy.to(gpu)
        if y.dim() == 4:
            gyd = gyd.to(y.dtype).contiguous(memory_format=torch.channels_last)
        y.backward(gyd)
        grads_ref = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
        grads = {k: p.grad for k, p in m.named_parameters() if k in grads_ref}
        name = f"sequential_model[{a}:{b}] ({type(ref.sequential_model[b - 1]).__name__})"
        _check(name, y, y_ref, xin.grad if a > 0 else None, t_in.grad if a > 0 else None, grads, grads_ref)
    assert pkg.ops.CALLS["vendor_conv"] == 0


def test_narrow_option_b_blocks_take_the_two_launch_backward(pkg, orc, gpu):
    """a narrow option-B ResNet (16 -> 32 -> 64 projection blocks) in bf16 channels-last.  The fused
    3x3/s2 + 1x1/s2 input gradient (afan_conv_dgrad_sc_nhwc_bf16) has no small-channel form, so the arena must not hand such
    blocks its [Ci][10][Co] operand — the backward of every block runs (two launches) and matches the bf16-emulating oracle."""
    torch.manual_seed(3)
    ref = orc.SlicedResNet([16, 32, 64], [1, 1, 1], "B")
    ref.train()
    m = pkg.resnet_s.ResNet(pkg.resnet_s.BasicBlock, [1, 1, 1], widths=(16, 32, 64), option="B")
    m.load_state_dict(ref.state_dict())
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    arena = pkg.arena.ParamArena(m)
    narrow = [b for b in m.modules() if getattr(b, "_sc_kind", None) == "conv"]
    assert len(narrow) == 2 and all(getattr(b._chain()[0][0], "_arena_wt10", None) is None for b in narrow)
    wide = pkg.resnet_s.ARCHS["resnet18"][0]().set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu)
    pkg.arena.ParamArena(wide)
    assert all(getattr(b._chain()[0][0], "_arena_wt10", None) is not None for b in wide.modules() if getattr(b, "_sc_kind", None) == "conv")
    torch.manual_seed(5)
    x = torch.rand(32, 3, 32, 32)
    g = torch.Generator().manual_seed(11)
    n = len(ref.sequential_model)
    for a in range(5, n - 3):                      # the two projection blocks
        with orc.emulate_bf16():
            with torch.no_grad():
                t_in = ref(x, end_point=a, start_point=0)
            t_in = t_in.detach().clone().requires_grad_(True)
            for p in ref.parameters():
                p.grad = None
            y_ref = ref(t_in, end_point=a + 1, start_point=a)
            gy = _bf(torch.randn(y_ref.shape, generator=g) * 1e-2)
            y_ref.backward(gy)
        arena.zero_grad()
        pkg.ops.acc_reset(gpu)
        xin = _to_dev(t_in.detach(), gpu).requires_grad_(True)
        y = m(xin, end_point=a + 1, start_point=a)
        y.backward(gy.to(gpu).to(y.dtype).contiguous(memory_format=torch.channels_last))
        grads_ref = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
        grads = {k: p.grad for k, p in m.named_parameters() if k in grads_ref}
        # (16 / 32-channel BatchNorm parameter gradients: sums over few, bf16-rounded terms — measured 2.6e-2 on bn2.bias)
        _check(f"narrow option-B block {a}", y, y_ref, xin.grad, t_in.grad, grads, grads_ref, dwt=4e-2)


# ------------------------------------------------------------------------------------------ DeepLabv3+ R101, configs[3]
def _emu_cbr(orc, conv, bn, t, relu=True):
    o = bn(orc._conv(conv, t))
    return orc._r(F.relu(o) if relu else o)


def _emu_bottleneck(orc, b, t):
    """Segmentation/network/backbone/resnet.py:97-119 with the product's bf16 rounding points."""
    o = _emu_cbr(orc, b.conv1, b.bn1, t)
    o = _emu_cbr(orc, b.conv2, b.bn2, o)
    o = b.bn3(orc._conv(b.conv3, o))
    res = _emu_cbr(orc, b.downsample[0], b.downsample[1], t, relu=False) if b.downsample is not None else t
    return orc._r(F.relu(o + res))


def _emu_aspp(orc, aspp, t):
    """_deeplab.py:165-193 (dropout off); the pooling branch in fp32, as the product runs it."""
    res = [_emu_cbr(orc, c[0], c[1], t) for c in list(aspp.convs)[:4]]
    pool = aspp.convs[4]
    p = t.mean(dim=(2, 3), keepdim=True)
    p = F.relu(pool[2](F.conv2d(p, pool[1].weight)))
    res.append(orc._r(p.expand(-1, -1, t.shape[2], t.shape[3])))
    return _emu_cbr(orc, aspp.project[0], aspp.project[1], torch.cat(res, dim=1))


def _emu_decoder(orc, head, low, hi):
    """_deeplab.py:47-80: low-level projection, resize + concat, 3x3 conv, classifier (fp32 logits)."""
    lo = _emu_cbr(orc, head.project[0], head.project[1], low)
    up = orc._r(F.interpolate(hi, size=lo.shape[2:], mode="bilinear", align_corners=False))
    c = _emu_cbr(orc, head.classifier[0], head.classifier[1], torch.cat([lo, up], dim=1))
    return F.conv2d(c, head.classifier[3].weight, head.classifier[3].bias)


@pytest.mark.parametrize("which", ["layer1.0", "layer1.1", "layer2.0", "layer2.1", "layer3.0", "layer3.1", "layer4.0", "layer4.1",
                                   "aspp", "decoder"])
def test_deeplab_blocks_match_bf16_oracle_blocks(pkg, orc, gpu, which):
    """513 x 513 images, batch 2, output stride 16: layer1 at 129 x 129, layer2 at 65 x 65, layer3 / atrous layer4 / ASPP at
    33 x 33, decoder at 129 x 129.  Inputs: post-ReLU-like random maps of the block's input shape (the same for both sides)."""
    torch.manual_seed(3)
    ref = orc.deeplabv3plus_resnet101(21, 16)
    ref.classifier.aspp.project[3].p = 0.0
    ref.train()
    m = pkg.deeplab.deeplabv3plus_resnet101(21, 16)
    m.load_state_dict(ref.state_dict())
    m.classifier.aspp.project[3].p = 0.0
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    pkg.arena.ParamArena(m, skip=())
    g = torch.Generator().manual_seed(7)
    shapes = {"layer1.0": (2, 64, 129, 129), "layer1.1": (2, 256, 129, 129), "layer2.0": (2, 256, 129, 129), "layer2.1": (2, 512, 65, 65),
              "layer3.0": (2, 512, 65, 65), "layer3.1": (2, 1024, 33, 33), "layer4.0": (2, 1024, 33, 33), "layer4.1": (2, 2048, 33, 33),
              "aspp": (2, 2048, 33, 33)}
    for p in ref.parameters():
        p.grad = None
    pkg.ops.acc_reset(gpu)
    if which == "decoder":
        low = _bf(torch.randn(2, 256, 129, 129, generator=g).relu())
        hi = _bf(torch.randn(2, 256, 33, 33, generator=g).relu())
        low_r, hi_r = low.clone().requires_grad_(True), hi.clone().requires_grad_(True)
        with orc.emulate_bf16():
            y_ref = _emu_decoder(orc, ref.classifier, low_r, hi_r)
            gy = torch.randn(y_ref.shape, generator=g) * 1e-3
            y_ref.backward(gy)
        low_d, hi_d = _to_dev(low, gpu).requires_grad_(True), _to_dev(hi, gpu).requires_grad_(True)
        y = m.classifier({"low_level": low_d, "adv": hi_d}, "aspp_tail")
        y.backward(gy.to(gpu).contiguous(memory_format=torch.channels_last))
        mod_ref, prefix = ref.classifier, "classifier."
        dx, dx_ref = torch.cat([low_d.grad.flatten(), hi_d.grad.flatten()]), torch.cat([low_r.grad.flatten(), hi_r.grad.flatten()])
        skip = ("aspp.",)
    else:
        t = _bf(torch.randn(*shapes[which], generator=g).relu() * 0.7)
        t_r = t.clone().requires_grad_(True)
        t_d = _to_dev(t, gpu).requires_grad_(True)
        with orc.emulate_bf16():
            if which == "aspp":
                mod_ref, mod, prefix = ref.classifier.aspp, m.classifier.aspp, "classifier.aspp."
                y_ref = _emu_aspp(orc, mod_ref, t_r)
            else:
                ln, bi = which.split(".")
                mod_ref, mod, prefix = getattr(ref.backbone, ln)[int(bi)], getattr(m.backbone, ln)[int(bi)], f"backbone.{which}."
                y_ref = _emu_bottleneck(orc, mod_ref, t_r)
            gy = _bf(torch.randn(y_ref.shape, generator=g) * 1e-2)
            y_ref.backward(gy)
        y = mod(t_d)
        y.backward(_to_dev(gy, gpu))
        dx, dx_ref, skip = t_d.grad, t_r.grad, ()
    grads_ref = {prefix + k: p.grad for k, p in mod_ref.named_parameters() if p.grad is not None and not k.startswith(skip)}
    params = dict(m.named_parameters())
    grads = {k: params[k].grad for k in grads_ref}
    assert grads_ref, "no parameter gradients to compare"
    _check(which, y, y_ref, dx, dx_ref, grads, grads_ref)
    assert pkg.ops.CALLS["vendor_conv"] == 0
