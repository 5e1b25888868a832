"""CPU ORACLE for the A-FAN hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this file; the shipped
package (cv_a-fan_amd/) never does.  It restates, in plain torch-CPU fp32 ops, what the reference
(VITA-Group/CV_A-FAN) computes on the path BASELINE.json names, each function citing the reference
file:line it follows (paths relative to the reference root).

Parity status: PINNED.  The reference's own tests hold nothing for this path (its only
test is Detection NMS), so the pin is `oracle/gen_golden.py`: it imports the reference's
Classification/{attack_algo,resnet_s}.py and Segmentation/attack_algo.py in the build container, runs
fixed-seed cases and stores inputs + outputs under tests/golden/*.npz; tests/test_oracle_golden.py
checks every function below against those vectors bit for bit (CPU fp32 is run-to-run deterministic).
The conv / BN / CE / SGD arithmetic itself lives in PyTorch (a third-party dependency of the reference,
README.md:39 pins torch 1.5.0; this image has 2.10.0) — parity is against this image's torch CPU kernels.
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# bf16 emulation (for checking the product's bf16 channels-last path tighter than against the fp32 reference):
# inside `with emulate_bf16():` the ResNets below round to bf16 at the points where the product stores bf16 —
# the low-precision weight copy, every convolution output, every BatchNorm(+residual)(+ReLU) output, the normalised
# image, the feature map PGD hands to the tail — and the gradients flowing back through those same points; sums,
# statistics, the loss, parameter gradients and the SGD update stay fp32, as in the product.  Everything else in this
# file is unchanged, so with emulation off the functions remain the bit-exact restatement of the reference.
# --------------------------------------------------------------------------------------------------
_EMU = [False]


_EMU_ROUND_PROJECTION = [False]


class emulate_bf16:
    """round_projection: the BasicBlock's projection-shortcut BatchNorm output is a stored bf16 tensor (the product's
    per-launch path: slab statistics, grouped passes) instead of staying in registers inside the block's last launch (the
    product's default since round 3: afan_bn_train_forward_acc_dual)."""

    def __init__(self, round_projection=False):
        self.round_projection = bool(round_projection)

    def __enter__(self):
        self.old = (_EMU[0], _EMU_ROUND_PROJECTION[0])
        _EMU[0] = True
        _EMU_ROUND_PROJECTION[0] = self.round_projection
        return self

    def __exit__(self, *exc):
        _EMU[0], _EMU_ROUND_PROJECTION[0] = self.old
        return False


class _RoundBoth(torch.autograd.Function):
    """bf16 round-to-nearest-even of the value (forward) and of the gradient (backward): a tensor stored in bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundFwd(torch.autograd.Function):
    """bf16 copy of an fp32 master weight: rounded value forward, fp32 gradient backward."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _r(t):
    return _RoundBoth.apply(t) if _EMU[0] else t


def _conv(c, t):
    """c(t) for a bias-free nn.Conv2d; under emulation: bf16 weights, bf16-rounded output (fp32 accumulate)."""
    if not _EMU[0]:
        return c(t)
    return _RoundBoth.apply(F.conv2d(t, _RoundFwd.apply(c.weight), None, c.stride, c.padding, c.dilation))


def _emu_sequential(layers, x):
    """sequential_model[s:e](x) with the product's rounding points (only called under emulation)."""
    i, n = 0, len(layers)
    while i < n:
        L = layers[i]
        if isinstance(L, nn.Conv2d):
            x = _conv(L, x)
        elif isinstance(L, nn.BatchNorm2d):
            if i + 1 < n and isinstance(layers[i + 1], nn.ReLU):
                x = _r(F.relu(L(x)))
                i += 1
            else:
                x = _r(L(x))
        elif isinstance(L, ChannelNormalize):
            x = _r(L(x))
        else:
            x = L(x)          # blocks round inside; pooling / flatten / the fp32 classifier do not round
        i += 1
    return x


# --------------------------------------------------------------------------------------------------
# model protocol: model(x, end_point, start_point) == sequential_model[start_point:end_point](x)
# --------------------------------------------------------------------------------------------------
class ChannelNormalize(nn.Module):
    """advertorch.utils.NormalizeByChannelMeanStd as used at Classification/resnet_s.py:87:
    buffers `mean`, `std`; (x - mean[None,:,None,None]) / std[None,:,None,None]."""

    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean, dtype=torch.float32))
        self.register_buffer("std", torch.tensor(std, dtype=torch.float32))

    def forward(self, t):
        return (t - self.mean[None, :, None, None]) / self.std[None, :, None, None]


class _PadShortcut(nn.Module):
    """Option-A shortcut, Classification/resnet_s.py:64-65: subsample by 2, zero-pad planes//4 channels each side."""

    def __init__(self, planes):
        super().__init__()
        self.pad = planes // 4

    def forward(self, t):
        return F.pad(t[:, :, ::2, ::2], (0, 0, 0, 0, self.pad, self.pad), "constant", 0)


class Block(nn.Module):
    """BasicBlock, Classification/resnet_s.py:48-77 (option 'A' pad shortcut or 'B' 1x1 conv + BN)."""

    def __init__(self, cin, cout, stride, option):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.shortcut = nn.Sequential()
        if stride != 1 or cin != cout:
            if option == "A":
                self.shortcut = _PadShortcut(cout)
            else:
                self.shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, t):
        if _EMU[0]:
            o = _r(F.relu(self.bn1(_conv(self.conv1, t))))
            o = self.bn2(_conv(self.conv2, o))
            sc = self.shortcut
            if isinstance(sc, nn.Sequential) and len(sc) == 2:
                res = sc[1](_conv(sc[0], t))
                res = _r(res) if _EMU_ROUND_PROJECTION[0] else res
            else:
                res = sc(t)
            return _r(F.relu(o + res))
        o = F.relu(self.bn1(self.conv1(t)))
        o = self.bn2(self.conv2(o))
        o += self.shortcut(t)  # resnet_s.py:75 (in-place add, then relu)
        return F.relu(o)


class Bottleneck(nn.Module):
    """torchvision-style ResNet-v1.5 bottleneck (1x1 -> 3x3 (stride) -> 1x1 x4, projection shortcut when the shape
    changes).  Build-defined: the reference has no ImageNet-shape classifier (SURVEY.md warning 3); it follows the
    same slice protocol so the reference's PGD drives it unchanged."""
    expansion = 4

    def __init__(self, cin, planes, stride):
        super().__init__()
        cout = planes * 4
        self.conv1 = nn.Conv2d(cin, planes, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, cout, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.shortcut = nn.Sequential()
        if stride != 1 or cin != cout:
            self.shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, t):
        if _EMU[0]:
            o = _r(F.relu(self.bn1(_conv(self.conv1, t))))
            o = _r(F.relu(self.bn2(_conv(self.conv2, o))))
            o = self.bn3(_conv(self.conv3, o))
            sc = self.shortcut
            res = _r(sc[1](_conv(sc[0], t))) if len(sc) == 2 else t
            return _r(F.relu(o + res))
        o = F.relu(self.bn1(self.conv1(t)))
        o = F.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        o += self.shortcut(t)
        return F.relu(o)


class SlicedResNet50(nn.Module):
    """ImageNet-shape ResNet-50 as one flat Sequential with the slice protocol: 0 normalise, 1 conv7x7/2, 2 BN, 3 ReLU,
    4 maxpool, 5-7 layer1, 8-11 layer2, 12-17 layer3, 18-20 layer4, 21 avgpool, 22 flatten, 23 fc."""

    def __init__(self, num_classes=1000, blocks=(3, 4, 6, 3)):
        super().__init__()
        layers = [ChannelNormalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),
                  nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1)]
        cin = 64
        for si, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
            for bi in range(nb):
                layers.append(Bottleneck(cin, planes, 2 if (bi == 0 and si > 0) else 1))
                cin = planes * 4
        layers += [nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(cin, num_classes)]
        self.sequential_model = nn.Sequential(*layers)
        self.all_layers = 9
        self.w = nn.Parameter(torch.full((self.all_layers,), 1.0))
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x, end_point=None, start_point=0):
        if end_point is None:
            end_point = len(self.sequential_model)
        if _EMU[0]:
            return _emu_sequential(list(self.sequential_model[start_point:end_point]), x)
        return self.sequential_model[start_point:end_point](x)


class SlicedResNet(nn.Module):
    """Flat nn.Sequential ResNet with the slice-forward protocol of Classification/resnet_s.py:79-121.
    widths/blocks/option select ResNet-20s/56s (16-32-64, option A: the reference's class) or the
    build-defined ResNet-18-CIFAR (64-128-256-512, option B; SURVEY.md warning 3).  State-dict keys match the
    reference's (`w`, `sequential_model.<i>...`)."""

    def __init__(self, widths, blocks, option, num_classes=10, init_weight=1):
        super().__init__()
        layers = [ChannelNormalize([0.4914, 0.4822, 0.4465], [0.2470, 0.2435, 0.2616]),
                  nn.Conv2d(3, widths[0], 3, 1, 1, bias=False), nn.BatchNorm2d(widths[0]), nn.ReLU()]
        cin = widths[0]
        for si, (wd, nb) in enumerate(zip(widths, blocks)):
            for bi in range(nb):
                layers.append(Block(cin, wd, 2 if (bi == 0 and si > 0) else 1, option))
                cin = wd
        layers += [nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(cin, num_classes)]
        self.sequential_model = nn.Sequential(*layers)
        self.all_layers = 9
        self.w = nn.Parameter(torch.full((self.all_layers,), float(init_weight)))  # resnet_s.py:113-114
        for m in self.modules():  # resnet_s.py:34-38 via self.apply(_weights_init): module traversal order
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, x, end_point=None, start_point=0):
        if end_point is None:
            end_point = len(self.sequential_model)
        if _EMU[0]:
            return _emu_sequential(list(self.sequential_model[start_point:end_point]), x)
        return self.sequential_model[start_point:end_point](x)


def resnet20s():
    return SlicedResNet([16, 32, 64], [3, 3, 3], "A")


def resnet56s(init_weight_eta=1):  # Classification/resnet_s.py:123-124
    return SlicedResNet([16, 32, 64], [9, 9, 9], "A", init_weight=init_weight_eta)


def resnet18_cifar():
    return SlicedResNet([64, 128, 256, 512], [2, 2, 2, 2], "B")


def resnet50(num_classes=1000):
    return SlicedResNet50(num_classes)


ARCHS = {"resnet20s": (resnet20s, 7), "resnet56s": (resnet56s, 13), "resnet18": (resnet18_cifar, 6),
         "resnet50": (resnet50, 8)}


# --------------------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------------------
def tensor_clamp_(t, lo, hi):
    """Classification/attack_algo.py:9-19: lower clamp first, then upper; NaN passes through."""
    t.copy_(torch.where(t < lo, lo, t))
    t.copy_(torch.where(t > hi, hi, t))
    return t


def linf_project_(center, radius, t):
    """Classification/attack_algo.py:35-36: the two bounds are materialised (one rounding each) first."""
    return tensor_clamp_(t, center - radius, center + radius)


def pgd_step_(x_adv, grad, gamma, x_clean=None, eps=0.0, clip=False):
    """One iteration body of Classification/attack_algo.py:53-56 given the gradient."""
    x_adv.add_(gamma * torch.sign(grad))
    if clip:
        linf_project_(x_clean, eps, x_adv)
    return x_adv


def randinit_(x_adv, eps, u=None):
    """Classification/attack_algo.py:42-44; `u` = torch.rand(x_adv.shape) from the CPU default generator."""
    if u is None:
        u = torch.rand(x_adv.shape)
    x_adv += (2.0 * u - 1.0) * eps
    return x_adv


def PGD(x, loss_fn, y=None, model=None, steps=3, gamma=None, start_idx=1, layer_number=16, eps=(2 / 255),
        randinit=False, clip=False, trace=None):
    """Classification/attack_algo.py:38-58 restated; `trace` (list) collects (grad, x_adv) per step."""
    x_adv = x.clone()
    if randinit:
        randinit_(x_adv, eps)
    x_adv.requires_grad_(True)
    for _ in range(steps):
        out = model(_r(x_adv), end_point=layer_number, start_point=start_idx)    # (_r: identity unless emulate_bf16())
        loss = loss_fn(out, y)
        g = torch.autograd.grad(loss, x_adv, only_inputs=True)[0]
        with torch.no_grad():
            pgd_step_(x_adv.data, g.data, gamma, x, eps, clip)
        if trace is not None:
            trace.append((g.detach().clone(), x_adv.detach().clone()))
    return x_adv


def perturb_norms(x_adv, x):
    """Classification/main_perturb.py:188-192: per-sample L2 and Linf of the perturbation."""
    d = (x_adv - x).clone().detach().reshape(x.shape[0], -1)
    return torch.norm(d, p=2, dim=1), torch.norm(d, p=float("inf"), dim=1)


def mix_feature(clean, adv, eps=1e-5):
    """Segmentation/attack_algo.py:121-130 (== Detection/attack_algo.py:254-265)."""
    mc = clean.mean(dim=1, keepdim=True)
    sc = (clean.var(dim=1, keepdim=True) + eps).sqrt()
    ma = adv.mean(dim=1, keepdim=True)
    sa = (adv.var(dim=1, keepdim=True) + eps).sqrt()
    return (clean - mc) / sc * sa + ma


def get_sample_points(px, py, number):
    """Segmentation/attack_algo.py:108-118."""
    percent = 1.0 / (number - 1)
    pts = [px]
    for i in range(1, number - 1):
        pts.append(torch.lerp(px, py, i * percent))
    pts.append(py)
    return pts


# --------------------------------------------------------------------------------------------------
# training step (Classification/main_perturb.py:165-209) and its schedule helpers
# --------------------------------------------------------------------------------------------------
def setup_seed(seed):
    """Classification/main_perturb.py:310-315."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def warmup_lr(step, optimizer, warm_up_steps, max_lr):
    """Classification/main_perturb.py:288-293."""
    lr = min(step * max_lr / (warm_up_steps - 1), max_lr)
    for g in optimizer.param_groups:
        g["lr"] = lr
    return lr


def make_optimizer(model, lr=0.1, momentum=0.9, weight_decay=5e-4):
    """Classification/main_perturb.py:72-74."""
    return torch.optim.SGD(model.parameters(), lr, momentum=momentum, weight_decay=weight_decay)


def afan_train_step(model, optimizer, criterion, inp, target, *, steps, gamma, eps, perturb_idx, layer_number,
                    randinit=False, clip=False):
    """One iteration of Classification/main_perturb.py:173-201 (model already in train mode).
    gamma/eps are the raw flags (divided by 255 here, lines 180/183). Returns a dict of observables."""
    feature_map = model(inp, end_point=perturb_idx, start_point=0).detach()
    feature_map_adv = PGD(feature_map, criterion, y=target, model=model, steps=steps, gamma=gamma / 255,
                          start_idx=perturb_idx, layer_number=layer_number, eps=eps / 255, randinit=randinit,
                          clip=clip)
    l2, linf = perturb_norms(feature_map_adv, feature_map)
    out_adv = model(_r(feature_map_adv), end_point=layer_number, start_point=perturb_idx)
    out_clean = model(inp, end_point=layer_number, start_point=0)
    loss_adv = criterion(out_adv, target)
    loss_clean = criterion(out_clean, target)
    loss = (loss_adv + loss_clean) / 2
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    prec1 = (out_clean.argmax(dim=1) == target).float().sum() * (100.0 / target.shape[0])
    return {"loss": loss.detach(), "loss_adv": loss_adv.detach(), "loss_clean": loss_clean.detach(),
            "l2": l2, "linf": linf, "prec1": prec1, "x_adv": feature_map_adv.detach(),
            "feature_map": feature_map, "out_clean": out_clean.detach()}


# ---------------------------------------------------------------------------------- learnable multi-layer A-FAN (N3)
LEARNABLE_IDX = (4, 8, 11, 14, 18, 21, 24, 28, 31)   # Classification/main_learnable.py:59 (ResNet-56s, layer_number 34)


def sum_project(b, K=9):
    """Classification/main_learnable.py:369-378: shift so that the entries sum to one."""
    return b - (torch.sum(b, dim=0) - 1) / K


def make_learnable_optimizers(model, lr=0.1, w_lr=0.01, momentum=0.9, weight_decay=5e-4):
    """Classification/main_learnable.py:82-90: backbone SGD over sequential_model only; a second SGD for the 9 mixing
    weights `w` (own lr, no weight decay)."""
    opt = torch.optim.SGD(model.sequential_model.parameters(), lr, momentum=momentum, weight_decay=weight_decay)
    opt_w = torch.optim.SGD([{"params": model.w, "lr": w_lr, "weight_decay": 0}], w_lr, momentum=momentum, weight_decay=0)
    return opt, opt_w


def learnable_train_step(model, optimizer, optimizer_w, criterion, inp, target, *, steps, gamma, eps,
                         idx_list=LEARNABLE_IDX, layer_number=34, l1_coef=1.0, randinit=False, clip=False):
    """One iteration of Classification/main_learnable.py:196-252 (model in train mode): 9 x (head forward, K-step PGD),
    9 mixed tail forwards `clean + w[i]*(adv - clean)` (:226-227), clean forward, loss = (clean + adv/9)/2 + l1*|w|_1
    (:240-245), both optimizers step, `w` projected back onto sum = 1 (:254-255)."""
    clean, adv = [], []
    for num in idx_list:
        fea = model(inp, end_point=num, start_point=0).detach()
        clean.append(fea)
        adv.append(PGD(fea, criterion, y=target, model=model, steps=steps, gamma=gamma / 255, start_idx=num,
                       layer_number=layer_number, eps=eps / 255, randinit=randinit, clip=clip))
    outs, l2s, linfs = [], [], []
    for i, num in enumerate(idx_list):
        l2, linf = perturb_norms(adv[i], clean[i])
        l2s.append(l2)
        linfs.append(linf)
        mixed = clean[i] + model.w[i] * (adv[i] - clean[i])
        outs.append(model(mixed, end_point=layer_number, start_point=num))
    out_clean = model(inp, end_point=layer_number, start_point=0)
    loss_adv = 0
    for o in outs:
        loss_adv = loss_adv + criterion(o, target)
    loss_clean = criterion(out_clean, target)
    l1 = torch.norm(model.w, p=1)
    loss = (loss_clean + loss_adv / len(idx_list)) / 2 + l1 * l1_coef
    optimizer.zero_grad()
    optimizer_w.zero_grad()
    loss.backward()
    optimizer.step()
    optimizer_w.step()
    with torch.no_grad():
        model.w.data = sum_project(model.w.data, K=len(idx_list))
    prec1 = (out_clean.argmax(dim=1) == target).float().sum() * (100.0 / target.shape[0])
    return {"loss": loss.detach(), "loss_clean": loss_clean.detach(), "loss_adv": loss_adv.detach(), "l1": l1.detach(),
            "l2": torch.stack(l2s), "linf": torch.stack(linfs), "prec1": prec1, "w": model.w.detach().clone(),
            "out_clean": out_clean.detach()}


# ----------------------------------------------------------------------- Segmentation operators and step (N1, first slice)
class _ConvBNReLU(nn.Sequential):
    def __init__(self, ci, co, k=3, stride=1):
        super().__init__(nn.Conv2d(ci, co, k, stride, k // 2, bias=False), nn.BatchNorm2d(co), nn.ReLU())


class TinySegNet(nn.Module):
    """A small network that follows the reference's dict-dispatch protocol exactly — `Segmentation/network/utils.py:14-47`
    (flag head / tail / clean, integer or "aspp|concat"_"head|tail" out_idx), `backbone/resnet.py:198-304` (head up to
    layer out_idx, tail from there, low_level = layer1 output) and `_deeplab.py:48-80` (project + aspp + classifier with
    the aspp/concat split) — on a 4-stage toy backbone.  Test infrastructure: it lets the reference's own
    `Segmentation/attack_algo.py` functions and the build's operators run on the same model; DeepLabv3+ itself is the
    next slice."""

    def __init__(self, classes=5, w=8):
        super().__init__()
        self.stem = _ConvBNReLU(3, w, 3, 2)
        self.layer1 = _ConvBNReLU(w, w)
        self.layer2 = _ConvBNReLU(w, 2 * w, 3, 2)
        self.layer3 = _ConvBNReLU(2 * w, 2 * w)
        self.layer4 = _ConvBNReLU(2 * w, 4 * w)
        self.project = _ConvBNReLU(w, w // 2, 1)
        self.aspp = _ConvBNReLU(4 * w, 2 * w, 1)
        self.classifier = nn.Sequential(_ConvBNReLU(2 * w + w // 2, 2 * w), nn.Conv2d(2 * w, classes, 1))

    def backbone(self, d):
        layers = [self.layer1, self.layer2, self.layer3, self.layer4]
        out = {}
        if d["flag"] == "head":
            x = self.layer1(self.stem(d["x"]))
            out["low_level"] = x
            for L in layers[1:d["out_idx"]]:
                x = L(x)
            out["out"] = x
            return out
        if d["flag"] == "tail":
            x = d["adv"]
            for L in layers[d["out_idx"]:]:
                x = L(x)
            return {"out": x, "low_level": d["low_level_feat"]}
        assert d["flag"] == "clean"
        x = self.layer1(self.stem(d["x"]))
        out["low_level"] = x
        for L in layers[1:]:
            x = L(x)
        out["out"] = x
        return out

    def head(self, f, return_type=None):
        import torch.nn.functional as F
        if return_type == "aspp_head":
            return self.aspp(f["out"])
        if return_type == "concat_tail":
            return self.classifier(f["adv"])
        low = self.project(f["low_level"])
        o = f["adv"] if return_type == "aspp_tail" else self.aspp(f["out"])
        o = F.interpolate(o, size=low.shape[2:], mode="bilinear", align_corners=False)
        cat = torch.cat([low, o], dim=1)
        if return_type == "concat_head":
            return cat
        assert return_type in (None, "aspp_tail")
        return self.classifier(cat)

    def forward(self, d):
        import torch.nn.functional as F
        if d["flag"] == "head":
            return self.backbone(d)
        assert d["flag"] in ("tail", "clean")
        if isinstance(d["out_idx"], int):
            x = self.head(self.backbone(d))
            return F.interpolate(x, size=d["x"].shape[-2:], mode="bilinear", align_corners=False)
        if d["out_idx"] in ("aspp_head", "concat_head"):
            f = self.backbone(d)
            f["adv"] = self.head(f, d["out_idx"])
            return f
        assert d["out_idx"] in ("aspp_tail", "concat_tail")
        x = self.head(d["adv"], d["out_idx"])
        return F.interpolate(x, size=d["x"].shape[-2:], mode="bilinear", align_corners=False)


def seg_PGD(x, image_batch, low_level_feat, criterion, y=None, model=None, steps=3, eps=None, gamma=None, idx=1,
            randinit=False, clip=False):
    """Segmentation/attack_algo.py:40-59."""
    x_adv = x.clone()
    if randinit:
        x_adv = randinit_(x_adv, eps)
    x_adv.requires_grad_(True)
    for _ in range(steps):
        logits = model({"x": image_batch, "adv": x_adv, "out_idx": idx, "flag": "tail", "low_level_feat": low_level_feat})
        grad0 = torch.autograd.grad(criterion(logits, y), x_adv, only_inputs=True)[0]
        with torch.no_grad():
            pgd_step_(x_adv.data, grad0.data, gamma, x, eps, clip)
    return x_adv


def seg_decoder_PGD(input_dict, image_batch, criterion, y=None, model=None, steps=3, eps=None, gamma=None, idx=1,
                    randinit=False, clip=False):
    """Segmentation/attack_algo.py:61-84.  (With clip=True the reference raises NameError: its projection refers to an
    undefined `x`; restated as such.)"""
    x_adv = input_dict["adv"].detach().clone()
    if randinit:
        x_adv = randinit_(x_adv, eps)
    x_adv.requires_grad_(True)
    input_dict["adv"] = x_adv
    for _ in range(steps):
        logits = model({"x": image_batch, "adv": input_dict, "out_idx": idx + "_tail", "flag": "clean"})
        grad = torch.autograd.grad(criterion(logits, y), x_adv, only_inputs=True)[0]
        with torch.no_grad():
            pgd_step_(x_adv.data, grad.data, gamma, None, eps, False)
        if clip:
            raise NameError("name 'x' is not defined")
    input_dict["adv"] = x_adv
    return input_dict


def seg_adv_input(x=None, criterion=None, y=None, model=None, steps=3, eps=None, gamma=None, randinit=False, clip=False):
    """Segmentation/attack_algo.py:86-105: image-space PGD, clamped to [0, 1] at the end."""
    x_adv = x.clone()
    if randinit:
        x_adv = randinit_(x_adv, eps)
    x_adv.requires_grad_(True)
    for _ in range(steps):
        logits = model({"x": x_adv, "adv": None, "out_idx": 0, "flag": "clean", "low_level_feat": None})
        grad0 = torch.autograd.grad(criterion(logits, y), x_adv, only_inputs=True)[0]
        with torch.no_grad():
            pgd_step_(x_adv.data, grad0.data, gamma, x, eps, clip)
    return torch.clamp(x_adv, 0, 1.0)


def seg_train_step(model, optimizer, criterion, images, labels, *, steps=1, eps=2.0, gamma_se=0.5, gamma_sd=0.5,
                   pertub_idx_se=3, pertub_idx_sd="aspp", mix_layer="11", mix_sd=True, randinit=False, clip=False):
    """One iteration of Segmentation/main_aug_final.py:158-232 (noise_sd = 0): SE feature PGD + SD decoder PGD, 3 SAT
    sample points, mix_feature, four forwards, loss = 0.7*clean + 0.1*(se1 + se2 + sd)."""
    f0, f1 = int(mix_layer[0]), int(mix_layer[1])
    optimizer.zero_grad()
    out_se = model({"x": images, "adv": None, "out_idx": pertub_idx_se, "flag": "head"})
    dec = model({"x": images, "adv": None, "out_idx": pertub_idx_sd + "_head", "flag": "clean"})
    fm_sd = dec["adv"].detach()
    low = out_se["low_level"]
    fm_se = out_se["out"].detach()
    adv_se = seg_PGD(fm_se, images, low, criterion, y=labels, model=model, steps=steps, eps=eps / 255,
                     gamma=gamma_se / 255, idx=pertub_idx_se, randinit=randinit, clip=clip)
    adv_sd_dict = seg_decoder_PGD(dec, images, criterion, y=labels, model=model, steps=steps, eps=eps / 255,
                                  gamma=gamma_sd / 255, idx=pertub_idx_sd, randinit=randinit, clip=clip)
    adv_sd = adv_sd_dict["adv"].detach()
    if mix_sd:
        adv_sd = mix_feature(fm_sd, adv_sd)
    adv_sd_dict["adv"] = adv_sd
    pts = get_sample_points(fm_se, adv_se, 3)
    if f0:
        pts[1] = mix_feature(fm_se, pts[1])
    if f1:
        pts[2] = mix_feature(fm_se, pts[2])
    o0 = model({"x": images, "adv": None, "out_idx": 0, "flag": "clean"})
    o1 = model({"x": images, "adv": pts[1], "out_idx": pertub_idx_se, "flag": "tail", "low_level_feat": low})
    o2 = model({"x": images, "adv": pts[2], "out_idx": pertub_idx_se, "flag": "tail", "low_level_feat": low})
    o3 = model({"x": images, "adv": adv_sd_dict, "out_idx": pertub_idx_sd + "_tail", "flag": "clean"})
    l0, l1, l2, l3 = (criterion(o, labels) for o in (o0, o1, o2, o3))
    loss = 0.7 * l0 + 0.1 * l1 + 0.1 * l2 + 0.1 * l3
    loss.backward()
    optimizer.step()
    return {"loss": loss.detach(), "losses": torch.stack([l0, l1, l2, l3]).detach(), "adv_se": adv_se.detach(),
            "adv_sd": adv_sd.detach(), "fm_se": fm_se, "fm_sd": fm_sd, "out_clean": o0.detach()}


# ------------------------------------------------------------- Detection (N2): step functions on a protocol-faithful toy detector
def synth_field(shape, seed):
    """A deterministic fp32 field in [-2, 2) from a closed form (splitmix64 of the flat index; 24 random bits per element, so
    every value is exact in fp32): lets a fixture name a 17 MB ROIAlign input by (shape, seed) instead of storing it.  Integer
    arithmetic only — independent of numpy's generator streams."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (((z >> np.uint64(40)).astype(np.float32) / np.float32(1 << 24)) * np.float32(4.0) - np.float32(2.0)).reshape(shape)


def roi_align_torch(inp, rois, output_size, spatial_scale, sampling_ratio):
    """Detection/support/src/cpu/ROIAlign_cpu.cpp:110-238 / cuda/ROIAlign_cuda.cu:10-122 (legacy, un-aligned ROIAlign) in
    differentiable torch ops: rois [K, 5] = (batch index, x1, y1, x2, y2) in image coordinates; roi extent clamped to
    >= 1; sampling_ratio > 0 samples per bin side (else ceil(roi size / bins)); a sample outside [-1, H] x [-1, W]
    contributes 0, coordinates are clamped to [0, size-1], bilinear weights from the clamped position; bin = mean of
    its samples.  Pinned to oracle_roi_align (C) in tests/test_det_oracle.py, which is itself pinned bit for bit to the
    reference's own CPU kernel (tests/golden/roi_align_fwd_*.npz)."""
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    n, c, h, w = inp.shape
    out = []
    for r in rois:
        b = int(r[0])
        x1, y1, x2, y2 = [float(v) * spatial_scale for v in r[1:]]
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        bh, bw = rh / ph, rw / pw
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / pw))
        ys = y1 + (torch.arange(ph, dtype=torch.float32)[:, None] + (torch.arange(gh, dtype=torch.float32)[None, :] + 0.5) / gh) * bh
        xs = x1 + (torch.arange(pw, dtype=torch.float32)[:, None] + (torch.arange(gw, dtype=torch.float32)[None, :] + 0.5) / gw) * bw
        ys, xs = ys.reshape(-1), xs.reshape(-1)                        # [ph*gh], [pw*gw]

        def axis(t, size):
            ok = (t >= -1.0) & (t <= size)
            t = t.clamp(min=0.0)
            lo = t.floor().long()
            hi = lo + 1
            top = lo >= size - 1
            lo = torch.where(top, torch.full_like(lo, size - 1), lo)
            hi = torch.where(top, torch.full_like(hi, size - 1), hi)
            t = torch.where(top, lo.float(), t)
            l = t - lo.float()
            return ok, lo, hi, l, 1.0 - l

        oky, ylo, yhi, ly, hy = axis(ys, h)
        okx, xlo, xhi, lx, hx = axis(xs, w)
        f = inp[b]                                                     # [c, h, w]
        g = lambda yy, xx: f[:, yy][:, :, xx]                          # [c, ph*gh, pw*gw]
        v = (g(ylo, xlo) * (hy[:, None] * hx[None, :]) + g(ylo, xhi) * (hy[:, None] * lx[None, :])
             + g(yhi, xlo) * (ly[:, None] * hx[None, :]) + g(yhi, xhi) * (ly[:, None] * lx[None, :]))
        v = v * (oky[:, None] & okx[None, :]).float()
        out.append(v.reshape(c, ph, gh, pw, gw).sum(dim=(2, 4)) / float(gh * gw))
    return torch.stack(out) if out else inp.new_zeros(0, c, ph, pw)


class TinyDetNet(nn.Module):
    """A small detector that follows the reference's dispatch protocol — `Detection/model.py:40-185` (flag head / tail /
    clean; integer out_idx or 'roi_head' / 'roi_tail'; training mode returns four per-image loss tensors; BatchNorm
    frozen in eval mode inside every training forward, `:46-47`) and `backbone/resnet101_ori.py:203-262` (head = up to
    layer out_idx, tail = the layers after it, out_idx 3 tail = identity) — on a three-stage toy backbone, a one-anchor
    RPN head and a ROIAlign + two-layer detection head over the ground-truth boxes.  Test infrastructure: it lets the
    reference's own `Detection/attack_algo.py` functions and the build's operators run on the same model; Faster-RCNN
    itself (rpn/, roi/, anchors, proposals) is out of scope.  `roi_align` is pluggable: the torch restatement above on
    the CPU, the library's HIP operator in the GPU tests."""

    def __init__(self, num_classes=4, w=8, roi_align=None):
        super().__init__()
        self.stem = _ConvBNReLU(3, w, 3, 2)
        self.layer1 = _ConvBNReLU(w, w)
        self.layer2 = _ConvBNReLU(w, 2 * w, 3, 2)
        self.layer3 = _ConvBNReLU(2 * w, 4 * w)
        self.rpn_obj = nn.Conv2d(4 * w, 1, 1)
        self.rpn_tr = nn.Conv2d(4 * w, 4, 1)
        self.hidden = nn.Linear(4 * w * 2 * 2, 16)
        self.cls = nn.Linear(16, num_classes)
        self.reg = nn.Linear(16, 4)
        self._bn_modules = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        self.roi_align = roi_align or roi_align_torch
        self.stride = 4

    def features(self, d):
        layers = [self.layer1, self.layer2, self.layer3]
        if d["flag"] == "head":
            assert d["out_idx"] in (1, 2, 3)
            x = self.stem(d["x"])
            for L in layers[:d["out_idx"]]:
                x = L(x)
            return x
        if d["flag"] == "tail":
            x = d["adv"]
            for L in layers[d["out_idx"]:]:
                x = L(x)
            return x
        assert d["flag"] == "clean"
        x = self.stem(d["x"])
        for L in layers:
            x = L(x)
        return x

    def rpn(self, f, bb):
        """Per-image objectness (BCE against 'pixel centre inside a ground-truth box') and transformer (smooth-L1 of the
        4 regression maps against the box's normalised centre offset / log size, on the positive pixels) losses."""
        import torch.nn.functional as F
        n, _, h, w = f.shape
        obj, tr = self.rpn_obj(f)[:, 0], self.rpn_tr(f)
        cy = (torch.arange(h, dtype=f.dtype, device=f.device) + 0.5) * self.stride
        cx = (torch.arange(w, dtype=f.dtype, device=f.device) + 0.5) * self.stride
        lo_obj, lo_tr = [], []
        for i in range(n):
            tgt = torch.zeros(h, w, dtype=f.dtype, device=f.device)
            reg = torch.zeros(4, h, w, dtype=f.dtype, device=f.device)
            for b in bb[i]:
                x1, y1, x2, y2 = [float(v) for v in b]
                m = ((cy[:, None] >= y1) & (cy[:, None] <= y2) & (cx[None, :] >= x1) & (cx[None, :] <= x2)).to(f.dtype)
                tgt = torch.maximum(tgt, m)
                bw_, bh_ = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
                t = torch.stack([((x1 + x2) / 2 - cx)[None, :].expand(h, w) / bw_, ((y1 + y2) / 2 - cy)[:, None].expand(h, w) / bh_,
                                 torch.full((h, w), float(np.log(bw_ / 16.0)), dtype=f.dtype, device=f.device),
                                 torch.full((h, w), float(np.log(bh_ / 16.0)), dtype=f.dtype, device=f.device)])
                reg = torch.where(m[None] > 0, t, reg)
            lo_obj.append(F.binary_cross_entropy_with_logits(obj[i], tgt))
            sl = F.smooth_l1_loss(tr[i], reg, reduction="none") * tgt[None]
            lo_tr.append(sl.sum() / tgt.sum().clamp(min=1.0))
        return torch.stack(lo_obj), torch.stack(lo_tr)

    def detection(self, f, bb=None, lb=None, return_type=None):
        import torch.nn.functional as F
        if return_type == "tail":
            d = f
            roi, bb, lb = d["roi_feature_map"], d["bboxes"], d["labels"]
        else:
            n, g = bb.shape[0], bb.shape[1]
            idx = torch.arange(n, dtype=bb.dtype, device=bb.device)[:, None].expand(n, g).reshape(-1, 1)
            rois = torch.cat([idx, bb.reshape(-1, 4)], dim=1)
            roi = self.roi_align(f, rois, (2, 2), 1.0 / self.stride, 2)
            if return_type == "head":
                return {"roi_feature_map": roi, "bboxes": bb, "labels": lb}
        n, g = bb.shape[0], bb.shape[1]
        hid = F.relu(self.hidden(roi.flatten(1)))
        ce = F.cross_entropy(self.cls(hid), lb.reshape(-1), reduction="none").reshape(n, g).mean(dim=1)
        tgt = torch.stack([(bb[..., 2] - bb[..., 0]) / 32.0, (bb[..., 3] - bb[..., 1]) / 32.0,
                           (bb[..., 0] + bb[..., 2]) / 64.0, (bb[..., 1] + bb[..., 3]) / 64.0], dim=-1).reshape(-1, 4)
        sl = F.smooth_l1_loss(self.reg(hid), tgt, reduction="none").sum(dim=1).reshape(n, g).mean(dim=1)
        return ce, sl

    def forward(self, input_dict, gt_bboxes_batch=None, gt_classes_batch=None):
        d = input_dict
        if d["flag"] == "head":
            for m in self._bn_modules:
                m.eval()
            return self.features(d)
        assert d["flag"] in ("tail", "clean") and self.training
        if type(d["out_idx"]) == int:
            for m in self._bn_modules:
                m.eval()
            f = self.features(d)
            ao, at = self.rpn(f, gt_bboxes_batch)
            pc, pt = self.detection(f, gt_bboxes_batch, gt_classes_batch)
            return ao, at, pc, pt
        if d["out_idx"] == "roi_head":
            for m in self._bn_modules:
                m.eval()
            f = self.features(d)
            ao, at = self.rpn(f, gt_bboxes_batch)
            return {"anchor_objectness_losses": ao, "anchor_transformer_losses": at,
                    "roi_output_dict": self.detection(f, gt_bboxes_batch, gt_classes_batch, return_type="head")}
        assert d["out_idx"] == "roi_tail"
        u = d["adv"]
        pc, pt = self.detection(u["roi_output_dict"], return_type="tail")
        return u["anchor_objectness_losses"], u["anchor_transformer_losses"], pc, pt


def det_compute_loss(l1, l2, l3, l4):
    """Detection/attack_algo.py:21-27."""
    return l1.mean() + l2.mean() + l3.mean() + l4.mean()


def det_PGD(x, image_batch, y=None, model=None, steps=3, eps=None, gamma=None, idx=1, randinit=False, clip=False):
    """Detection/attack_algo.py:48-74."""
    x_adv = x.clone()
    if randinit:
        x_adv += (2.0 * torch.rand(x_adv.shape) - 1.0) * eps
    x_adv.requires_grad_(True)
    for _ in range(steps):
        loss = det_compute_loss(*model.train().forward({"x": image_batch, "adv": x_adv, "out_idx": idx, "flag": "tail"}, y["bb"], y["lb"]))
        g = torch.autograd.grad(loss, x_adv, only_inputs=True)[0]
        x_adv.data.add_(gamma * torch.sign(g.data))
        if clip:
            linf_project_(x, eps, x_adv.data)
    return x_adv


def det_roi_PGD(rpn_roi_output_dict, y=None, model=None, steps=1, eps=None, gamma=None, randinit=False, clip=False,
                only_roi_loss=True):
    """Detection/attack_algo.py:77-116 (layer='roi'): perturbs roi_output_dict['roi_feature_map'] in the dict.  clip=True
    names an undefined `rpn_feature1` there (:110) and raises NameError after the first step — kept."""
    d = rpn_roi_output_dict
    x_adv = d["roi_output_dict"]["roi_feature_map"].detach().clone()
    if randinit:
        x_adv += (2.0 * torch.rand(x_adv.shape) - 1.0) * eps
    x_adv.requires_grad_(True)
    d["roi_output_dict"]["roi_feature_map"] = x_adv
    for _ in range(steps):
        ao, at, pc, pt = model.train().forward({"adv": d, "out_idx": "roi_tail", "flag": "clean"}, y["bb"], y["lb"])
        loss = (pc.mean() + pt.mean()) if only_roi_loss else (ao.mean() + at.mean() + pc.mean() + pt.mean())
        g = torch.autograd.grad(loss, x_adv, only_inputs=True)[0]
        x_adv.data.add_(gamma * torch.sign(g.data))
        if clip:
            raise NameError("name 'rpn_feature1' is not defined")
    d["roi_output_dict"]["roi_feature_map"] = x_adv
    return d


def det_adv_input(x=None, y=None, model=None, steps=3, eps=None, gamma=None, randinit=False, clip=False):
    """Detection/attack_algo.py:153-178: image-space PGD, clamped to [0, 1] at the end."""
    x_adv = x.clone()
    if randinit:
        x_adv += (2.0 * torch.rand(x_adv.shape) - 1.0) * eps
    x_adv.requires_grad_(True)
    for _ in range(steps):
        loss = det_compute_loss(*model.train().forward({"x": x_adv, "adv": None, "out_idx": -1, "flag": "clean"}, y["bb"], y["lb"]))
        g = torch.autograd.grad(loss, x_adv, only_inputs=True)[0]
        x_adv.data.add_(gamma * torch.sign(g.data))
        if clip:
            linf_project_(x, eps, x_adv.data)
    return torch.clamp(x_adv, 0, 1.0)


def det_train_step(model, optimizer, image_batch, bboxes_batch, labels_batch, loss_settings=1):
    """One iteration of Detection/train_aug_sat_muti_advt.py:70-172: adversarial image (5 steps, randinit, clip), three
    feature maps + the ROI dict, three one-step feature PGDs, five SAT sample points (two of them mixed), one-step ROI
    feature PGD + mix_feature, eight forwards, the weighted loss of :141-153."""
    y = {"bb": bboxes_batch, "lb": labels_batch}
    fwd = lambda d: model.train().forward(d, bboxes_batch, labels_batch)
    adv_image = det_adv_input(x=image_batch, y=y, model=model, steps=5, eps=(2.0 / 255), gamma=(0.3 / 255), randinit=True, clip=True)
    fm = [fwd({"x": image_batch, "adv": None, "out_idx": i, "flag": "head"}).detach() for i in (1, 2, 3)]
    rr = fwd({"x": image_batch, "adv": None, "out_idx": "roi_head", "flag": "clean"})
    clean_sd = rr["roi_output_dict"]["roi_feature_map"].detach()
    adv1 = det_PGD(fm[0], image_batch, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=1)
    adv2 = det_PGD(fm[1], image_batch, y=y, model=model, steps=1, eps=(0.1 / 255), gamma=(0.001 / 255), idx=2)
    adv3 = det_PGD(fm[2], image_batch, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(1.0 / 255), idx=3)
    pts = get_sample_points(fm[2], adv3, 5)
    pts[1] = mix_feature(fm[2], pts[1])
    pts[2] = mix_feature(fm[2], pts[2])
    adv_rr = det_roi_PGD(rr, y=y, model=model, steps=1, eps=(2.0 / 255), gamma=(0.2 / 255), only_roi_loss=False)
    adv_sd = mix_feature(clean_sd, adv_rr["roi_output_dict"]["roi_feature_map"].detach())
    adv_rr["roi_output_dict"]["roi_feature_map"] = adv_sd
    dicts = [{"x": adv_image, "adv": None, "out_idx": 0, "flag": "clean"},
             {"x": image_batch, "adv": adv1, "out_idx": 1, "flag": "tail"},
             {"x": image_batch, "adv": adv2, "out_idx": 2, "flag": "tail"}] + \
            [{"x": image_batch, "adv": pts[j], "out_idx": 3, "flag": "tail"} for j in (1, 2, 3, 4)] + \
            [{"adv": adv_rr, "out_idx": "roi_tail", "flag": "clean"}]
    L = [det_compute_loss(*fwd(d)) for d in dicts]
    loss_clean_adv = 0.9 * (0.2333 * (L[0] + L[3] + L[4] + L[5] + L[6]) + 0.1 * L[7]) + 0.05 * (L[1] + L[2])
    w = {1: (1.0, 0.0), 2: (0.5, 0.5), 3: (0.4, 0.6), 4: (0.3, 0.7)}[loss_settings]
    loss = loss_clean_adv if loss_settings == 1 else w[0] * loss_clean_adv + w[1] * L[0]
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return {"loss": loss.detach(), "losses": torch.stack(L).detach(), "adv_image": adv_image.detach(), "adv1": adv1.detach(),
            "adv2": adv2.detach(), "adv3": adv3.detach(), "adv_sd": adv_sd.detach(), "fm3": fm[2]}


# ------------------------------------------------------------- DeepLabv3+ (N1, second slice): the reference's network restated
class SegBottleneck(nn.Module):
    """Segmentation/network/backbone/resnet.py:76-119 (groups=1, base_width=64)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class SegResNet(nn.Module):
    """Segmentation/network/backbone/resnet.py:109-304: atrous ResNet with the head / tail / clean dispatch."""

    def __init__(self, layers, replace_stride_with_dilation=(False, False, False)):
        super().__init__()
        self.normal = ChannelNormalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2, replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(256, layers[2], 2, replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(512, layers[3], 2, replace_stride_with_dilation[2])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        downsample, prev = None, self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [SegBottleneck(self.inplanes, planes, stride, downsample, prev)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(SegBottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    def forward(self, d):
        stages = [self.layer1, self.layer2, self.layer3, self.layer4]
        out = {}
        if d["flag"] in ("head", "clean"):
            last = 4 if d["flag"] == "clean" else d["out_idx"]
            x = self.maxpool(self.relu(self.bn1(self.conv1(self.normal(d["x"])))))
            x = self.layer1(x)
            out["low_level"] = x
            for st in stages[1:last]:
                x = st(x)
            out["out"] = x
            return out
        assert d["flag"] == "tail"
        x = d["adv"]
        for st in stages[d["out_idx"]:]:
            x = st(x)
        return {"out": x, "low_level": d["low_level_feat"]}


class SegASPP(nn.Module):
    """Segmentation/network/_deeplab.py:143-193 (ASPPConv, ASPPPooling, ASPP)."""

    def __init__(self, cin, rates):
        super().__init__()
        mods = [nn.Sequential(nn.Conv2d(cin, 256, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True))]
        for r in rates:
            mods.append(nn.Sequential(nn.Conv2d(cin, 256, 3, padding=r, dilation=r, bias=False), nn.BatchNorm2d(256),
                                      nn.ReLU(inplace=True)))
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, 256, 1, bias=False), nn.BatchNorm2d(256),
                                  nn.ReLU(inplace=True)))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(5 * 256, 256, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
                                     nn.Dropout(0.1))

    def forward(self, x):
        res = [c(x) for c in list(self.convs)[:4]]
        res.append(F.interpolate(self.convs[4](x), size=x.shape[-2:], mode="bilinear", align_corners=False))
        return self.project(torch.cat(res, dim=1))


class SegHeadV3Plus(nn.Module):
    """Segmentation/network/_deeplab.py:28-90."""

    def __init__(self, cin, low_level_channels, num_classes, rates):
        super().__init__()
        self.project = nn.Sequential(nn.Conv2d(low_level_channels, 48, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU(inplace=True))
        self.aspp = SegASPP(cin, rates)
        self.classifier = nn.Sequential(nn.Conv2d(304, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
                                        nn.Conv2d(256, num_classes, 1))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, f, return_type=None):
        if return_type == "aspp_head":
            return self.aspp(f["out"])
        if return_type == "concat_tail":
            return self.classifier(f["adv"])
        low = self.project(f["low_level"])
        hi = f["adv"] if return_type == "aspp_tail" else self.aspp(f["out"])
        cat = torch.cat([low, F.interpolate(hi, size=low.shape[2:], mode="bilinear", align_corners=False)], dim=1)
        if return_type == "concat_head":
            return cat
        assert return_type in (None, "aspp_tail")
        return self.classifier(cat)


class SegDeepLabV3Plus(nn.Module):
    """Segmentation/network/utils.py:8-47 (`_SimpleSegmentationModel`) around the two modules above; built as
    network/modeling.py:6-29 does (output_stride 16: dilate layer4, ASPP rates 6/12/18; 8: layers 3-4, rates 12/24/36)."""

    def __init__(self, layers=(3, 4, 23, 3), num_classes=21, output_stride=16):
        super().__init__()
        rswd, rates = ((False, True, True), (12, 24, 36)) if output_stride == 8 else ((False, False, True), (6, 12, 18))
        self.backbone = SegResNet(layers, rswd)
        self.classifier = SegHeadV3Plus(2048, 256, num_classes, rates)

    def forward(self, d):
        if d["flag"] == "head":
            return self.backbone(d)
        assert d["flag"] in ("tail", "clean")
        if isinstance(d["out_idx"], int):
            x = self.classifier(self.backbone(d))
            return F.interpolate(x, size=d["x"].shape[-2:], mode="bilinear", align_corners=False)
        if d["out_idx"] in ("aspp_head", "concat_head"):
            f = self.backbone(d)
            f["adv"] = self.classifier(f, return_type=d["out_idx"])
            return f
        assert d["out_idx"] in ("aspp_tail", "concat_tail")
        x = self.classifier(d["adv"], return_type=d["out_idx"])
        return F.interpolate(x, size=d["x"].shape[-2:], mode="bilinear", align_corners=False)


def deeplabv3plus_resnet101(num_classes=21, output_stride=16):
    return SegDeepLabV3Plus((3, 4, 23, 3), num_classes, output_stride)


def deeplabv3plus_resnet50(num_classes=21, output_stride=16):
    return SegDeepLabV3Plus((3, 4, 6, 3), num_classes, output_stride)


def seg_make_optimizer(model, lr=0.01, weight_decay=1e-4):
    """Segmentation/main_aug_final.py:77-82: backbone BN momentum 0.01; SGD with two groups (backbone 0.1*lr, head lr)."""
    for m in model.backbone.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = 0.01
    return torch.optim.SGD(params=[{"params": model.backbone.parameters(), "lr": 0.1 * lr},
                                   {"params": model.classifier.parameters(), "lr": lr}], lr=lr, momentum=0.9,
                           weight_decay=weight_decay)


def poly_lr(base_lr, it, max_iters, power=0.9, min_lr=1e-6):
    """Segmentation/utils/scheduler.py:3-12 after `it` scheduler steps."""
    return max(base_lr * (1 - it / max_iters) ** power, min_lr)


def sharded_train_step(model, optimizer, criterion, inp, target, world, **kw):
    """N-GPU data-parallel emulation: each rank runs the step on its shard with per-shard BN
    statistics from the SAME starting weights, parameter gradients are averaged, one SGD update is applied.
    BN buffers of rank 0 persist (mirrors nn.DataParallel's replica-0 rule)."""
    import copy
    n = inp.shape[0] // world
    grads, outs, buffers0 = None, [], None
    state0 = copy.deepcopy(model.state_dict())
    for r in range(world):
        model.load_state_dict(state0)
        xs, ys = inp[r * n:(r + 1) * n], target[r * n:(r + 1) * n]
        fm = model(xs, end_point=kw["perturb_idx"], start_point=0).detach()
        fa = PGD(fm, criterion, y=ys, model=model, steps=kw["steps"], gamma=kw["gamma"] / 255,
                 start_idx=kw["perturb_idx"], layer_number=kw["layer_number"], eps=kw["eps"] / 255,
                 randinit=False, clip=kw.get("clip", False))
        oa = model(fa, end_point=kw["layer_number"], start_point=kw["perturb_idx"])
        oc = model(xs, end_point=kw["layer_number"], start_point=0)
        loss = (criterion(oa, ys) + criterion(oc, ys)) / 2
        optimizer.zero_grad()
        loss.backward()
        g = [None if p.grad is None else p.grad.detach().clone() for p in model.parameters()]
        grads = g if grads is None else [a if b is None else a + b for a, b in zip(grads, g)]
        outs.append(loss.detach())
        if r == 0:
            buffers0 = {k: v.clone() for k, v in model.named_buffers()}
    for p, g in zip(model.parameters(), grads):
        p.grad = None if g is None else g / world
    for k, v in model.named_buffers():
        v.copy_(buffers0[k])
    optimizer.step()
    return torch.stack(outs)
