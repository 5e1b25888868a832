"""Flat parameter arena + fused SGD (afan_sgd_step).

All trainable tensors of the model are views into ONE fp32 buffer; their gradients are views into a
second, the momentum into a third, and (bf16 backbone) a bf16 shadow of the parameters into a fourth,
which the SGD kernel refreshes in the same pass.  Consequences on MI355X:
  * optimizer.step() is one launch over ~N floats instead of ~3 launches per tensor;
  * the data-parallel exchange is an all-reduce of a few large contiguous chunks of `grad`
    (no bucket copy-in/copy-out), sized for xGMI's per-link bandwidth;
  * the learning rate lives in device memory, so warm-up (main_perturb.py:288-293) changes it without
    re-recording a captured step.
Semantics follow torch.optim.SGD as configured at Classification/main_perturb.py:72-74.  Parameters that
never receive a gradient (the reference's unused `w`, resnet_s.py:113-114) stay outside the arena, exactly
as torch.optim.SGD skips tensors whose .grad is None.
"""
import torch

from . import ops
from .resnet_s import Conv2d

_ALIGN = 64  # floats: every tensor starts on a 256-byte boundary of the arena


class ParamArena:
    def __init__(self, model, skip=("w",), bf16_shadow=None):
        named = [(n, p) for n, p in model.named_parameters() if n not in skip and p.requires_grad]
        if not named:
            raise ValueError("no parameters to manage")
        dev = named[0][1].device
        if dev.type != "cuda":
            raise ops.AfanLibraryError("ParamArena needs the model on the MI355X")
        self.names = [n for n, _ in named]
        self.offsets, off = [], 0
        for _, p in named:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.param)
        self.momentum_buf = torch.zeros_like(self.param)
        if bf16_shadow is None:
            bf16_shadow = getattr(model, "compute_dtype", torch.float32) == torch.bfloat16
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev) if bf16_shadow else None
        self.lr = torch.zeros(1, dtype=torch.float32, device=dev)
        self.params = []
        for (n, p), o in zip(named, self.offsets):
            view = self.param[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + p.numel()].view_as(p)
            self.params.append(p)
        if self.shadow is not None:
            ops.cast_bf16(self.param, self.shadow)
            mods = dict(model.named_modules())
            for (n, p), o in zip(named, self.offsets):
                m = mods.get(n.rsplit(".", 1)[0])
                if isinstance(m, Conv2d) and n.endswith(".weight"):
                    m._arena_shadow = self.shadow[o:o + p.numel()].view_as(p)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # re-attach views if someone set .grad = None
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def chunks(self, n_chunks):
        """Contiguous [start, end) ranges of the arena, cut at tensor boundaries, in BACKWARD order of use
        (last layers first) — the units of the overlapped gradient all-reduce."""
        bounds = self.offsets + [self.numel]
        target = self.numel / max(n_chunks, 1)
        cuts, acc_start = [], len(self.offsets)
        end = self.numel
        i = len(self.offsets) - 1
        while i >= 0:
            if end - bounds[i] >= target or i == 0:
                cuts.append((bounds[i], end, i))
                end = bounds[i]
            i -= 1
        return cuts  # (start, end, index of the first parameter in the chunk)


class ArenaSGD:
    """torch.optim.SGD-compatible surface (param_groups[0]['lr'], step, zero_grad, state_dict) on a ParamArena."""

    def __init__(self, arena, lr, momentum=0.9, weight_decay=5e-4):
        self.arena = arena
        self.param_groups = [{"lr": float(lr), "momentum": float(momentum), "weight_decay": float(weight_decay),
                              "dampening": 0, "nesterov": False, "params": list(range(len(arena.params)))}]
        self._lr_on_device = None
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def _sync_lr(self):
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self.arena.lr.fill_(lr)
            self._lr_on_device = lr

    def step(self):
        g = self.param_groups[0]
        self._sync_lr()
        a = self.arena
        ops.sgd_step_(a.param, a.grad, a.momentum_buf, a.lr, g["momentum"], g["weight_decay"], self.grad_scale,
                      a.shadow)

    def state_dict(self):
        """Same layout torch.optim.SGD.state_dict() produces (main_perturb.py:124,132 stores it in checkpoints)."""
        a = self.arena
        state = {i: {"momentum_buffer": a.momentum_buf[o:o + p.numel()].view_as(p).clone()}
                 for i, (p, o) in enumerate(zip(a.params, a.offsets))}
        groups = [dict(self.param_groups[0])]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        a = self.arena
        for i, (p, o) in enumerate(zip(a.params, a.offsets)):
            st = sd["state"].get(i)
            buf = a.momentum_buf[o:o + p.numel()].view_as(p)
            if st is not None and st.get("momentum_buffer") is not None:
                buf.copy_(st["momentum_buffer"])
            else:
                buf.zero_()
        for k in ("lr", "momentum", "weight_decay"):
            self.param_groups[0][k] = sd["param_groups"][0][k]
        self._lr_on_device = None
