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
from .resnet_s import Conv2d, _Flags

_ALIGN = 64  # floats: every tensor starts on a 256-byte boundary of the arena


class ParamArena:
    def __init__(self, model, skip=("w",), bf16_shadow=None, allow_cpu=False, channels_last=None, index_root=None):
        """index_root: the module whose .parameters() the REFERENCE hands to torch.optim.SGD (default: the model itself,
        main_perturb.py:72; main_learnable.py:82 passes model.sequential_model) — optimizer state is keyed by a
        parameter's position in that list (ArenaSGD.state_dict)."""
        all_named = list(model.named_parameters())
        named = [(n, p) for n, p in all_named if n not in skip and p.requires_grad]
        if not named:
            raise ValueError("no parameters to manage")
        dev = named[0][1].device
        if dev.type != "cuda" and not allow_cpu:
            raise ops.AfanLibraryError("ParamArena needs the model on the MI355X (allow_cpu=True only lays out the "
                                       "buffers for host-side tests; no kernel can run on them)")
        self.names = [n for n, _ in named]
        # index of every managed parameter in model.parameters() order (what torch.optim.SGD(model.parameters())
        # uses as state keys, main_perturb.py:72 — the skipped `w` is index 0 there)
        if index_root is None:
            root_ids = [id(p) for _, p in all_named]
        else:
            root_ids = [id(p) for p in index_root.parameters()]
        self.model_index = [root_ids.index(id(p)) for _, p in named]     # ValueError: a managed parameter outside the root
        self.n_model_params = len(root_ids)
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
        # 4-D (convolution) weights live in the arena in KRSC order (logical [K,C,R,S] with channels-last strides):
        # K = (r, s, c) contiguous is what the implicit-GEMM convolutions read; SGD is element-wise, so it does not care.
        if channels_last is None:
            channels_last = bool(getattr(model, "channels_last", False))
        self.channels_last = channels_last
        for (n, p), o in zip(named, self.offsets):
            view = self._view(self.param, o, p)
            view.copy_(p.data)
            p.data = view
            p.grad = self._view(self.grad, o, p)
            p._afan_arena_grad = not allow_cpu   # kernels may accumulate straight into .grad (resnet_s._accumulates_in_place)
            self.params.append(p)
        self.shadow_t, self._tdesc, self._t_ndesc, self._t_tiles = None, None, 0, 0
        if self.shadow is not None:
            mods = dict(model.named_modules())
            desc, t_off, tiles = [], 0, 0
            for (n, p), o in zip(named, self.offsets):
                m = mods.get(n.rsplit(".", 1)[0])
                if isinstance(m, Conv2d) and n.endswith(".weight"):
                    m._arena_shadow = self._view(self.shadow, o, p)
                    k_, c_, r_, s_ = p.shape
                    if self.channels_last and k_ % 16 == 0 and c_ % 16 == 0:   # the shapes the MFMA dgrad kernels take
                        desc.append((o, t_off, k_, r_ * s_, c_, tiles, m, p))
                        t_off += p.numel()
                        tiles += ((k_ + 63) // 64) * r_ * s_ * ((c_ + 63) // 64)
            # a residual block's 3x3 / stride-2 convolution and its 1x1 / stride-2 projection also share ONE [C][10][K]
            # operand (slots 0-8: the 3x3 taps, slot 9: the projection) for afan_conv_dgrad_sc_nhwc_bf16
            by_mod = {id(d[6]): d for d in desc}
            pairs = []
            for blk in model.modules():
                if getattr(blk, "_sc_kind", None) == "conv" and hasattr(blk, "_chain"):
                    c1, csc = blk._chain()[0][0], blk.shortcut[0]
                    d1, dsc = by_mod.get(id(c1)), by_mod.get(id(csc))
                    if (d1 is not None and dsc is not None and c1.kernel_size == (3, 3) and csc.kernel_size == (1, 1)
                            and c1.stride == (2, 2) and csc.stride == (2, 2) and c1.dilation == (1, 1)
                            and c1.out_channels == csc.out_channels and c1.in_channels == csc.in_channels
                            # afan_conv_dgrad_sc_nhwc_bf16 has no small-channel form (it declines ci < 40 or co < 40, like
                            # the plain dgrad's tiled kernels): narrower option-B blocks keep the two-launch backward
                            and c1.in_channels >= 40 and c1.out_channels >= 40):
                        pairs.append((c1, d1, dsc, t_off))
                        t_off += c1.in_channels * 10 * c1.out_channels
            if desc:
                # CRSK copies of the conv weights (dgrad operands), rebuilt by ONE launch after every SGD step
                self.shadow_t = torch.zeros(t_off, dtype=torch.bfloat16, device=dev)
                rows = [list(d[:6]) + [d[3], 0] for d in desc]
                for (c1, d1, dsc, to10) in pairs:
                    for d, rs0 in ((d1, 0), (dsc, 9)):
                        o, _, k_, rs, c_ = d[:5]
                        rows.append([o, to10, k_, rs, c_, tiles, 10, rs0])
                        tiles += ((k_ + 63) // 64) * rs * ((c_ + 63) // 64)
                    c1._arena_wt10 = self.shadow_t[to10:to10 + c1.in_channels * 10 * c1.out_channels]
                self._tdesc = torch.tensor(rows, dtype=torch.int64, device=dev).contiguous()
                self._t_ndesc, self._t_tiles = len(rows), tiles
                for (o, to, k_, rs, c_, _, m, p) in desc:
                    r_ = p.shape[2]
                    m._arena_wt = self.shadow_t[to:to + p.numel()].view(c_, r_, rs // r_, k_).permute(0, 3, 1, 2)
            self.refresh_shadow()

    def refresh_shadow(self):
        """Re-derive the bf16 shadow from the fp32 parameters (after load_state_dict / manual edits)."""
        if self.shadow is not None:
            ops.cast_bf16(self.param, self.shadow)
            self.refresh_transposed()
        _Flags.weight_epoch += 1

    def refresh_transposed(self):
        if self.shadow_t is not None:
            ops.transpose_weights(self.shadow, self.shadow_t, self._tdesc, self._t_ndesc, self._t_tiles)

    def _view(self, buf, o, p):
        flat = buf[o:o + p.numel()]
        if self.channels_last and p.dim() == 4:
            k, c, r, s_ = p.shape
            return flat.view(k, r, s_, c).permute(0, 3, 1, 2)
        return flat.view(p.shape)

    def view(self, buf, i):
        return self._view(buf, self.offsets[i], self.params[i])

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # re-attach views if someone set .grad = None
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self._view(self.grad, o, p)


class ArenaSGD(torch.optim.Optimizer):
    """torch.optim.SGD's surface (param_groups, step, zero_grad, state_dict in the SAME layout, so lr schedulers
    and the reference's checkpoints work) executed as ONE afan_sgd_step launch per parameter group over the arena.

    groups: None = one group over everything (main_perturb.py:72-74), or [(name_prefix, lr), ...] — consecutive runs of
    the arena's parameters by name prefix, each with its own learning rate (Segmentation/main_aug_final.py:79-82:
    backbone 0.1*lr, classifier lr).  Every group's lr lives in device memory (arena.lr[g])."""

    def __init__(self, arena, lr, momentum=0.9, weight_decay=5e-4, groups=None):
        self.arena = arena
        base = dict(lr=float(lr), momentum=float(momentum), dampening=0, weight_decay=float(weight_decay), nesterov=False)
        if groups is None:
            spec = [dict(params=arena.params)]
            self._ranges = [(0, arena.numel)]
            self._members = [list(range(len(arena.params)))]
        else:
            spec, self._ranges, self._members = [], [], []
            bounds = arena.offsets + [arena.numel]
            i = 0
            for prefix, glr in groups:
                j = i
                while j < len(arena.names) and arena.names[j].startswith(prefix):
                    j += 1
                if j == i:
                    raise ValueError(f"no parameters (left) with prefix {prefix!r}: groups must follow the arena's order")
                spec.append(dict(params=arena.params[i:j], lr=float(glr)))
                self._ranges.append((bounds[i], bounds[j]))
                self._members.append(list(range(i, j)))
                i = j
            if i != len(arena.names):
                raise ValueError("parameter groups do not cover the arena")
        super().__init__(spec, base)
        if arena.lr.numel() < len(self.param_groups):
            arena.lr = torch.zeros(len(self.param_groups), dtype=torch.float32, device=arena.param.device)
        self._lr_on_device = [None] * len(self.param_groups)
        self.grad_scale = 1.0
        if ops.GRID_BN_ALLOWED_AT_IMPORT and arena.param.is_cuda:
            ops.grid_guard_word(arena.param.device)      # the update's device-side guard (ops.sgd_step_): allocated outside any capture

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def _sync_lr(self):
        for gi, g in enumerate(self.param_groups):
            lr = float(g["lr"])
            if lr != self._lr_on_device[gi]:
                self.arena.lr[gi:gi + 1].fill_(lr)
                self._lr_on_device[gi] = lr

    @torch.no_grad()
    def step(self, closure=None):
        if not torch.cuda.is_current_stream_capturing():
            self._sync_lr()     # (inside a capture the fill would be frozen into the graph: callers sync before replay)
        a = self.arena
        for gi, (g, (lo, hi)) in enumerate(zip(self.param_groups, self._ranges)):
            ops.sgd_step_(a.param[lo:hi], a.grad[lo:hi], a.momentum_buf[lo:hi], a.lr[gi:gi + 1], g["momentum"],
                          g["weight_decay"], self.grad_scale, a.shadow[lo:hi] if a.shadow is not None else None)
        a.refresh_transposed()
        _Flags.weight_epoch += 1   # module-level caches of transposed weights (no arena) are stale now

    def state_dict(self):
        """Layout of torch.optim.SGD(<root>.parameters()).state_dict() (main_perturb.py:124,132 store it): state keyed
        by the parameter's index in the root's parameter list (ParamArena index_root); parameters outside the arena
        (`w`) have no state.  With several groups: each group lists its own indices, as torch does."""
        a = self.arena
        state = {mi: {"momentum_buffer": a.view(a.momentum_buf, i).clone()} for i, mi in enumerate(a.model_index)}
        out_groups = []
        for gi, g in enumerate(self.param_groups):
            group = {k: v for k, v in g.items() if k != "params"}
            if len(self.param_groups) == 1:
                group["params"] = list(range(a.n_model_params))
            else:
                group["params"] = [a.model_index[i] for i in self._members[gi]]
            out_groups.append(group)
        return {"state": state, "param_groups": out_groups}

    def load_state_dict(self, sd):
        a = self.arena
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for i, mi in enumerate(a.model_index):
            st = sd["state"].get(mi)
            buf = a.view(a.momentum_buf, i)
            if st is not None and st.get("momentum_buffer") is not None:
                buf.copy_(st["momentum_buffer"])
            else:
                buf.zero_()
        for g, src in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "momentum", "weight_decay"):
                g[k] = src[k]
            if "initial_lr" in src:
                g["initial_lr"] = src["initial_lr"]
        self._lr_on_device = [None] * len(self.param_groups)
