"""The A-FAN joint training step (Classification/main_perturb.py:165-209), one process per MI355X.

    head fwd (detached) -> K-step feature PGD -> [norms fused in the last PGD step, kept on device]
    -> adv tail fwd + clean full fwd -> (CE_adv + CE_clean)/2 -> backward -> SGD

Order of the forwards, train-mode BN side effects (head BN: 2 updates, tail BN: K+2 updates per
iteration) and the optimizer semantics follow the reference line by line; what changes is
execution: no host synchronisation inside the step (the reference copies the whole perturbation to the
host every iteration, main_perturb.py:190, and calls .item() twice, :208-209), fused HIP kernels for
everything that is not a convolution, one fused SGD launch over the parameter arena, and — with
world_size > 1 — a chunked RCCL all-reduce of the flat gradient arena that overlaps the rest of the
backward (data parallel: the PGD loop itself needs no communication).
"""
import torch
import torch.distributed as dist

from . import ops
from .arena import ArenaSGD, ParamArena
from .attack_algo import PGD, last_norms
from .grid_guard import GuardedTrainer


def warmup_lr(step, optimizer, warm_up_steps=200, max_lr=0.1):
    """main_perturb.py:288-293."""
    lr = step * max_lr / (warm_up_steps - 1)
    lr = min(lr, max_lr)
    for p in optimizer.param_groups:
        p["lr"] = lr
    return lr


class GradAllReducer:
    """Chunked all-reduce (SUM; the 1/world factor is folded into afan_sgd_step's grad_scale) of the flat
    gradient arena on a side stream.  A chunk is launched as soon as autograd has accumulated the gradient
    of the chunk's FIRST parameter (the last one the backward reaches), so the exchange of the deep layers
    overlaps the backward of the shallow ones.  Few, large, contiguous messages: xGMI links are
    point-to-point (~153 GB/s each), per-message latency matters more than on a switch."""

    def __init__(self, arena, n_chunks=4, group=None):
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.chunks = _cut_chunks(arena, n_chunks)
        self.on_cuda = arena.grad.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=arena.grad.device) if self.on_cuda else None
        self._pending = []
        self._hooks = []
        self._need, self._done = [], []
        # diag = True: every exchange is bracketed by events on the exchange stream and the compute stream's wait for it in
        # finish() by events on the compute stream; last_diag() turns them into milliseconds (bench.py's N > 1 line)
        self.diag = False
        self._diag_ranges, self._diag_wait, self._diag_host = [], None, []
        if self.world > 1:
            # a chunk is ready when EVERY parameter in it has its final gradient (head parameters finish during the
            # clean pass, tail parameters only after the adv pass too), so count completions per chunk.
            bounds = arena.offsets + [arena.numel]
            for ci, (start, end, first_param) in enumerate(self.chunks):
                members = [i for i in range(len(arena.params)) if start <= bounds[i] < end]
                self._need.append(len(members))
                self._done.append(0)
                for i in members:
                    self._hooks.append(arena.params[i].register_post_accumulate_grad_hook(self._make_hook(ci)))

    enabled = False

    def _make_hook(self, ci):
        def hook(_param):
            if not self.enabled:
                return
            self._done[ci] += 1
            if self._done[ci] == self._need[ci]:
                start, end, _ = self.chunks[ci]
                self._launch(start, end)
        return hook

    def begin(self, explicit=False):
        """explicit=True: the caller announces finished ranges with launch_params(); the autograd hooks stay off."""
        self.enabled = not explicit
        self._done = [0] * len(self.chunks)
        self._covered = []
        self._diag_ranges, self._diag_wait, self._diag_host = [], None, []
        ops.exchange_in_flight(False)

    def launch_params(self, p_lo, p_hi):
        """Start the exchange of the gradients of parameters [p_lo, p_hi) (arena order) NOW, on the side stream: the
        caller guarantees that every kernel writing them has been issued on the current stream.  The library's layers
        add their parameter gradients straight into the arena (no AccumulateGrad node runs, so no autograd hook can
        tell when a gradient is final): the step announces finished ranges itself, segment by segment of its backward."""
        if p_hi <= p_lo:
            return
        bounds = self.arena.offsets + [self.arena.numel]
        start, end = bounds[p_lo], bounds[p_hi]
        self._covered.append((start, end))
        self._launch(start, end)

    def _launch(self, start, end):
        buf = self.arena.grad[start:end]
        # from here to finish() RCCL's channel kernels may sit on CUs beside the compute stream's launches: no grid barrier may be
        # issued meanwhile (the in-launch BatchNorm needs every workgroup of its launch resident: it would wait for the exchange to END)
        ops.exchange_in_flight(True)
        if self.on_cuda:
            ev = torch.cuda.Event(enable_timing=self.diag)
            ev.record(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                if self.diag:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.diag:
                    if self.world > 1 and dist.get_backend(self.group) != "nccl":
                        w.wait()                   # (gloo on device tensors: the copy back is host-driven; bracket it whole)
                    e1.record(self.stream)
                    self._diag_ranges.append((start, end, ev, e0, e1))
        else:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(w)

    _covered = ()

    def finish(self):
        self.enabled = False
        if self._covered:                                    # explicit ranges: reduce whatever they left out, then wait
            pos = 0
            for start, end in sorted(self._covered):
                if start > pos:
                    self._launch(pos, start)
                pos = max(pos, end)
            if pos < self.arena.numel:
                self._launch(pos, self.arena.numel)
            self._covered = []
        else:
            for ci, (start, end, _) in enumerate(self.chunks):  # parameters that got no gradient this step
                if self._done[ci] < self._need[ci]:
                    self._launch(start, end)
        cur = torch.cuda.current_stream(self.arena.grad.device) if self.on_cuda else None
        if self.diag and self.on_cuda:
            wa, wb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            wa.record(cur)
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if self.on_cuda:
            cur.wait_stream(self.stream)
            if self.diag:
                wb.record(cur)
                self._diag_wait = (wa, wb)
        ops.exchange_in_flight(False)

    def note_host_gap(self, seconds):
        """(diag) host time between two graph replays around an exchange launch."""
        if self.diag:
            self._diag_host.append(seconds)

    def last_diag(self):
        """After a synchronize: the last step's exchanges — per announced range its bytes, the time from the compute stream's
        announcement to the exchange's start on its stream (queueing behind earlier ranges) and the exchange's own duration;
        `exposed_allreduce_ms` = how long the compute stream stood in finish() before the optimizer could run; `host_gap_us`
        = host time between consecutive graph replays (the RCCL launch sits in it)."""
        if not (self.diag and self.on_cuda and self._diag_wait):
            return None
        rs = [{"bytes": (end - start) * 4, "queued_ms": round(ev.elapsed_time(e0), 3), "allreduce_ms": round(e0.elapsed_time(e1), 3)}
              for start, end, ev, e0, e1 in self._diag_ranges]
        return {"ranges": rs, "exposed_allreduce_ms": round(self._diag_wait[0].elapsed_time(self._diag_wait[1]), 3),
                "host_gap_us": [round(h * 1e6, 1) for h in self._diag_host]}


class NullReducer:
    """GradAllReducer's surface with no exchange behind it: `AfanTrainer(emulate_dp=True)` on ONE GPU then runs exactly the program a
    rank of a data-parallel job runs — the backward cut into phases, one hipGraph per phase with a host call between two replays, the
    in-launch BatchNorm off from the first announced range on — so that its cost can be timed without a second GPU (bench.py's
    `dp_schedule`) and the multi-GPU efficiency anchored on the right single-GPU number.  Records what was announced."""
    world = 1
    diag = False
    stream = None

    def __init__(self, arena):
        self.arena = arena
        self.announced = []             # (p_lo, p_hi) of the last step, in order
        self.fused_while_in_flight = 0  # convolution + BatchNorm launches issued between the first announcement and finish() (must stay 0)
        self._mark = None

    def begin(self, explicit=False):
        self.announced, self._mark = [], None
        ops.exchange_in_flight(False)

    def launch_params(self, p_lo, p_hi):
        if p_hi <= p_lo:
            return
        self.announced.append((p_lo, p_hi))
        if self._mark is None:
            self._mark = ops.CALLS["conv_bn_fused"]
        ops.exchange_in_flight(True)

    def finish(self):
        if self._mark is not None:
            self.fused_while_in_flight += ops.CALLS["conv_bn_fused"] - self._mark
        ops.exchange_in_flight(False)

    def note_host_gap(self, seconds):
        pass

    def last_diag(self):
        return None


def _cut_chunks(arena, n_chunks):
    """[start, end, first_param_index) ranges cut at tensor boundaries, roughly equal in size."""
    bounds = arena.offsets + [arena.numel]
    target = arena.numel / max(n_chunks, 1)
    cuts, end = [], arena.numel
    for i in range(len(arena.offsets) - 1, -1, -1):
        if end - bounds[i] >= target or i == 0:
            cuts.append((bounds[i], end, i))
            end = bounds[i]
    return cuts


def _out_shape(mod, c, h, w):
    """(C, H, W) after one entry of the model's flat Sequential (convolutions / pooling change it; the rest keep it)."""
    import torch.nn as nn
    if isinstance(mod, nn.Conv2d):
        st = mod.stride[0]
        return mod.out_channels, (h + 2 * mod.padding[0] - mod.kernel_size[0]) // st + 1, (w + 2 * mod.padding[1] - mod.kernel_size[1]) // st + 1
    if isinstance(mod, nn.MaxPool2d):
        k, st, p = mod.kernel_size, mod.stride, mod.padding
        return c, (h + 2 * p - k) // st + 1, (w + 2 * p - k) // st + 1
    if hasattr(mod, "_chain"):           # residual block: the main chain's strides and its last conv's channels
        for conv, _ in mod._chain():
            st = conv.stride[0]
            h, w = (h + st - 1) // st, (w + st - 1) // st
            c = conv.out_channels
        return c, h, w
    return c, h, w


class AfanTrainer(GuardedTrainer):
    """Owns the arena, the optimizer and the per-step schedule of one rank.

    use_graph=True (default): after `graph_warmup` eager iterations the whole iteration body — head forward, K PGD
    steps, both final forwards, backward and (single GPU) the SGD step, ~1300 launches — is captured once into a
    hipGraph and replayed from then on: static shapes, the learning rate in device memory and on-device metrics make
    the step capturable, and replay removes the host launch cost that otherwise bounds the step (PyTorch's eager
    dispatch + autograd tape is ~15 us per launch).  With world_size > 1 the graph stops after the backward and the
    gradient all-reduce + SGD run eagerly.  randinit draws on the host generator every step, so it stays eager."""

    def __init__(self, model, criterion, *, steps=5, gamma=0.5, eps=2.0, perturb_idx=13, layer_number=None,
                 randinit=False, clip=False, lr=0.1, momentum=0.9, weight_decay=5e-4, allreduce_chunks=4,
                 group=None, use_graph=True, graph_warmup=3, batch_final=True,
                 share_head=True, fold_clean=None, segmented=None, dual_bn=False, emulate_dp=False):
        self.model, self.criterion = model, criterion
        # dual-BN option (off = the reference's single BatchNorm set): adversarial features — every PGD pass and the
        # adversarial final pass — are normalised by an auxiliary BatchNorm set.  The first PGD pass is then no longer the
        # clean pass and the two final passes no longer share their affine parameters: no fold, no grouped final pass.
        self.dual_bn = bool(dual_bn)
        if self.dual_bn:
            from . import resnet_s
            resnet_s.enable_dual_bn(model)
            fold_clean, batch_final = False, False
        self.steps, self.gamma, self.eps = steps, gamma, eps
        self.perturb_idx = perturb_idx
        self.layer_number = layer_number if layer_number is not None else model.layer_number
        self.randinit, self.clip = randinit, clip
        self.arena = ParamArena(model)
        self.optimizer = ArenaSGD(self.arena, lr, momentum, weight_decay)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reducer = GradAllReducer(self.arena, allreduce_chunks, group) if self.world > 1 else None
        if self.world > 1:
            self.optimizer.grad_scale = 1.0 / self.world
            self.sync_replicas()
        elif emulate_dp:
            self.reducer = NullReducer(self.arena)      # one GPU, the data-parallel program (see NullReducer)
        # a grid barrier of the in-launch BatchNorm that gives up: detected every step without a host synchronisation, no update
        # applied meanwhile (the optimizer's device-side guard), the lost steps run again on the two-launch forms (grid_guard.py)
        self._guard_init(model, self.arena.param.device)
        import os
        # exchange the tail's gradients stage by stage while the rest of the backward runs (folded schedule); 0: one
        # blocking all-reduce after the backward (A/B, and the fallback if a piece-wise capture fails)
        self.ddp_overlap = self.reducer is not None and os.environ.get("AFAN_DDP_OVERLAP", "1") != "0"
        self.segmented = bool(segmented)          # True: run the segmented (piece-wise) step on one GPU too (tests)
        self._pieces = None
        self.batch_final = bool(batch_final)      # adv + clean final passes as one grouped pass over the tail
        self._groupable_key, self._groupable = None, False
        self.share_head = bool(share_head)
        self.fold_clean = fold_clean if fold_clean is None else bool(fold_clean)
        self._fold_auto = {}
        self.use_graph = bool(use_graph) and not randinit
        self.graph_warmup = graph_warmup
        self._graph = None
        self._graph_failed = None
        self._graph_unsafe = None
        self._eager_steps = 0
        self._static_in = None
        self._static_out = None
        self._shape_key = None
        self._stream = None

    def sync_replicas(self, src=0):
        """Every rank starts from rank `src`'s parameters, momentum and BatchNorm buffers (the reference's nn.DataParallel
        replicates module 0 every iteration; here replicas only ever see identical updates, so once is enough).  Without
        it an unseeded launch (`--seed` unset, or 0: main_perturb.py:61) would average gradients of DIFFERENT weights."""
        if not dist.is_initialized() or self.world <= 1:
            return
        a = self.arena
        for t in (a.param, a.momentum_buf):
            dist.broadcast(t, src=src, group=self.group)
        for b in self.model.buffers():
            dist.broadcast(b, src=src, group=self.group)
        a.refresh_shadow()

    # ------------------------------------------------------------------------------------------------ body
    FOLD_MIN_ELEMS = 0            # feature-map elements (batch x C x H x W) from which fold_clean=None folds (0: always)

    def _fold_ok(self, inp):
        """fold_clean: True / False force it; None (default) folds whenever the schedule applies (FOLD_MIN_ELEMS can
        restrict it to large feature maps).  Before the clean and the adversarial pass shared their weight-gradient
        launches the launch-bound networks lost with it (ResNet-56s, batch 128: 11.19 -> 11.40 ms); with the shared
        launches everything measured wins: ResNet-56s 10.99 -> 10.76 ms, ResNet-20s 3.87 -> 3.71, ResNet-18 batch 256
        11.10 -> 10.03, ResNet-50/224 batch 64 31.7 -> 26.9."""
        if self.fold_clean is False or not (self._share_head(inp) and self.steps >= 1 and not self.randinit):
            return False
        if self.fold_clean is None:
            key = tuple(inp.shape)
            if self._fold_auto.get(key) is None:
                self._fold_auto[key] = self._feature_elems(inp) >= self.FOLD_MIN_ELEMS
            return self._fold_auto[key]
        return True

    def _feature_elems(self, inp):
        """Elements of the feature map PGD perturbs for this input shape (from the model's layer shapes, no launch)."""
        m = self.model
        c, h, w = inp.shape[1], inp.shape[2], inp.shape[3]
        for mod in m.sequential_model[:self.perturb_idx]:
            c, h, w = _out_shape(mod, c, h, w)
        return inp.shape[0] * c * h * w

    def _forward_backward_folded(self, inp, target, overlap_allreduce):
        """The iteration with ONE clean tail pass.  Without randinit PGD starts AT the clean feature map
        (attack_algo.py:40-41), so its first pass (attack_algo.py:50-52: tail forward, CE, gradient w.r.t. the feature map)
        and the clean forward/backward of the joint loss (main_perturb.py:196-200) evaluate the same function of the same
        weights at the same point: same logits, same CE, same BatchNorm batch moments, and d(CE)/d(activations) differing
        by the joint loss's factor 1/2 only — which sign() ignores.  So the clean tail pass runs ONCE, with parameter
        gradients and root gradient 1/2: it yields loss_clean, the clean half of every tail parameter gradient, and
        g0 = (1/2) dCE/dfeature, which is both PGD's first ascent direction (PGD(grad0=g0)) and the gradient the clean
        branch sends into the head.  Then PGD steps 1..K-1, the adversarial pass (root 1/2), and the head backward from
        g0.  BatchNorm running statistics: the clean pass updates them first (as PGD's first pass does) and its moments
        are applied once more after the adversarial pass (ops.record_bn_updates: the reference's final clean forward
        comes last) — K + 2 updates per tail layer, in the reference's order.  Two tail forwards, one input-gradient
        pass and — against the grouped form — nothing but launch geometry are saved: 16 -> 14 tail pass-units."""
        from . import ops, resnet_s
        m, idx, ln = self.model, self.perturb_idx, self.layer_number
        ops.acc_reset(inp.device)
        self.optimizer.zero_grad()                      # before the first backward of the iteration
        crit = resnet_s.fused_criterion(self.criterion, m)
        half = ops.half(inp.device)
        with ops.bn_running_updates(2):                 # main_perturb.py:173 + :196: the head, once for both
            fm_clean = m(inp, end_point=idx, start_point=0)
        xin = fm_clean.detach().requires_grad_(True)
        with ops.record_bn_updates() as clean_bn:
            output_clean = m(xin, end_point=ln, start_point=idx)
        loss_clean = crit(output_clean, target)
        with resnet_s.stash_wgrad():                    # the tail's weight gradients wait for the adversarial pass's operands
            torch.autograd.backward(loss_clean, grad_tensors=half)
        g0 = xin.grad
        feature_map = fm_clean.detach()
        feature_map = feature_map.float() if feature_map.dtype != torch.float32 else feature_map
        feature_map_adv = PGD(feature_map, self.criterion, y=target, model=m, steps=self.steps,
                              gamma=(self.gamma / 255), start_idx=idx, layer_number=ln, eps=(self.eps / 255),
                              randinit=False, clip=self.clip, with_norms=True, grad0=g0)
        l2, linf = last_norms()
        adv_in = getattr(feature_map_adv, "_afan_shadow", None)
        if adv_in is None:
            adv_in = feature_map_adv.detach()
        output_adv = m(adv_in, end_point=ln, start_point=idx)                    # main_perturb.py:195
        loss_adv = crit(output_adv, target)
        torch.autograd.backward(loss_adv, grad_tensors=half)    # weight gradients: clean + adversarial operands, one launch
        resnet_s.flush_wgrad(m)
        clean_bn.replay()                               # main_perturb.py:196's BatchNorm side effect, last in order
        if overlap_allreduce:
            self.reducer.begin()
        fm_clean.backward(g0)                           # the clean branch's gradient through the head
        with torch.no_grad():
            loss = (loss_adv + loss_clean) / 2                                   # main_perturb.py:197
            prec1 = (output_clean.argmax(dim=1) == target).float().sum() * (100.0 / target.shape[0])
        return {"loss": loss.detach(), "loss_adv": loss_adv.detach(), "loss_clean": loss_clean.detach(),
                "prec1": prec1, "l2": l2, "linf": linf, "x_adv": feature_map_adv.detach(),
                "feature_map": feature_map, "out_clean": output_clean.detach()}

    def _tail_segments(self):
        """[(a, b), ...] cutting the tail at its stage transitions (blocks that change the shape): the backward finishes
        the LAST segment's parameter gradients first — 75 % of ResNet-18's parameters sit in the last stage."""
        m, idx, ln = self.model, self.perturb_idx, self.layer_number
        cuts = [idx]
        for i in range(idx + 1, ln):
            mod = m.sequential_model[i]
            if getattr(mod, "_sc_kind", "identity") != "identity":
                cuts.append(i)
        cuts.append(ln)
        return [(cuts[k], cuts[k + 1]) for k in range(len(cuts) - 1)]

    def _param_range(self, a, b):
        """Arena parameter indices [lo, hi) of sequential_model[a:b] (named_parameters order is the Sequential's order)."""
        pre = tuple(f"sequential_model.{i}." for i in range(a, b))
        idxs = [i for i, n in enumerate(self.arena.names) if n.startswith(pre)]
        return (idxs[0], idxs[-1] + 1) if idxs else (0, 0)

    def _phases_folded(self, inp, target, out):
        """The folded iteration (see _forward_backward_folded) as a generator for data-parallel runs: it yields the arena
        parameter range whose gradients have just become final — the tail stage by stage from the last one (the tail is
        run in segments, detached at the stage transitions, so that its backward can be issued piece by piece), then the
        head — and the caller starts that range's all-reduce while the next piece runs.  Same arithmetic as the unsegmented
        step; only the BatchNorm-backward sums across a cut are reduced stand-alone instead of in the next dgrad's epilogue.
        `out`: dict filled with the step's observables at the end."""
        from . import ops, resnet_s
        m, idx, ln = self.model, self.perturb_idx, self.layer_number
        segs = self._tail_segments()
        ops.acc_reset(inp.device)
        self.optimizer.zero_grad()
        crit = resnet_s.fused_criterion(self.criterion, m)
        half = ops.half(inp.device)

        def run_segments(x0):
            ins, outs, h = [x0], [], x0
            for k, (a, b) in enumerate(segs):
                o = m(h, end_point=b, start_point=a)
                outs.append(o)
                if k + 1 < len(segs):
                    h = o.detach().requires_grad_(True)
                    ins.append(h)
            return ins, outs

        with ops.bn_running_updates(2):
            fm_clean = m(inp, end_point=idx, start_point=0)
        xin = fm_clean.detach().requires_grad_(True)
        with ops.record_bn_updates() as clean_bn:
            c_ins, c_outs = run_segments(xin)
        loss_clean = crit(c_outs[-1], target)
        with resnet_s.stash_wgrad():
            torch.autograd.backward(loss_clean, grad_tensors=half)
            for k in range(len(segs) - 2, -1, -1):
                c_outs[k].backward(c_ins[k + 1].grad)
        g0 = xin.grad
        feature_map = fm_clean.detach()
        feature_map = feature_map.float() if feature_map.dtype != torch.float32 else feature_map
        feature_map_adv = PGD(feature_map, self.criterion, y=target, model=m, steps=self.steps,
                              gamma=(self.gamma / 255), start_idx=idx, layer_number=ln, eps=(self.eps / 255),
                              randinit=False, clip=self.clip, with_norms=True, grad0=g0)
        l2, linf = last_norms()
        adv_in = getattr(feature_map_adv, "_afan_shadow", None)
        if adv_in is None:
            adv_in = feature_map_adv.detach()
        a_ins, a_outs = run_segments(adv_in)
        loss_adv = crit(a_outs[-1], target)
        torch.autograd.backward(loss_adv, grad_tensors=half)                 # last segment (+ classifier head)
        resnet_s.flush_wgrad(m.sequential_model[segs[-1][0]:segs[-1][1]])
        yield self._param_range(*segs[-1])
        # From here on an exchange may be in flight on the reducer's stream: its resident kernels would make a grid barrier wait
        # for the exchange to END (the in-launch BatchNorm needs every workgroup of its launch on the chip at once), so the rest of
        # the backward takes the two-launch forms when ranks exchange gradients (ops.grid_bn; same bits either way).
        with ops.grid_bn(self.reducer is None):
            for k in range(len(segs) - 2, -1, -1):
                a_outs[k].backward(a_ins[k + 1].grad)
                resnet_s.flush_wgrad(m.sequential_model[segs[k][0]:segs[k][1]])
                if k == 0:
                    clean_bn.replay()                   # main_perturb.py:196's BatchNorm side effect, last in order
                yield self._param_range(*segs[k])
            if len(segs) == 1:
                clean_bn.replay()
            fm_clean.backward(g0)                       # the clean branch's gradient through the head
        with torch.no_grad():
            loss = (loss_adv + loss_clean) / 2
            prec1 = (c_outs[-1].argmax(dim=1) == target).float().sum() * (100.0 / target.shape[0])
        out.update({"loss": loss.detach(), "loss_adv": loss_adv.detach(), "loss_clean": loss_clean.detach(),
                    "prec1": prec1, "l2": l2, "linf": linf, "x_adv": feature_map_adv.detach(),
                    "feature_map": feature_map, "out_clean": c_outs[-1].detach()})
        yield self._param_range(0, idx)

    def _ddp_phases_ok(self, inp):
        want = self.segmented or (self.ddp_overlap and self.reducer is not None)
        return bool(want and self._fold_ok(inp) and hasattr(self.model, "sequential_model"))

    def _forward_backward(self, inp, target, overlap_allreduce):
        if self._fold_ok(inp):
            return self._forward_backward_folded(inp, target, overlap_allreduce)
        m, idx, ln = self.model, self.perturb_idx, self.layer_number
        if inp.is_cuda:
            from . import ops
            ops.acc_reset(inp.device)   # BatchNorm accumulator arena: one memset per step, blocks are bump-allocated
        # main_perturb.py:173 runs the head detached for PGD and :196 runs it again inside the clean forward: same images,
        # same weights, same values; each head BatchNorm updates its running statistics twice from the same moments.
        # On the channels-last kernels ONE head pass (with its autograd graph) stands for both: the BatchNorm launches apply
        # their running-statistics update twice (ops.bn_running_updates).  Otherwise: a no_grad pass here, a second below.
        fm_clean = None
        if self._share_head(inp):
            from . import ops
            with ops.bn_running_updates(2):
                fm_clean = m(inp, end_point=idx, start_point=0)
            feature_map = fm_clean.detach()
        else:
            with torch.no_grad():  # (.detach()): values and BN side effects are identical
                feature_map = m(inp, end_point=idx, start_point=0)
        feature_map = feature_map.float() if feature_map.dtype != torch.float32 else feature_map
        from . import resnet_s as _rsb
        with _rsb.bn_branch(m, "adv"):
            feature_map_adv = PGD(feature_map, self.criterion, y=target, model=m, steps=self.steps,
                                  gamma=(self.gamma / 255), start_idx=idx, layer_number=ln, eps=(self.eps / 255),
                                  randinit=self.randinit, clip=self.clip, with_norms=True)
        l2, linf = last_norms()
        # The reference feeds the requires_grad leaf itself and so also computes a never-used d(loss)/d(x_adv)
        # (SURVEY.md §9.7); feeding the detached tensor (its bf16 shadow on the bf16 path) is parity-neutral.
        adv_in = getattr(feature_map_adv, "_afan_shadow", None)
        if adv_in is None:
            adv_in = feature_map_adv.detach()
        if self.batch_final and self._tail_groupable(adv_in):
            # main_perturb.py:195-196 as ONE pass over the tail: [adv | clean] concatenated, convolutions and weight
            # gradients run once over both halves (each layer's weights are fetched once instead of twice, twice the
            # row tiles per launch), BatchNorm treats the halves as the two separate passes they are — statistics,
            # running-stat updates (adv first, then clean) and backward sums per half (resnet_s.bn_groups).
            from . import resnet_s
            if fm_clean is None:
                fm_clean = m(inp, end_point=idx, start_point=0)
            both = torch.cat([adv_in, fm_clean.to(adv_in.dtype)], dim=0)
            with resnet_s.bn_groups(2):
                out_both = m(both, end_point=ln, start_point=idx)
            nb = adv_in.shape[0]
            output_adv, output_clean = out_both[:nb], out_both[nb:]
        else:
            with _rsb.bn_branch(m, "adv"):
                output_adv = m(adv_in, end_point=ln, start_point=idx)            # main_perturb.py:195
            output_clean = (m(fm_clean, end_point=ln, start_point=idx) if fm_clean is not None
                            else m(inp, end_point=ln, start_point=0))            # main_perturb.py:196
        from . import resnet_s as _rs
        crit = _rs.fused_criterion(self.criterion, m)
        loss_adv = crit(output_adv.contiguous() if output_adv.dim() == 2 else output_adv, target)
        loss_clean = crit(output_clean.contiguous() if output_clean.dim() == 2 else output_clean, target)
        loss = (loss_adv + loss_clean) / 2                                   # main_perturb.py:197
        self.optimizer.zero_grad()
        if overlap_allreduce:
            self.reducer.begin()
        from . import resnet_s
        if loss.is_cuda and loss.dtype == torch.float32 and loss.dim() == 0:
            from . import ops as _ops
            loss.backward(gradient=_ops.one(loss.device))
        else:
            loss.backward()
        with torch.no_grad():
            prec1 = (output_clean.argmax(dim=1) == target).float().sum() * (100.0 / target.shape[0])
        return {"loss": loss.detach(), "loss_adv": loss_adv.detach(), "loss_clean": loss_clean.detach(),
                "prec1": prec1, "l2": l2, "linf": linf, "x_adv": feature_map_adv.detach(),
                "feature_map": feature_map, "out_clean": output_clean.detach()}

    def _share_head(self, inp):
        m = self.model
        return bool(self.share_head and inp.is_cuda and self.perturb_idx > 0 and getattr(m, "channels_last", False)
                    and hasattr(m, "sequential_model") and m.training)

    def _tail_groupable(self, fea):
        """Can the tail run the adversarial and the clean pass as one grouped pass?  bf16 channels-last feature map, every
        BatchNorm of the tail inside a one-node residual block on the library's kernels, and at every resolution the
        half-batch's pixel count a multiple of the largest row tile (so no tile straddles the halves)."""
        from . import ops as _ops
        key = (tuple(fea.shape), fea.dtype, fea.is_contiguous(memory_format=torch.channels_last), _ops.BN_ACC)
        if self._groupable_key == key:
            return self._groupable
        from . import resnet_s
        from . import ops
        ok = (fea.is_cuda and fea.dim() == 4 and fea.dtype == torch.bfloat16 and key[2] and resnet_s._Flags.block_fusion
              and ops.BN_ACC)                                  # per-half sums live in the f64 accumulator blocks
        n, _, h, w = fea.shape if fea.dim() == 4 else (0, 0, 0, 0)
        if ok:
            for mod in self.model.sequential_model[self.perturb_idx:self.layer_number]:
                if isinstance(mod, (resnet_s.BasicBlock, resnet_s.Bottleneck)):
                    probe = torch.empty((2, mod._chain()[0][0].in_channels, 2, 2), dtype=torch.bfloat16,
                                        device=fea.device).contiguous(memory_format=torch.channels_last)
                    if not (mod.training and resnet_s._block_fast_path_ok(mod, probe)):
                        ok = False
                        break
                    for c, _ in mod._chain():
                        if (n * h * w) % 128 or (c.stride[0] == 2 and ((h | w) & 1)):
                            ok = False
                        h, w = (h + c.stride[0] - 1) // c.stride[0], (w + c.stride[0] - 1) // c.stride[0]
                        if (n * h * w) % 128:
                            ok = False
                    if not ok:
                        break
                elif any(isinstance(q, torch.nn.modules.batchnorm._BatchNorm) for q in mod.modules()):
                    ok = False
                    break
        self._groupable_key, self._groupable = key, bool(ok)
        return self._groupable

    def _graph_safe(self):
        """Every configuration runs on the library's own kernels (tuned bf16 or general f32 MFMA) and is capturable; a
        convolution that left the library would be listed by resnet_s.vendor_convs (empty by construction) and keep the
        step eager: rounds 1-2 measured a captured vendor input-gradient pass reading memory the graph did not own."""
        if self._graph_unsafe is None:
            from . import resnet_s
            self._graph_unsafe = resnet_s.vendor_convs(self.model) if hasattr(self.model, "sequential_model") else []
            if self._graph_unsafe:
                self.use_graph = False
        return not self._graph_unsafe

    def _step_eager(self, inp, target):
        self.optimizer._sync_lr()
        if self._ddp_phases_ok(inp):
            out = {}
            if self.reducer is not None:
                self.reducer.begin(explicit=True)
            for rng in self._phases_folded(inp, target, out):
                if self.reducer is not None:
                    self.reducer.launch_params(*rng)
            if self.reducer is not None:
                self.reducer.finish()
            self._pre_sgd()
            self.optimizer.step()
            return out
        out = self._forward_backward(inp, target, overlap_allreduce=self.reducer is not None)
        if self.reducer is not None:
            self.reducer.finish()
        self._pre_sgd()
        self.optimizer.step()
        return out

    def _pre_sgd(self):
        """Data parallel: the grid barrier's error word becomes the maximum over the ranks before the (device-guarded) optimizer
        launch reads it — every rank skips the same updates (grid_guard.GridGuard.sync_ranks)."""
        if self._guard is not None and self.world > 1:
            self._guard.sync_ranks(self.group)

    def _drop_graphs(self):
        """After a grid barrier gave up: the captured graphs hold in-launch-BatchNorm launches; capture again (two-launch forms)."""
        self._graph = self._pieces = self._static_in = self._static_out = self._shape_key = None
        self._graph_failed = None

    # ----------------------------------------------------------------------------------------------- graph
    def _capture_pieces(self, inp, target):
        """Data-parallel capture: one hipGraph per phase of _phases_folded; between two replays the host starts the
        all-reduce of the range the finished piece completed (RCCL on the side stream), so the exchange of the last
        stage's gradients — most of the bytes — runs under the rest of the backward.  The pieces share one memory pool."""
        dev = inp.device
        self._stream = torch.cuda.Stream(device=dev)
        self._static_in = (inp.clone(), target.clone())
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        out, pieces, pool, fused = {}, [], None, []
        gen = self._phases_folded(self._static_in[0], self._static_in[1], out)
        try:
            for _ in range(len(self._tail_segments()) + 1):      # one piece per tail segment + the head (nothing follows the last)
                n0 = ops.CALLS["conv_bn_fused"]
                g = torch.cuda.CUDAGraph()
                with ops.no_gc_during_capture(), torch.cuda.graph(g, pool=pool, stream=self._stream, capture_error_mode="thread_local"):
                    rng = next(gen)
                fused.append(ops.CALLS["conv_bn_fused"] - n0)
                pieces.append((g, rng))
                pool = pieces[0][0].pool()
                if self.reducer is not None:
                    # the replay starts an exchange HERE: whatever is captured from now on runs beside RCCL's resident kernels and
                    # must not contain a grid barrier (the eager path gets the same switch from the reducer's own launches)
                    ops.exchange_in_flight(True)
        finally:
            ops.exchange_in_flight(False)
        gen.close()
        self._pieces_fused = fused          # convolution + BatchNorm launches captured per piece (data parallel: 0 after the first)
        self._pieces, self._static_out = pieces, out
        self._graph = pieces[0][0]
        self._shape_key = (tuple(inp.shape), inp.dtype, tuple(target.shape))

    def _capture(self, inp, target):
        if self._ddp_phases_ok(inp):
            return self._capture_pieces(inp, target)
        dev = inp.device
        self._stream = torch.cuda.Stream(device=dev)
        self._static_in = (inp.clone(), target.clone())
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()
        with ops.no_gc_during_capture(), torch.cuda.graph(g, stream=self._stream, capture_error_mode="thread_local"):
            out = self._forward_backward(self._static_in[0], self._static_in[1], overlap_allreduce=False)
            if self.world == 1:
                self.optimizer.step()
        self._graph, self._static_out = g, out
        self._shape_key = (tuple(inp.shape), inp.dtype, tuple(target.shape))

    def _step_graph(self, inp, target):
        self._static_in[0].copy_(inp, non_blocking=True)
        self._static_in[1].copy_(target, non_blocking=True)
        self.optimizer._sync_lr()          # lr lives in device memory: the graph reads it, the host only writes it here
        if self._pieces is not None:
            if self.reducer is not None:
                self.reducer.begin(explicit=True)
            import time as _time
            for g, rng in self._pieces:
                g.replay()
                t_ = _time.perf_counter()
                if self.reducer is not None:
                    self.reducer.launch_params(*rng)
                    self.reducer.note_host_gap(_time.perf_counter() - t_)
            if self.reducer is not None:
                self.reducer.finish()
            self._pre_sgd()
            self.optimizer.step()
        else:
            self._graph.replay()
            if self.world > 1:
                dist.all_reduce(self.arena.grad, op=dist.ReduceOp.SUM, group=self.group)
                self._pre_sgd()
                self.optimizer.step()
        small = ("loss", "loss_adv", "loss_clean", "prec1", "l2", "linf")
        # graph-owned outputs are overwritten by the next replay: hand out copies of the small ones
        return {k: (v.clone() if k in small else v) for k, v in self._static_out.items()}

    def step(self, inp, target):
        """One iteration. Returns device tensors only: loss, loss_adv, loss_clean, prec1, l2[N], linf[N] (+ x_adv,
        feature_map, out_clean — in graph mode these three are views of graph-owned buffers, valid until the next step)."""
        return self._guarded((inp, target), self._step_once)

    def _step_once(self, inp, target):
        if self._graph is not None and self._shape_key == (tuple(inp.shape), inp.dtype, tuple(target.shape)):
            return self._step_graph(inp, target)
        if (self.use_graph and self._graph is None and self._graph_failed is None
                and self._eager_steps >= self.graph_warmup and inp.is_cuda and self.model.training
                and self._graph_safe()):
            try:
                self._capture(inp, target)
                return self._step_graph(inp, target)
            except Exception as e:  # noqa: BLE001 — stay correct: fall back to eager launches, loudly
                import warnings
                self._graph, self._graph_failed = None, e
                warnings.warn(f"hipGraph capture of the A-FAN step failed ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize()
        self._eager_steps += 1
        return self._step_eager(inp, target)
