"""The Segmentation A-FAN iteration (Segmentation/main_aug_final.py:149-232) on one MI355X: owner of the parameter arena,
the two-group SGD and the PolyLR schedule around `seg_attack_algo.seg_train_step`.

    head pass (SE, out_idx) + clean decoder-head pass (SD) -> K-step SE feature PGD + K-step SD decoder PGD
    -> 3 SAT sample points + mix_feature -> clean / SE1 / SE2 / SD forwards -> 0.7/0.1/0.1/0.1 loss -> backward -> SGD

Reference details kept: BatchNorm momentum 0.01 in the backbone (main_aug_final.py:77), SGD(momentum 0.9) with the
backbone at 0.1 x lr (:79-82), PolyLR(power 0.9) stepped once per iteration (:261), CrossEntropyLoss(ignore_index=255).
After `graph_warmup` eager iterations the whole iteration body is captured into a hipGraph and replayed (bf16
every configuration: all convolutions are the library's own, see resnet_s.vendor_convs)."""
import torch
import torch.nn as nn

from . import ops, resnet_s
from .arena import ArenaSGD, ParamArena
from .deeplab import PolyLR, set_bn_momentum
from .grid_guard import GuardedTrainer
from .seg_attack_algo import seg_train_phases, seg_train_step


class SegTrainer(GuardedTrainer):
    def __init__(self, model, criterion=None, *, steps=1, eps=2.0, gamma_se=0.5, gamma_sd=0.5, pertub_idx_se=3,
                 pertub_idx_sd="aspp", mix_layer="11", mix_sd=False, noise_sd=0.0, randinit=False, clip=False, lr=0.01,
                 momentum=0.9, weight_decay=1e-4, total_itrs=30000, lr_policy="poly", step_size=10000,
                 backbone_bn_momentum=0.01, use_graph=True, graph_warmup=2, dual_bn=False, fold_clean=None, group=None,
                 allreduce_chunks=4, fold_pgd0=None, segmented=None, wgrad_stream=None, batch_tails=None):
        self.model = model
        if dual_bn:      # BASELINE configs[3] "+ dual-BN": an option the reference does not have (resnet_s.enable_dual_bn); default off
            resnet_s.enable_dual_bn(model)
        self.criterion = criterion if criterion is not None else nn.CrossEntropyLoss(ignore_index=255, reduction="mean")
        self.kw = dict(steps=steps, eps=eps, gamma_se=gamma_se, gamma_sd=gamma_sd, pertub_idx_se=pertub_idx_se,
                       pertub_idx_sd=pertub_idx_sd, mix_layer=mix_layer, mix_sd=mix_sd, noise_sd=noise_sd, randinit=randinit,
                       clip=clip, dual_bn=bool(dual_bn), fold_clean=fold_clean, fold_pgd0=fold_pgd0, batch_tails=batch_tails)
        if backbone_bn_momentum is not None:
            set_bn_momentum(model.backbone, backbone_bn_momentum)
        self.arena = ParamArena(model, skip=())
        self.optimizer = ArenaSGD(self.arena, lr, momentum, weight_decay,
                                  groups=[("backbone.", 0.1 * lr), ("classifier.", lr)])
        if lr_policy == "poly":
            self.scheduler = PolyLR(self.optimizer, total_itrs, power=0.9)
        else:
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=step_size, gamma=0.1)
        # data parallel (BASELINE configs[3] is 4 GPUs x 2 images): minibatch sharding, BatchNorm per replica like the
        # reference's nn.DataParallel, ONE exchange per iteration — the fp32 gradient arena, summed over the ranks in
        # chunks on a side stream after the backward, 1/world folded into the SGD kernel.  The iteration's graph ends
        # before the optimizer step; the all-reduce and the one SGD launch follow it.
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reducer = None
        if self.world > 1:
            from .train_step import GradAllReducer
            self.reducer = GradAllReducer(self.arena, allreduce_chunks, group)
            self.optimizer.grad_scale = 1.0 / self.world
            for t in (self.arena.param, self.arena.momentum_buf):      # replicas start from rank 0's state
                dist.broadcast(t, src=0, group=group)
            for b in model.buffers():
                dist.broadcast(b, src=0, group=group)
            self.arena.refresh_shadow()
            self.kw["defer_step"] = True
        self._guard_init(model, self.arena.param.device)      # a grid barrier that gives up: grid_guard.py
        self.use_graph = bool(use_graph) and not randinit and noise_sd == 0
        self.graph_warmup = graph_warmup
        self._graph = self._graph_failed = self._static = self._out = self._key = self._pieces = None
        self.segmented = bool(segmented)      # True: run the two-phase (cut) schedule on one GPU too (tests)
        if self.segmented:
            self.kw["defer_step"] = True
        self._eager_steps = 0
        # weight gradients on a side stream / parallel graph branch (resnet_s._WgradStream): None = by workload size — it
        # pays from about 8 images of 513 x 513 per GPU (55.2 -> 52.6 ms), not at the 2-image share (25.3 -> 25.5 ms)
        self.wgrad_stream = wgrad_stream

    WGRAD_STREAM_MIN_PIXELS = 1 << 21

    def _wgrad_side(self, images):
        if self.wgrad_stream is not None:
            return bool(self.wgrad_stream)
        return images.is_cuda and images.shape[0] * images.shape[2] * images.shape[3] >= self.WGRAD_STREAM_MIN_PIXELS

    def _body(self, images, labels):
        return seg_train_step(self.model, self.optimizer, self.criterion, images, labels, **self.kw)

    # ---- data parallel: the tail's gradients (everything behind the SE point: layer4, ASPP, decoder — 53 % of DeepLabv3+
    # ResNet-101's 58.7 M parameters, the LAST contiguous range of the arena) are final when seg_train_phases yields "tail";
    # their all-reduce starts there, on the side stream, and runs under the head's backward.  The rest follows at finish().
    def _tail_range(self):
        se = self.kw["pertub_idx_se"]
        if type(se) != int:
            return None
        pre = tuple(f"backbone.layer{k}." for k in range(se + 1, 5)) + ("classifier.",)
        idx = [i for i, n in enumerate(self.arena.names) if n.startswith(pre)]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        return idx[0], idx[-1] + 1

    def _phased(self):
        return self.segmented or self.reducer is not None

    def _run_phases(self, images, labels):
        """Eager iteration through seg_train_phases: the tail's exchange is launched at the yield."""
        out, rng = {}, self._tail_range()
        if self.reducer is not None:
            self.reducer.begin(explicit=True)
        for ph in seg_train_phases(self.model, self.optimizer, self.criterion, images, labels, out, **self.kw):
            if ph == "tail" and self.reducer is not None and rng is not None:
                self.reducer.launch_params(*rng)
        return out

    def _drop_graphs(self):
        self._graph = self._graph_failed = self._static = self._out = self._key = self._pieces = None

    def _exchange_and_step(self):
        if self.reducer is not None:
            self.reducer.finish()      # whatever no launch_params() announced is reduced here
            if self.world > 1:
                self._guard_sync_ranks(self.group)
            self.optimizer.step()
        elif self.segmented:
            self.optimizer.step()

    def _graph_safe(self):
        if resnet_s.vendor_convs(self.model):
            self.use_graph = False
        return self.use_graph

    def _capture(self, images, labels):
        dev = images.device
        self._static = (images.clone(), labels.clone())
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        self.optimizer._sync_lr()
        if not self._phased():
            g = torch.cuda.CUDAGraph()
            with ops.no_gc_during_capture(), torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                out = self._body(self._static[0], self._static[1])
            self._pieces = [(g, None)]
            return out
        # one hipGraph per phase (shared memory pool): between two replays the host starts the tail's all-reduce
        out, pieces, pool = {}, [], None
        gen = seg_train_phases(self.model, self.optimizer, self.criterion, self._static[0], self._static[1], out, **self.kw)
        done, fused = False, []
        try:
            while not done:
                n0 = ops.CALLS["conv_bn_fused"]
                g = torch.cuda.CUDAGraph()
                with ops.no_gc_during_capture(), torch.cuda.graph(g, pool=pool, stream=stream, capture_error_mode="thread_local"):
                    try:
                        ph = next(gen)
                    except StopIteration:
                        ph, done = None, True
                pieces.append((g, ph))
                fused.append(ops.CALLS["conv_bn_fused"] - n0)
                pool = pieces[0][0].pool()
                if ph == "tail" and self.reducer is not None:
                    # the replay starts the tail's exchange HERE: what is captured from now on runs beside RCCL's resident kernels
                    # and must not contain a grid barrier (ops.exchange_in_flight; the eager path gets it from the reducer itself)
                    ops.exchange_in_flight(True)
        finally:
            ops.exchange_in_flight(False)
        self._pieces, self._pieces_fused = pieces, fused
        return out

    def step(self, images, labels):
        """One iteration (device tensors only; the scheduler is NOT stepped here — call `trainer.scheduler.step()` once
        per iteration like main_aug_final.py:261).  `flush_guard()` at every logging interval (grid_guard.py)."""
        return self._guarded((images, labels), self._step_once)

    def _step_once(self, images, labels):
        key = (tuple(images.shape), images.dtype, tuple(labels.shape))
        if self._graph is not None and self._key == key:
            self._static[0].copy_(images, non_blocking=True)
            self._static[1].copy_(labels, non_blocking=True)
            self.optimizer._sync_lr()
            rng = self._tail_range()
            if self.reducer is not None:
                self.reducer.begin(explicit=True)
            for g, ph in self._pieces:
                g.replay()
                if ph == "tail" and self.reducer is not None and rng is not None:
                    self.reducer.launch_params(*rng)
            self._exchange_and_step()
            small = ("loss", "losses")
            return {k: (v.clone() if k in small else v) for k, v in self._out.items()}
        if (self.use_graph and self._graph is None and self._graph_failed is None and images.is_cuda
                and self._eager_steps >= self.graph_warmup and self.model.training and self._graph_safe()):
            try:
                with resnet_s.wgrad_stream(self._wgrad_side(images)):
                    out = self._capture(images, labels)
                self._graph, self._out, self._key = self._pieces[0][0], out, key
                return self._step_once(images, labels)
            except Exception as e:  # noqa: BLE001 — stay correct: fall back to eager launches, loudly
                import warnings
                self._graph, self._graph_failed = None, e
                warnings.warn(f"hipGraph capture of the segmentation A-FAN step failed ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize()
        self._eager_steps += 1
        self.optimizer._sync_lr()
        with resnet_s.wgrad_stream(self._wgrad_side(images)):
            out = self._run_phases(images, labels) if self._phased() else self._body(images, labels)
        self._exchange_and_step()
        return out
