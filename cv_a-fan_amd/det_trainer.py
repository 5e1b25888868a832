"""The Detection multi-layer SAT iteration (Detection/train_aug_sat_muti_advt.py:26-209) on one MI355X per rank: owner of the
parameter arena and the SGD around `det_attack_algo.det_train_phases`, and of the data-parallel exchange.

The reference shards this step with `nn.DataParallel(Model(...).cuda())` (:36-43): one process, replicas rebuilt every forward,
gradients reduced onto GPU 0.  Here: one process per GPU (torch.distributed, RCCL over xGMI), minibatch sharding, frozen
BatchNorm (nothing to synchronise), ONE exchange per iteration — the fp32 gradient arena summed over the ranks, 1/world folded
into the SGD kernel — started TAIL FIRST: the joint backward is cut at the backbone's output, so the gradients of layer4 / RPN /
the two heads (the arena's suffix, `tail_range`) are final
while the backbone's backward (layer3 / layer2 over the three passes that reach it) still runs, and their all-reduce flies on a
side stream under it; the backbone's gradients follow at finish().  SGD(lr 0.001, momentum 0.9, weight decay 5e-4) as
config/train_config.py; the learning-rate schedule (WarmUpMultiStepLR, :48) steps outside, once per iteration."""
import torch

from .arena import ArenaSGD, ParamArena
from .det_attack_algo import det_train_phases
from .grid_guard import GuardedTrainer

# everything behind the conv4 feature map: layer4 (= detection.hidden), the RPN and the two heads.  The backbone's unused ImageNet
# classifier `features.fc` sits between them in the reference's parameter order and never receives a gradient: optim.SGD skips a
# tensor whose .grad is None (no weight decay, no momentum: it keeps its values), so it stays OUTSIDE the arena (UNUSED_PARAMS) —
# the prefix is kept for arenas built with skip=() by callers that want the reference's full parameter list.
TAIL_PREFIXES = ("features.layer4.", "features.fc.", "rpn.", "detection.")
UNUSED_PARAMS = ("features.fc.weight", "features.fc.bias")


class DetTrainer(GuardedTrainer):      # (frozen BatchNorm: no grid barrier on this path; flush_guard() is the uniform no-op)
    def __init__(self, model, *, lr=0.001, momentum=0.9, weight_decay=0.0005, loss_settings=1, group=None, allreduce_chunks=4,
                 segmented=None, arena=None, noise_ahead=False):
        import torch.distributed as dist
        self.model, self.loss_settings, self.group = model, int(loss_settings), group
        self.arena = arena if arena is not None else ParamArena(model, skip=UNUSED_PARAMS)
        self.optimizer = ArenaSGD(self.arena, lr, momentum, weight_decay)
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
        # the next iteration's image-PGD noise drawn behind this iteration's backward (det_attack_algo.NoiseAhead): same generator
        # stream provided the caller draws nothing from the host generator between iterations
        from .det_attack_algo import NoiseAhead
        self.noise_ahead = NoiseAhead() if noise_ahead else None
        self.segmented = bool(segmented)        # True: run the two-part backward on one GPU too (tests)
        self._graph = None                      # (bench.py asks every trainer whether its step is a hipGraph replay)

    def tail_range(self):
        """[lo, hi) in arena-parameter order of everything behind the backbone's output — a suffix, or None if the model's
        parameter order does not make it one (then the whole arena is exchanged at finish())."""
        return tail_range(self.arena.names)

    def _phased(self):
        return self.segmented or self.reducer is not None

    def step(self, images, bboxes, labels):
        out, rng = {}, self.tail_range()
        cut = self._phased() and rng is not None and hasattr(self.model, "cut_features")
        if self.reducer is not None:
            self.reducer.begin(explicit=True)
        for ph in det_train_phases(self.model, self.optimizer, images, bboxes, labels, out, loss_settings=self.loss_settings,
                                   cut=cut, defer_step=True, noise_ahead=self.noise_ahead):
            if ph == "tail" and self.reducer is not None:
                self.reducer.launch_params(*rng)
        if self.reducer is not None:
            self.reducer.finish()          # whatever no launch_params() announced (the backbone) is reduced here
        self.optimizer.step()
        return out


def tail_range(names):
    idx = [i for i, n in enumerate(names) if n.startswith(TAIL_PREFIXES)]
    if not idx or idx != list(range(idx[0], len(names))):
        return None
    return idx[0], len(names)
