"""What happens when a grid barrier of the in-launch BatchNorm gives up (csrc/afan_conv.hip, GridBar): detection without a host
synchronisation inside the step, no corrupted update ever applied, and an exact recovery on the two-launch forms.

The convolution + BatchNorm launches (afan_conv_*_bn_*) meet at a grid-wide barrier, which needs every workgroup of the launch on
the chip at once.  The launcher checks that against the occupancy API, but the launches are not cooperative launches: another
process's kernels on the GPU, or a preempted queue, can still keep a workgroup out.  The spin is bounded (0.2 s); a workgroup that
gives up sets the barrier's error word and goes on with PARTIAL batch totals.  From there:

* **device side, same step** — `afan_sgd_step_guarded` reads the word and skips the update: parameters, momentum and the bf16
  shadow stay as they were.  The word is sticky, so every later step's update is skipped too until the host has noticed.
* **device side, step start** — `afan_guarded_copy` snapshots the BatchNorm buffers (running statistics are updated INSIDE the
  launches that may give up) at the start of every step, unless the word is set: after a failure the snapshot holds the buffers as
  they were at the START of the failed step.
* **host side, every step** — at the end of every step the word is copied, asynchronously, into that step's slot of a small pinned
  ring; `GridGuard.check()` at the start of step s waits for the event of step s - depth (normally long complete: it only bounds how
  far the host runs ahead) and reads that step's slot — a plain host load.  A failure is therefore seen exactly `depth` steps late,
  at the same step on every rank of a data-parallel run (the words are equalised across ranks inside every step), and the inputs
  of those `depth` steps are kept (clones) to run them again.
* **recovery** — `ops.grid_bn_disable()` for the process (loud warning), the trainer's graphs are dropped, the BatchNorm buffers
  are restored from the snapshot, the word and the counter are cleared, and the lost steps are run again on the two-launch forms
  with their own inputs and learning rates: weights, momentum and running statistics end up equal to a run that never used the
  in-launch form (tests/test_grid_guard_gpu.py forces a give-up with a resident spinner kernel and checks exactly that).
  The observables (`loss`, ...) already RETURNED for the lost steps were computed from partial totals and cannot be taken back;
  `GridGuard.lost_steps` counts them and the warning names them.  Host-drawn randomness (randinit noise, dropout masks) is drawn
  anew for a re-run step.

The reference has no counterpart (its BatchNorm is cuDNN's, one launch per layer: Classification/resnet_s.py:72-77)."""
import weakref

import torch

from . import ops


class BufferArena:
    """The model's BatchNorm running statistics as views into ONE flat fp32 tensor (+ one int64 tensor for num_batches_tracked), so
    that the pre-step snapshot is one small launch.  state_dict keys and values are unchanged (load_state_dict copies in place)."""

    def __init__(self, model):
        bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats
               and m.running_mean is not None]
        self.modules = bns
        if not bns:
            self.f32 = self.i64 = None
            return
        dev = bns[0].running_mean.device
        off, offs = 0, []
        for m in bns:
            c = m.running_mean.numel()
            offs.append(off)
            off += 2 * ((c + 3) // 4 * 4)                          # 16-byte pieces
        self.f32 = torch.zeros(max(off, 4), dtype=torch.float32, device=dev)
        self.i64 = torch.zeros((len(bns) + 1) // 2 * 2, dtype=torch.int64, device=dev)
        for k, (m, o) in enumerate(zip(bns, offs)):
            c = m.running_mean.numel()
            cp = (c + 3) // 4 * 4
            for name, lo in (("running_mean", o), ("running_var", o + cp)):
                v = self.f32[lo:lo + c].view(m._buffers[name].shape)
                v.copy_(m._buffers[name])
                m._buffers[name] = v
            nb = self.i64[k:k + 1].view(m.num_batches_tracked.shape)
            nb.copy_(m.num_batches_tracked)
            m._buffers["num_batches_tracked"] = nb
        self.snap_f32, self.snap_i64 = self.f32.clone(), self.i64.clone()

    def snapshot_guarded(self, counter):
        if self.f32 is not None:
            ops.guarded_copy_(self.snap_f32, self.f32, counter)
            ops.guarded_copy_(self.snap_i64, self.i64, None)

    def restore(self):
        if self.f32 is not None:
            self.f32.copy_(self.snap_f32)
            self.i64.copy_(self.snap_i64)


def _keep(t):
    if torch.is_tensor(t):
        return t.clone()
    if isinstance(t, (list, tuple)):
        return type(t)(_keep(u) for u in t)
    return t


class GridGuard:
    """One per trainer.  The trainer's step():

        for inputs, lrs in guard.check():     # normally empty; after a give-up: the lost steps to run again (GRID_BN is off by then)
            <run them>
        guard.begin(inputs, optimizer)        # snapshot (guarded), keep the inputs
        ... the step ... guard.sync_ranks(group) before the optimizer (data parallel) ...
        guard.end()                           # mirror the error word into this step's host slot, asynchronously

    The rule is deterministic — at the start of step s the slot of step s - depth is read, after a wait for that step's event — so
    that the ranks of a data-parallel run (whose words are equalised inside every step, sync_ranks) recover at the same step."""

    def __init__(self, model, device, depth=3, on_failure=None):
        self.device = torch.device(device)
        self.depth = max(int(depth), 1)
        self.buffers = BufferArena(model)
        self.word = ops.grid_guard_word(self.device)
        self._slots = torch.zeros(self.depth + 1, dtype=torch.int32).pin_memory()
        self._next_slot = 0
        self.on_failure = on_failure               # called once per failure before the re-runs (the trainer drops its graphs there)
        self._ring = []                            # [inputs, lrs, event, slot index] of the steps not yet verified
        self.lost_steps = 0                        # steps whose returned observables were invalid (run again since)
        self.failures = 0
        self._dd = torch.zeros(4, dtype=torch.int32, device=self.device)

    # ---- per step
    def begin(self, inputs, optimizer=None, state=None):
        """state: whatever else the trainer must put back before running this step again (host-side clones of small tensors the
        device guard does not cover: the learnable trainer's mixing weights and their optimizer state)."""
        lrs = [float(g["lr"]) for g in optimizer.param_groups] if optimizer is not None else None
        kept = tuple(_keep(t) for t in inputs)
        self.buffers.snapshot_guarded(None)
        self._ring.append([(kept, lrs, state), None, None, None])

    def sync_ranks(self, group=None):
        """Data parallel, between the gradient exchange and the optimizer: the error word becomes the MAXIMUM over the ranks, so
        that every rank skips the same updates, keeps the same snapshot and finds the same slot set (one 4-byte all-reduce)."""
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.word, op=dist.ReduceOp.MAX, group=group)

    def end(self):
        if not self._ring or self._ring[-1][3] is not None:
            return
        k = self._next_slot
        self._next_slot = (k + 1) % self._slots.numel()
        self._slots[k:k + 1].copy_(self.word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._ring[-1][2], self._ring[-1][3] = ev, k

    def check(self):
        """[] while no barrier has given up; else recover and return the lost steps as [(inputs, lrs, state), ...], oldest first.  Waits for
        the step `depth` back only (normally long complete): the host runs at most `depth` steps ahead of the device."""
        while len(self._ring) >= self.depth:
            _, _, ev, k = self._ring[0]
            if ev is None:                         # (a step that never reached end(): an exception in the trainer) — nothing to verify
                self._ring.pop(0)
                continue
            ev.synchronize()
            if int(self._slots[k]) != 0:
                return self._recover(0)
            self._ring.pop(0)
        return []

    def flush(self):
        """Where the host reads from the device anyway (a logging interval, the end of an epoch, before a checkpoint): wait for
        everything issued and verify every step still in the ring.  Returns the lost steps like check()."""
        torch.cuda.synchronize(self.device)
        for i, (_, _, ev, k) in enumerate(self._ring):
            if k is not None and int(self._slots[k]) != 0:
                return self._recover(i)
        self._ring = []
        return []

    def _recover(self, first):
        torch.cuda.synchronize(self.device)
        lost = [e[0] for e in self._ring[first:]]
        self.failures += 1
        ops.grid_bn_disable(f"a grid barrier gave up (workgroups of a convolution + BatchNorm launch not co-resident on {self.device}); "
                            f"no update has been applied since; the last {len(lost)} step(s) run again on the two-launch forms — the "
                            "observables already returned for them were invalid")
        self.buffers.restore()                     # the buffers at the START of the failed step (the snapshot stopped there)
        ops.grid_barrier_error(self.device)        # clears the barrier's words
        self._slots.zero_()
        self._ring = []
        self.lost_steps += len(lost)
        cb = self.on_failure() if isinstance(self.on_failure, weakref.WeakMethod) else self.on_failure
        if cb is not None:
            cb()
        return lost


def make(model, device, **kw):
    """A GridGuard if the in-launch BatchNorm can run in this process at all, else None."""
    if not (ops.GRID_BN_ALLOWED_AT_IMPORT and torch.device(device).type == "cuda"):
        return None
    return GridGuard(model, device, **kw)


class rerun_lrs:
    """Context: the optimizer's learning rates as they were when a lost step was first issued."""

    def __init__(self, optimizer, lrs):
        self.opt, self.lrs = optimizer, lrs

    def __enter__(self):
        if self.opt is None:
            self.lrs = None
        if self.lrs is not None:
            self.old = [g["lr"] for g in self.opt.param_groups]
            for g, lr in zip(self.opt.param_groups, self.lrs):
                g["lr"] = lr
        return self

    def __exit__(self, *exc):
        if self.lrs is not None:
            for g, lr in zip(self.opt.param_groups, self.old):
                g["lr"] = lr
        return False


class GuardedTrainer:
    """What the trainers share: step() -> self._guarded(inputs, run) wraps one iteration in the guard's check / begin / end."""
    _guard = None

    def _guard_init(self, model, device, **kw):
        # (a WEAK reference to the bound method: a strong one would make trainer <-> guard a reference cycle, and a trainer — its
        # hipGraphs, streams and events — would then only die in a cyclic garbage collection at some later point, possibly in the middle
        # of another trainer's graph capture, where destroying a graph or a stream aborts the process)
        self._guard = make(model, device, on_failure=weakref.WeakMethod(self._drop_graphs), **kw)

    def _guard_state(self):
        return None

    def _guard_restore(self, state):
        pass

    def _drop_graphs(self):
        pass

    def _guard_sync_ranks(self, group=None):
        """Data parallel, between the gradient exchange and the optimizer launch."""
        if self._guard is not None:
            self._guard.sync_ranks(group)

    def _guarded(self, inputs, run):
        g = self._guard
        if g is None or not any(torch.is_tensor(t) and t.is_cuda for t in inputs):
            return run(*inputs)
        self._rerun(g.check(), run)
        g.begin(inputs, getattr(self, "optimizer", None), self._guard_state())
        out = run(*inputs)
        g.end()
        return out

    def _rerun(self, lost, run):
        # (a grid barrier gave up: GRID_BN is off by now, the graphs are dropped, the BatchNorm buffers are back at the first lost step's start)
        for n, (inputs, lrs, state) in enumerate(lost):
            if n == 0 and state is not None:
                self._guard_restore(state)
            with rerun_lrs(getattr(self, "optimizer", None), lrs):
                run(*inputs)

    def flush_guard(self, run=None):
        """Verify every step issued so far (synchronises; call where the host reads results anyway: a logging interval, the end of
        an epoch, before a checkpoint).  Steps that ran on a given-up barrier's partial totals are run again.  Returns how many."""
        if self._guard is None:
            return 0
        lost = self._guard.flush()
        self._rerun(lost, run if run is not None else self._step_once)
        return len(lost)
