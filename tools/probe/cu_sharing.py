"""What a gradient exchange's resident kernels cost the backward beside them — measurable on ONE GPU.

RCCL's all-reduce runs as a few long-lived workgroups ("channels") that sit on CUs for the whole exchange.  The data-parallel step
(train_step.AfanTrainer with a reducer) overlaps the last stage's exchange with the rest of the backward, whose convolution launches
are EXACTLY one workgroup per CU x 256 CUs (149 KB of LDS each): a CU whose LDS a channel kernel holds cannot take its workgroup, which
then runs as a second round — up to 2x for every launch issued while the exchange is in flight.  This probe runs the data-parallel
program on one GPU (emulate_dp) with a stand-in for the channels: afan_occupy_cus(W workgroups, L bytes of LDS each) started on a
side stream where the real reducer starts RCCL (first announced range) for about as long as the rest of the backward takes.

    python tools/probe/cu_sharing.py            -> table: ms per step by (W, L)
"""
import importlib, os, sys, time
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("cv_a-fan_amd")
ops, ts = pkg.ops, pkg.train_step
dev = torch.device("cuda:0")


class ChannelReducer(ts.NullReducer):
    """NullReducer + a spinner standing for RCCL's channel kernels from the first announcement on."""

    def __init__(self, arena, wgs, lds, us):
        super().__init__(arena)
        self.wgs, self.lds, self.us = wgs, lds, us
        self.side = torch.cuda.Stream(device=dev)
        self._started = False
        self.t_announce = self.t_finish = None

    def begin(self, explicit=False):
        super().begin(explicit)
        self._started = False

    def launch_params(self, lo, hi):
        super().launch_params(lo, hi)
        if not self._started and hi > lo:
            self._started = True
            cur = torch.cuda.current_stream(dev)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(cur)
            self.t_announce = ev
            if self.wgs:
                self.side.wait_event(ev)
                ops.occupy_cus(self.wgs, self.lds, self.us, stream=self.side)

    def finish(self):
        cur = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(cur)
        self.t_finish = ev
        if self.wgs:
            cur.wait_stream(self.side)
        super().finish()


def run(wgs, lds, us, steps=20):
    torch.manual_seed(3)
    m = pkg.resnet_s.resnet18()
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(dev).train()
    tr = ts.AfanTrainer(m, nn.CrossEntropyLoss(), steps=5, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.1, emulate_dp=True)
    tr.reducer = ChannelReducer(tr.arena, wgs, lds, us)
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(256, 3, 32, 32, generator=g).to(dev), torch.randint(0, 10, (256,), generator=g).to(dev)
    for _ in range(6):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    rest = tr.reducer.t_announce.elapsed_time(tr.reducer.t_finish) if tr.reducer.t_announce is not None else float("nan")
    assert tr.flush_guard() == 0 and not ops.grid_barrier_error(dev)
    return ms, rest


base, rest = run(0, 1024, 0)
print(f"data-parallel program, no channel kernels: {base:.3f} ms per step; first announcement -> end of backward: {rest * 1e3:.0f} us")
us = int(rest * 1e3 * 0.9)
print(f"channel stand-in: W workgroups x L bytes of LDS, resident for {us} us from the first announcement (90 % of that stretch)")
print(f"{'W':>4s} {'L':>8s} {'ms/step':>9s} {'vs none':>8s} {'stretch us':>11s}")
for lds in (1024, 16 * 1024, 64 * 1024):
    for wgs in (8, 16, 32, 64):
        ms, r = run(wgs, lds, us)
        print(f"{wgs:4d} {lds:8d} {ms:9.3f} {ms / base:8.3f} {r * 1e3:11.0f}")
