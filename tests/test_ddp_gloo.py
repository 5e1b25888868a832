"""N>1 path on CPU: world_size-2 / -4 / -8 gloo processes exercise the flat-arena gradient all-reduce (GradAllReducer), the
1/world scaling contract, the per-rank sharding of a global batch and the explicit range schedules of the three trainers
(AfanTrainer's backward stages, SegTrainer's / DetTrainer's arena suffix behind the cut) incl. ranges whose sizes the world does
not divide and empty ranges.  No GPU kernels run here: the model is plain
torch modules (the reducer is generic over any nn.Module's parameters); the fused SGD step itself is covered on the GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model():
    """(odd sizes on purpose: 3*7*9 = 189, 7, 7, 7*7*9 = 441, 7, 7*36*5 = 1260, 5 elements — none divides by 8)"""
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 7, 3, padding=1), torch.nn.BatchNorm2d(7), torch.nn.ReLU(),
                               torch.nn.Conv2d(7, 7, 3, padding=1), torch.nn.Flatten(), torch.nn.Linear(7 * 6 * 6, 5))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_pkg
        pkg = load_pkg()
        model = _make_model()
        arena = pkg.arena.ParamArena(model, skip=(), bf16_shadow=False, allow_cpu=True)
        reducer = pkg.train_step.GradAllReducer(arena, n_chunks=3)
        assert reducer.world == world and len(reducer.chunks) >= 2
        # the global batch, and this rank's shard of it (main_perturb.DeviceLoader slices the same way)
        g = torch.Generator().manual_seed(11)
        xs, ys = torch.randn(16, 3, 6, 6, generator=g), torch.randint(0, 5, (16,), generator=g)
        per = 16 // world
        x, y = xs[rank * per:(rank + 1) * per], ys[rank * per:(rank + 1) * per]
        crit = torch.nn.CrossEntropyLoss()
        for it in range(2):                      # two iterations: hooks must re-arm
            arena.zero_grad()
            reducer.begin()
            crit(model(x), y).backward()
            reducer.finish()
            local_sum = arena.grad.clone()       # SUM over ranks; the SGD kernel applies grad_scale = 1/world
            # reference: every rank recomputes both shards' gradients itself
            ref = torch.zeros_like(arena.grad)
            for r in range(world):
                m2 = _make_model()
                m2.load_state_dict(model.state_dict())
                a2 = pkg.arena.ParamArena(m2, skip=(), bf16_shadow=False, allow_cpu=True)
                crit(m2(xs[r * per:(r + 1) * per]), ys[r * per:(r + 1) * per]).backward()
                ref += a2.grad
            torch.testing.assert_close(local_sum, ref, rtol=1e-5, atol=1e-6)
        # every rank holds the same reduced gradient
        gathered = [torch.zeros_like(arena.grad) for _ in range(world)]
        dist.all_gather(gathered, arena.grad)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        # explicit ranges (what the segmented A-FAN step uses: the library's layers add parameter gradients straight into
        # the arena, so no autograd hook can announce them): launched before finish(), in backward order, no hook fires;
        # whatever the ranges leave out is reduced by finish(); nothing is reduced twice
        # [(npar - 2, npar)] alone is SegTrainer's schedule (seg_trainer.py: the arena SUFFIX behind the SE point is announced at
        # seg_train_phases' "tail" yield, the head's range is left to finish()); the three-range form is AfanTrainer's stages
        npar = len(arena.params)
        # (tensor sizes are ragged — 189, 7, 441, 1260, 5 elements — but every tensor starts on a 64-float boundary of the arena, so
        # any announced range is a multiple of 256 bytes whatever the world size: no rank ever sees a remainder)
        assert all(o % 64 == 0 for o in arena.offsets + [arena.numel]) and any(p.numel() % world for p in arena.params)
        # ... DetTrainer's: the suffix behind the backbone's output (det_trainer.tail_range; the parent test checks it on the real
        # Faster-RCNN names) with the backbone left to finish(); an EMPTY range (a stage without parameters) is a no-op
        for covered in ([(npar - 2, npar), (2, npar - 2), (0, 2)], [(npar - 2, npar)], [(npar - 3, npar)], [(3, 3), (npar - 1, npar)],
                        [(0, npar)], []):
            arena.zero_grad()
            reducer.begin(explicit=True)
            crit(model(x), y).backward()
            assert not reducer._pending                                   # hooks are off in explicit mode
            assert not pkg.ops._grid_exchange                             # (no exchange yet: grid barriers may still be issued)
            for lo, hi in covered:
                before = len(reducer._pending)
                reducer.launch_params(lo, hi)
                assert len(reducer._pending) == before + (1 if hi > lo else 0)   # started before finish(); empty range: nothing
                # round 6: from the first real exchange to finish() the in-launch BatchNorm is off for the process
                # (ops.exchange_in_flight: RCCL's resident kernels would keep a grid barrier waiting until the exchange ends)
                assert pkg.ops._grid_exchange == bool(reducer._pending) and not (pkg.ops._grid_exchange and pkg.ops.GRID_BN)
            reducer.finish()
            assert not pkg.ops._grid_exchange
            torch.testing.assert_close(arena.grad, ref, rtol=1e-5, atol=1e-6)
        # a chunk whose parameters got no gradient this step is still reduced by finish()
        arena.zero_grad()
        reducer.begin()
        reducer.finish()
        assert float(arena.grad.abs().sum()) == 0.0
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001 — report to the parent
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_grad_allreduce_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    os.environ.setdefault("OMP_NUM_THREADS", "1")          # 8 ranks on 8 host CPUs
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        if p.is_alive():
            p.kill()
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


def test_trainers_tail_ranges_are_arena_suffixes(pkg):
    """The ranges the Segmentation / Detection trainers announce at their "tail" yield, on the real models' parameter names:
    contiguous suffixes of the arena (what GradAllReducer.launch_params(lo, hi) needs), the rest left to finish()."""
    m = pkg.det_model.fasterrcnn_resnet101(21)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    lo, hi = pkg.det_trainer.tail_range(names)
    assert hi == len(names) and names[lo] == "features.layer4.0.conv1.weight" and names[lo - 1].startswith("features.layer3.")
    assert all(n.startswith(("features.layer2.", "features.layer3.")) for n in names[:lo])
    assert pkg.det_trainer.tail_range(names[::-1]) is None and pkg.det_trainer.tail_range(names[:lo]) is None
    sz = [p.numel() for n, p in m.named_parameters() if p.requires_grad]
    assert 0.4 < sum(sz[lo:]) / sum(sz) < 0.5                          # 22.0 of 49.2 M parameters fly under the backbone's backward


def test_trainer_sets_grad_scale_from_world(pkg):
    """1/world lives in the fused SGD kernel's grad_scale (the all-reduce is a plain SUM)."""
    import inspect
    src = inspect.getsource(pkg.train_step.AfanTrainer.__init__)
    assert "grad_scale = 1.0 / self.world" in src
