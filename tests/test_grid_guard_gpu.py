"""Round 6: what happens when a grid barrier of the in-launch BatchNorm gives up, and what keeps it from being issued beside a
gradient exchange (cv_a-fan_amd/grid_guard.py, ops.exchange_in_flight, train_step.NullReducer).

The in-launch BatchNorm (afan_conv_*_bn_*: the chain conv - bn - relu - conv - bn - (+shortcut) - relu of
Classification/resnet_s.py:72-77 as one launch per convolution) meets its launch's other workgroups at a grid-wide barrier.
A spinner kernel that holds half of the chip's LDS (afan_occupy_cus) keeps half of such a launch's workgroups out for longer than
the barrier's bounded spin: the launch goes on with partial totals — and the trainer must end up with exactly the weights of a
run that never used the form."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _model(pkg, gpu, arch="resnet18"):
    torch.manual_seed(3)
    m = pkg.resnet_s.ARCHS[arch][0]()
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    return m


def _batches(gpu, n, batch=256):
    gen = torch.Generator().manual_seed(17)
    return [(torch.rand(batch, 3, 32, 32, generator=gen).to(gpu), torch.randint(0, 10, (batch,), generator=gen).to(gpu)) for _ in range(n)]


def _state(tr, m):
    return (tr.arena.param.clone(), tr.arena.momentum_buf.clone(), {k: v.clone() for k, v in m.state_dict().items()})


@pytest.fixture
def grid_restored(pkg):
    """grid_bn_disable() is process-wide and permanent by design: put the switches back for the tests that follow."""
    ops = pkg.ops
    old = (ops.GRID_BN_ALLOWED, ops.GRID_BN_SC)
    yield ops
    ops.GRID_BN_ALLOWED, ops.GRID_BN_SC = old
    ops.exchange_in_flight(False)
    ops._grid_refresh()
    ops.grid_barrier_error()


@pytest.mark.parametrize("graph", [True, False])
def test_barrier_give_up_is_detected_and_recovered_exactly(pkg, gpu, grid_restored, graph):
    """Six iterations at the benched shape (ResNet-18, batch 256, bf16 channels-last, K = 3), the learning rate changing every
    iteration like the warm-up's (main_perturb.py:288-293).  Before iteration 3 a spinner kernel takes 128 CUs' LDS for 0.5 s on a side
    stream: the next convolution + BatchNorm launch gets half of its 256 workgroups resident, their barrier gives up after 0.2 s.
    Expected: no exception, the guard notices by itself (exactly `depth` steps late, or at flush_guard()), switches the form off for
    the process, runs the lost iterations again — and parameters, momentum and every BatchNorm buffer equal, bit for bit, those of the
    same six iterations run on the two-launch forms from the start."""
    ops = grid_restored
    if not ops.GRID_BN_ALLOWED:
        pytest.skip("the in-launch BatchNorm is switched off in this process (AFAN_GRID_BN=0)")
    ops.GRID_BN_SC = False          # (the projection's in-launch backward sums in another order: not part of a bit comparison)
    data = _batches(gpu, 6)
    side = torch.cuda.Stream(device=gpu)
    res = {}
    for spin in (False, True):
        with ops.grid_bn(spin):     # the reference run: two-launch forms throughout
            m = _model(pkg, gpu)
            tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=3, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.05,
                                            use_graph=graph, graph_warmup=1)
            assert tr._guard is not None
            before = ops.CALLS["conv_bn_fused"]
            for i, (x, y) in enumerate(data):
                for g in tr.optimizer.param_groups:
                    g["lr"] = 0.01 * (i + 1)
                if spin and i == 3:
                    torch.cuda.synchronize()
                    ops.occupy_cus(128, 160 * 1024, 500000, stream=side)
                tr.step(x, y)
            n_lost = tr.flush_guard()
            torch.cuda.synchronize()
            if spin:
                assert ops.CALLS["conv_bn_fused"] > before, "the in-launch form never ran: nothing was tested"
                assert tr._guard.failures == 1, "the spinner did not make a barrier give up"
                assert tr._guard.lost_steps >= 1 and tr._guard.lost_steps <= 3
                assert not ops.GRID_BN_ALLOWED and not ops.GRID_BN, "the form must be off for the process after a give-up"
                assert n_lost + tr._guard.lost_steps >= 1
            else:
                assert tr._guard.failures == 0 and n_lost == 0
            assert not ops.grid_barrier_error(gpu)
            res[spin] = _state(tr, m)
    a, b = res[False], res[True]
    assert torch.equal(a[0], b[0]), f"parameters differ after the recovery: max |d| {float((a[0] - b[0]).abs().max()):.3e}"
    assert torch.equal(a[1], b[1]), "momentum buffers differ after the recovery"
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), f"state_dict entry {k} differs after the recovery"


def test_guarded_update_and_snapshot_skip_while_the_word_is_set(pkg, gpu, grid_restored):
    """The device side by itself: with the barrier's error word set, afan_sgd_step_guarded leaves parameters / momentum / shadow
    alone and afan_guarded_copy leaves the snapshot alone; with the word clear both do their work (same bits as afan_sgd_step)."""
    ops = grid_restored
    if not ops.GRID_BN_ALLOWED_AT_IMPORT:
        pytest.skip("no device-side guard without the in-launch BatchNorm")
    n = 4096 + 64
    gen = torch.Generator().manual_seed(5)
    p0, g0, m0 = (torch.randn(n, generator=gen).to(gpu) for _ in range(3))
    lr = torch.full((1,), 0.1, device=gpu)
    word = ops.grid_guard_word(gpu)
    outs = {}
    for setw in (0, 1):
        p, m, sh = p0.clone(), m0.clone(), torch.zeros(n, dtype=torch.bfloat16, device=gpu)
        word.fill_(setw)
        ops.sgd_step_(p, g0, m, lr, 0.9, 5e-4, 1.0, sh)
        dst, src = torch.zeros(64, device=gpu), torch.arange(64, dtype=torch.float32, device=gpu)
        cnt = torch.zeros(1, dtype=torch.int32, device=gpu)
        ops.guarded_copy_(dst, src, cnt)
        torch.cuda.synchronize()
        outs[setw] = (p, m, sh, dst, int(cnt))
    word.zero_()
    lib = pkg._lib.load()
    p, m, sh = p0.clone(), m0.clone(), torch.zeros(n, dtype=torch.bfloat16, device=gpu)
    pkg._lib.check(lib.afan_sgd_step(ops._ptr(p), ops._ptr(g0), ops._ptr(m), ops._ptr(sh), n, ops._ptr(lr), 0.9, 5e-4, 1.0, 0, ops._stream(p)), "sgd")
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], p) and torch.equal(outs[0][1], m) and torch.equal(outs[0][2], sh)
    assert torch.equal(outs[0][3], torch.arange(64, dtype=torch.float32, device=gpu)) and outs[0][4] == 1
    assert torch.equal(outs[1][0], p0) and torch.equal(outs[1][1], m0) and float(outs[1][2].float().abs().max()) == 0.0
    assert float(outs[1][3].abs().max()) == 0.0 and outs[1][4] == 0


def test_grid_bn_contexts_only_narrow(pkg, grid_restored):
    """grid_bn(True) inside grid_bn(False) — or while an exchange is in flight — must not switch the form back on."""
    ops = grid_restored
    allowed = ops.GRID_BN_ALLOWED
    with ops.grid_bn(False):
        assert not ops.GRID_BN
        with ops.grid_bn(True):
            assert not ops.GRID_BN
        assert not ops.GRID_BN
    assert ops.GRID_BN == allowed
    ops.exchange_in_flight(True)
    with ops.grid_bn(True):
        assert not ops.GRID_BN
    ops.exchange_in_flight(False)
    assert ops.GRID_BN == allowed


@pytest.mark.parametrize("graph", [False, True])
def test_no_grid_barrier_is_issued_while_an_exchange_is_in_flight(pkg, gpu, grid_restored, graph):
    """The data-parallel program on one GPU (AfanTrainer(emulate_dp=True): train_step.NullReducer stands where GradAllReducer starts
    RCCL): the backward in phases, the tail's ranges announced from the last stage on.  From the first announcement to finish() no
    convolution + BatchNorm launch may be issued (RCCL's resident channel kernels would keep the barrier waiting until the exchange
    ENDS) — eager (counted between the reducer's calls) and captured (counted per graph piece: only the first piece has any) — while
    before the first announcement the form IS used; and the step's results equal the unsegmented single-GPU step's up to the summation
    order of the BatchNorm-backward sums at the cuts."""
    ops = grid_restored
    if not ops.GRID_BN_ALLOWED:
        pytest.skip("the in-launch BatchNorm is switched off in this process")
    ops.GRID_BN_SC = False
    data = _batches(gpu, 4)
    res = {}
    for dp in (False, True):
        m = _model(pkg, gpu)
        tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=3, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.05,
                                        use_graph=graph, graph_warmup=1, emulate_dp=dp)
        before = ops.CALLS["conv_bn_fused"]
        for x, y in data:
            r = tr.step(x, y)
        torch.cuda.synchronize()
        assert ops.CALLS["conv_bn_fused"] > before
        if dp:
            red = tr.reducer
            assert red.fused_while_in_flight == 0
            assert len(red.announced) >= 3 and red.announced[0][1] == len(tr.arena.params), "the last stage's range comes first"
            if graph:
                assert tr._pieces is not None and len(tr._pieces) == len(red.announced)
                assert tr._pieces_fused[0] > 0 and all(n == 0 for n in tr._pieces_fused[1:]), tr._pieces_fused
        assert tr.flush_guard() == 0 and not ops.grid_barrier_error(gpu)
        res[dp] = (r["loss"].clone(), _state(tr, m))
    # (the phased backward reduces the BatchNorm-backward sums across a cut stand-alone instead of in the next dgrad's epilogue:
    # another summation order — the bound of test_train_step_gpu's segmented-vs-plain tests)
    la, lb = float(res[False][0]), float(res[True][0])
    assert abs(la - lb) <= 2e-3 * max(1.0, abs(la)), (la, lb)
    pa, pb = res[False][1][0], res[True][1][0]
    assert float((pa - pb).abs().max()) <= 5e-3 * float(pa.abs().max())      # (four iterations at lr 0.05 on a fresh network)


def test_segmentation_capture_keeps_grid_barriers_out_of_the_exchange(pkg, gpu, grid_restored):
    """SegTrainer's two-part schedule with a reducer (seg_trainer.py:103): the head's backward, issued or captured
    after the tail's exchange has started, takes the two-launch forms."""
    ops = grid_restored
    if not ops.GRID_BN_ALLOWED:
        pytest.skip("the in-launch BatchNorm is switched off in this process")
    torch.manual_seed(3)
    m = pkg.deeplab.MODELS["deeplabv3plus_resnet50"](num_classes=21, output_stride=16)
    m.set_compute_dtype(torch.bfloat16).set_channels_last(True).to(gpu).train()
    tr = pkg.seg_trainer.SegTrainer(m, steps=1, lr=0.01, use_graph=True, graph_warmup=1, segmented=True)
    tr.reducer = pkg.train_step.NullReducer(tr.arena)
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 129, 129, generator=gen).to(gpu)
    y = torch.randint(0, 21, (2, 129, 129), generator=gen).to(gpu)
    for _ in range(3):
        tr.step(x, y)
    torch.cuda.synchronize()
    assert tr.reducer.fused_while_in_flight == 0
    assert tr._graph is not None and tr._pieces_fused is not None
    tail_at = [i for i, (_, ph) in enumerate(tr._pieces) if ph == "tail"]
    assert tail_at, "the schedule never announced its tail"
    assert all(n == 0 for n in tr._pieces_fused[tail_at[0] + 1:]), tr._pieces_fused
    assert tr.flush_guard() == 0 and not ops.grid_barrier_error(gpu)


def test_second_consumer_of_a_block_output_is_refused(pkg, gpu, grid_restored):
    """ADVICE round 5 (resnet_s.py hand-off by attribute): a block whose output feeds TWO block nodes must not take the form in which
    the consumer runs the producer's last-BatchNorm backward inside its own launch — each consumer would do it from its share alone.
    With two consumers the plain path runs (consumers == 2), and the gradients equal those with block fusion off."""
    ops, rs = grid_restored, pkg.resnet_s
    m = _model(pkg, gpu)
    blocks = [b for b in m.sequential_model if isinstance(b, rs.BasicBlock)]
    b0, b1, b2 = blocks[2], blocks[3], blocks[3]        # the first stage-2 pair: b0's output read twice by b1
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(64, b0._chain()[0][0].in_channels, 32, 32, generator=gen).to(gpu).bfloat16().contiguous(memory_format=torch.channels_last)
    outs = {}
    for fusion in (True, False):
        old = rs._Flags.block_fusion
        rs._Flags.block_fusion = fusion
        try:
            for p in m.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            a = b0(xi)
            y = b1(a).float().sum() + b2(a).float().mul(0.5).sum()
            y.backward()
            torch.cuda.synchronize()
            outs[fusion] = (xi.grad.clone(), [p.grad.clone() for p in b0.parameters()])
        finally:
            rs._Flags.block_fusion = old
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
    assert rel(outs[True][0], outs[False][0]) <= 2e-2, rel(outs[True][0], outs[False][0])     # (bf16 tensors, two summation orders)
    for ga, gb in zip(outs[True][1], outs[False][1]):
        assert rel(ga, gb) <= 2e-2, rel(ga, gb)


def test_trainers_die_by_reference_count(pkg, gpu, grid_restored):
    """A dropped trainer (its hipGraphs, streams, events) must be freed at once, not by a later cyclic garbage collection: one that
    ran in the middle of another trainer's graph capture aborted the -m gpu suite once this round (grid_guard held a strong reference
    to the trainer's bound method)."""
    import gc
    import weakref
    gc.collect()
    gc.disable()
    try:
        m = _model(pkg, gpu)
        tr = pkg.train_step.AfanTrainer(m, nn.CrossEntropyLoss(), steps=1, gamma=0.5, eps=2.0, perturb_idx=6, lr=0.05, use_graph=True,
                                        graph_warmup=1)
        x, y = _batches(gpu, 1, batch=64)[0]
        for _ in range(3):
            tr.step(x, y)
        torch.cuda.synchronize()
        assert tr._graph is not None
        probe = weakref.ref(tr)
        del tr
        assert probe() is None, "the trainer is only reachable through a reference cycle"
    finally:
        gc.enable()
