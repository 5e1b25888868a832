"""Host-side logic that needs no GPU: model layout/state_dict contract, schedule helpers, arena bookkeeping,
entry-point flags."""
import numpy as np
import pytest
import torch

import os

from conftest import ROOT, golden


def test_state_dict_keys_match_reference(pkg):
    g = golden("step_r56s_k5")
    torch.manual_seed(3)
    m = pkg.resnet_s.resnet56()
    keys = list(m.state_dict().keys())
    assert keys == [str(k) for k in g["keys"]] and len(keys) == 335
    assert m.layer_number == 34
    g20 = golden("step_r20s_k1")
    m20 = pkg.resnet_s.resnet20()
    assert list(m20.state_dict().keys()) == [str(k) for k in g20["keys"]] and m20.layer_number == 16
    m18 = pkg.resnet_s.resnet18()
    assert list(m18.state_dict().keys()) == [str(k) for k in golden("step_r18_k5")["keys"]] and m18.layer_number == 15


def test_seeded_init_equals_reference(pkg):
    """Same seed, same construction order => the reference's initial weights (golden fingerprint + full tensors)."""
    g = golden("step_r20s_k1")
    torch.manual_seed(3)
    m = pkg.resnet_s.resnet20()
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), g["sd0/" + k], err_msg=k)
    for arch, case in (("resnet56s", "step_r56s_k5"), ("resnet18", "step_r18_k5")):
        torch.manual_seed(3)
        m = pkg.resnet_s.ARCHS[arch][0]()
        ck = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in m.state_dict().values()])
        np.testing.assert_array_equal(ck, golden(case)["ck0"])


def test_warmup_lr_schedule(pkg):
    class Opt:
        param_groups = [{"lr": 0.5}, {"lr": 0.5}]
    o = Opt()
    assert pkg.train_step.warmup_lr(0, o, 351, 0.1) == 0.0          # first step of epoch 0 runs with lr = 0
    assert pkg.train_step.warmup_lr(175, o, 351, 0.1) == pytest.approx(0.05)
    assert pkg.train_step.warmup_lr(350, o, 351, 0.1) == pytest.approx(0.1)
    assert pkg.train_step.warmup_lr(9999, o, 351, 0.1) == 0.1
    assert all(g["lr"] == 0.1 for g in o.param_groups)
    lrs = golden("traj_r20s")["lrs"]
    for i, lr in enumerate(lrs):
        assert pkg.train_step.warmup_lr(i, o, 5, 0.1) == float(lr)


def test_cpu_tensors_are_rejected_not_silently_computed(pkg):
    m = pkg.resnet_s.resnet20()
    with pytest.raises(pkg.AfanLibraryError):
        m(torch.rand(2, 3, 32, 32), end_point=16, start_point=0)
    with pytest.raises(pkg.AfanLibraryError):
        pkg.mix_feature(torch.rand(1, 4, 2, 2), torch.rand(1, 4, 2, 2))
    with pytest.raises(pkg.AfanLibraryError):
        pkg.arena.ParamArena(m)


def test_arena_layout_on_host(pkg):
    """Arena bookkeeping (offsets, views, chunk cuts) with plain CPU tensors: no kernel is launched."""
    m = torch.nn.Sequential(torch.nn.Linear(10, 7), torch.nn.Linear(7, 3))
    a = pkg.arena.ParamArena(m, skip=(), bf16_shadow=False, allow_cpu=True)
    assert a.numel % 64 == 0 and all(o % 64 == 0 for o in a.offsets)
    for p, o in zip(a.params, a.offsets):
        assert p.data_ptr() == a.param.data_ptr() + 4 * o and p.grad.data_ptr() == a.grad.data_ptr() + 4 * o
    m(torch.randn(4, 10)).sum().backward()
    assert float(a.grad.abs().sum()) > 0          # autograd accumulated straight into the arena
    a.zero_grad()
    assert float(a.grad.abs().sum()) == 0
    cuts = pkg.train_step._cut_chunks(a, 2)
    assert cuts[0][1] == a.numel and cuts[-1][0] == 0
    assert all(c[0] == n[1] for c, n in zip(cuts[:-1], cuts[1:]))      # contiguous, back to front


def test_main_perturb_flags(pkg):
    from importlib import import_module
    mp = import_module("cv_a-fan_amd.main_perturb")
    a = mp.parser.parse_args([])
    # reference defaults, main_perturb.py:28-49
    assert (a.batch_size, a.lr, a.momentum, a.weight_decay, a.epochs, a.decreasing_lr) == (128, 0.1, 0.9, 5e-4, 200, "50,150")
    assert (a.steps, a.perturb_idx, a.gamma, a.eps, a.randinit, a.clip) == (5, 13, 1.5, 2, False, False)
    assert (a.print_freq, a.seed, a.gpu, a.resume, a.save_dir) == (50, None, 0, False, "res56s_adv_aug")
    b = mp.parser.parse_args("--seed 3 --save_dir x --gamma 0.5".split())      # cmd/run_perturb.sh
    assert (b.seed, b.save_dir, b.gamma) == (3, "x", 0.5)


def test_no_convolution_leaves_the_library(pkg):
    """resnet_s.general_convs lists the convolutions that run on the general fp32-arithmetic kernels (fp32 parity mode: all
    16+-channel convolutions; the image stem is never listed); resnet_s.vendor_convs is empty by construction and the
    package source holds no aten / functional convolution call for GPU tensors."""
    m = pkg.resnet_s.resnet20()
    names = pkg.resnet_s.general_convs(m)
    convs = [n for n, mod in m.named_modules() if isinstance(mod, pkg.resnet_s.Conv2d) and mod.in_channels > 4]
    assert names == convs and len(names) == 18
    assert pkg.resnet_s.vendor_convs(m) == []
    assert set(pkg.ops.CALLS) == {"conv_fwd", "conv_dgrad", "conv_wgrad", "conv_general", "vendor_conv", "conv_bn_fused"}
    import glob
    import os
    import re
    src_dir = os.path.dirname(pkg.__file__)
    for path in glob.glob(os.path.join(src_dir, "*.py")):
        txt = open(path).read()
        assert not re.search(r"aten\.convolution|convolution_backward|cudnn_convolution|miopen_convolution", txt), path
        assert not re.search(r"functional\.conv2d|F\.conv2d|F\.linear|functional\.linear", txt), path


def test_faster_rcnn_seeded_construction_equals_the_reference(pkg):
    """cv_a-fan_amd/det_model.py creates, default-initialises and re-initialises its modules in the reference's order
    (Detection/model.py:23-38, backbone/resnet101_ori.py:130-171, rpn/region_proposal_network.py:15-36): from the same seed it
    holds the reference's tensors bit for bit, under the reference's 1 268 state_dict keys — including the `_bn_modules.*` and
    `detection.hidden.*` aliases (tests/golden/det_frcnn_r101.npz `ck_init`, from the reference's own Model)."""
    import numpy as np
    import torch
    from conftest import golden
    g = golden("det_frcnn_r101")
    torch.manual_seed(7)
    m = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode="pooling", anchor_sizes=(64,), rpn_pre_nms_top_n=200, rpn_post_nms_top_n=64)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]] and len(sd) == 1268
    ck = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
    np.testing.assert_array_equal(ck, g["ck_init"])
    assert sd["detection.hidden.0.conv1.weight"].data_ptr() == sd["features.layer4.0.conv1.weight"].data_ptr()
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert any(n.startswith("features.layer1.") for n in frozen) and "features.conv1.weight" in frozen
    assert all(("bn" in n or "downsample.1" in n or n.startswith(("features.conv1", "features.layer1"))) for n in frozen)


def test_seg_trainer_tail_range_is_the_arena_suffix_behind_the_se_point(pkg):
    """What SegTrainer hands to the all-reduce at seg_train_phases' "tail" yield: the contiguous arena range of the layers
    behind the SE point (backbone.layer{se+1..4} + classifier) — the LAST parameters in registration order, 53 % of
    DeepLabv3+ ResNet-101; every earlier parameter belongs to the head and is reduced at finish()."""
    import types
    model = pkg.deeplab.deeplabv3plus_resnet101(num_classes=21, output_stride=16)
    arena = pkg.arena.ParamArena(model, skip=(), bf16_shadow=False, allow_cpu=True)
    bounds = arena.offsets + [arena.numel]
    for se, frac in ((3, (0.5, 0.56)), (2, (0.9, 1.0)), (4, (0.1, 0.4))):
        stub = types.SimpleNamespace(kw={"pertub_idx_se": se}, arena=arena)
        lo, hi = pkg.seg_trainer.SegTrainer._tail_range(stub)
        assert hi == len(arena.params)
        assert all(not n.startswith(tuple(f"backbone.layer{k}." for k in range(se + 1, 5)) + ("classifier.",)) for n in arena.names[:lo])
        share = (bounds[hi] - bounds[lo]) / arena.numel
        assert frac[0] < share < frac[1], (se, share)
    assert pkg.seg_trainer.SegTrainer._tail_range(types.SimpleNamespace(kw={"pertub_idx_se": "aspp"}, arena=arena)) is None


def test_bench_starts_its_own_ranks_as_child_processes(monkeypatch):
    """`python bench.py --gpus N` with no launcher: N ranks through `python -m torch.distributed.run` as a CHILD (subprocess.call —
    never an exec of a process that may have touched the GPU), rendezvous on 127.0.0.1, dmabuf IPC kept in the environment, the
    user's flags passed through; decided before torch is imported."""
    import importlib
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7 and len(calls) == 1                         # the child's exit code is ours
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src


def test_bench_device_identity_is_16_bytes():
    import importlib
    import sys
    import types
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")

    class P:
        name, pci_bus_id, pci_device_id, pci_domain_id = "MI355X", 5, 0, 0
    fake = types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda d: P()))
    a = bench.device_uuid(fake, types.SimpleNamespace(index=0))
    P.pci_bus_id = 6
    b = bench.device_uuid(fake, types.SimpleNamespace(index=1))
    assert len(a) == len(b) == 16 and a != b


def test_tiled_convolution_kernels_use_no_scratch():
    """Every instantiation of the tiled convolution compiles without a private segment (tools/check_isa.py's rule, here for the
    one source whose 768- and 1 024-thread variants sit exactly at their register caps: in round 4 an epilogue change pushed
    them over — nine spilled registers in the K loop, the headline step 8.8 -> 11.6 ms — and only the final bench run noticed)."""
    import re
    import subprocess
    import tempfile
    # ... and the instantiations with the in-launch BatchNorm (afan_conv_bnf.hip, round 5): their epilogues hold three prefetched
    # operand tiles and a third column sum — the 768-thread x 128-column one spilled until its prefetch went two-phase
    for name, least in (("afan_conv.hip", 40), ("afan_conv_bnf.hip", 10)):
        src = os.path.join(ROOT, "cv_a-fan_amd", "csrc", name)
        with tempfile.NamedTemporaryFile(suffix=".s") as f:
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-w", "-S",
                            "--cuda-device-only", "-o", f.name, src], check=True)
            isa = open(f.name).read()
        kernels = re.findall(r"\.set (\S+)\.private_seg_size, (\d+)", isa)
        assert len(kernels) >= least, (name, len(kernels))
        assert [k for k in kernels if int(k[1]) > 0] == [], name


def test_detection_bottleneck_copies_without_its_launch_plans(pkg):
    """det_model.Bottleneck caches launch plans (ctypes pointer objects) on the module; copy.deepcopy / pickling must still work
    (EMA copies, torch.save of a module) and the copy must start without the cache."""
    import copy
    import ctypes
    import pickle
    blk = pkg.det_model.Bottleneck(64, 16)
    blk._plans, blk._plan_epoch = {"key": ctypes.c_void_p(1234)}, 7
    blk._params = tuple(blk.parameters())
    c = copy.deepcopy(blk)
    assert c._plans is None and c._plan_epoch == -1 and c._params is None
    assert [tuple(p.shape) for p in c.parameters()] == [tuple(p.shape) for p in blk.parameters()]
    assert pickle.loads(pickle.dumps(blk))._plans is None
    assert blk._plans is not None                      # the original keeps its cache


def test_padded_nms_contract_rejects_cpu_and_device_is_checked_first(pkg):
    """det_ops.nms has no CPU path; the padded form's empty-set answer is a (keep, count) pair like every other padded answer
    (checked on the GPU in tests/test_det_gpu.py)."""
    import pytest
    import torch
    with pytest.raises(pkg.ops.AfanLibraryError):
        pkg.det_ops.nms(torch.zeros(0, 4), torch.zeros(0), 0.7, padded=True)


def test_grid_bn_switches_compose_and_only_narrow(pkg):
    """ops.GRID_BN = process switch AND every enclosing grid_bn() AND no exchange in flight (a nested grid_bn(True)
    re-enabled the form inside grid_bn(False); SegTrainer's head backward ran grid barriers beside the tail's exchange)."""
    ops = pkg.ops
    allowed, ex = ops.GRID_BN_ALLOWED, ops._grid_exchange
    try:
        ops.GRID_BN_ALLOWED = True
        ops.exchange_in_flight(False)
        assert ops.GRID_BN
        with ops.grid_bn(False):
            with ops.grid_bn(True):
                assert not ops.GRID_BN
        assert ops.GRID_BN
        assert ops.exchange_in_flight(True) is False and not ops.GRID_BN
        with ops.grid_bn(True):
            assert not ops.GRID_BN
        assert ops.exchange_in_flight(False) is True and ops.GRID_BN
        import warnings
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ops.grid_bn_disable("test")
        assert w and not ops.GRID_BN and not ops.GRID_BN_ALLOWED
        with ops.grid_bn(True):
            assert not ops.GRID_BN
    finally:
        ops.GRID_BN_ALLOWED = allowed
        ops.exchange_in_flight(ex)
        ops._grid_refresh()


def test_null_reducer_flags_the_exchange(pkg):
    """From the first announced range to finish() the process-wide switch says "exchange in flight" (no grid barrier may be issued);
    GradAllReducer does the same around its real all-reduces (tests/test_ddp_gloo.py)."""
    import torch
    ops, ts = pkg.ops, pkg.train_step
    lin = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
    arena = pkg.arena.ParamArena(lin, skip=(), allow_cpu=True)
    red = ts.NullReducer(arena)
    red.begin(explicit=True)
    assert not ops._grid_exchange
    red.launch_params(2, 4)
    assert ops._grid_exchange and red.announced == [(2, 4)]
    red.launch_params(3, 3)                           # an empty range announces nothing
    assert red.announced == [(2, 4)]
    red.finish()
    assert not ops._grid_exchange and red.fused_while_in_flight == 0


def test_buffer_arena_keeps_the_state_dict_contract(pkg):
    """grid_guard.BufferArena: the BatchNorm running statistics become views of one flat tensor (one launch snapshots them); keys,
    values, load_state_dict and the dual-BN set exchange behave as before."""
    import torch
    gg = pkg.grid_guard
    m = pkg.resnet_s.resnet20()
    for b in m.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            b.running_mean.uniform_(-1, 1)
            b.running_var.uniform_(0.5, 2)
            b.num_batches_tracked.fill_(7)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    ba = gg.BufferArena(m)
    sd1 = m.state_dict()
    assert list(sd1) == list(sd0) and all(torch.equal(sd1[k], sd0[k]) for k in sd0)
    bn = next(b for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))
    assert bn.running_mean.data_ptr() >= ba.f32.data_ptr() and bn.running_mean.data_ptr() < ba.f32.data_ptr() + ba.f32.numel() * 4
    bn.running_mean.add_(1.0)
    assert float(ba.f32.sum()) != float(ba.snap_f32.sum())
    ba.restore()
    assert all(torch.equal(m.state_dict()[k], sd0[k]) for k in sd0)
    sd2 = {k: (v + 1 if v.dtype.is_floating_point else v) for k, v in sd0.items()}
    m.load_state_dict(sd2)                            # in place: the views stay views
    assert bn.running_mean.data_ptr() >= ba.f32.data_ptr() and torch.equal(bn.running_mean, sd2[[k for k in sd2 if k.endswith("running_mean")][0]])


def test_trainer_and_guard_form_no_reference_cycle(pkg):
    """A trainer must die by reference count (its hipGraphs, streams and events with it): grid_guard.GridGuard keeps only a WEAK
    reference to the trainer's `_drop_graphs` — a strong one made every trainer wait for a cyclic garbage collection, which once ran
    in the middle of another trainer's graph capture and aborted the process (round 6, the -m gpu suite)."""
    import gc
    import weakref
    gg = pkg.grid_guard

    class T(gg.GuardedTrainer):
        def _drop_graphs(self):
            self.dropped = True

    t = T()
    t._guard = gg.GridGuard.__new__(gg.GridGuard)                # (no device here: only the reference structure matters)
    t._guard.on_failure = weakref.WeakMethod(t._drop_graphs)
    probe = weakref.ref(t)
    gc.disable()
    try:
        del t
        assert probe() is None, "trainer kept alive by a reference cycle"
    finally:
        gc.enable()
    import inspect
    assert "WeakMethod(self._drop_graphs)" in inspect.getsource(gg.GuardedTrainer._guard_init)


def test_host_placement_picks_a_block_of_allowed_cores(pkg, monkeypatch):
    """cv_a-fan_amd/host.py: the rank's block lies inside the process's allowed CPUs, different local ranks take different blocks when
    the node has room, AFAN_HOST_CPUS overrides, AFAN_HOST_PIN=0 leaves everything alone; place_rank / restore round-trip."""
    import os
    import torch
    host = pkg.host
    allowed = sorted(os.sched_getaffinity(0))
    blk = host.rank_cpus(0, cores=2)
    if blk is not None:
        assert len(blk) == 2 and set(blk) <= set(allowed)
        other = host.rank_cpus(1, cores=2)
        assert other is None or set(other) <= set(allowed)
    assert host.rank_cpus(0, cores=10 ** 6) is None
    monkeypatch.setenv("AFAN_HOST_CPUS", f"{allowed[0]}")
    assert host.rank_cpus(0, cores=8) == [allowed[0]]
    monkeypatch.delenv("AFAN_HOST_CPUS")
    assert host._parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    monkeypatch.setenv("AFAN_HOST_PIN", "0")
    assert host.place_rank(0) == {"pinned": False}
    monkeypatch.delenv("AFAN_HOST_PIN")
    n0 = torch.get_num_threads()
    info = host.place_rank(0, cores=2, threads=1)
    try:
        if info["pinned"]:
            assert sorted(os.sched_getaffinity(0)) == host.rank_cpus(0, cores=2) and torch.get_num_threads() == 1
    finally:
        host.restore(info)
    assert sorted(os.sched_getaffinity(0)) == allowed and torch.get_num_threads() == n0
