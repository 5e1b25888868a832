#!/usr/bin/env python3
"""bench.py — images/sec of the A-FAN train step (BASELINE.json metric) on N MI355X of one node.

Default workload (configs[1]): ResNet-18 (CIFAR stem), A-FAN K=5 at the end of stage 1 (perturb_idx 6: 64x32x32
feature map), bf16 backbone, batch 256 PER GPU (weak scaling), synthetic 3x32x32 inputs in [0,1),
gamma 0.5/255, eps 2/255, no clip / no randinit (= cmd/run_perturb.sh flags), SGD(0.1, 0.9, 5e-4).
A "step" is one full iteration of main_perturb.py:165-201: head fwd -> 5 x (tail fwd, CE, dgrad, sign
step) -> norms -> adv tail fwd + clean full fwd -> joint loss -> backward -> [all-reduce] -> SGD.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Second workload (configs[3], the Segmentation A-FAN iteration of main_aug_final.py:149-232 on DeepLabv3+ / ResNet-101,
output stride 16, 3x513x513 inputs, 21 classes with ~5 % ignore labels, K=3 SE + SD feature PGD, mix_feature, four
forwards, two-group SGD + PolyLR):

  python bench.py --arch deeplabv3plus_resnet101 --batch 2 --pgd_steps 3 --steps 10 --warmup 4

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     — the dominant hand-written kernel by time: achieved algorithmic FLOP/s (MFMA-bound) or GB/s (HBM-bound),
                 timed per launch with HIP events on the launch stream in a separate instrumented pass (not in `value`);
                 `traffic` = HBM bytes per launch from the newest committed rocprofv3 --pmc summary that names the kernel,
                 flagged `traffic_stale` when the kernel sources changed since that summary was taken;
  conv_mfma    — EXECUTED convolution FLOPs per step (summed over the launches of the instrumented pass) / step time;
  hbm_kernels  — the A-FAN element-wise kernels (PGD step, mix_feature, lerp, ...) as fractions of the 8 TB/s HBM peak;
  cpu_baseline — the CPU oracle (oracle/afan_oracle.py, bit-identical to the reference's Python on CPU) timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured achievable)
BF16_DENSE_PEAK_TFLOPS = 2500.0
F32_MFMA_PEAK_TFLOPS = 157.3
# reference-schedule convolution / linear FLOPs per image (4H + (2K+6)T), for the secondary figure only
REF_GFLOP_PER_IMAGE = {("resnet18", 5): 14.1, ("resnet56s", 5): 2.99, ("resnet50", 3): 85.5}
ARCH_INPUT = {"resnet50": (224, 1000)}                                        # (image side, classes); default (32, 10)
SEG_ARCHS = ("deeplabv3plus_resnet101", "deeplabv3plus_resnet50")
DET_ARCHS = ("fasterrcnn_resnet101",)
HBM_KERNELS = ("pgd_step_kernel", "pgd_step_norms_kernel", "mix_feature_nhwc_kernel", "mix_feature_kernel", "lerp_points_kernel",
               "sgd_kernel", "cast_bf16_kernel", "upsample_bilinear_fwd_kernel", "upsample_bilinear_bwd_kernel", "ce2d_kernel",
               "maxpool_fwd_kernel", "maxpool_bwd_kernel", "bn_nhwc_apply_kernel", "bn_nhwc_bwd_apply_kernel")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="resnet18")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 256; 2 for the DeepLab workload)")
    ap.add_argument("--pgd_steps", type=int, default=None, help="K (default 5; 3 for the DeepLab workload)")
    ap.add_argument("--side", type=int, default=None, help="image side of the DeepLab workload (default 513)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--layout", default="nhwc", choices=["nhwc", "nchw"], help="internal activation layout")
    ap.add_argument("--no_graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--no_batch_final", action="store_true", help="run the adv and clean final passes separately (A/B)")
    ap.add_argument("--no_fold_clean", action="store_true", help="separate first PGD pass and final clean pass (A/B)")
    ap.add_argument("--no_batch_tails", action="store_true", help="DeepLab: the two sample-point forwards as two passes instead of one concatenated pass (A/B)")
    ap.add_argument("--no_fold_pgd0", action="store_true", help="DeepLab: keep the first pass of both PGD loops separate from the clean pass (A/B)")
    ap.add_argument("--dropout", type=float, default=None, help="DeepLab: override the head's nn.Dropout p (reference: 0.1)")
    ap.add_argument("--force_fold_clean", action="store_true", help="one clean tail pass regardless of the size heuristic (A/B)")
    ap.add_argument("--no_share_head", action="store_true", help="run the head twice per step like the reference's text (A/B)")
    ap.add_argument("--dual_bn", action="store_true", help="A/B: auxiliary BatchNorm set for adversarial features (an option the "
                    "reference does not have; the headline line is measured without it)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--no_literal", action="store_true", help="skip the after-the-fact run of the reference's literal schedule "
                    "(main_perturb.py:173,195-196 as written: two head passes, separate first PGD pass and final clean pass)")
    ap.add_argument("--literal_steps", type=int, default=12)
    ap.add_argument("--no_dp_schedule", action="store_true", help="skip the after-the-fact run of the data-parallel program on this one GPU")
    ap.add_argument("--cpu_steps", type=int, default=None, help="timed CPU-oracle steps (default 3; 1 for the DeepLab workload)")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


CPU_THREAD_SWEEP = (8, 16, 32, 64, 128)


def pick_threads(run_small, torch):
    """Best torch thread count for the oracle on this host: one small step per candidate (<= the machine's logical CPUs, plus the
    default), fastest wins — the default (all logical CPUs) oversubscribes 32 x 32 convolutions on a 128-thread host."""
    default = torch.get_num_threads()
    cands = sorted({t for t in CPU_THREAD_SWEEP if t <= (os.cpu_count() or default)} | {default})
    timing = {}
    for t in cands:
        torch.set_num_threads(t)
        run_small()                                   # first call at this count: pool start-up
        t0 = time.perf_counter()
        run_small()
        timing[t] = round(time.perf_counter() - t0, 3)
    best = min(timing, key=timing.get)
    torch.set_num_threads(best)
    return best, timing, default


def cpu_baseline(arch, batch, pgd_steps, idx, timed_steps):
    """The oracle's full step on the host cores (fp32), at the best thread count of a sweep, bounded sample."""
    import torch
    import torch.nn as nn
    from oracle import afan_oracle as orc
    torch.manual_seed(3)
    model = orc.ARCHS[arch][0]()
    model.train()
    opt = orc.make_optimizer(model)
    crit = nn.CrossEntropyLoss()
    ln = len(model.sequential_model)
    side, ncls = ARCH_INPUT.get(arch, (32, 10))
    x, y = torch.rand(batch, 3, side, side), torch.randint(0, ncls, (batch,))
    kw = dict(steps=pgd_steps, gamma=0.5, eps=2.0, perturb_idx=idx, layer_number=ln)
    nsmall = min(batch, 32)
    best, sweep, default = pick_threads(lambda: orc.afan_train_step(model, opt, crit, x[:nsmall], y[:nsmall], **kw), torch)
    per = []
    for _ in range(timed_steps):
        t0 = time.perf_counter()
        orc.afan_train_step(model, opt, crit, x, y, **kw)
        per.append(time.perf_counter() - t0)
    dt = sum(per)
    torch.set_num_threads(default)
    return {"value": round(batch * timed_steps / dt, 2), "unit": "images/sec", "cores": best,
            "kind": "port", "cpu": cpu_model(), "thread_sweep_s_per_small_step": sweep,
            "sample": f"{timed_steps} full steps of the same workload (batch {batch}, K={pgd_steps}, fp32) at {best} threads = the fastest "
                      f"of a sweep over {sorted(sweep)} threads on one batch-{nsmall} step each; per step {[round(p, 2) for p in per]} s; "
                      f"host {os.cpu_count()} logical CPUs"}


def cpu_baseline_seg(arch, batch, pgd_steps, side, timed_steps):
    """The oracle's Segmentation iteration (oracle.seg_train_step on oracle.SegDeepLabV3Plus) on the host cores."""
    import torch
    import torch.nn as nn
    from oracle import afan_oracle as orc
    torch.manual_seed(3)
    model = orc.deeplabv3plus_resnet101(21, 16) if arch.endswith("101") else orc.deeplabv3plus_resnet50(21, 16)
    opt = orc.seg_make_optimizer(model, lr=0.01)
    model.train()
    crit = nn.CrossEntropyLoss(ignore_index=255, reduction="mean")
    x, y = synth_seg(batch, side, torch.Generator().manual_seed(3))
    kw = dict(steps=pgd_steps, eps=2.0, gamma_se=0.5, gamma_sd=0.5, pertub_idx_se=3, pertub_idx_sd="aspp", mix_layer="11", mix_sd=True)
    xs_, ys_ = x[:2, :, :129, :129].contiguous(), y[:2, :129, :129].contiguous()
    best, sweep, default = pick_threads(lambda: orc.seg_train_step(model, opt, crit, xs_, ys_, **kw), torch)
    per = []
    for _ in range(timed_steps):
        t0 = time.perf_counter()
        orc.seg_train_step(model, opt, crit, x, y, **kw)
        per.append(time.perf_counter() - t0)
    dt = sum(per)
    torch.set_num_threads(default)
    return {"value": round(batch * timed_steps / dt, 3), "unit": "images/sec", "cores": best,
            "kind": "port", "cpu": cpu_model(), "thread_sweep_s_per_small_step": sweep,
            "sample": f"{timed_steps} full iteration(s) of the same workload (batch {batch}, {side}x{side}, K={pgd_steps}, fp32) at {best} "
                      f"threads = the fastest of a sweep over {sorted(sweep)} on one 2x129x129 iteration each; per step "
                      f"{[round(p, 1) for p in per]} s; host {os.cpu_count()} logical CPUs"}


def synth_seg(batch, side, g):
    """rand(N,3,S,S) images, labels randint(0,21) with ~5 % set to 255 (ignore)."""
    import torch
    x = torch.rand(batch, 3, side, side, generator=g)
    y = torch.randint(0, 21, (batch, side, side), generator=g)
    y[torch.rand(batch, side, side, generator=g) < 0.05] = 255
    return x, y


def _csrc_files():
    return sorted(glob.glob(os.path.join(ROOT, "cv_a-fan_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cv_a-fan_amd", "csrc", "*.h")))


def kernel_sources_sha():
    h = hashlib.sha256()
    for f in _csrc_files():
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def source_shas():
    """{source file: sha} of every kernel source: what a counter summary's `_meta.kernel_sources_sha_files` records (tools/pmc_summary.py,
    tools/pmc_mfma_summary.py), so that staleness is judged per KERNEL (below), not by any edit anywhere in csrc/."""
    return {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()[:16] for f in _csrc_files()}


# kernel label (afan_profile / rocprofv3 symbol prefix) -> the sources that kernel is compiled from; anything else: every file
_CONV_SRC = ("afan_conv.hip", "afan_conv_bnf.hip", "afan_conv_params.h", "afan_common.h")
KERNEL_SOURCES = {"conv_bn_": _CONV_SRC, "conv_igemm_": _CONV_SRC, "conv_wgrad": ("afan_wgrad.hip", "afan_common.h"),
                  "wgrad_": ("afan_wgrad.hip", "afan_common.h"), "conv3x3_c64": ("afan_conv_c64.hip", "afan_conv_c64.h", "afan_common.h"),
                  "conv_f32": ("afan_conv_f32.hip", "afan_common.h"), "pgd_": ("afan_pgd.hip", "afan_common.h"),
                  "sgd_": ("afan_sgd.hip", "afan_common.h")}


def summary_stale(kernel_name, meta):
    """Has a source of `kernel_name` changed since the counter summary with this `_meta` was taken?"""
    old = meta.get("kernel_sources_sha_files")
    if not old:                       # (summaries of rounds 1-5: one hash over all of csrc/)
        return meta.get("kernel_sources_sha") != kernel_sources_sha()
    now = source_shas()
    files = next((v for k, v in KERNEL_SOURCES.items() if kernel_name.startswith(k)), tuple(now))
    return any(old.get(f) != now.get(f) for f in files)


def pmc_traffic(kernel_name, arch):
    """HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes: the guide's gfx950 correction) of `kernel_name` from the
    newest profiles/*_pmc_hbm_traffic*.json collected on THIS workload (its `_meta.command` names the same --arch; summaries
    without `_meta` predate round 2 and count as the default workload) that has it; (bytes, source file, stale?, command) or None."""
    kernel_name = kernel_name.replace("conv_bn_", "conv_igemm_")      # (the traffic summaries key the tiled kernel by its symbol)
    base = kernel_name.replace("_fwd_kernel", "_kernel").replace("_dgrad_kernel", "_kernel")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json")), reverse=True):
        try:
            pmc = json.load(open(f))
        except (OSError, ValueError):
            continue
        cmd = (pmc.get("_meta", {}).get("command") or "")
        f_arch = cmd.split("--arch", 1)[1].split()[0] if "--arch" in cmd else "resnet18"
        if f_arch != arch:
            continue
        key = next((k for k in pmc if not k.startswith("_") and (k == kernel_name or k == base or k in kernel_name)), None)
        if key and "FETCH_SIZE" in pmc[key] and "WRITE_SIZE" in pmc[key]:
            meta = pmc.get("_meta", {})
            stale = summary_stale(kernel_name, meta)
            return (round((2 * pmc[key]["FETCH_SIZE"]["avg"] + pmc[key]["WRITE_SIZE"]["avg"]) * 1024),
                    os.path.relpath(f, ROOT), stale, meta.get("command"))
    return None


def pmc_mfma(kernel_name, arch):
    """MFMA utilisation of `kernel_name` (all its variants, weighted by launches) from the newest profiles/r*_pmc_mfma.json taken on
    this workload (tools/gpu_pmc_mfma.sh): SQ_VALU_MFMA_BUSY_CYCLES / (GPU-active cycles x 4 SIMDs x 256 CUs)."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")), reverse=True):
        try:
            pmc = json.load(open(f))
        except (OSError, ValueError):
            continue
        meta = pmc.get("_meta", {})
        cmd = meta.get("command") or ""
        f_arch = cmd.split("--arch", 1)[1].split()[0] if "--arch" in cmd else "resnet18"
        if f_arch != arch:
            continue
        # "conv_bn_*" = the tiled kernel's instantiations with the in-launch BatchNorm (last template argument true)
        fused, base = kernel_name.startswith("conv_bn_"), kernel_name.replace("conv_bn_", "conv_igemm_")
        vs = [(k, e) for k, e in pmc.items() if not k.startswith("_") and k.split("<")[0] == base and e.get("mfma_util") is not None
              and (k.endswith(",true>") == fused or k.count(",") < 9)]
        n = sum(e["launches"] for _, e in vs)
        if n:
            w = lambda key: round(sum((e.get(key) or 0.0) * e["launches"] for _, e in vs) / n, 4)
            return {"mfma_util": w("mfma_util"), "wait_any": w("wait_any"), "wait_inst": w("wait_inst"), "lds_issue": w("lds_issue"),
                    "variants": {k: e["mfma_util"] for k, e in vs}, "source": os.path.relpath(f, ROOT),
                    "stale": summary_stale(kernel_name, meta)}
    return None


def _second_run(pkg, torch, nn, args, dev, idx, xs, ys, timed_steps, **trainer_kw):
    """A fresh model of the same architecture from the same seed, run over EXACTLY the batches of the headline run (warm-up
    0..W-1, then 0..K-1) so that its final loss is comparable with the headline's; the last `timed_steps` of them are timed."""
    ctor, _ = pkg.resnet_s.ARCHS[args.arch]
    torch.manual_seed(3)
    model = ctor()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.set_compute_dtype(dtype).set_channels_last(args.layout == "nhwc").to(dev).train()
    tr = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=args.pgd_steps, gamma=0.5, eps=2.0, perturb_idx=idx, lr=0.1,
                                    use_graph=not args.no_graph, **trainer_kw)
    nb = len(xs)
    seq = list(range(args.warmup)) + list(range(args.steps))
    timed_steps = max(1, min(timed_steps, len(seq) - 1))
    for i in seq[:-timed_steps]:
        tr.step(xs[i % nb], ys[i % nb])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in seq[-timed_steps:]:
        r = tr.step(xs[i % nb], ys[i % nb])
    torch.cuda.synchronize()
    return tr, r, time.perf_counter() - t0, timed_steps, len(seq)


def literal_schedule(pkg, torch, nn, args, dev, idx, xs, ys, folded_loss):
    """The reference's schedule as written (main_perturb.py:173,195-196: head forward for PGD AND inside the clean forward, K PGD
    passes, adversarial + clean final passes), on a fresh model of the same architecture, hipGraph replay, after the timed region."""
    tr, r, dt, n_timed, n_total = _second_run(pkg, torch, nn, args, dev, idx, xs, ys, args.literal_steps, share_head=False, fold_clean=False)
    args_literal_steps = n_timed
    pkg.ops.profile_enable(True)
    tr._step_eager(xs[0], ys[0])
    torch.cuda.synchronize()
    prof = pkg.ops.profile_collect()
    pkg.ops.profile_enable(False)
    cf = sum(q["flops"] for k, q in prof.items() if k.startswith("conv_"))
    return {"images_per_s": round(args.batch * args_literal_steps / dt, 1), "ms_per_step": round(dt / args_literal_steps * 1e3, 3),
            "steps": args_literal_steps, "executed_GFLOP_per_step": round(cf / 1e9, 1), "hipgraph": tr._graph is not None,
            "final_loss": round(float(r["loss"]), 4), "final_loss_folded": round(folded_loss, 4), "steps_from_seed": n_total,
            "loss_note": "both losses after the SAME number of steps on the same batches from the same seed (value-identical schedules: "
                         "they differ by bf16 summation noise amplified over the steps)",
            "schedule": "2 head passes, %d PGD passes + adversarial and clean final passes%s (bench.py --no_fold_clean --no_share_head)"
                        % (args.pgd_steps, " (one grouped pass over the tail)" if getattr(tr, "_groupable", False) else "")}


def dp_schedule(pkg, torch, nn, args, dev, idx, xs, ys, folded_loss):
    """One GPU running EXACTLY the code path of a rank of a data-parallel job (AfanTrainer(emulate_dp=True)): the backward cut into
    phases at the tail's stage transitions, one hipGraph per phase with a host call between two replays (where RCCL's all-reduce of
    the finished range would be started), the in-launch BatchNorm off from the first announced range on — everything but the
    exchange itself.  The 1 -> N efficiency of a SCALE run is to be read against THIS single-GPU number: the difference to `value`
    is what the data-parallel schedule costs before a single byte moves."""
    tr, r, dt, n_timed, n_total = _second_run(pkg, torch, nn, args, dev, idx, xs, ys, args.literal_steps, emulate_dp=True)
    red = tr.reducer
    b = tr.arena.offsets + [tr.arena.numel]
    return {"images_per_s": round(args.batch * n_timed / dt, 1), "ms_per_step": round(dt / n_timed * 1e3, 3), "steps": n_timed,
            "final_loss": round(float(r["loss"]), 4), "final_loss_headline": round(folded_loss, 4), "steps_from_seed": n_total,
            "graph_pieces": len(tr._pieces) if tr._pieces is not None else 0,
            "fused_launches_per_piece": getattr(tr, "_pieces_fused", None),
            "announced_ranges_MiB": [round((b[hi] - b[lo]) * 4 / 2 ** 20, 1) for lo, hi in red.announced],
            "in_launch_batchnorm_while_exchange_in_flight": red.fused_while_in_flight,
            "note": "world 1, no exchange: train_step.NullReducer stands where GradAllReducer starts RCCL; same graphs, same host calls"}


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves — `python -m torch.distributed.run`
    as a CHILD process (never an exec), decided before anything in this process touches the GPU — relay its output and exit
    with its code.  Under torch.distributed.run (WORLD_SIZE set) this is not reached."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_uuid(torch, dev):
    """16 bytes naming the physical GPU behind `dev` (uuid when this torch exposes it, else the PCI location)."""
    prop = torch.cuda.get_device_properties(dev)
    u = getattr(prop, "uuid", None)
    raw = getattr(u, "bytes", None)
    if raw is None:
        raw = hashlib.md5(f"{getattr(prop, 'pci_domain_id', 0)}:{getattr(prop, 'pci_bus_id', dev.index)}:"
                          f"{getattr(prop, 'pci_device_id', 0)}:{prop.name}".encode()).digest()
    return bytes(raw)[:16].ljust(16, b"\0")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist
    import torch.nn as nn
    pkg = importlib.import_module("cv_a-fan_amd")
    # one process per GPU on ONE block of cores of the GPU's NUMA node, a small intra-op pool (cv_a-fan_amd/host.py): before the GPU is
    # initialised, so that the runtime's threads inherit the mask; restored around the CPU baseline, which wants the whole machine
    placement = pkg.host.place_rank()
    seg = args.arch in SEG_ARCHS
    det = args.arch in DET_ARCHS
    if det:
        args.no_cpu_baseline = True           # (no CPU restatement of Faster-RCNN travels: the goldens come from the reference itself)
    if args.batch is None:
        args.batch = 2 if seg else (1 if det else 256)
    if args.pgd_steps is None:
        args.pgd_steps = 3 if seg else 5
    if args.side is None:
        args.side = 513
    if args.cpu_steps is None:
        args.cpu_steps = 1 if seg else 3

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if os.environ.get("AFAN_BENCH_ONE_DEVICE"):     # functional test of the N>1 path on a 1-GPU box (gloo, all ranks on cuda:0)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("AFAN_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm (over xGMI)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    ranks_seen = None
    if world > 1:
        # proof of what the collective saw: every rank contributes the identity of its GPU over the SAME group the gradient
        # exchange uses; rank 0 reports them (distinct devices == world on a real node; 1 under AFAN_BENCH_ONE_DEVICE)
        mine = torch.tensor(list(device_uuid(torch, dev)), dtype=torch.uint8, device=dev if dist.get_backend() == "nccl" else "cpu")
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        ids = [bytes(t.cpu().tolist()).hex() for t in seen]
        ranks_seen = {"world": world, "backend": dist.get_backend(), "gathered": len(ids), "distinct_devices": len(set(ids)),
                      "device_ids": ids}
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(3)                      # same initial weights on every rank (reference --seed 3)
    g = torch.Generator().manual_seed(3 + rank)          # each rank its own shard of the synthetic stream
    nbuf = 4
    idx = None
    if det:
        # BASELINE configs[4]: Faster-RCNN / ResNet-101, multi-layer SAT feature perturbation (train_aug_sat_muti_advt.py:70-172),
        # VOC-shape images (config/config.py: min side 600, max side 1000), 21 classes, one image per GPU (config/train_config.py)
        model = pkg.det_model.fasterrcnn_resnet101(21, pooler_mode="align")
        # no pretrained weights offline: the frozen BatchNorms are identities and 33 undamped residual blocks overflow the
        # activations (no proposals survive); every block's last-BatchNorm weight x 0.2 puts the random network in the O(1)
        # regime pretrained weights are in (the same device as tests/golden/det_frcnn_r101.npz).  Work per step is unchanged.
        for b_ in model.modules():
            if isinstance(b_, pkg.det_model.Bottleneck):
                b_.bn3.weight.data.mul_(0.2)
        model.set_compute_dtype(dtype).set_channels_last(args.layout == "nhwc").to(dev).train()
        # data parallel like the reference's nn.DataParallel (train_aug_sat_muti_advt.py:36): det_trainer.DetTrainer — replicas from
        # rank 0, the fp32 gradient arena summed tail first under the backbone's backward, 1/world in the SGD launch
        trainer = pkg.det_trainer.DetTrainer(model, lr=0.001, momentum=0.9, weight_decay=0.0005, loss_settings=1, noise_ahead=True)
        side, ncls = (600, 904), 21
        xs, ys = [], []
        for _ in range(nbuf):
            xs.append(torch.rand(args.batch, 3, side[0], side[1], generator=g).to(dev))
            x0 = torch.rand(args.batch, 6, 1, generator=g) * (side[1] - 260)
            y0 = torch.rand(args.batch, 6, 1, generator=g) * (side[0] - 260)
            wh = 60 + torch.rand(args.batch, 6, 2, generator=g) * 200
            ys.append((torch.cat([x0, y0, x0 + wh[..., :1], y0 + wh[..., 1:]], dim=-1).to(dev),
                       torch.randint(1, 21, (args.batch, 6), generator=g).to(dev)))

        def one(i):
            return trainer.step(xs[i % nbuf], ys[i % nbuf][0], ys[i % nbuf][1])

        def one_eager(i):          # the instrumented pass times launches with events: the backbone stages un-replayed
            old_ = pkg.det_model._StageGraphs.ON
            pkg.det_model._StageGraphs.ON = False
            try:
                return one(i)
            finally:
                pkg.det_model._StageGraphs.ON = old_
    elif seg:
        model = pkg.deeplab.MODELS[args.arch](num_classes=21, output_stride=16)
        model.set_compute_dtype(dtype).set_channels_last(args.layout == "nhwc").to(dev).train()
        trainer = pkg.seg_trainer.SegTrainer(model, nn.CrossEntropyLoss(ignore_index=255, reduction="mean"), steps=args.pgd_steps,
                                             eps=2.0, gamma_se=0.5, gamma_sd=0.5, pertub_idx_se=3, pertub_idx_sd="aspp",
                                             mix_layer="11", mix_sd=True, lr=0.01, use_graph=not args.no_graph,
                                             dual_bn=args.dual_bn,
                                             fold_clean=False if args.no_fold_clean else None,
                                             fold_pgd0=False if (args.no_fold_clean or args.no_fold_pgd0) else None,
                                             batch_tails=False if args.no_batch_tails else None)
        if args.dropout is not None:               # (the reference's DeepLab head has nn.Dropout(0.1), _deeplab.py:185)
            for m_ in model.modules():
                if isinstance(m_, nn.Dropout):
                    m_.p = args.dropout
        side, ncls = args.side, 21
        data = [synth_seg(args.batch, side, g) for _ in range(nbuf)]
        xs, ys = [d[0].to(dev) for d in data], [d[1].to(dev) for d in data]

        def one(i):
            r = trainer.step(xs[i % nbuf], ys[i % nbuf])
            trainer.scheduler.step()               # main_aug_final.py:261: PolyLR once per iteration
            return r

        def one_eager(i):
            trainer.optimizer._sync_lr()
            return trainer._body(xs[i % nbuf], ys[i % nbuf])
    else:
        ctor, idx = pkg.resnet_s.ARCHS[args.arch]
        model = ctor()
        model.set_compute_dtype(dtype).set_channels_last(args.layout == "nhwc").to(dev).train()
        trainer = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=args.pgd_steps, gamma=0.5, eps=2.0,
                                             perturb_idx=idx, lr=0.1, use_graph=not args.no_graph,
                                             batch_final=not args.no_batch_final, share_head=not args.no_share_head,
                                             fold_clean=(False if args.no_fold_clean else (True if args.force_fold_clean else None)),
                                             dual_bn=args.dual_bn)
        side, ncls = ARCH_INPUT.get(args.arch, (32, 10))
        xs = [torch.rand(args.batch, 3, side, side, generator=g).to(dev) for _ in range(nbuf)]
        ys = [torch.randint(0, ncls, (args.batch,), generator=g).to(dev) for _ in range(nbuf)]

        def one(i):
            return trainer.step(xs[i % nbuf], ys[i % nbuf])

        def one_eager(i):
            return trainer._step_eager(xs[i % nbuf], ys[i % nbuf])

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        r = one(i)
    sync()
    marks = [] if os.environ.get("AFAN_BENCH_STEP_TIMES") == "1" else None      # host time at each step's return (stderr; diagnosis only)
    t0 = time.perf_counter()
    for i in range(args.steps):
        r = one(i)
        if marks is not None:
            marks.append(time.perf_counter())
    sync()
    dt = time.perf_counter() - t0
    if marks and rank == 0:
        import gc
        per = [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)]
        print("step times (ms, host clock at return):", per, " gc counts", gc.get_count(), " gc stats", gc.get_stats(), file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(r["loss"])
    assert loss == loss, "loss is NaN"
    step_ms = dt / args.steps * 1e3
    # the convolution + BatchNorm launches meet at a grid-wide barrier whose spin is bounded: a spin that gave up (workgroups of a
    # launch not co-resident) invalidates everything after it — fail loudly rather than print a number
    assert not pkg.ops.grid_barrier_error(dev), "a grid barrier of the in-launch BatchNorm gave up (another process's kernels on this GPU?)"

    # ---- instrumented pass (not part of `value`): per-launch HIP-event timing of the hand-written kernels ----
    roof, kernels, conv_exec, hbm = None, None, None, None
    graphed = trainer._graph is not None or (det and bool(pkg.det_model._StageGraphs.cache))
    NP = 2
    if not args.no_roofline:
        # per-launch event timing needs eager launches: the same step body, un-captured.  EVERY rank runs it (the step
        # contains the gradient all-reduce); only rank 0 records.
        if rank == 0:
            pkg.ops.profile_enable(True)
        for i in range(NP):
            one_eager(i)
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        prof = pkg.ops.profile_collect()
        pkg.ops.profile_enable(False)
        # every instrumented duration = kernel + one empty (event, event) bracket: measure that bracket and take it out
        ev_us = pkg.ops.profile_event_overhead(512)
        for v_ in prof.values():
            v_["ms_raw"] = v_["ms"]
            v_["ms"] = max(v_["ms"] - v_["launches"] * ev_us * 1e-3, 0.25 * v_["ms"])

        def _k(v):
            d = {"launches_per_step": v["launches"] // NP, "avg_us": round(v["ms"] * 1e3 / v["launches"], 2),
                 "ms_per_step": round(v["ms"] / NP, 3)}
            if v["flops"] > 0:
                d["algo_TFLOPs"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)
            else:
                d["algo_GBps"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None
            return d
        kernels = {k: _k(v) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        name, v = max(prof.items(), key=lambda kv: kv[1]["ms"])
        hand_ms = round(sum(q["ms"] for q in prof.values()) / NP, 3)
        if v["flops"] > 0:      # MFMA-bound kernel: algorithmic FLOPs / measured launch time vs the dense bf16 peak
            ach = v["flops"] / (v["ms"] * 1e-3) / 1e12
            f32k = "_f32_" in name      # the general kernels multiply in fp32 (v_mfma_f32_32x32x2_f32): priced against the f32 matrix peak
            peak = F32_MFMA_PEAK_TFLOPS if f32k else BF16_DENSE_PEAK_TFLOPS
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "frac_raw": round(v["flops"] / (v["ms_raw"] * 1e-3) / 1e12 / peak, 4),
                    "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2),
                    "algo_flops_per_launch": round(v["flops"] / v["launches"]),
                    "handwritten_ms_per_step": hand_ms, "event_overhead_us": round(ev_us, 2),
                    "avg_launch_us_raw": round(v["ms_raw"] * 1e3 / v["launches"], 2),
                    "timing_note": "HIP events around each eager launch on its stream, minus the measured empty-bracket time "
                                   "(event_overhead_us) per launch; un-graphed launches still run with cold instruction caches "
                                   "and no overlap, so the sum can exceed the graph-replayed ms_per_step: frac is a lower bound",
                    "peak_note": ("f32-input MFMA peak (MI355X_MICROARCH.md: 157.3 TFLOP/s, 1/16 of the bf16 rate): fp32 parity mode"
                                  if f32k else
                                  "nominal dense bf16 MFMA peak (MI355X_MICROARCH.md); measured on this chip: bare MFMA loop "
                                  "1.75-2.1 PFLOP/s, LDS -> MFMA consumer loop on random operands 1.13-1.41 PFLOP/s "
                                  "(tools/probe/mfma_rate.hip, lds_mfma.hip; NOTES.md 9.5)")}
        else:
            ach = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "frac_raw": round(v["bytes"] / (v["ms_raw"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2),
                    "algo_bytes_per_launch": round(v["bytes"] / v["launches"]), "handwritten_ms_per_step": hand_ms,
                    "event_overhead_us": round(ev_us, 2), "avg_launch_us_raw": round(v["ms_raw"] * 1e3 / v["launches"], 2)}
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process; the newest committed
        # summary of the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes supplies it (per launch)
        tr = pmc_traffic(name, args.arch)
        if tr is not None:
            roof["traffic"], roof["traffic_source"], roof["traffic_stale"] = tr[0], tr[1], tr[2]
            roof["traffic_note"] = ("rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE per launch; stale = the kernel sources changed "
                                    "since that summary was taken" + (f"; measured on: {tr[3]}" if tr[3] else ""))
        mu = pmc_mfma(name, args.arch) if roof["bound"] == "mfma" else None
        if mu is not None:
            roof["mfma_util"] = mu.pop("mfma_util")
            roof["mfma_counters"] = mu
        # executed convolution FLOPs of one step: every conv launch records its own 2*M*N*K
        cf = sum(q["flops"] for k, q in prof.items() if k.startswith("conv_")) / NP
        conv_ms = sum(q["ms"] for k, q in prof.items() if k.startswith("conv_")) / NP
        if cf > 0:
            tf = cf / (step_ms * 1e-3) / 1e12
            cpeak = F32_MFMA_PEAK_TFLOPS if args.dtype == "fp32" else BF16_DENSE_PEAK_TFLOPS
            conv_exec = {"executed_GFLOP_per_step": round(cf / 1e9, 1), "achieved_TFLOPs": round(tf, 1),
                         "peak_TFLOPs": cpeak, "frac": round(tf / cpeak, 4),
                         "conv_kernel_ms_per_step_eager": round(conv_ms, 3),
                         "in_kernel_TFLOPs": round(cf / (conv_ms * 1e-3) / 1e12, 1) if conv_ms > 0 else None,
                         "note": "FLOPs of the convolution launches the step EXECUTES (summed over the instrumented pass), "
                                 "divided by the timed step; in_kernel = the same FLOPs over the summed conv kernel time"}
        hbm = {}
        for k in HBM_KERNELS:
            if k in prof and prof[k]["ms"] > 0 and prof[k]["bytes"] > 0:
                gbs = prof[k]["bytes"] / (prof[k]["ms"] * 1e-3) / 1e9
                hbm[k] = {"achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4),
                          "algo_bytes_per_launch": round(prof[k]["bytes"] / prof[k]["launches"]),
                          "avg_us": round(prof[k]["ms"] * 1e3 / prof[k]["launches"], 2)}
    # ---- N > 1: three more steps with every exchange bracketed by events (not part of `value`) ----
    ddp_diag = None
    red = getattr(trainer, "reducer", None)
    if world > 1 and red is not None:
        red.diag = True
        steps_ = []
        for i in range(3):
            one(i)
            torch.cuda.synchronize()
            d_ = red.last_diag()
            if d_ is not None:
                steps_.append(d_)
        red.diag = False
        if rank == 0 and steps_:
            ddp_diag = {"steps": steps_, "note": "rank 0, three steps after the timed region: per announced gradient range the bytes, the wait "
                        "behind earlier ranges on the exchange stream (queued_ms) and the all-reduce's own duration; exposed_allreduce_ms = "
                        "how long the compute stream stood waiting for the exchange stream before SGD; host_gap_us = host time spent "
                        "launching the exchange between two graph replays"}
    if world > 1:
        dist.barrier()

    literal = None
    if rank == 0 and world == 1 and not (seg or det) and not args.no_literal and trainer._fold_ok(xs[0]):
        literal = literal_schedule(pkg, torch, nn, args, dev, idx, xs, ys, loss)
    # ---- the data-parallel PROGRAM on this one GPU (train_step.NullReducer): what a rank of an N > 1 job runs, minus the exchange
    dp_sched = None
    if rank == 0 and world == 1 and not (seg or det) and not args.no_dp_schedule and trainer._fold_ok(xs[0]):
        dp_sched = dp_schedule(pkg, torch, nn, args, dev, idx, xs, ys, loss)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        pkg.host.restore(placement)
        cpu = (cpu_baseline_seg(args.arch, args.batch, args.pgd_steps, side, args.cpu_steps) if seg
               else cpu_baseline(args.arch, args.batch, args.pgd_steps, idx, args.cpu_steps))

    if rank == 0:
        ips = args.batch * world * args.steps / dt
        default_cfg = args.arch == "resnet18" and args.pgd_steps == 5
        if det:
            metric = f"images/sec (whole node) Detection multi-layer SAT A-FAN train step, {args.arch} {side[0]}x{side[1]}"
            workload = (f"{args.arch} (frozen-BatchNorm ResNet-101, RPN 9 anchors, ROIAlign pooler), VOC-shape {side[0]}x{side[1]} synthetic, "
                        f"image PGD 5 steps + 3 one-step feature PGDs + ROI-feature PGD, 5 SAT points, {args.dtype}, batch {args.batch}/GPU, "
                        f"internal layout {args.layout} (BASELINE configs[4], per-GPU share)")
            sched = ("train_aug_sat_muti_advt.py:70-172 as written (eight training forwards, host-side proposal sampling); the backbone's stages "
                     "(frozen-BatchNorm bottlenecks, BatchNorm in the convolution epilogues) replayed from hipGraphs forward and backward, "
                     "RPN / proposal / ROI heads launched eagerly (their shapes follow the proposals)"
                     if pkg.det_model._StageGraphs.ON and pkg.det_model._StageGraphs.cache else
                     "train_aug_sat_muti_advt.py:70-172 as written (eight training forwards, host-side proposal sampling; the image PGD's host noise "
                     "draw of iteration i + 1 issued behind iteration i's backward: the same generator stream); eager launches")
        elif seg:
            metric = f"images/sec (whole node) Segmentation A-FAN K={args.pgd_steps} train step, {args.arch} {side}x{side}"
            workload = (f"{args.arch} output-stride 16, VOC-shape {side}x{side} synthetic, SE (layer3) + SD (aspp) feature PGD K="
                        f"{args.pgd_steps}, mix_feature 11 + mix_sd, {args.dtype}, batch {args.batch}/GPU, internal layout "
                        f"{args.layout} (BASELINE configs[3], per-GPU share)")
            # which passes ran (seg_train_step reports its decisions): the reference's text is head pass (:166) + clean decoder-head
            # pass (:167) + K SE + K SD PGD passes + clean / SE1 / SE2 / SD forwards (:193-209)
            K_ = args.pgd_steps
            drop_p = max([m_.p for m_ in model.modules() if isinstance(m_, nn.Dropout)] + [0.0])
            se12 = "SE1 + SE2 as ONE concatenated pass (BatchNorm statistics and dropout per half)" if r.get("batch_tails") else "SE1 / SE2"
            if r.get("fold_pgd0"):
                sched = (f"ONE clean pass standing for :166, :167, :193 AND the first pass of both PGD loops (dropout p = {drop_p}: "
                         f"identical passes), {K_ - 1} SE + {K_ - 1} SD PGD passes, {se12} / SD forwards, one joint backward")
            elif r.get("fold_clean"):
                sched = (f"ONE clean pass standing for :166, :167 and :193 (two dropout draws on its ASPP output, p = {drop_p}), "
                         f"{K_} SE + {K_} SD PGD passes, {se12} / SD forwards, one joint backward")
            else:
                sched = (f"main_aug_final.py:158-232 as written: head pass + clean decoder-head pass, {K_} SE + {K_} SD PGD passes, "
                         "clean / SE1 / SE2 / SD forwards, one joint backward")
            if trainer._phased() and r.get("fold_clean"):
                lo_, hi_ = trainer._tail_range()
                b_ = trainer.arena.offsets + [trainer.arena.numel]
                sched += (f"; data parallel: the backward runs in two captured parts, the all-reduce of the tail's gradients "
                          f"({(b_[hi_] - b_[lo_]) * 4 / 2 ** 20:.0f} MiB of {trainer.arena.numel * 4 / 2 ** 20:.0f} MiB: layer4, ASPP, decoder) "
                          "starts between them on a side stream, the head's follows the second part")
        else:
            metric = ("images/sec (whole node) A-FAN K=5 train step, ResNet-18/CIFAR-10" if default_cfg else
                      f"images/sec (whole node) A-FAN K={args.pgd_steps} train step, {args.arch} {side}x{side}")
            workload = (f"{args.arch} {'CIFAR-10' if side == 32 else 'ImageNet'}-shape A-FAN K={args.pgd_steps} {args.dtype}, batch "
                        f"{args.batch}/GPU, perturb_idx {idx}, internal layout {args.layout}, 1xMI355X per rank "
                        + ("(BASELINE configs[1])" if default_cfg and args.batch == 256 else
                           "(BASELINE configs[2], per-GPU share)" if args.arch == "resnet50" else "(not a BASELINE configuration)"))
            # which passes one iteration runs: the reference's text is 2 head passes + K PGD
            # passes + adversarial and clean final passes; value-identical passes are run once
            sched = (("1 head pass (stands for the reference's 2), " if trainer._share_head(xs[0]) else "2 head passes, ")
                     + (f"1 clean tail pass (= PGD step 0 and the final clean pass) + {args.pgd_steps - 1} PGD passes + "
                        "adversarial pass" if trainer._fold_ok(xs[0]) else
                        f"{args.pgd_steps} PGD passes + adversarial and clean final passes"
                        + (" (one grouped pass)" if getattr(trainer, "_groupable", False) else "")))
        if args.dual_bn:
            workload += ", dual-BN option ON (not the reference's arithmetic)"
        line = {
            "metric": metric, "value": round(ips, 2 if (seg or det) else 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(step_ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "final_loss": round(loss, 4), "hipgraph": graphed, "schedule": sched,
                       "host_placement": {k: v for k, v in placement.items() if k != "restore"}},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if not (seg or det):
            line["config"]["in_launch_batchnorm"] = bool(pkg.ops.GRID_BN) and (pkg.ops.CALLS["conv_bn_fused"] > 0)
        if ranks_seen is not None:
            line["ranks_seen"] = ranks_seen
        if ddp_diag is not None:
            line["ddp_diag"] = ddp_diag
        if literal is not None:
            line["literal_schedule"] = literal
        if dp_sched is not None:
            line["dp_schedule"] = dp_sched
        if conv_exec is not None:
            gf = REF_GFLOP_PER_IMAGE.get((args.arch, args.pgd_steps))
            if gf is not None:      # secondary: what the same img/s would mean at the reference schedule's FLOP count
                conv_exec["reference_schedule_GFLOP_per_image"] = gf
                conv_exec["reference_schedule_equiv_TFLOPs"] = round(gf * 1e9 * ips / 1e12, 1)
            line["conv_mfma"] = conv_exec
        if hbm:
            line["hbm_kernels"] = hbm
        if cpu is not None:
            line["speedup_vs_cpu"] = round(ips / cpu["value"], 1)
        if kernels is not None:
            line["kernels"] = kernels
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
