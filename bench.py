#!/usr/bin/env python3
"""bench.py — images/sec of the A-FAN train step (BASELINE.json metric) on N MI355X of one node.

Workload (configs[1]): ResNet-18 (CIFAR stem), A-FAN K=5 at the end of stage 1 (perturb_idx 6: 64x32x32
feature map), bf16 backbone, batch 256 PER GPU (weak scaling), synthetic 3x32x32 inputs in [0,1),
gamma 0.5/255, eps 2/255, no clip / no randinit (= cmd/run_perturb.sh flags), SGD(0.1, 0.9, 5e-4).
A "step" is one full iteration of main_perturb.py:165-201: head fwd -> 5 x (tail fwd, CE, dgrad, sign
step) -> norms -> adv tail fwd + clean full fwd -> joint loss -> backward -> [all-reduce] -> SGD.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — the dominant hand-written kernel, achieved algorithmic GB/s vs the 8 TB/s HBM peak, timed
                 per launch with HIP events on the launch stream in a separate instrumented pass (not in `value`);
  cpu_baseline — the CPU oracle (oracle/afan_oracle.py, bit-identical to the reference's Python on CPU) timed on
                 this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
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
GFLOP_PER_IMAGE = {"resnet18": 14.1, "resnet20s": None, "resnet56s": 2.99}   # 4H + (2K+6)T at K=5
GFLOP_PER_IMAGE_K3 = {"resnet50": 85.5}                                       # K=3, 224^2, perturb after layer1
ARCH_INPUT = {"resnet50": (224, 1000)}                                        # (image side, classes); default (32, 10)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="resnet18")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--pgd_steps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--layout", default="nhwc", choices=["nhwc", "nchw"], help="internal activation layout")
    ap.add_argument("--no_graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--async_wgrad", action="store_true", help="weight gradients on a side stream (measured: no gain)")
    ap.add_argument("--no_batch_final", action="store_true", help="run the adv and clean final passes separately (A/B)")
    ap.add_argument("--no_fold_clean", action="store_true", help="separate first PGD pass and final clean pass (A/B)")
    ap.add_argument("--force_fold_clean", action="store_true", help="one clean tail pass regardless of the size heuristic (A/B)")
    ap.add_argument("--no_share_head", action="store_true", help="run the head twice per step like the reference's text (A/B)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--cpu_steps", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(arch, batch, pgd_steps, idx, timed_steps):
    """The oracle's full step on the host cores (fp32, all default torch threads), bounded sample."""
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
    orc.afan_train_step(model, opt, crit, x[:16], y[:16], **kw)      # thread-pool / allocator warm-up (small)
    t0 = time.perf_counter()
    for _ in range(timed_steps):
        orc.afan_train_step(model, opt, crit, x, y, **kw)
    dt = time.perf_counter() - t0
    return {"value": round(batch * timed_steps / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{timed_steps} full steps of the same workload (batch {batch}, K={pgd_steps}, fp32) after one "
                      f"batch-16 warm-up step; {dt / timed_steps * 1e3:.0f} ms/step; host {os.cpu_count()} logical CPUs"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import torch.nn as nn
    pkg = importlib.import_module("cv_a-fan_amd")

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

    ctor, idx = pkg.resnet_s.ARCHS[args.arch]
    torch.manual_seed(3)                      # same initial weights on every rank (reference --seed 3)
    model = ctor()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.set_compute_dtype(dtype).set_channels_last(args.layout == "nhwc").to(dev).train()
    trainer = pkg.train_step.AfanTrainer(model, nn.CrossEntropyLoss(), steps=args.pgd_steps, gamma=0.5, eps=2.0,
                                         perturb_idx=idx, lr=0.1, use_graph=not args.no_graph, batch_final=not args.no_batch_final, share_head=not args.no_share_head, fold_clean=(False if args.no_fold_clean else (True if args.force_fold_clean else None)),
                                         async_wgrad=args.async_wgrad)
    g = torch.Generator().manual_seed(3 + rank)          # each rank its own shard of the synthetic stream
    nbuf = 4
    side, ncls = ARCH_INPUT.get(args.arch, (32, 10))
    xs = [torch.rand(args.batch, 3, side, side, generator=g).to(dev) for _ in range(nbuf)]
    ys = [torch.randint(0, ncls, (args.batch,), generator=g).to(dev) for _ in range(nbuf)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        r = trainer.step(xs[i % nbuf], ys[i % nbuf])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        r = trainer.step(xs[i % nbuf], ys[i % nbuf])
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(r["loss"])
    assert loss == loss, "loss is NaN"

    # ---- instrumented pass (not part of `value`): per-launch HIP-event timing of the hand-written kernels ----
    roof, kernels = None, None
    graphed = trainer._graph is not None
    if not args.no_roofline:
        # per-launch event timing needs eager launches: the same step body, un-captured.  EVERY rank runs it (the step
        # contains the gradient all-reduce); only rank 0 records.
        if rank == 0:
            pkg.ops.profile_enable(True)
        for i in range(2):
            trainer._step_eager(xs[i % nbuf], ys[i % nbuf])
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        prof = pkg.ops.profile_collect()
        pkg.ops.profile_enable(False)
        def _k(v):
            d = {"launches_per_step": v["launches"] // 2, "avg_us": round(v["ms"] * 1e3 / v["launches"], 2),
                 "ms_per_step": round(v["ms"] / 2, 3)}
            if v["flops"] > 0:
                d["algo_TFLOPs"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)
            else:
                d["algo_GBps"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None
            return d
        kernels = {k: _k(v) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        name, v = max(prof.items(), key=lambda kv: kv[1]["ms"])
        hand_ms = round(sum(q["ms"] for q in prof.values()) / 2, 3)
        if v["flops"] > 0:      # MFMA-bound kernel: algorithmic FLOPs / measured launch time vs the dense bf16 peak
            ach = v["flops"] / (v["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": BF16_DENSE_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                    "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2),
                    "algo_flops_per_launch": round(v["flops"] / v["launches"]),
                    "handwritten_ms_per_step": hand_ms}
        else:
            ach = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_launch_us": round(v["ms"] * 1e3 / v["launches"], 2),
                    "algo_bytes_per_launch": round(v["bytes"] / v["launches"]), "handwritten_ms_per_step": hand_ms}
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process; the committed summary
        # of the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command supplies it (per launch)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01q_pmc_hbm_traffic.json")))
            key = next((k for k in pmc if k in name or name.replace("_fwd", "").replace("_dgrad", "") .startswith(k.split("_kernel")[0])), None)
            if key:
                roof["traffic"] = round((2 * pmc[key]["FETCH_SIZE"]["avg"] + pmc[key]["WRITE_SIZE"]["avg"]) * 1024)
                roof["traffic_source"] = "profiles/r01q_pmc_hbm_traffic.json (rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE per launch)"
        except (OSError, KeyError, ValueError):
            pass
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.arch, args.batch, args.pgd_steps, idx, args.cpu_steps)

    if rank == 0:
        ips = args.batch * world * args.steps / dt
        gf = GFLOP_PER_IMAGE.get(args.arch) if args.pgd_steps == 5 else None
        if gf is None and args.pgd_steps == 3:
            gf = GFLOP_PER_IMAGE_K3.get(args.arch)
        default_cfg = args.arch == "resnet18" and args.pgd_steps == 5
        line = {
            "metric": "images/sec (whole node) A-FAN K=5 train step, ResNet-18/CIFAR-10" if default_cfg else
                      f"images/sec (whole node) A-FAN K={args.pgd_steps} train step, {args.arch} {side}x{side}",
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.arch} {'CIFAR-10' if side == 32 else 'ImageNet'}-shape A-FAN K={args.pgd_steps} {args.dtype}, batch "
                                   f"{args.batch}/GPU, perturb_idx {idx}, internal layout {args.layout}, 1xMI355X per rank "
                                   f"(BASELINE configs[1])",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "final_loss": round(loss, 4),
                       "hipgraph": graphed,
                       # which passes one iteration runs: the reference's text is 2 head passes + K PGD
                       # passes + adversarial and clean final passes; value-identical passes are run once
                       "schedule": ("1 head pass (stands for the reference's 2), " if trainer._share_head(xs[0]) else "2 head passes, ")
                                   + (f"1 clean tail pass (= PGD step 0 and the final clean pass) + {args.pgd_steps - 1} PGD passes + "
                                      "adversarial pass" if trainer._fold_ok(xs[0]) else
                                      f"{args.pgd_steps} PGD passes + adversarial and clean final passes"
                                      + (" (one grouped pass)" if getattr(trainer, "_groupable", False) else ""))},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if gf is not None:
            tf = gf * 1e9 * ips / 1e12
            line["conv_mfma"] = {"achieved_TFLOPs": round(tf, 1), "peak_TFLOPs": BF16_DENSE_PEAK_TFLOPS,
                                 "frac": round(tf / BF16_DENSE_PEAK_TFLOPS, 4),
                                 "note": "algorithmic conv/linear FLOPs (4H+(2K+6)T, BASELINE.md §4) / step time"}
        if cpu is not None:
            line["speedup_vs_cpu"] = round(ips / cpu["value"], 1)
        if kernels is not None:
            line["kernels"] = kernels
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
