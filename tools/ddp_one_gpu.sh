#!/bin/bash
# Functional run of the N>1 code path of bench.py on a 1-GPU box: two ranks (gloo, both on cuda:0).  `python bench.py --gpus 2`
# with no launcher around it starts its own ranks (bench.launch_ranks: torch.distributed.run as a child process, before any GPU
# call).  Proves that the piece-wise captured step, the tail-first all-reduce on the side stream, the replica broadcast and the
# ranks_seen gather execute on hardware; RCCL itself needs 2 GPUs.
#   [LOG=name] [WORLD=2] bash tools/ddp_one_gpu.sh [bench.py args]   -> gpurun_out/${LOG:-ddp2_one_gpu}.log   (later args win: --batch 2 ...)
# Round 5: the N > 1 line carries `ddp_diag` (per announced range: bytes, queueing behind earlier ranges, all-reduce duration; the compute
# stream's wait for the exchange before SGD; host time between graph replays) — WORLD=4 / 8 exercise the same path with more ranks.
# (AFAN_BENCH_ONE_DEVICE also switches the in-launch BatchNorm off: several processes' grid barriers on one GPU could starve each other.)
mkdir -p gpurun_out
WORLD=${WORLD:-2}
LOG=gpurun_out/${LOG:-ddp${WORLD}_one_gpu}.log
export AFAN_BENCH_ONE_DEVICE=1 AFAN_DIST_BACKEND=gloo
python bench.py --gpus $WORLD --steps 6 --warmup 5 --batch 128 --no_cpu_baseline "$@" > $LOG 2>&1
echo "rc=$?"; grep "^{" $LOG | cut -c1-900; grep -i "error\|Traceback" -A8 $LOG | head -30
