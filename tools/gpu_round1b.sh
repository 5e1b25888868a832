#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_roofline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r1a -name "*stats*" | head; du -sh gpurun_out/prof_r1a
# keep only the stats csv (kernel trace can be large)
find gpurun_out/prof_r1a -name "*kernel_trace*" -size +20M -delete
tail -30 gpurun_out/pytest_gpu.log
