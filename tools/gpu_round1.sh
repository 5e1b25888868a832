#!/bin/bash
# first GPU pass: tests, smoke, short bench, conv probe
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
timeout 900 python tools/conv_probe.py > gpurun_out/conv_probe.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log; tail -8 gpurun_out/conv_probe.log
