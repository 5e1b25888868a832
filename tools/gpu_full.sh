#!/bin/bash
# round-end style pass: all gpu tests, smoke, default bench (with cpu baseline), rocprof stats
TAG=${1:-r1x}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log; grep -n "^E  " gpurun_out/pytest_gpu.log | head -20
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py 2>&1 | grep "^{" > gpurun_out/bench_$TAG.json; cut -c1-600 gpurun_out/bench_$TAG.json
bash tools/gpu_prof.sh $TAG
