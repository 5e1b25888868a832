#!/bin/bash
# quick GPU pass: the BN / conv / train-step tests, then the default bench (nhwc) once per BN reduction scheme
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log; grep -n "^E  " gpurun_out/pytest_gpu.log | head -20
for ACC in 1 0; do
  AFAN_BN_ACC=$ACC timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline $EXTRA 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_acc$ACC.log
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_acc$ACC.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("acc=$ACC", d["value"], "img/s", d["ms_per_step"], "ms/step", "handwritten ms", d["roofline"]["handwritten_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"])
    for k,v in d["kernels"].items(): print("   ", k, v)
else:
    print(open("gpurun_out/bench_acc$ACC.log").read()[-2000:])
PY
done
