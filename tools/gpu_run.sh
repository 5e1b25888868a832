#!/bin/bash
# generic GPU pass: gpu tests, then bench in both layouts
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log; grep -n "^E  " gpurun_out/pytest_gpu.log | head -20
for L in nhwc nchw; do
  timeout 600 python bench.py --steps 20 --warmup 6 --no_cpu_baseline --layout $L $EXTRA 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_$L.log
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$L.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$L", d["value"], "img/s", d["ms_per_step"], "ms/step", "handwritten ms", d["roofline"]["handwritten_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"])
    for k,v in d["kernels"].items(): print("   ", k, v)
else:
    print(open("gpurun_out/bench_$L.log").read()[-2000:])
PY
done
