#!/bin/bash
# round 5, first GPU call: barrier probe, baseline bench line (with the literal schedule), MFMA counters, the new tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05a; mkdir -p $OUT; cd $R
timeout 120 tools/probe/_bin/gb2 > $OUT/gb2.txt 2>&1; cat $OUT/gb2.txt
timeout 600 python3 bench.py > $OUT/r18_bench.json 2> $OUT/r18_bench.err; tail -c 600 $OUT/r18_bench.err; python3 -c "
import json;d=json.loads(open('$OUT/r18_bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d.get('literal_schedule'))"
timeout 1500 bash tools/gpu_pmc_mfma.sh r18
timeout 900 python3 -m pytest tests/test_det_model_gpu.py tests/test_det_gpu.py -x -q -k "freed or full_size or padded" 2>&1 | tail -5
