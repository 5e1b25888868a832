#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05c; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -5
AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_stamp.so timeout 300 python3 tools/probe/conv_stamps.py > $OUT/conv_stamps.txt 2>&1; grep -v "^   *[0-9]*:" $OUT/conv_stamps.txt
timeout 600 python3 bench.py --no_cpu_baseline --no_literal > $OUT/r18_bench.json 2> $OUT/r18_bench.err; tail -c 300 $OUT/r18_bench.err; python3 -c "
import json;d=json.loads(open('$OUT/r18_bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'], {k:(v['avg_us'],v['ms_per_step']) for k,v in list(d['kernels'].items())[:8]})"
