#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05h; mkdir -p $OUT; cd $R
timeout 1200 python3 -m pytest tests/test_det_model_gpu.py tests/test_det_gpu.py tests/test_det_targets_gpu.py -x -q 2>&1 | tail -6
for rep in 1 2; do
for V in "0" "1"; do
AFAN_DET_MERGE_READS=$V timeout 600 python3 bench.py --arch fasterrcnn_resnet101 --steps 12 --warmup 3 --no_roofline > $OUT/frcnn_m$V.json 2> $OUT/frcnn.err; python3 -c "
import json;d=json.loads(open('$OUT/frcnn_m$V.json').read().strip().splitlines()[-1]);print('merge_reads=$V (noise ahead on)', d['value'],d['ms_per_step'])"
done; done
