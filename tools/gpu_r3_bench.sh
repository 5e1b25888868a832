#!/bin/bash
# Round-3 measurement pass: bench lines + rocprofv3 kernel stats for the three workloads (run on the GPU box through gpurun).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/r18_bench.json 2> $OUT/r18_bench.err
python3 bench.py --arch deeplabv3plus_resnet101 --steps 10 --warmup 4 > $OUT/dl101_bench.json 2> $OUT/dl101_bench.err
python3 bench.py --arch deeplabv3plus_resnet101 --batch 8 --steps 6 --warmup 4 --no_cpu_baseline > $OUT/dl101_b8_bench.json 2> $OUT/dl101_b8_bench.err
python3 bench.py --arch resnet50 --batch 64 --pgd_steps 3 --steps 10 --warmup 4 --cpu_steps 1 > $OUT/r50_bench.json 2> $OUT/r50_bench.err
python3 bench.py --arch fasterrcnn_resnet101 --steps 5 --warmup 2 > $OUT/frcnn_bench.json 2> $OUT/frcnn_bench.err
export TMPDIR=/tmp
for W in "r18:" "dl101:--arch deeplabv3plus_resnet101" "frcnn:--arch fasterrcnn_resnet101 --steps 4"; do
  N=${W%%:*}; A=${W#*:}
  rm -rf /tmp/prof_$N
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -- python3 $R/bench.py --steps 10 --warmup 4 --no_cpu_baseline --no_roofline $A > $OUT/${N}_prof.log 2>&1)
  F=$(find /tmp/prof_$N -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/${N}_kernel_stats.csv
done
for f in $OUT/*_bench.json; do echo "== $f"; python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["metric"], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms | roofline", (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"),
          "| conv", (d.get("conv_mfma") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), "| sched:", d["config"].get("schedule", "")[:110])
except Exception as e:
    print("unreadable:", e)
PY
done
