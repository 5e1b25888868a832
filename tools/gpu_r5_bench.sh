#!/bin/bash
# Round-5 measurement pass: bench lines + rocprofv3 kernel stats for the workloads (run on the GPU box through gpurun).
#   bash tools/gpu_r5_bench.sh TAG [which...]     which in: r18 fp32 dl101 dl101b8 r50 frcnn pmc (default: all but pmc)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r05a}; shift
WHICH=${@:-r18 fp32 dl101 dl101b8 r50 frcnn}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
declare -A ARGS=( [r18]="" [fp32]="--dtype fp32 --steps 10 --warmup 4 --no_cpu_baseline --no_literal"
                  [dl101]="--arch deeplabv3plus_resnet101 --steps 10 --warmup 4"
                  [dl101b8]="--arch deeplabv3plus_resnet101 --batch 8 --steps 6 --warmup 4 --no_cpu_baseline"
                  [r50]="--arch resnet50 --batch 64 --pgd_steps 3 --steps 10 --warmup 4 --cpu_steps 1 --no_literal"
                  [frcnn]="--arch fasterrcnn_resnet101 --steps 5 --warmup 3" )
for N in $WHICH; do
  if [ "$N" = "pmc" ]; then
    bash tools/gpu_pmc_mfma.sh r18 > $OUT/pmc_mfma.log 2>&1; cp gpurun_out/pmcm_r18/summary.json $OUT/r18_pmc_mfma.json
    bash tools/gpu_pmc_bench.sh r18 --no_literal > $OUT/pmc_hbm.log 2>&1; cp gpurun_out/pmcb_r18/summary.json $OUT/r18_pmc_hbm_traffic.json
    continue
  fi
  python3 bench.py ${ARGS[$N]} > $OUT/${N}_bench.json 2> $OUT/${N}_bench.err
  A=$(echo "${ARGS[$N]}" | sed 's/--steps [0-9]*//; s/--warmup [0-9]*//; s/--cpu_steps [0-9]*//; s/--no_cpu_baseline//; s/--no_literal//')
  rm -rf /tmp/prof_$N
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -- python3 $R/bench.py --steps 10 --warmup 4 --no_cpu_baseline --no_roofline --no_literal $A > $OUT/${N}_prof.log 2>&1)
  F=$(find /tmp/prof_$N -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/${N}_kernel_stats.csv
done
for f in $OUT/*_bench.json; do echo "== $f"; python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["metric"], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms | roofline", (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"),
          "| conv", (d.get("conv_mfma") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"),
          "| literal", (d.get("literal_schedule") or {}).get("images_per_s"))
except Exception as e:
    print("unreadable:", e)
PY
done
