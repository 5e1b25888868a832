#!/bin/bash
# Round 6's GPU calls, one parameterised script (replaces the dozen gpu_r5_*.sh one-offs):  bash tools/gpu_r6.sh <what> [args...]
#   stamps [section]        launch-phase / per-tap stamps of the stamped build (tools/build_stamp.sh first)     -> gpurun_out/r06/conv_stamps_<section>.txt
#   libs W "a b tree"       several BUILDS of the library on one box, interleaved twice (tools/build_variant.sh / build_prev.sh; tree = working tree)
#   env W VAR "0 1"         one environment variable's settings, interleaved twice
#   bench TAG [which...]    the measurement pass: bench line + rocprofv3 --kernel-trace --stats per workload (r18 fp32 dl101 dl101b8 r50 frcnn;
#                           "pmc" adds the MFMA-utilisation and HBM-traffic counter passes of the headline step) -> gpurun_out/TAG/ -> profiles/TAG_*
#   ddp                     python bench.py --gpus 2 / 4 / 8 with all ranks on this ONE GPU over gloo (functional: the N > 1 line, ddp_diag)
#   suite [pytest args]     the -m gpu suite + smoke()
#   parity                  every PARITY line of the golden tests (measured delta, reference-vs-reference floor, bound) -> gpurun_out/r06/parity_measurements.txt
#   py script.py [args]     any probe script
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
declare -A ARGS=( [r18]="--steps 30" [r18lit]="--steps 20 --no_fold_clean --no_share_head" [dl101]="--arch deeplabv3plus_resnet101 --steps 12 --warmup 4"
                  [dl101b8]="--arch deeplabv3plus_resnet101 --batch 8 --steps 6 --warmup 4"
                  [frcnn]="--arch fasterrcnn_resnet101 --steps 10 --warmup 3" [r50]="--arch resnet50 --batch 64 --pgd_steps 3 --steps 10 --warmup 4" )
bench1() { env $3 timeout 900 python3 bench.py --no_cpu_baseline --no_literal --no_roofline ${ARGS[$1]} > $OUT/b.json 2> $OUT/bench.err; python3 -c "
import json;d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]);print('$1 $2', d['value'],d['ms_per_step'])" 2>/dev/null || { echo "$1 $2 FAILED"; tail -3 $OUT/bench.err; }; }
declare -A BARGS=( [r18]="" [fp32]="--dtype fp32 --steps 10 --warmup 4 --no_cpu_baseline --no_literal --no_dp_schedule"
                   [dl101]="--arch deeplabv3plus_resnet101 --steps 10 --warmup 4"
                   [dl101b8]="--arch deeplabv3plus_resnet101 --batch 8 --steps 6 --warmup 4 --no_cpu_baseline"
                   [r50]="--arch resnet50 --batch 64 --pgd_steps 3 --steps 10 --warmup 4 --cpu_steps 1 --no_literal --no_dp_schedule"
                   [frcnn]="--arch fasterrcnn_resnet101 --steps 30 --warmup 5" )
case "$1" in
  bench) TAG=${2:-r06a}; shift; shift; WHICH=${@:-r18 fp32 dl101 dl101b8 r50 frcnn}; O=$R/gpurun_out/$TAG; mkdir -p $O
         for N in $WHICH; do
           if [ "$N" = "pmc" ]; then
             bash tools/gpu_pmc_mfma.sh r18 --no_dp_schedule > $O/pmc_mfma.log 2>&1; cp gpurun_out/pmcm_r18/summary.json $O/r18_pmc_mfma.json
             bash tools/gpu_pmc_bench.sh r18 --no_literal --no_dp_schedule > $O/pmc_hbm.log 2>&1; cp gpurun_out/pmcb_r18/summary.json $O/r18_pmc_hbm_traffic.json
             continue
           fi
           python3 bench.py ${BARGS[$N]} > $O/${N}_bench.json 2> $O/${N}_bench.err
           A=$(echo "${BARGS[$N]}" | sed 's/--steps [0-9]*//; s/--warmup [0-9]*//; s/--cpu_steps [0-9]*//; s/--no_cpu_baseline//; s/--no_literal//; s/--no_dp_schedule//')
           rm -rf /tmp/prof_$N
           (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$N -- python3 $R/bench.py --steps 10 --warmup 4 --no_cpu_baseline --no_roofline --no_literal --no_dp_schedule $A > $O/${N}_prof.log 2>&1)
           F=$(find /tmp/prof_$N -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $O/${N}_kernel_stats.csv
         done
         for f in $O/*_bench.json; do echo "== $f"; python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(d["metric"], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms | roofline", r.get("kernel"), r.get("frac"), "stale", r.get("traffic_stale"), (r.get("mfma_counters") or {}).get("stale"),
          "| conv", (d.get("conv_mfma") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"),
          "| literal", (d.get("literal_schedule") or {}).get("images_per_s"), "| dp", (d.get("dp_schedule") or {}).get("ms_per_step"))
except Exception as e:
    print("unreadable:", e)
PY
         done ;;
  ddp) for W in 2 4 8; do B=$((256 / W)); LOG=r06_ddp${W}_one_gpu_r18 WORLD=$W timeout 1500 bash tools/ddp_one_gpu.sh --batch $B --steps 2 --warmup 4 --no_roofline 2>&1 | cut -c1-1500; done ;;
  stamps) S=${2:-all}; CONV_STAMPS=$S AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_stamp.so timeout 900 python3 tools/probe/conv_stamps.py > $OUT/conv_stamps_$S.txt 2>&1
          grep -v "^   *[0-9]*:" $OUT/conv_stamps_$S.txt | tail -60 ;;
  libs) W=$2; for rep in 1 2; do for L in $3; do
          if [ $L = tree ]; then unset AFAN_HIP_LIB; else export AFAN_HIP_LIB=$R/tools/probe/_bin/libafan_hip_$L.so; fi
          bench1 $W $L "${ENVS:-_X=1}"; done; done; unset AFAN_HIP_LIB ;;
  env) W=$2; for rep in 1 2; do for V in $4; do bench1 $W "$3=$V" "$3=$V"; done; done ;;
  suite) shift; timeout 2700 python3 -m pytest tests/ -x -q -m gpu "$@" > $OUT/gpu_suite.txt 2>&1; tail -15 $OUT/gpu_suite.txt
         timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt ;;
  parity) timeout 2400 python3 -m pytest tests/test_train_step_gpu.py tests/test_deeplab_gpu.py tests/test_det_model_gpu.py -q -s -m gpu -k "matches_reference or trajectory or contractive or perturbation_given or golden" > $OUT/parity_raw.txt 2>&1
          grep -o "PARITY.*\|[0-9]* passed.*\|[0-9]* failed.*" $OUT/parity_raw.txt | sort -u > $OUT/parity_measurements.txt; tail -3 $OUT/parity_raw.txt; grep -c PARITY $OUT/parity_measurements.txt; grep "det_frcnn" $OUT/parity_measurements.txt ;;
  py) shift; timeout 1500 python3 "$@" 2>&1 | tee $OUT/py_$(basename $1 .py).txt | tail -80 ;;
  *) echo "unknown: $1"; exit 2 ;;
esac
