#!/bin/bash
# Round 6's GPU calls, one parameterised script (replaces the dozen gpu_r5_*.sh one-offs):  bash tools/gpu_r6.sh <what> [args...]
#   stamps [section]        launch-phase / per-tap stamps of the stamped build (tools/build_stamp.sh first)     -> gpurun_out/r06/conv_stamps_<section>.txt
#   libs W "a b tree"       several BUILDS of the library on one box, interleaved twice (tools/build_variant.sh / build_prev.sh; tree = working tree)
#   env W VAR "0 1"         one environment variable's settings, interleaved twice
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
case "$1" in
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
