#!/bin/bash
# the PARITY lines of the golden tests (measured |delta| beside the reference-vs-reference floor and the bound) + the full suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05_parity; mkdir -p $OUT; cd $R
timeout 1800 python3 -m pytest tests/test_train_step_gpu.py tests/test_deeplab_gpu.py -q -s -m gpu -k "matches_reference or trajectory or contractive or perturbation_given or golden" > $OUT/raw.txt 2>&1
grep -o "PARITY.*\|[0-9]* passed.*\|[0-9]* failed.*" $OUT/raw.txt | sort -u > $OUT/parity_measurements.txt; tail -3 $OUT/raw.txt; grep -c PARITY $OUT/parity_measurements.txt
grep "traj_\|loss_adv" $OUT/parity_measurements.txt | head -40
timeout 2400 python3 -m pytest tests/ -x -q -m gpu > $OUT/gpu_suite.txt 2>&1; tail -6 $OUT/gpu_suite.txt
