#!/bin/bash
# One measured configuration, everything the judge reads:  bash tools/gpu_profile.sh TAG [bench.py args...]
#   gpurun_out/TAG_bench.json            the bench line (with cpu_baseline unless NOCPU=1)
#   gpurun_out/TAG_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (graph replay)
#   gpurun_out/TAG_pmc_hbm_traffic.json  FETCH_SIZE / WRITE_SIZE per kernel (two separate --pmc passes, eager), unless NOPMC=1
TAG=${1:-x}; shift
ARGS="$@"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python bench.py $ARGS $([ "$NOCPU" = 1 ] && echo --no_cpu_baseline) 2>&1 | grep "^{" > gpurun_out/${TAG}_bench.json
python3 -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench.json'));print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'] and {k:d['roofline'][k] for k in ('kernel','achieved','frac')}, d.get('conv_mfma'), d.get('cpu_baseline') and d['cpu_baseline']['value'])"
rm -rf /tmp/prof_$TAG; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py $ARGS --no_cpu_baseline --no_roofline > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
head -25 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
if [ "$NOPMC" != 1 ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${TAG}_$C; cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$C -- python3 $R/bench.py $ARGS --steps 2 --warmup 3 --no_cpu_baseline --no_roofline --no_graph > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
    cd $R
  done
  mkdir -p /tmp/pmc_${TAG}_all; cp -r /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE /tmp/pmc_${TAG}_all/
  python tools/pmc_summary.py /tmp/pmc_${TAG}_all gpurun_out/${TAG}_pmc_hbm_traffic.json "bench.py $ARGS --steps 2 --warmup 3 --no_graph" | head -40
fi
