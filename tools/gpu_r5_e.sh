#!/bin/bash
# kernel stats of the step with / without the in-launch BatchNorm
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05e; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for GB in 1 0; do
  rm -rf /tmp/prof_gb$GB
  export AFAN_GRID_BN=$GB
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gb$GB -- python3 $R/bench.py --steps 10 --warmup 4 --no_cpu_baseline --no_roofline --no_literal > $OUT/prof_gb$GB.log 2>&1)
  F=$(find /tmp/prof_gb$GB -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/r18_gb${GB}_kernel_stats.csv
  tail -1 $OUT/prof_gb$GB.log | cut -c1-200
done
python3 - <<'PY'
import csv,re
for gb in (1,0):
    rows=list(csv.DictReader(open(f'gpurun_out/r05e/r18_gb{gb}_kernel_stats.csv')))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f'== GRID_BN={gb}: total kernel time {tot/14/1e6:.3f} ms per step')
    bn=0
    for r in rows[:26]:
        n=re.sub(r'\(anonymous namespace\)::|void |afan_nhwc::|unsigned short, 8, ','',r['Name']).split('(')[0][:70]
        print(f"  {int(r['Calls'])/14:6.1f}/step {float(r['AverageNs'])/1e3:7.1f} us  {float(r['TotalDurationNs'])/14/1e6:6.3f} ms  {n}")
    for r in rows:
        if any(k in r['Name'] for k in ('apply','bwd_reduce','finalize_kernel','stats_kernel')) and 'wgrad' not in r['Name']:
            bn+=float(r['TotalDurationNs'])
    print(f'  BN-family: {bn/14/1e6:.3f} ms per step')
PY
