#!/bin/bash
# per-dispatch durations of a bench run, aggregated by (kernel, grid): gpurun_out/${TAG}_trace_by_grid.txt
#   [ALL=1 TOP=150] bash tools/gpu_trace.sh TAG [bench.py args...]      (ALL: every kernel, not only the convolutions)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
TAG=${1:-x}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/trace_out; cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 6 --warmup 3 --no_cpu_baseline --no_roofline > /tmp/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob('/tmp/trace_out/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
for r in rows:
    name = r['Kernel_Name']
    if not os.environ.get('ALL') and 'conv' not in name and 'wgrad' not in name and 'stem' not in name: continue
    short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-70:]
    key = (short, r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'), r.get('Workgroup_Size_X', r.get('Workgroup_Size')))
    agg[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = []
for k, v in agg.items():
    v = sorted(v)
    out.append((sum(v), k, len(v), v[len(v)//2], v[0], v[-1]))
out.sort(reverse=True)
with open('gpurun_out/%s_trace_by_grid.txt' % os.environ.get('TAG', 'x'), 'w') as fo:
    for tot, k, n, med, lo, hi in out[:int(os.environ.get('TOP', 60))]:
        line = f"{tot/1e6:9.3f} ms  n={n:5d} med={med/1e3:7.1f}us min={lo/1e3:7.1f} max={hi/1e3:7.1f}  {k}"
        print(line); fo.write(line + "\n")
PY
