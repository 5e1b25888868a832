#!/bin/bash
# Which kernels run right before / after the small device-to-device copies and ATen copy kernels of a bench step?
#   bash tools/probe/copy_neighbours.sh [bench.py args...]  -> gpurun_out/copy_neighbours.txt   (rocprofv3 --kernel-trace)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
rm -rf /tmp/cn_out; cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/cn_out -- python3 $R/bench.py "$@" --steps 4 --warmup 3 --no_cpu_baseline --no_roofline > /tmp/cn.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/copy_neighbours.txt
import csv, glob, collections
f = glob.glob('/tmp/cn_out/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-60:]
pairs = collections.Counter()
for i, r in enumerate(rows):
    n = r['Kernel_Name']
    if 'copyBuffer' in n or 'direct_copy' in n or 'CatArray' in n:
        prev = short(rows[i - 1]['Kernel_Name']) if i else ''
        nxt = short(rows[i + 1]['Kernel_Name']) if i + 1 < len(rows) else ''
        pairs[(short(n)[-28:], r.get('Grid_Size_X', r.get('Grid_Size')), prev, nxt)] += 1
for (k, g, p, n), c in pairs.most_common(60):
    print(f"{c:5d}  {k:28s} grid {g:>9s}  after [{p}]  before [{n}]")
PY
