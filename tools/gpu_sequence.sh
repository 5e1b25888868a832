#!/bin/bash
# the launches of ONE replayed step in order (kernel, grid, duration, gap to the previous one): gpurun_out/${TAG}_sequence.txt
#   bash tools/gpu_sequence.sh TAG [bench.py args...]
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
TAG=${1:-x}; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/trace_seq; cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_seq -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 4 --warmup 3 --no_cpu_baseline --no_roofline > /tmp/bench_seq.log 2>&1
cd $GRAFT_REPO_ROOT
TAG=$TAG python3 - <<'PY'
import csv, glob, os
f = glob.glob('/tmp/trace_seq/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')))
               for r in csv.DictReader(open(f))))
# the last step: from the last sgd launch but one to the last
sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r[2]]
hi = sgd[-1] + 1 if sgd else len(rows)
prev_ = [i for i in sgd if hi - i > 50]          # (a step may end in several optimizer launches: one per parameter group)
lo = prev_[-1] + 1 if prev_ else 0
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').replace('afan_nhwc::', '').replace('at::native::', '').split('(')[0][:72]
with open('gpurun_out/%s_sequence.txt' % os.environ.get('TAG', 'x'), 'w') as fo:
    prev = rows[lo][0]
    fo.write(f"{hi - lo} launches, {(rows[hi - 1][1] - rows[lo][0]) / 1e6:.3f} ms\n")
    for s, e, n, g, w in rows[lo:hi]:
        fo.write(f"{(e - s) / 1e3:7.1f} us  gap {max(0, s - prev) / 1e3:5.1f}  grid {g:>8} x {w:<4} {short(n)}\n")
        prev = e
PY
head -3 gpurun_out/${TAG}_sequence.txt
