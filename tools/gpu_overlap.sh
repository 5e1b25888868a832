#!/bin/bash
# do kernels of the replayed step overlap in time (parallel graph branches)?  kernel trace -> overlapped time per step
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/trace_ov; cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_ov -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no_cpu_baseline --no_roofline "$@" > /tmp/bench_ov.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/trace_ov/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-50:]) for r in csv.DictReader(open(f))))
rows = rows[len(rows) // 2:]            # steady state
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
ov, pairs, prev_end, prev = 0, 0, 0, None
names = {}
for s, e, n in rows:
    if s < prev_end:
        ov += min(e, prev_end) - s
        pairs += 1
        names[(prev, n)] = names.get((prev, n), 0) + 1
    if e > prev_end:
        prev_end, prev = e, n
print("kernels %d  sum of durations %.3f ms  wall span %.3f ms  overlapped %.3f ms in %d pairs" % (len(rows), busy / 1e6, span / 1e6, ov / 1e6, pairs))
for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:8]:
    print("   ", v, k)
PY
