#!/bin/bash
# where the GPU idles inside a step: kernel trace of bench.py (any arguments) -> idle time between consecutive kernels,
# grouped by (kernel before -> kernel after), largest groups first.  Written for the Faster-RCNN step (--arch
# fasterrcnn_resnet101), whose ~50 host reads per iteration leave the device waiting for the next launches.
#   bash tools/gpu_gaps.sh TAG --arch fasterrcnn_resnet101 --steps 4 --warmup 3      -> gpurun_out/TAG_gaps.txt
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
rm -rf /tmp/trace_gaps; cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_gaps -- python3 $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_roofline "$@" > /tmp/bench_gaps.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 /tmp/bench_gaps.log | cut -c1-300
python3 - > gpurun_out/${TAG}_gaps.txt <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/trace_gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '').split('(')[0][:64])
              for r in csv.DictReader(open(f)))
rows = rows[len(rows) * 2 // 3:]            # the last third: steady state steps
span = rows[-1][1] - rows[0][0]
busy, end = 0, rows[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
for s, e, n in rows:
    if s > end:
        g = s - end
        key = (prev, n)
        gaps[key][0] += g
        gaps[key][1] += 1
        hist[min(int(g / 1e3).bit_length(), 12)] += g
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end:
        end, prev = e, n
print(f"{len(rows)} launches over {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms ({100 * (span - busy) / span:.0f} %)")
print("idle time by gap length:")
for k in sorted(hist):
    print(f"   < {2 ** k:5d} us: {hist[k] / 1e6:7.2f} ms")
print("idle time by (kernel before -> kernel after), top 40:")
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"  {t / 1e6:7.2f} ms  x{c:5d}  avg {t / c / 1e3:7.1f} us   {a}  ->  {b}")
PY
head -60 gpurun_out/${TAG}_gaps.txt
