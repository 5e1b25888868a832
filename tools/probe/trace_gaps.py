"""Idle time between consecutive kernels of a rocprofv3 kernel trace (all queues merged): total, and by the kernel that ENDS each gap — which
launch the GPU was waiting for.   python tools/probe/trace_gaps.py DIR [min_gap_us] [top]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.search(r"^([\w:]+)", n)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)[-48:] if m else n[:48]))
rows.sort()
busy_end, gaps, total_gap = rows[0][1], collections.defaultdict(list), 0
prev_name = rows[0][2]
for s, e, name in rows[1:]:
    if s > busy_end:
        g = (s - busy_end) / 1e3
        if g < 5e4:                                   # (the pause between warm-up and the timed region and the like are not gaps)
            total_gap += g
            if g >= min_gap:
                gaps[(prev_name, name)].append(g)
    if e > busy_end:
        busy_end, prev_name = e, name
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"span {span / 1e3:.1f} ms, idle between kernels {total_gap / 1e3:.1f} ms ({100 * total_gap / span:.1f} %), gaps >= {min_gap} us listed by (kernel before -> kernel after)")
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{a:48s} -> {b:48s} n={len(v):4d}  total {sum(v) / 1e3:7.2f} ms  median {sorted(v)[len(v) // 2]:7.1f} us")
