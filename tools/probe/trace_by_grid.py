"""Kernel trace (rocprofv3 --kernel-trace --output-format csv -d DIR) grouped by kernel and grid: calls, total, median — the launches of
one shape of a templated kernel separated.   python tools/probe/trace_by_grid.py DIR [top]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(\w+)<([^>]*)>", n)
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.search(r"^(\w+)<([^>]*)>", n)
    name = (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else re.sub(r"\(.*", "", n)
    wg = int(r["Workgroup_Size_X"]) or 1
    key = (name[-70:], int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(len(v) for v in d.values())} launches")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    v = sorted(v)
    print(f"{k[0]:72s} wgs {k[1]:5d}x{k[2]:4d}x{k[3]:2d}  n={len(v):5d}  total {sum(v) / 1e6:8.3f} ms ({100 * sum(v) / tot:4.1f} %)  med {v[len(v) // 2] / 1e3:7.2f} us")
