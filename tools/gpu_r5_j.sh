#!/bin/bash
# kernel trace of DeepLab at batch 8 (weight gradients on the side stream): which launches run long, and what ran beside them
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05j; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_j
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_j -- python3 $R/bench.py --arch deeplabv3plus_resnet101 --batch 8 --steps 2 --warmup 2 --no_cpu_baseline --no_roofline --no_literal --no_graph > $OUT/run.log 2>&1
F=$(find /tmp/prof_j -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY' > $OUT/long_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
long_ = [r for r in rows if r["e"] - r["s"] > 5_000_000]
print(len(rows), "kernels;", len(long_), "longer than 5 ms")
for r in long_[:6]:
    print("LONG", (r["e"] - r["s"]) / 1e6, "ms", r["Kernel_Name"][:110], "grid", r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], "wg", r["Workgroup_Size_X"], "lds", r["LDS_Block_Size"], "queue", r["Queue_Id"])
    for o in rows:
        if o is not r and o["s"] < r["e"] and o["e"] > r["s"]:
            print("     beside:", (o["e"] - o["s"]) / 1e3, "us", o["Kernel_Name"][:90], "grid", o["Grid_Size_X"], o["Grid_Size_Y"], o["Grid_Size_Z"], "wg", o["Workgroup_Size_X"], "lds", o["LDS_Block_Size"], "queue", o["Queue_Id"], "starts +", (o["s"] - r["s"]) / 1e3, "us")
PY
head -40 $OUT/long_kernels.txt | cut -c1-330
