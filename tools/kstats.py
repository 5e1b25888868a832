"""Print a rocprofv3 kernel_stats.csv as share / calls / average (us) / name:  python tools/kstats.py FILE [ROWS]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    print("%5.1f%% %6d %7.1fus  %s" % (float(r["TotalDurationNs"]) / tot * 100, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:120]))
