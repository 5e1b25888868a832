import csv, glob, collections, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "conv_igemm" not in n and "c64" not in n: continue
    m = re.search(r"conv_igemm_(\w+)_kernel<([^>]*)>", n)
    key = ((m.group(1) + " <" + m.group(2).replace(" ", "") + ">") if m else n[:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v = sorted(v)
    print(f"{k[0]:62s} grid {k[1]:>6s}x{k[2]:>4s}x{k[3]:>2s} wg {k[4]:>4s}  n={len(v):3d}  med {v[len(v)//2]/1e3:6.2f} us  min {v[0]/1e3:6.2f}")
