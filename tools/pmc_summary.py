"""Aggregate rocprofv3 --pmc counter_collection.csv files into a per-kernel summary (avg per launch)."""
import collections, csv, glob, json, sys
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for key in ("conv_igemm_kernel", "wgrad_kernel", "wgrad_reduce", "stats_kernel", "finalize_kernel", "bwd_reduce_kernel", "bwd_apply_kernel",
                    "apply_kernel", "pgd_step_norms_kernel", "pgd_step_kernel", "sgd_kernel", "cast_bf16", "transpose_weights"):
            if key in name:
                out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
summary = {k: {c: {"launches": len(v), "avg": sum(v) / len(v)} for c, v in cs.items()} for k, cs in out.items()}
json.dump(summary, open(sys.argv[2], "w"), indent=1)
for k, cs in summary.items():
    print(k, {c: round(v["avg"], 1) for c, v in cs.items()})
