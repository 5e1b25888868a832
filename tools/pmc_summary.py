"""Aggregate rocprofv3 --pmc counter_collection.csv files into a per-kernel summary (avg per launch).
usage: pmc_summary.py <dir with the counter CSVs> <out.json> ["<command the counters were collected on>"]
The `_meta` entry records the command and a hash of the kernel sources, so that bench.py can tell when a summary is stale."""
import collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("conv_igemm_fwd_kernel", "conv_igemm_dgrad_kernel", "conv_f32_kernel", "wgrad_f32_kernel", "conv3x3_c64_kernel", "conv_small_kernel", "stem7_fwd_kernel", "stem7_wgrad_kernel", "stem_fwd_kernel",
        "wgrad_small_kernel", "wgrad_reduce", "wgrad_kernel", "stats_kernel", "finalize_kernel", "bwd_reduce_kernel", "bwd_apply_acc_kernel",
        "bwd_apply_kernel", "apply_acc_kernel", "apply_kernel", "pgd_step_norms_kernel", "pgd_step_kernel", "sgd_kernel", "cast_bf16",
        "transpose_weights", "mix_feature_nhwc_kernel", "mix_feature_kernel", "lerp_points_kernel", "upsample_fwd_kernel", "upsample_bwd_kernel",
        "ce2d_up_kernel", "ce2d_kernel", "maxpool_fwd_kernel", "maxpool_bwd_kernel", "pointwise_fwd_kernel", "pointwise_dx_kernel", "pointwise_dw_kernel")
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for key in KEYS:
            if key in name:
                out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
summary = {k: {c: {"launches": len(v), "avg": sum(v) / len(v)} for c, v in cs.items()} for k, cs in out.items()}
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(ROOT, "cv_a-fan_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cv_a-fan_amd", "csrc", "*.h"))):
    h.update(open(f, "rb").read())
sys.path.insert(0, ROOT)
import bench  # noqa: E402 (stdlib-only at import time): per-file hashes, so that staleness is judged per kernel
summary["_meta"] = {"kernel_sources_sha": h.hexdigest()[:16], "kernel_sources_sha_files": bench.source_shas(), "command": sys.argv[3] if len(sys.argv) > 3 else None,
                    "unit": "KiB per launch (rocprofv3); HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950"}
json.dump(summary, open(sys.argv[2], "w"), indent=1)
for k, cs in summary.items():
    if not k.startswith("_"):
        print(k, {c: round(v["avg"], 1) for c, v in cs.items()})
