#!/bin/bash
# L2 request counters of the tiled convolution with and without the LDS-resident halo (AFAN_CONV_HALO=1 / 0), forward launches
# of the step's 8x8 and 4x4 stages: TCC_REQ_sum (128-byte L2 requests), TCP_TCC_READ_REQ_sum, TCC_BUSY_sum per launch.
#   [BASE_LIB=path/to/an/older/libafan_hip.so] bash tools/gpu_pmc_halo.sh   -> gpurun_out/pmc_halo/summary.json
# (separate --pmc passes, kernel-trace only; BASE_LIB adds a third column "base": another build of the library, e.g. the
#  tap-outer K order of the commit before the halo form)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_halo
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for SH in 256,256,8,3,1 512,512,4,3,1; do
for HALO in 0 1 ${BASE_LIB:+2}; do
for SET in "TCC_REQ_sum TCC_BUSY_sum" "TCP_TCC_READ_REQ_sum"; do
  TAG=$(echo $SH | tr ',' '_')_h${HALO}_$(echo $SET | cut -d' ' -f1)
  rm -rf /tmp/pmch_$TAG
  export SHAPE=$SH N=256 AFAN_CONV_HALO=$HALO
  if [ $HALO = 2 ]; then export AFAN_HIP_LIB=$BASE_LIB; else unset AFAN_HIP_LIB; fi
  timeout 90 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmch_$TAG -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py > $OUT/$TAG.log 2>&1
done
done
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, json, os, re
out = collections.defaultdict(dict)
for d in sorted(glob.glob('/tmp/pmch_*')):
    m = re.match(r'/tmp/pmch_(\d+_\d+_\d+_\d+_\d+)_h(\d)_', d)
    shape, halo = m.group(1), {"0": "per_tap", "1": "halo", "2": "base"}[m.group(2)]
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_igemm' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            out[shape].setdefault(halo, {})[k] = sorted(v)[len(v) // 2]
res = {"what": "median per forward launch, batch 256, bf16, 3x3 / stride 1; TCC_REQ_sum counts 128-byte L2 requests",
       "command": "bash tools/gpu_pmc_halo.sh", "shapes": {}}
for shape, d in out.items():
    e = dict(d)
    if "halo" in d and "per_tap" in d and "TCC_REQ_sum" in d["halo"]:
        e["TCC_REQ_ratio_halo_over_per_tap"] = round(d["halo"]["TCC_REQ_sum"] / d["per_tap"]["TCC_REQ_sum"], 4)
        e["L2_request_MB"] = {k: round(d[k]["TCC_REQ_sum"] * 128 / 1e6, 1) for k in ("base", "per_tap", "halo") if k in d}
        if "base" in d:
            e["TCC_REQ_ratio_halo_over_base"] = round(d["halo"]["TCC_REQ_sum"] / d["base"]["TCC_REQ_sum"], 4)
    res["shapes"]["ci_co_h_k_s=" + shape] = e
json.dump(res, open(os.path.join('gpurun_out/pmc_halo', 'summary.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
