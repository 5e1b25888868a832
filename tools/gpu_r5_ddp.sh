#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for W in 2 4 8; do
  B=$((256 / W))
  LOG=r05_ddp${W}_one_gpu_r18 WORLD=$W timeout 1500 bash tools/ddp_one_gpu.sh --batch $B --steps 2 --warmup 4 --no_roofline 2>&1 | cut -c1-1500
done
