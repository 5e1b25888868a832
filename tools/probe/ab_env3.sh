#!/bin/bash
# A/B of environment settings on given bench args: usage  ab_env3.sh "A=1 B=2" "A=0" ... -- bench args...   (2 rounds, interleaved)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
for rep in 1 2; do
for v in "${SETS[@]}"; do
    env $v python3 bench.py --no_cpu_baseline --no_roofline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v |', d['metric'][27:80], d['ms_per_step'], 'ms')"
done
done
