#!/bin/bash
# A/B runs of one bench configuration under different environment settings on the GPU box:
#   tools/ab_bench.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...      ("-" = no setting)
R=${GRAFT_REPO_ROOT:-$PWD}
ARGS=$1; shift
for SET in "$@"; do
  [ "$SET" = "-" ] && SET=""
  echo "=== $SET"
  env $SET python3 $R/bench.py $ARGS --no-cpu-baseline --no-streaming --no-rlc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']
print(round(d['value']/1e6,2),'M reads/s', d['ms_per_step'],'ms/step | serial ms:', {k:v['ms'] for k,v in pk.items()}, '| dfs frac', pk['k_dfs']['frac_of_hbm_peak'])
"
done
