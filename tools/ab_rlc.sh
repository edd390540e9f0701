#!/bin/bash
# A/B runs of the configs[4] stand-in (b-move backend) under different environment settings on the GPU box:
#   tools/ab_rlc.sh "<bench args>" "VAR=a" "CMB_LIB=..." ...      ("-" = no setting)
R=${GRAFT_REPO_ROOT:-$PWD}
ARGS=$1; shift
for SET in "$@"; do
  [ "$SET" = "-" ] && SET=""
  echo "=== $SET"
  env $SET python3 $R/bench.py --config rlc $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(round(d['value']/1e3,1),'k reads/s', d['ms_per_step'],'ms/step | serial ms:', {k:v['ms'] for k,v in r.get('per_kernel',{}).items()}, '| frac', r['frac'], '| overlap', r.get('overlap'))
"
done
