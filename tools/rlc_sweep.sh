#!/bin/bash
# A/B runs of the b-move search on one index: tools/rlc_sweep.sh "ENV=VAL ..." ["ENV=VAL ..."]...   (GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
for cfg in "$@"; do
  ( for kv in $cfg; do export "$kv"; done
    python3 $R/bench.py --config rlc --haplotypes 32 --base-mbp 2 --reads 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-streaming --no-rlc 2>/dev/null |
    python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['per_kernel'].items()})" )
done
