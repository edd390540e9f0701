#!/bin/bash
# The headline bench with larger seed tables (-K, alignparameters.cpp:440-470: 0 ... 15; default 10): tools/kmer_sweep.sh 10 12 13 14
R=${GRAFT_REPO_ROOT:-$PWD}
for K in "$@"; do
  python3 $R/bench.py --kmer-size $K --sparseness ${SPARSENESS:-4} --in-text-switch ${SWITCH:-4} --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-streaming --no-rlc 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('K=$K', d['value'], d['ms_per_step'], d['config']['index_bytes_hbm'], {k:v['ms'] for k,v in d['roofline']['per_kernel'].items()}, d['config'].get('occurrences'))"
done
