#!/bin/bash
# usage: kstat.sh <tag> [env...]  -> per-kernel avg ms of k_parts / k_exact / k_verify under rocprofv3
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
CMB_SERIAL_SUBBATCHES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-rlc > /tmp/ks_$TAG.log 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob('/tmp/ks_$TAG/**/*_kernel_stats.csv', recursive=True): rows += list(csv.DictReader(open(f)))
print("== $TAG")
for r in rows:
    n=r["Name"]
    if any(x in n for x in ("k_parts","k_exact","k_verify<","k_bfs_pass","k_fmocc")):
        print(f'{n[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f}')
PY
