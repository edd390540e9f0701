#!/bin/bash
# kernel-time summary of one bench configuration:  tools/prof_stats.sh <tag> [bench args...]  -> gpurun_out/<tag>_kernel_stats.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py "$@" --no-cpu-baseline --no-streaming --no-rlc > $R/gpurun_out/${TAG}_bench.log 2>&1
python3 - <<PY > $R/gpurun_out/${TAG}_kernel_stats.csv
import csv, glob
rows = []
for f in glob.glob('/tmp/prof_$TAG/**/*_kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for r in rows:
    n = r["Name"].split("(")[0]
    if n.startswith("cmb::") or n.startswith("void cmb::") or "rocprim" in n:
        print(",".join([n[:80].replace(",", ";"), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
PY
tail -2 $R/gpurun_out/${TAG}_bench.log
