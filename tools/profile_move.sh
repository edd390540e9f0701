#!/bin/bash
# rocprofv3 evidence for the b-move kernels:  tools/profile_move.sh <tag>   (writes gpurun_out/profiles_move_<tag>/)
# kernel statistics of tools/move_bench.py, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (only this library's
# b-move kernels are instrumented).
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles_move_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_stats -- python3 $R/tools/move_bench.py > $OUT/move_bench_under_rocprof.log 2>&1
python3 - <<PY > $OUT/kernel_stats.csv
import csv, glob
rows = []
for f in glob.glob('/tmp/pm_stats/**/*_kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs")
for r in rows:
    n = r["Name"].split("(")[0]
    if "k_move" in n or "k_posset" in n or "k_run_map" in n:
        print(",".join([n[:80].replace(",", ";"), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]]))
PY
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex 'k_move' --output-format csv -d /tmp/pm_$C -- python3 $R/tools/move_bench.py > $OUT/pmc_$C.log 2>&1
done
python3 - <<PY > $OUT/pmc_summary.txt
import csv, glob, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob('/tmp/pm_%s/**/*counter_collection.csv' % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                k = r["Kernel_Name"].split("(")[0]
                acc[k][0] += 1
                acc[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(acc.items()):
        print(f"{C} {k[:60]:60s} dispatch-rows {n:6d} total_KiB {v:.4g}")
PY
cat $OUT/kernel_stats.csv $OUT/pmc_summary.txt
