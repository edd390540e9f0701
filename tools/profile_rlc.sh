#!/bin/bash
# rocprofv3 evidence for `bench.py --config rlc` (BASELINE configs[4] stand-in on the b-move index):
#   tools/profile_rlc.sh r03 [bench args...]   -> gpurun_out/profiles_<round>_rlc/  (copy into profiles/ and commit)
set -u
ROUND=${1:-rXX}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles_${ROUND}_rlc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config rlc $@"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rlc_stats -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-streaming --no-rlc > $OUT/bench_under_rocprof.log 2>&1
python3 - <<PY > $OUT/kernel_stats.csv
import csv, glob
rows = []
for f in glob.glob('/tmp/rlc_stats/**/*_kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for r in rows:
    n = r["Name"].split("(")[0]
    if "cmb::" in n or "rocprim" in n or "hipcub" in n:
        print(",".join([n[:90].replace(",", ";"), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
PY
# HBM traffic counters: every run starts with a warm-up that sizes the pools, so one step = (two timed steps) - (one timed step)
for C in FETCH_SIZE WRITE_SIZE; do
  for N in 1 2; do
    rocprofv3 --pmc $C --kernel-include-regex 'k_mvs|k_move' --output-format csv -d /tmp/rlc_${C}_$N -- python3 $R/bench.py $ARGS --steps $N --warmup 1 --no-cpu-baseline --no-streaming --no-rlc > $OUT/pmc_${C}_$N.log 2>&1
  done
done
cd $R && python3 bench.py $ARGS > $OUT/bench_line.json 2> $OUT/bench_stderr.log
python3 $R/tools/pmc_traffic_rlc.py $OUT/bench_line.json /tmp/rlc_FETCH_SIZE_1 /tmp/rlc_FETCH_SIZE_2 /tmp/rlc_WRITE_SIZE_1 /tmp/rlc_WRITE_SIZE_2 > $OUT/pmc_traffic.json
ls -la $OUT
