#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:  tools/profile_round.sh r01 [bench args...]
# Writes small summaries under gpurun_out/profiles_<round>/ (copy them into profiles/ and commit).
set -u
ROUND=${1:-rXX}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/profiles_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$@"
# 1) per-kernel time (same command as the bench line)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-streaming --no-rlc > $OUT/bench_under_rocprof.log 2>&1
python3 - <<PY > $OUT/kernel_stats.csv
import csv, glob
rows = []
for f in glob.glob('/tmp/prof_stats/**/*_kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for r in rows:
    n = r["Name"].split("(")[0]
    if n.startswith("cmb::") or n.startswith("void cmb::") or "rocprim" in n:
        print(",".join([n[:80].replace(",", ";"), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
PY
# 1b) the same with the sub-batches one after the other (the kernel durations bench.py's roofline table is built on)
CMB_SERIAL_SUBBATCHES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_serial -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-streaming --no-rlc > $OUT/bench_under_rocprof_serial.log 2>&1
python3 - <<PY > $OUT/kernel_stats_serial.csv
import csv, glob
rows = []
for f in glob.glob('/tmp/prof_stats_serial/**/*_kernel_stats.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for r in rows:
    n = r["Name"].split("(")[0]
    if n.startswith("cmb::") or n.startswith("void cmb::") or "rocprim" in n:
        print(",".join([n[:80].replace(",", ";"), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]))
PY
# 2) HBM traffic counters, each in its own pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass);
#    only this library's kernels are instrumented (the harness' torch kernels on > 2^31-element tensors crashed under --pmc)
for C in FETCH_SIZE WRITE_SIZE; do
  CMB_SERIAL_SUBBATCHES=1 rocprofv3 --pmc $C --kernel-include-regex 'cmb::' --output-format csv -d /tmp/prof_$C -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-streaming --no-rlc > $OUT/pmc_$C.log 2>&1
done
CMB_SERIAL_SUBBATCHES=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM --kernel-include-regex 'cmb::' --output-format csv -d /tmp/prof_SQ -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline --no-streaming --no-rlc > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE /tmp/prof_SQ > $OUT/pmc_summary.txt 2>&1
# 3) the plain bench line (not under the profiler), then the per-kernel traffic table bench.py reads back
cd $R && python3 bench.py $ARGS > $OUT/bench_line.json 2> $OUT/bench_stderr.log
python3 $R/tools/pmc_traffic.py $OUT/bench_line.json /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE /tmp/prof_SQ > $OUT/pmc_traffic.json
ls -la $OUT
