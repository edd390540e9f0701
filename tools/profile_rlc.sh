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
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex 'k_mvs|k_move' --output-format csv -d /tmp/rlc_$C -- python3 $R/bench.py $ARGS --steps 1 --warmup 1 --no-cpu-baseline --no-streaming --no-rlc > $OUT/pmc_$C.log 2>&1
done
python3 - <<'PY' > $OUT/pmc_traffic.txt
import csv, glob, collections
def per_kernel(d, name):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cmb::", "")
                agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    return agg, cnt
f, fc = per_kernel("/tmp/rlc_FETCH_SIZE", "FETCH_SIZE")
w, _ = per_kernel("/tmp/rlc_WRITE_SIZE", "WRITE_SIZE")
print("bench.py --config rlc --steps 1 --warmup 1: two runs of the hot path (the warm-up run includes the re-runs that size the pools); KiB -> GB, FETCH_SIZE x 2 (gfx950)")
for k in sorted(f, key=lambda k: -f[k]):
    print(f"{k:40s} dispatches {fc[k]:6d}  2 x FETCH_SIZE {2 * f[k] * 1024 / 1e9:9.2f} GB  WRITE_SIZE {w.get(k, 0) * 1024 / 1e9:9.2f} GB")
PY
cd $R && python3 bench.py $ARGS > $OUT/bench_line.json 2> $OUT/bench_stderr.log
ls -la $OUT
