#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per 128-byte line for scattered loads of 16 / 32 / 64 / 128 bytes and for a coalesced stream
# (tools/fetch_granule.hip).   usage (GPU box): tools/calibrate_fetch_granule.sh   -> gpurun_out/fetch_granule.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_granule $R/tools/fetch_granule.hip || exit 1
cd /tmp && export TMPDIR=/tmp
/tmp/fetch_granule > /tmp/fg_plain.log 2>&1 || { cat /tmp/fg_plain.log; exit 1; }
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fg_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/fg_$C -- /tmp/fetch_granule > /tmp/fg_$C.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "rocprofv3 --pmc $C failed ($rc)"; tail -5 /tmp/fg_$C.log; exit $rc; fi
done
python3 - <<'PY' > $R/gpurun_out/fetch_granule.txt
import csv, glob, collections
N = 1 << 24
def per_kernel(d, name):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: agg[k] / cnt[k] for k in agg}, cnt
f, fc = per_kernel("/tmp/fg_FETCH_SIZE", "FETCH_SIZE")
w, _ = per_kernel("/tmp/fg_WRITE_SIZE", "WRITE_SIZE")
print("timings without a profiler attached:")
print(open("/tmp/fg_plain.log").read())
print("raw counters per launch / N lines (FETCH_SIZE, WRITE_SIZE in KiB -> bytes); the 4-byte line index of every line is")
print("a coalesced stream and is tallied at half its bytes (2 B per line) by the guide's rule")
for k in sorted(f):
    print(f"{k:20s} launches {fc[k]:2d}  FETCH_SIZE {f[k] * 1024 / N:7.1f} B per line   WRITE_SIZE {w.get(k, 0) * 1024 / N:6.2f} B per line")
PY
cat $R/gpurun_out/fetch_granule.txt
