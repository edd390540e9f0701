#!/bin/bash
# Calibration of the FETCH_SIZE / WRITE_SIZE counters for SCATTERED 16-byte loads (the access shape of the rank kernels):
# tools/extend_bench.py expands 2^24 random parents per launch — "wide" ranges touch two 128-byte rank blocks per parent
# (4 x 16 B requested from each), "narrow" ranges one — and streams 16 B in / 68 B out per parent, coalesced.
# Known bytes per parent vs what the counters report gives the factor for this shape (MI355X_MICROARCH.md documents x2
# for wide coalesced streams only).   usage (GPU box): tools/calibrate_fetch.sh [genome bp]  -> gpurun_out/fetch_calibration.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-1e9}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  rocprofv3 --pmc $C --kernel-include-regex 'k_extend' --output-format csv -d /tmp/cal_$C -- python3 $R/tools/extend_bench.py $N > /tmp/cal_$C.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "rocprofv3 --pmc $C failed ($rc)"; tail -5 /tmp/cal_$C.log; exit $rc; fi
done
python3 - <<'PY' > $R/gpurun_out/fetch_calibration.txt
import csv, glob
N = 1 << 24
def per_dispatch(d, name):
    rows = []
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "k_extend" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]
f, w = per_dispatch("/tmp/cal_FETCH_SIZE", "FETCH_SIZE"), per_dispatch("/tmp/cal_WRITE_SIZE", "WRITE_SIZE")
print("dispatches:", len(f), len(w), "(3 modes x 11 launches per shape: wide first, then narrow)")
half = len(f) // 2
for name, lo, hi, lines in (("wide", 0, half, 2), ("narrow", half, len(f), 1)):
    fs = sum(f[lo:hi]) / max(hi - lo, 1) * 1024 / N      # KiB -> bytes per parent
    ws = sum(w[lo:hi]) / max(hi - lo, 1) * 1024 / N
    print(f"{name}: FETCH_SIZE {fs:.1f} B per parent, WRITE_SIZE {ws:.1f} B per parent")
    print(f"   known: {lines} rank line(s) of 128 B with 64 B requested in 4 x 16 B each, + 16 B parent read (coalesced); "
          f"written 68 B (coalesced)")
    rank = fs - 16 / 2   # (the coalesced 16-B parent stream is reported at half its bytes: the guide's x2 rule)
    print(f"   FETCH_SIZE per rank line = {rank / lines:.1f} B (line 128 B, requested 64 B)")
PY
cat $R/gpurun_out/fetch_calibration.txt
