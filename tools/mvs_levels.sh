#!/bin/bash
# duration of every k_mvs_pass dispatch of `bench.py --config rlc` (rocprofv3 --kernel-trace): a histogram by duration and the totals
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rocprofv3 --kernel-trace --output-format csv -d /tmp/mvs_trace -- python3 $R/bench.py --config rlc --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
d = []
for f in glob.glob('/tmp/mvs_trace/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_mvs_pass' in r['Kernel_Name']:
            d.append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
d.sort()
us = [x[1] for x in d]
half = us[len(us) // 2:]  # (the second run = the timed step, roughly)
print(f"{len(us)} dispatches in two runs; second half: {len(half)} dispatches, {sum(half) / 1e3:.1f} ms")
edges = [0, 20, 40, 80, 160, 320, 640, 1280, 2560, 5120, 1e9]
for a, b in zip(edges, edges[1:]):
    sel = [x for x in half if a <= x < b]
    print(f"  {a:>6.0f} .. {b:<8.0f} us: {len(sel):5d} dispatches, {sum(sel) / 1e3:8.1f} ms")
gaps = [d[i + 1][0] - (d[i][0] + int(d[i][1] * 1e3)) for i in range(len(d) // 2, len(d) - 1)]
gaps = [g / 1e3 for g in gaps if g < 200e3]
print(f"  gaps between consecutive dispatches (< 200 us): {len(gaps)}, mean {sum(gaps) / max(len(gaps), 1):.1f} us, total {sum(gaps) / 1e3:.1f} ms")
PY
