#!/bin/bash
# Per-level durations of the frontier search (k_bfs_pass dispatches of one serial step), development helper.
# usage (on the GPU box): tools/bfs_levels.sh  -> gpurun_out/bfs_levels.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMB_SERIAL_SUBBATCHES=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/bfs_levels_bench.log 2>&1
python3 - <<PY > $R/gpurun_out/bfs_levels.txt
import csv, glob
rows = []
for f in glob.glob('/tmp/prof_lv/**/*_kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if 'k_bfs_pass' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print('dispatches', len(d), 'total ms', round(sum(d) / 1e3, 2))
# the last serial step = the last 3 runs of ~150 levels; print the last run level by level
n = 0
for i in range(len(d) - 1, 0, -1):
    n += 1
    if d[i - 1] < 20 and d[i] > 100: break   # (a new run starts with a long pass after short tail passes)
run = d[len(d) - n:]
print('levels of the last sub-batch run:', len(run), 'sum ms', round(sum(run) / 1e3, 2))
print(' '.join(f'{x:.0f}' for x in run))
PY
