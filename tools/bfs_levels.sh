#!/bin/bash
# Per-pass profile of the frontier search: nodes and events consumed by every k_bfs_pass of the last sub-batch of one
# serial step (CMB_VERBOSE=2) next to the pass's duration (rocprofv3 --kernel-trace).  Development helper.
# usage (on the GPU box): [CMB_BFS_CHAIN=n ...] tools/bfs_levels.sh [tag]  -> gpurun_out/bfs_levels_<tag>.txt
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lv
export CMB_SERIAL_SUBBATCHES=1 CMB_VERBOSE=2
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-streaming --no-rlc > $R/gpurun_out/bfs_levels_bench_$TAG.log 2>&1
python3 - $R/gpurun_out/bfs_levels_bench_$TAG.log <<'PY' > $R/gpurun_out/bfs_levels_$TAG.txt
import csv, glob, re, sys
rows = []
for f in glob.glob('/tmp/prof_lv/**/*_kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if 'k_bfs_pass' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print('dispatches', len(d), 'total ms', round(sum(d) / 1e3, 2))
# the passes of the last search run, from the log ("  pass p: N nodes, E events" after a "[bfs]" line)
runs, cur = [], None
for line in open(sys.argv[1], errors='replace'):
    if line.startswith('[bfs]'):
        cur = []
        runs.append((line.strip(), cur))
    m = re.match(r'\s+pass (\d+): (\d+) nodes, (\d+) events', line)
    if m and cur is not None:
        cur.append((int(m.group(2)), int(m.group(3))))
hdr, passes = runs[-1]
print(hdr)
npass = int(re.search(r'(\d+) passes', hdr).group(1))
run = d[len(d) - npass:]
print('passes of the last sub-batch run:', len(run), 'sum ms', round(sum(run) / 1e3, 2))
print('pass  nodes  events  us  nodes/us')
tot = 0
for i, us in enumerate(run):
    n, e = passes[i] if i < len(passes) else (0, 0)
    tot += n
    print(i, n, e, round(us), round(n / us, 1) if us else 0)
print('nodes consumed', tot)
PY
