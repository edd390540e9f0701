#!/bin/bash
# per-kernel time of the verification path at 8 ... 10 errors (k_wide_filter / k_verify_wide / sorts):  tools/dp_stats.sh ["k = 9"]
R=${GRAFT_REPO_ROOT:-$PWD}
SEL=${1:-k = 9}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_stats -- python3 $R/tools/config_table.py 3000 "$SEL" > /tmp/dp_stats.log 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob('/tmp/dp_stats/**/*_kernel_stats.csv', recursive=True): rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in [r for r in rows if "cmb::" in r["Name"]][:10] + rows[:6]:
    print(f'{r["Name"].split("(")[0][:70]:70s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"])/1e6:10.1f}')
PY
tail -2 /tmp/dp_stats.log | cut -c1-300
