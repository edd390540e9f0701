cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM --kernel-include-regex 'k_mvs_pass' --output-format csv -d /tmp/rlc_SQ -- python3 $R/bench.py --config rlc --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(float)
for f in glob.glob('/tmp/rlc_SQ/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(agg))
PY
