"""Per-kernel HBM traffic of ONE bench step from the rocprofv3 --pmc passes of tools/profile_round.sh.

    python tools/pmc_traffic.py <bench_line.json> <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [steps run]  > *_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are collected in separate passes (they do not fit one pass, MI355X_MICROARCH.md) of
`bench.py --steps 1 --warmup 0`, which runs the hot path 1 + SERIAL_TABLE_STEPS times (one timed step + the serial steps its
kernel table is the median of; the constant is read from bench.py): the sums over all dispatches of a kernel are divided by the
number of steps run, and the division is checked against the dispatch count of k_parts (one launch per sub-batch and step).  Units: KiB.
bench.py (load_traffic) applies the gfx950 correction (FETCH_SIZE x 2) when it reports `roofline.traffic`."""
import collections, csv, glob, hashlib, json, os, sys


def source_digest():  # the same digest bench.py computes: a profile only speaks for the kernels it was measured on
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "columba_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.startswith(("move_", "pair_")):
            continue  # translation units of their own (b-move backend, paired-end records): none of the kernels timed here
        with open(os.path.join(d, fn), "rb") as f:
            h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
import re
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_serial = int(re.search(r"^SERIAL_TABLE_STEPS = (\d+)", open(os.path.join(_root, "bench.py")).read(), re.M).group(1))
steps_run = float(sys.argv[4]) if len(sys.argv) > 4 and not os.path.isdir(sys.argv[4]) else 1.0 + _serial
sq_dir = next((a for a in sys.argv[4:] if os.path.isdir(a)), None)  # the SQ_* pass (wave-instructions per kernel)
out = {"workload": {k: line["config"][k] for k in ("genome_bp", "reads_per_gpu", "read_len", "k")},
       "kernel_src_sha": source_digest(),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one pass each, bench.py --steps 1 --warmup 0 "
                 "(sub-batches one after the other); per step",
       "kernels": collections.defaultdict(dict)}
for d, name in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg, calls = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if not k.startswith("cmb::"):
                continue
            k = k[len("cmb::"):]
            agg[k] += float(r["Counter_Value"])
            calls[k] += 1
        for k, v in agg.items():
            out["kernels"][k][name + "_KiB"] = round(v / steps_run, 1)
            out["kernels"][k]["dispatches_per_step"] = calls[k] / steps_run
if sq_dir:
    for f in glob.glob(sq_dir + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k.startswith("cmb::"):
                agg[k[len("cmb::"):]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in agg.items():
            for c, x in v.items():
                out["kernels"][k][c] = round(x / steps_run, 1)
out["steps_run"] = steps_run
R = line["config"]["reads_per_gpu"]  # (columba_amd.hip: 3 sub-batches from 8 M reads, 2 from 2 M, CMB_SUBBATCHES overrides)
sub = int(os.environ.get("CMB_SUBBATCHES", 3 if R >= 8000000 else 2 if R >= 2000000 else 1))
if "k_parts" in out["kernels"] and abs(out["kernels"]["k_parts"]["dispatches_per_step"] - sub) > 1e-6:
    sys.exit(f"k_parts was launched {out['kernels']['k_parts']['dispatches_per_step']} times per assumed step, the batch has {sub} sub-batches: "
             f"wrong number of steps ({steps_run})")
print(json.dumps(out, indent=1))
