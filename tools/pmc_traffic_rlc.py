"""Per-kernel HBM traffic of ONE step of `bench.py --config rlc` from the rocprofv3 --pmc passes of tools/profile_rlc.sh.

    python tools/pmc_traffic_rlc.py <bench_line.json> <FETCH 1-step dir> <FETCH 2-step dir> <WRITE 1-step dir> <WRITE 2-step dir>  > *_rlc_pmc_traffic.json

Every run starts with a warm-up step that also sizes the pools (re-runs included), so the counters of a run are not a multiple of a
step's: a step is the DIFFERENCE between the run with two timed steps and the run with one.  Units: KiB; bench.py (load_rlc_traffic)
applies the gfx950 correction (FETCH_SIZE x 2)."""
import collections, csv, glob, hashlib, json, os, sys


def move_source_digest():  # the digest bench.py computes for the b-move backend's translation unit
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "columba_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.startswith("pair_") or fn == "columba_amd.hip":
            continue
        with open(os.path.join(d, fn), "rb") as f:
            h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def per_kernel(d, name):
    agg, calls = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cmb::", "")
                agg[k] += float(r["Counter_Value"])
                calls[k] += 1
    return agg, calls


line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
wl = line["config"]["workload"]
cfg = line["config"]
out = {"workload": {"reads": cfg["reads_per_gpu"], "read_len": cfg["read_len"], "k": cfg["k"], "description": wl},
       "kernel_src_sha": move_source_digest(),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, bench.py --config rlc: (--steps 2 --warmup 1) minus (--steps 1 --warmup 1); per step",
       "kernels": collections.defaultdict(dict)}
for name, d1, d2 in (("FETCH_SIZE", sys.argv[2], sys.argv[3]), ("WRITE_SIZE", sys.argv[4], sys.argv[5])):
    a1, c1 = per_kernel(d1, name)
    a2, c2 = per_kernel(d2, name)
    for k in a2:
        out["kernels"][k][name + "_KiB"] = round(a2[k] - a1.get(k, 0.0), 1)
        out["kernels"][k]["dispatches_per_step"] = c2[k] - c1.get(k, 0)
print(json.dumps(out, indent=1))
