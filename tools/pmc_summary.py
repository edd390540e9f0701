"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (development helper)."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in agg.items():
            if "cmb::" in k:
                print(d.split("/")[-1], k, {a: "%.4g" % b for a, b in sorted(v.items())})
    for f in glob.glob(d + "/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cmb::" in r["Name"]:
                print("stats", r["Name"].split("(")[0], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
