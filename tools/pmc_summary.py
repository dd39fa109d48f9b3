"""Summarize a rocprofv3 --pmc counter_collection.csv: per kernel name, summed counters (development aid)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0][-50:]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[name].add(r["Dispatch_Id"])
for name, v in agg.items():
    print(f"{name:52s} calls={len(calls[name]):3d} " + " ".join(f"{k}={x:.4g}" for k, x in sorted(v.items())))
