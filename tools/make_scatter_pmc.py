"""Derive profiles/<tag>_scatter_pmc.json (bench.py's roofline.traffic) from the two rocprofv3 PMC passes.

usage: make_scatter_pmc.py <pmc_fetch_dir> <pmc_write_dir> <out.json> <K> <log2n>
FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE counts half the bytes of 16-B-per-lane coalesced streaming reads -> doubled; WRITE_SIZE as is.
"""
import csv, glob, json, sys


def per_launch(d, counter, kernel):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot, ids = 0.0, set()
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            ids.add(r["Dispatch_Id"])
    return tot / max(len(ids), 1)


fetch_dir, write_dir, out, K, log2n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
fk = per_launch(fetch_dir, "FETCH_SIZE", "k_scatter_coarse")
wk = per_launch(write_dir, "WRITE_SIZE", "k_scatter_coarse")
json.dump({
    "kernel": "k_scatter_coarse",
    "workload": f"BLS12-377 G1 2^{log2n}, no GLV, K={K} (tools/profile_msm.py under rocprofv3 --pmc, separate passes)",
    "FETCH_SIZE_KB_per_launch": fk,
    "WRITE_SIZE_KB_per_launch": wk,
    "correction": "gfx950: FETCH_SIZE counts 1/2 of the bytes of 16-B-per-lane coalesced streaming reads "
                  "(MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE taken as is",
    "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
    "algorithmic_bytes_per_launch": K * (1 << log2n) * 8,
}, open(out, "w"), indent=1)
print(open(out).read())
