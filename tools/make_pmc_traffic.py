"""Derive profiles/<tag>_pmc_traffic.json (bench.py's `traffic` fields) from the two rocprofv3 PMC passes.

usage: make_pmc_traffic.py <pmc_fetch_dir> <pmc_write_dir> <out.json> <msms-per-run> <log2n>
FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE counts half the bytes of 16-B-per-lane coalesced streaming reads -> doubled; WRITE_SIZE as is.  For
kernels whose reads are not wide streams (gathers of 96-byte records) the doubled figure is an upper bound: both are
recorded.
"""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    tot, ids = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void msmz::", "").replace("msmz::", "").split("<")[0]
        tot[name] += float(r["Counter_Value"])
        ids[name].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in ids.items()}


fetch_dir, write_dir, out, reps, log2n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
fk, fcalls = per_kernel(fetch_dir, "FETCH_SIZE")
wk, wcalls = per_kernel(write_dir, "WRITE_SIZE")
KB = 1024


def entry(names, per):
    f = sum(fk.get(n, 0.0) for n in names) * KB / per
    w = sum(wk.get(n, 0.0) for n in names) * KB / per
    return {"kernels": names, "fetch_bytes_as_counted": int(f), "write_bytes": int(w),
            "hbm_bytes": int(2 * f + w), "hbm_bytes_fetch_uncorrected": int(f + w)}


sort_kernels = ["k_hist", "k_bin_scan", "k_coarse", "k_fine"]
res = {
    "workload": f"BLS12-377 G1 2^{log2n}, no GLV (tools/profile_msm.py --reps {reps} under rocprofv3 --pmc, separate passes)",
    "correction": "gfx950: FETCH_SIZE counts 1/2 of the bytes of 16-B-per-lane coalesced streaming reads "
                  "(MI355X_MICROARCH.md, HBM section) -> doubled in hbm_bytes; WRITE_SIZE taken as is",
    "k_coarse": dict(entry(["k_coarse"], max(fcalls.get("k_coarse", 1), 1)), per="launch"),
    "sort": dict(entry(sort_kernels, reps), per="MSM"),
    "k_batch_add": dict(entry(["k_batch_add"], reps), per="MSM (all tree rounds)"),
    "k_plan": dict(entry(["k_plan_count", "k_plan_emit"], reps), per="MSM"),
    "reduce": dict(entry(["k_reduce2d_partial", "k_reduce2d_partial_acc", "k_pairsum", "k_pairsum_x4", "k_fill_neutral", "k_reduce_first",
                          "k_reduce_quad", "k_reduce_quad16", "k_reduce_tail"], reps), per="MSM"),
}
json.dump(res, open(out, "w"), indent=1)
print(open(out).read())
