"""Size sweep 2^14 .. 2^24 on one GPU and the window sweep of the reference's scripts/evaluate-msm-377.ts:15-62.

    python tools/sweep.py [out.json]

Per size (BLS12-377 G1, msmUnsafe, reference protocol: warm-up, then `runs` MSMs with fresh scalars, median):
ms per MSM, Mpoint-adds/s, the window the engine chose (c, K), the bucket-scatter kernel's fraction of the 8 TB/s HBM
peak (32 n + 4 E algorithmic bytes over its HIP-event time), the same for the whole sort, and the tree rounds'
field-multiplication rate against the measured 72 Gmodmul/s.  GLV on (the reference's default) and off.
Window sweep: for n in 14 .. 22 every c in [n-6, n] within [7, 19] (evaluate-msm-377.ts sweeps c around n-1 for
its CPU cost model; here the optimum sits near n-3).
"""
import json, os, statistics, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msm_zprize_amd as m

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep.json"
m.startThreads()
C = m.Weierstrass.create(m.curves.bls12377Params)


def evaluate(n, opts, runs):
    pts = C.Parallel.randomPointsFast(n, 1)
    warm = C.Parallel.randomScalars(n, 2)
    C.Parallel.msmUnsafe(warm, pts, n, True, opts)
    warm.free()
    wall, stats = [], []
    for i in range(runs):
        sc = C.Parallel.randomScalars(n, 100 + i)
        t0 = time.perf_counter()
        out = C.Parallel.msmUnsafe(sc, pts, n, True, opts)
        wall.append((time.perf_counter() - t0) * 1e3)
        stats.append(out["stats"])
        sc.free()
    pts.free()
    st = stats[-1]
    mean = lambda f: statistics.mean(f(s) for s in stats)
    ms = statistics.median(wall)
    entries = mean(lambda s: float(s.n_entries))
    coarse_ms = mean(lambda s: float(s.scatter_kernel_ms))
    sort_ms = mean(lambda s: float(s.stage_ms[0] + s.stage_ms[1] + s.stage_ms[2]))
    acc_ms = mean(lambda s: float(s.stage_ms[4]))
    pairs = mean(lambda s: float(s.n_pairs))
    sb = 32 * n + 4 * entries
    return {"log2n": n.bit_length() - 1, "glv": opts.get("glv", 0), "c": st.c, "K": st.K, "rounds": st.rounds,
            "ms_per_msm": ms, "stdev_ms": statistics.stdev(wall) if len(wall) > 1 else 0.0,
            "mpoint_adds_per_s": entries / ms / 1e3,
            "stage_ms": {nm: mean(lambda s, i=i: float(s.stage_ms[i])) for i, nm in
                         enumerate(["digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"])},
            "scatter_frac_of_8TBps": sb / (coarse_ms * 1e-3) / 8e12 if coarse_ms > 0 else None,
            "sort_frac_of_8TBps": sb / (sort_ms * 1e-3) / 8e12 if sort_ms > 0 else None,
            "modmul_frac_of_72G": pairs * 6 / (acc_ms * 1e-3) / 72e9 if acc_ms > 0 else None}


res = {"sizes": [], "window_sweep": []}
for lg in range(14, 25):
    for glv in (1, 0):
        if glv and lg > 23:
            continue
        r = evaluate(1 << lg, {"glv": glv}, 8 if lg <= 22 else 4)
        res["sizes"].append(r)
        print(json.dumps(r), flush=True)
for lg in range(14, 23):
    for c in range(max(lg - 6, 7), min(lg, 19) + 1):
        for glv in (1, 0):
            r = evaluate(1 << lg, {"glv": glv, "c": c}, 5)
            res["window_sweep"].append({k: r[k] for k in ("log2n", "glv", "c", "K", "rounds", "ms_per_msm", "stdev_ms")})
            print(json.dumps(res["window_sweep"][-1]), flush=True)
C.close()
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
