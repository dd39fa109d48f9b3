"""ms per MSM for every BASELINE.json configuration that fits one GPU (development / DESIGN.md numbers).
Protocol of the reference's scripts/msm-weierstrass.ts:22-48: warm-up, 15 runs with fresh scalars, drop 5,
median +- sample standard deviation."""
import statistics, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import msm_zprize_amd as m

def evaluate(curve, fn, n, runs=15, drop=5, **opts):
    pts = curve.Parallel.randomPointsFast(n, 1)
    warm = curve.Parallel.randomScalars(n, 2)
    fn(curve)(warm, pts, n, True, opts) if fn.__name__ != "proj" else curve.Parallel.msmProjective(warm, pts, n, opts)
    times, last = [], None
    for i in range(runs):
        sc = curve.Parallel.randomScalars(n, 100 + i)
        t0 = time.perf_counter()
        out = curve.Parallel.msmProjective(sc, pts, n, opts) if fn.__name__ == "proj" else fn(curve)(sc, pts, n, True, opts)
        dt = (time.perf_counter() - t0) * 1e3
        if i >= drop: times.append(dt)
        last = out["stats"]; sc.free()
    pts.free(); warm.free()
    return statistics.median(times), statistics.stdev(times), last

def unsafe(c): return c.Parallel.msmUnsafe
def safe(c): return c.Parallel.msm
def proj(c): return None

m.startThreads()
rows = []
C = m.Weierstrass.create(m.curves.bls12377Params)
for n, lab, fn, o in [(14, "cfg1 shape: BLS12-377 2^14 GLV (reference default)", unsafe, {"glv": 1}),
                      (16, "BLS12-377 2^16 GLV", unsafe, {"glv": 1}),
                      (20, "cfg2: BLS12-377 2^20 no GLV affine msmUnsafe", unsafe, {"glv": 0}),
                      (20, "      BLS12-377 2^20 no GLV affine msm (safe)", safe, {"glv": 0}),
                      (20, "      BLS12-377 2^20 GLV affine msmUnsafe", unsafe, {"glv": 1}),
                      (20, "      BLS12-377 2^20 msmProjective", proj, {}),
                      (23, "cfg5 shard: BLS12-377 2^23 no GLV affine", unsafe, {"glv": 0})]:
    rows.append((lab,) + evaluate(C, fn, 1 << n, **o))
C.close()
C = m.Weierstrass.create(m.curves.pallasParams)
rows.append(("cfg3: Pallas 2^22 msmProjective",) + evaluate(C, proj, 1 << 22))
rows.append(("      Pallas 2^22 GLV affine msmUnsafe",) + evaluate(C, unsafe, 1 << 22, glv=1))
C.close()
C = m.Weierstrass.create(m.curves.bls12381Params)
rows.append(("BLS12-381 2^20 GLV affine msmUnsafe",) + evaluate(C, unsafe, 1 << 20, glv=1))
C.close()
C = m.TwistedEdwards.create(m.curves.edOnBls12377Params)
rows.append(("cfg4: ed-on-bls12-377 2^24 (extended buckets, no GLV)",) + evaluate(C, safe, 1 << 24, runs=8, drop=3))
C.close()
for lab, med, sd, st in rows:
    print(f"{lab:58s} {med:9.2f} ms +- {sd:5.2f}   c={st.c} K={st.K} rounds={st.rounds} entries={st.n_entries} ({st.n_entries / med / 1e3:.0f} Mpoint-adds/s)")
