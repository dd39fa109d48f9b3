import os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import msm_zprize_amd as m
m.startThreads()
C = m.Weierstrass.create(m.curves.bls12377Params)
for lg in (12, 14, 16, 18):
    n = 1 << lg
    pts = C.Parallel.randomPointsFast(n, 1)
    for name, fn in (("affine glv", lambda sc: C.Parallel.msmUnsafe(sc, pts, n, True, {"glv": 1})),
                     ("affine noglv", lambda sc: C.Parallel.msmUnsafe(sc, pts, n, True, {"glv": 0})),
                     ("projective", lambda sc: C.Parallel.msmProjective(sc, pts, n, {}))):
        ts = []
        for i in range(10):
            sc = C.Parallel.randomScalars(n, 7 + i)
            t0 = time.perf_counter(); out = fn(sc); ts.append((time.perf_counter() - t0) * 1e3); sc.free()
        st = out["stats"]
        print(f"2^{lg} {name:14s} {statistics.median(ts[2:]):.3f} ms c={st.c} K={st.K} stages=" + " ".join(f"{x:.3f}" for x in st.stage_ms[:8]))
    pts.free()
