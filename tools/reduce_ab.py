"""A/B of the two first levels of the bucket reduction: XYZZ running sums (default) against the
batched-affine lock-step running sums (reduceBucketsAffine, opt reduceAffine = 1).  Prints per-stage times and the
reduce stage for both, on the same inputs; results must agree.   python tools/reduce_ab.py > profiles/r02_reduce_ab.txt"""
import os, statistics, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msm_zprize_amd as m
m.startThreads()
C = m.Weierstrass.create(m.curves.bls12377Params)
names = ["digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"]
print("BLS12-377 G1, msmUnsafe; stage times in ms (mean of 5 after 2 warm-ups); `reduce` = all reduction kernels")
for log2n, glv in ((20, 0), (20, 1), (16, 1), (23, 0)):
    n = 1 << log2n
    pts = C.Parallel.randomPointsFast(n, 1)
    res = {}
    for mode in (0, 1):
        acc = []
        for i in range(7):
            sc = C.Parallel.randomScalars(n, 50 + i)
            out = C.Parallel.msmUnsafe(sc, pts, n, True, {"glv": glv, "reduceAffine": mode})
            sc.free()
            if i >= 2:
                acc.append([out["stats"].stage_ms[j] for j in range(8)])
            res[(mode, i)] = out["result"]
        mean = [statistics.mean(a[j] for a in acc) for j in range(8)]
        st = out["stats"]
        print(f"2^{log2n} glv={glv} c={st.c} K={st.K} rounds={st.rounds} first level = {'batched-affine (f2)' if mode else 'XYZZ running sums'}: "
              + "  ".join(f"{nm}={mean[j]:.3f}" for j, nm in enumerate(names)))
    assert all(res[(0, i)] == res[(1, i)] for i in range(7)), "variants disagree"
    pts.free()
C.close()
