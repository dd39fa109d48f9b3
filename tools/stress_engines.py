import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import msm_zprize_amd as m
n = 1 << 20
def run(engines, iters):
    m.startThreads(devices=[0] * engines)
    C = m.Weierstrass.create(m.curves.bls12377Params)
    pts = C.Parallel.randomPointsFast(n, 11)
    res = []
    for i in range(iters):
        sc = C.Parallel.randomScalars(n, 500 + i)
        res.append(C.Parallel.msmUnsafe(sc, pts, n, False, {"glv": i & 1})["result"])
        sc.free()
    pts.free(); C.close(); m.stopThreads()
    return res
ref = run(1, 24)
for e in (2, 3, 4):
    got = run(e, 24)
    bad = [i for i in range(24) if got[i] != ref[i]]
    print("engines", e, "mismatches", bad)
