import sys, random
sys.path.insert(0, '.')
import msm_zprize_amd as m
from oracle import params as P, bigint_ref as B
m.startThreads()
C = m.Weierstrass.create(m.curves.bls12377Params)
A = B.AffineWeierstrass(P.BLS12_377)
q = P.BLS12_377["order"]
pt = dict(P.KAT_BLS12_377_POINT, isZero=False)
ptt = (pt["x"], pt["y"], False)
def run(scalars, glv=0, c=0, safe=True):
    n = len(scalars)
    pts = C.Parallel.pointsFromBigints([pt] * n)
    sc = C.Parallel.scalarsFromBigints(scalars)
    f = C.Parallel.msm if safe else C.Parallel.msmUnsafe
    out = f(sc, pts, n, True, {"glv": glv, "c": c})
    r = out["result"]
    want = A.scale(sum(scalars) % q, ptt)
    ok = (r["x"], r["y"], r["isZero"]) == want or (r["isZero"] and want[2])
    st = out["stats"]
    print(f"n={n} glv={glv} c={st.c} K={st.K} rounds={st.rounds} maxb={st.max_bucket} ok={ok}", flush=True)
    return ok
for n in [2, 3, 4, 5, 8, 9, 16, 33]:
    run([1] * n)
run([1, q - 1]); run([1, 1, q - 1]); run([2, 2, q - 2, 5])
rng = random.Random(5)
for n in [4, 10, 50, 200, 1000]:
    s = [rng.randrange(q) for _ in range(n)]
    run(s, 0); run(s, 1)
    run(s, 0, 4); run(s, 0, 9)
