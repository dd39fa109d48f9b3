"""Randomized parity sweep (development aid, run on the GPU box): random curve / size / window / GLV / entry point against
the C oracle, bit-exact on the canonical affine result.  usage: fuzz_parity.py [cases=150] [seed=1]"""
import os
import random
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msm_zprize_amd as m
from oracle import c_oracle
from oracle import params as P

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
m.startThreads()
curves = {}


def curve(label):
    if label not in curves:
        params = m.curves.BY_LABEL[label]
        curves[label] = (m.Weierstrass if params["kind"] == "weierstrass" else m.TwistedEdwards).create(params)
    return curves[label]


def strip(p):
    return {"x": p["x"], "y": p["y"], "isZero": bool(p.get("isZero", False))}


t0 = time.time()
bad = 0
for case in range(cases):
    label = rng.choice(["bls12-377", "bls12-377", "pallas", "bls12-381", "ed-on-bls12-377"])
    kind = rng.random()
    n = rng.randint(1, 300) if kind < 0.3 else rng.randint(301, 6000) if kind < 0.85 else rng.choice([1 << 14, (1 << 15) + 37, 1 << 16, 100000])
    c = rng.choice([0, 0, 0] + list(range(2, 21)))
    if n > 20000 and 0 < c < 8:
        c = 0
    seed = rng.randrange(1 << 30)
    cv = curve(label)
    pts = cv.Parallel.randomPointsFast(n, seed)
    sc = cv.Parallel.randomScalars(n, seed + 1)
    want = strip(c_oracle.msm(P.CURVES[label], cv.Scalar.toBigints(sc), cv.Affine.toBigints(pts)))
    if label == "ed-on-bls12-377":
        how = "msm"
        got = cv.Parallel.msm(sc, pts, n, True, {"c": c})["result"]
    else:
        how = rng.choice(["msmUnsafe", "msmUnsafe", "msm", "msmProjective"])
        glv = rng.choice([0, 1, -1])
        if how == "msmProjective":
            got = cv.Parallel.msmProjective(sc, pts, n, {"c": c})["result"]
        elif how == "msm":
            got = cv.Parallel.msm(sc, pts, n, False, {"glv": glv, "c": c})["result"]
        else:
            got = cv.Parallel.msmUnsafe(sc, pts, n, True, {"glv": glv, "c": c})["result"]
        how += f" glv={glv}"
    ok = strip(got) == want
    bad += not ok
    if not ok or case % 25 == 0:
        print(f"case {case}: {label} n={n} c={c} {how}: {'ok' if ok else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
    pts.free()
    sc.free()
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
