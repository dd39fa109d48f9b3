"""Writes tests/golden/js_msm_fixtures.json: expected results of the JavaScript mirror of the reference's integration
test src/msm.test.ts:24-118 (four curves, N = 2^0, 2^2, ..., 2^12) on the engine's seeded inputs.

The reference draws unseeded random inputs and compares with its bigint MSM in the same process; node has no oracle here,
so the inputs are the seeded device generators (point i = a_i * G with a_i = oracle/prng.py point_multiplier(seed, i),
scalar i = prng.scalar(seed, i)) and the expected result is the closed form (sum_i s_i a_i mod q) * G computed by the
C oracle -- cross-checked against the oracle's windowed MSM over the explicit points for N <= 2^6.

    python tests/golden/make_js_fixtures.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle          # noqa: E402
from oracle import params as P       # noqa: E402
from oracle import prng              # noqa: E402

POINT_SEED, SCALAR_SEED = 1000, 2000


def main():
    out = []
    for label in ("ed-on-bls12-377", "pallas", "bls12-377", "bls12-381"):   # the order of msm.test.ts:25-31
        c = P.CURVES[label]
        q = c["order"]
        gen = {"x": c["generator"]["x"], "y": c["generator"]["y"], "isZero": False}
        for n in range(0, 14, 2):
            N = 1 << n
            ps, ss = POINT_SEED + n, SCALAR_SEED + n
            mult = [prng.point_multiplier(ps, i) for i in range(N)]
            scalars = [prng.scalar(ss, i, q) for i in range(N)]
            want = c_oracle.scale(c, sum(s * a for s, a in zip(scalars, mult)) % q, gen)
            if N <= 64:
                pts = [c_oracle.scale(c, a, gen) for a in mult]
                got = c_oracle.msm(c, scalars, pts)
                assert (got["x"], got["y"]) == (want["x"], want["y"]), (label, n)
            first = c_oracle.scale(c, mult[0], gen)
            out.append({"curve": label, "n": n, "pointSeed": ps, "scalarSeed": ss,
                        "firstPoint": {"x": hex(first["x"]), "y": hex(first["y"])}, "firstScalar": hex(scalars[0]),
                        "result": {"x": hex(want["x"]), "y": hex(want["y"]), "isZero": bool(want.get("isZero", False))}})
            print(label, n, "ok")
    path = os.path.join(ROOT, "tests", "golden", "js_msm_fixtures.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_js_fixtures.py", "cases": out}, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
