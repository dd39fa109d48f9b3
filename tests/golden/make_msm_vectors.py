"""Writes tests/golden/msm_vectors.json: seeded synthetic MSM inputs and the results the ORACLE gives for them.

Inputs are defined by (curve, seed, n) through the seeded generators restated in oracle/prng.py
(point i = splitmix64(seed, i) * G, scalar i = rejection-sampled 32 bytes); the expected result is computed
twice -- by the C restatement of src/bigint/msm.ts (oracle/msm_oracle.c) and by the closed form
(sum_i s_i a_i mod q) * G through the Python-integer oracle (oracle/bigint_ref.py) -- and written only when both
agree.  Nothing here imports or runs the reference (it cannot be built in this image); the
reference's own fixed vectors are in reference_fixtures.json.

    python tests/golden/make_msm_vectors.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bigint_ref as B   # noqa: E402
from oracle import c_oracle          # noqa: E402
from oracle import params as P       # noqa: E402
from oracle import prng              # noqa: E402

CASES = [(label, seed, n) for label in ("bls12-377", "pallas", "bls12-381", "ed-on-bls12-377")
         for seed, n in ((11, 1), (12, 7), (13, 64), (14, 300))]


def main():
    out = []
    for label, seed, n in CASES:
        c = P.CURVES[label]
        q = c["order"]
        gen = {"x": c["generator"]["x"], "y": c["generator"]["y"]}
        mult = [prng.point_multiplier(seed, i) for i in range(n)]
        scalars = [prng.scalar(seed, i, q) for i in range(n)]
        points = [c_oracle.scale(c, a, gen) for a in mult]
        got = c_oracle.msm(c, scalars, points)
        closed = c_oracle.scale(c, sum(s * a for s, a in zip(scalars, mult)) % q, gen)
        if c["kind"] == "weierstrass":
            A = B.AffineWeierstrass(c)
            k = sum(s * a for s, a in zip(scalars, mult)) % q
            py = A.scale(k, (gen["x"], gen["y"], False)) if hasattr(A, "scale") else None
            if py is not None:
                assert (py[0], py[1]) == (closed["x"], closed["y"]), label
        assert (got["x"], got["y"], bool(got["isZero"])) == (closed["x"], closed["y"], bool(closed["isZero"])), (label, n)
        out.append({"curve": label, "seed": seed, "n": n,
                    "first_point": {"x": hex(points[0]["x"]), "y": hex(points[0]["y"])},
                    "first_scalar": hex(scalars[0]),
                    "result": {"x": hex(got["x"]), "y": hex(got["y"]), "isZero": bool(got["isZero"])}})
        print(label, seed, n, "ok")
    path = os.path.join(ROOT, "tests", "golden", "msm_vectors.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_msm_vectors.py", "vectors": out}, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
