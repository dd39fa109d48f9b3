"""Writes tests/golden/reference_fixtures.json: the fixed vectors the reference's own files hold for the MSM path
(SURVEY.md section 8c), as data -- moduli, group orders, generators, endomorphism constants and the two known-answer
points of its ZPrize smoke tests.  The numbers are read from oracle/params.py (which cites the reference line each
was taken from); when /root/reference is present (this container, not the GPU box) every literal is also looked up
as text in the cited reference file, so a typo in the restatement cannot survive.

    python tests/golden/make_reference_fixtures.py
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import params as P   # noqa: E402

REF = "/root/reference"
SOURCES = {
    "bls12-377": "src/concrete/bls12-377.params.ts",
    "pallas": "src/concrete/pasta.params.ts",
    "bls12-381": "src/concrete/bls12-381.params.ts",
    "ed-on-bls12-377": "src/concrete/ed-on-bls12-377.params.ts",
}


def literals(path):
    """all integer literals (decimal or hex, with optional n suffix / underscores) of a reference file, as ints"""
    with open(os.path.join(REF, path)) as f:
        text = f.read()
    out = set()
    for m in re.finditer(r"0x[0-9a-fA-F_]+|\b[0-9][0-9_]{5,}", text):
        t = m.group(0).replace("_", "")
        out.add(int(t, 16) if t.startswith("0x") else int(t))
    return out


def main():
    fx = {"note": "data values held by the reference's own parameter and test files; see make_reference_fixtures.py",
          "curves": {}, "known_answers": {}}
    have_ref = os.path.isdir(REF)
    for label, c in P.CURVES.items():
        e = {"source": SOURCES[label], "modulus": hex(c["modulus"]), "order": hex(c["order"]),
             "generator": {"x": hex(c["generator"]["x"]), "y": hex(c["generator"]["y"])}}
        check = [c["modulus"], c["order"], c["generator"]["x"], c["generator"]["y"]]
        if c["kind"] == "weierstrass":
            e["b"] = c["b"]
            e["lambda"] = hex(c["endomorphism"]["lambda_"])
            e["beta"] = hex(c["endomorphism"]["beta"])
            if label == "bls12-377":
                check += [c["endomorphism"]["lambda_"], c["endomorphism"]["beta"]]
            elif label == "bls12-381":   # lambda = z^2 - 1 is derived there (bls12-381.params.ts:16-24)
                check += [c["endomorphism"]["beta"], 0xD201000000010000]
                assert c["endomorphism"]["lambda_"] == 0xD201000000010000 ** 2 - 1
            # pallas derives lambda / beta by formula (pasta.params.ts:19-32)
        else:
            e["a"] = -1   # "-x^2 + y^2 = 1 + d x^2 y^2" (ed-on-bls12-377.params.ts:8)
            e["d"] = c["d"]
            if have_ref:
                with open(os.path.join(REF, SOURCES[label])) as f:
                    assert ("const d = %dn" % c["d"]) in f.read()
        if have_ref:
            lits = literals(SOURCES[label])
            missing = [hex(v) for v in check if v not in lits and v > 1]
            assert not missing, (label, missing)
        fx["curves"][label] = e
    fx["known_answers"]["bls12-377"] = {"source": "scripts/zprize23/submission-test-bls377.ts:6-25",
                                        "point": {k: hex(v) for k, v in P.KAT_BLS12_377_POINT.items()},
                                        "relation": "msm([2, q - 1], [P, P]) == P"}
    fx["known_answers"]["ed-on-bls12-377"] = {"source": "scripts/zprize23/submission-test.ts:5-20",
                                              "point": {k: hex(v) for k, v in P.KAT_ED377_POINT.items()},
                                              "relation": "msm([2, q - 1], [P, P]) == P"}
    if have_ref:
        l1 = literals("scripts/zprize23/submission-test-bls377.ts")
        assert all(v in l1 for v in P.KAT_BLS12_377_POINT.values())
        l2 = literals("scripts/zprize23/submission-test.ts")
        assert all(v in l2 for v in P.KAT_ED377_POINT.values())
    path = os.path.join(ROOT, "tests", "golden", "reference_fixtures.json")
    with open(path, "w") as f:
        json.dump(fx, f, indent=1)
    print("wrote", path, "(literals checked against the reference text)" if have_ref else "(reference not present)")


if __name__ == "__main__":
    main()
