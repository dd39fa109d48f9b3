"""Committed fixtures under tests/golden/ (CPU part):
  reference_fixtures.json -- the fixed vectors the reference's own files hold for the MSM path; the oracle's
                             parameters must equal them and its arithmetic must satisfy the relations they pin
  msm_vectors.json        -- seeded inputs + expected MSM results; the oracle must reproduce them (regression pin),
                             the HIP path is checked against the same file in test_msm_gpu.py."""
import json
import os

import pytest

from oracle import bigint_ref as B
from oracle import c_oracle
from oracle import params as P
from oracle import prng

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


def test_params_equal_reference_fixtures():
    fx = _load("reference_fixtures.json")
    assert set(fx["curves"]) == set(P.CURVES)
    for label, e in fx["curves"].items():
        c = P.CURVES[label]
        assert int(e["modulus"], 16) == c["modulus"] and int(e["order"], 16) == c["order"]
        assert (int(e["generator"]["x"], 16), int(e["generator"]["y"], 16)) == (c["generator"]["x"], c["generator"]["y"])
        if c["kind"] == "weierstrass":
            assert e["b"] == c["b"]
            assert int(e["lambda"], 16) == c["endomorphism"]["lambda_"] and int(e["beta"], 16) == c["endomorphism"]["beta"]
        else:
            assert e["d"] == c["d"]
    assert {k: int(v, 16) for k, v in fx["known_answers"]["bls12-377"]["point"].items()} == P.KAT_BLS12_377_POINT
    assert {k: int(v, 16) for k, v in fx["known_answers"]["ed-on-bls12-377"]["point"].items()} == P.KAT_ED377_POINT


def test_known_answer_relations_from_fixture():
    """msm([2, q - 1], [P, P]) == P for both fixture points, through the C oracle"""
    fx = _load("reference_fixtures.json")
    for label in ("bls12-377", "ed-on-bls12-377"):
        c = P.CURVES[label]
        k = {n: int(v, 16) for n, v in fx["known_answers"][label]["point"].items()}
        pt = {"x": k["x"], "y": k["y"]}
        got = c_oracle.msm(c, [2, c["order"] - 1], [pt, pt])
        assert (got["x"], got["y"]) == (k["x"], k["y"]) and not got["isZero"]


@pytest.mark.parametrize("i", range(16))
def test_oracle_reproduces_msm_vectors(i):
    v = _load("msm_vectors.json")["vectors"][i]
    c = P.CURVES[v["curve"]]
    n, seed, q = v["n"], v["seed"], c["order"]
    gen = {"x": c["generator"]["x"], "y": c["generator"]["y"]}
    mult = [prng.point_multiplier(seed, j) for j in range(n)]
    scalars = [prng.scalar(seed, j, q) for j in range(n)]
    assert hex(scalars[0]) == v["first_scalar"]
    p0 = c_oracle.scale(c, mult[0], gen)
    assert (hex(p0["x"]), hex(p0["y"])) == (v["first_point"]["x"], v["first_point"]["y"])
    got = c_oracle.scale(c, sum(s * a for s, a in zip(scalars, mult)) % q, gen)   # closed form
    assert (hex(got["x"]), hex(got["y"]), bool(got["isZero"])) == (v["result"]["x"], v["result"]["y"], v["result"]["isZero"])
    if n <= 64:   # and the straight MSM on the expanded points
        pts = [c_oracle.scale(c, a, gen) for a in mult]
        r = c_oracle.msm(c, scalars, pts)
        assert (hex(r["x"]), hex(r["y"])) == (v["result"]["x"], v["result"]["y"])
