"""Parity of the HIP MSM (through the C ABI) against the oracle -- the build's version of the
reference's integration test src/msm.test.ts:65-82 (msmUnsafe == bigint msm for N = 2^0..2^12) and
of its known-answer smoke tests (scripts/zprize23/submission-test-bls377.ts)."""
import random

import pytest

from oracle import bigint_ref as B
from oracle import params as P
from oracle import prng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bls377():
    import msm_zprize_amd as m
    m.startThreads()
    curve = m.Weierstrass.create(m.curves.bls12377Params)
    yield curve
    curve.close()


def _oracle_msm(params, scalars, pts):
    Pr = B.ProjectiveWeierstrass(params)
    r = B.msm(Pr, scalars, [Pr.from_affine((p["x"], p["y"], p["isZero"])) for p in pts])
    x, y, z = Pr.to_affine(r)
    return {"x": x, "y": y, "isZero": z}


def test_random_inputs_match_their_spec(bls377):
    """device generators == their documented pure functions of (seed, index)"""
    c = P.BLS12_377
    A = B.AffineWeierstrass(c)
    n, seed = 300, 12345
    pts = bls377.Parallel.randomPointsFast(n, seed)
    got = bls377.Affine.toBigints(pts)
    for i in [0, 1, 2, 17, 299]:
        a = prng.point_multiplier(seed, i)
        x, y, z = A.scale(a, A.one)
        assert got[i] == {"x": x, "y": y, "isZero": z}, i
    for p in got:
        assert A.is_on_curve((p["x"], p["y"], p["isZero"]))
    sc = bls377.Parallel.randomScalars(n, seed)
    assert bls377.Scalar.toBigints(sc) == [prng.scalar(seed, i, c["order"]) for i in range(n)]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 16, 64, 257, 1024, 4096])
@pytest.mark.parametrize("glv", [0, 1])
def test_msm_unsafe_vs_oracle(bls377, n, glv):
    seed = 1000 + n
    pts = bls377.Parallel.randomPointsFast(n, seed)
    sc = bls377.Parallel.randomScalars(n, seed)
    want = _oracle_msm(P.BLS12_377, bls377.Scalar.toBigints(sc), bls377.Affine.toBigints(pts))
    for c in ([0] if n > 300 else [0, 2, 5]):
        got = bls377.Parallel.msmUnsafe(sc, pts, n, True, {"glv": glv, "c": c})["result"]
        assert got == want, (n, glv, c)
    got = bls377.Parallel.msm(sc, pts, n, False, {"glv": glv})["result"]
    assert got == want
    pts.free(); sc.free()


def test_known_answer_submission_bls377(bls377):
    """scripts/zprize23/submission-test-bls377.ts:6-45: 2P + (q-1)P = P; 1000 x same point."""
    q = P.BLS12_377["order"]
    pt = dict(P.KAT_BLS12_377_POINT, isZero=False)
    pts = bls377.Parallel.pointsFromBigints([pt, pt])
    sc = bls377.Parallel.scalarsFromBigints([2, q - 1])
    for glv in (0, 1):
        assert bls377.Parallel.msm(sc, pts, 2, False, {"glv": glv})["result"] == pt
    rng = random.Random(5)
    n = 1000
    scalars = [rng.randrange(q) for _ in range(n)]
    same = bls377.Parallel.pointsFromBigints([pt] * n)
    r2 = bls377.Parallel.msm(bls377.Parallel.scalarsFromBigints(scalars), same, n)["result"]
    one = bls377.Parallel.pointsFromBigints([pt])
    r3 = bls377.Parallel.msm(bls377.Parallel.scalarsFromBigints([sum(scalars) % q]), one, 1)["result"]
    assert r2 == r3
    A = B.AffineWeierstrass(P.BLS12_377)
    assert (r3["x"], r3["y"], r3["isZero"]) == A.scale(sum(scalars) % q, (pt["x"], pt["y"], False))
