"""Pins the oracle (oracle/bigint_ref.py + oracle/msm_oracle.c) against every fixed vector and
relation the reference's own tests hold for the MSM path.  CPU only."""
import random

import pytest

from oracle import bigint_ref as B
from oracle import c_oracle
from oracle import params as P

WEIER = [P.BLS12_377, P.PALLAS, P.BLS12_381]


def test_known_answer_bls12_377():
    """scripts/zprize23/submission-test-bls377.ts:6-45"""
    c = P.BLS12_377
    A, Pr = B.AffineWeierstrass(c), B.ProjectiveWeierstrass(c)
    pt = (P.KAT_BLS12_377_POINT["x"], P.KAT_BLS12_377_POINT["y"], False)
    assert A.is_on_curve(pt) and A.is_in_subgroup(pt)                      # :14-15
    r = B.msm(Pr, [2, c["order"] - 1], [Pr.from_affine(pt)] * 2)            # :17-25
    assert Pr.to_affine(r) == pt
    rng = random.Random(1)
    scalars = [rng.randrange(c["order"]) for _ in range(40)]               # :28-45 (1000 in the reference)
    r2 = B.msm(Pr, scalars, [Pr.from_affine(pt)] * 40)
    r3 = B.msm(Pr, [sum(scalars) % c["order"]], [Pr.from_affine(pt)])
    assert Pr.is_equal(r2, r3)


def test_known_answer_ed_on_bls12_377():
    """scripts/zprize23/submission-test.ts:5-20"""
    T = B.TwistedEdwards(P.ED_ON_BLS12_377)
    k = P.KAT_ED377_POINT
    pt = (k["x"], k["y"], 1, k["t"])
    assert T.is_on_curve(pt)
    assert T.to_affine(B.msm(T, [2, T.q - 1], [pt, pt])) == (k["x"], k["y"])


@pytest.mark.parametrize("c", WEIER, ids=lambda c: c["label"])
def test_generators_and_endomorphism(c):
    """generators on curve / in subgroup; lambda*G = (beta*x, y)  (bls12-377.params.ts:49-63)"""
    A = B.AffineWeierstrass(c)
    G = A.one
    assert A.is_on_curve(G) and A.is_in_subgroup(G)
    lam, beta = c["endomorphism"]["lambda_"], c["endomorphism"]["beta"]
    assert pow(lam, 3, c["order"]) == 1 and pow(beta, 3, c["modulus"]) == 1
    assert A.scale(lam, G) == B.endomorphism(G, beta, c["modulus"])


def test_te_generator():
    T = B.TwistedEdwards(P.ED_ON_BLS12_377)
    assert T.is_on_curve(T.one) and T.is_zero(T.scale(T.q, T.one))


@pytest.mark.parametrize("c", WEIER, ids=lambda c: c["label"])
def test_msm_relations(c):
    """bigint/msm.test.ts:36-56 and :62-101 (affine == projective)"""
    A, Pr = B.AffineWeierstrass(c), B.ProjectiveWeierstrass(c)
    q = c["order"]
    rng = random.Random(2)
    G = A.one
    pts = [A.scale(rng.randrange(1, 1 << 64), G) for _ in range(8)]
    scalars = [rng.randrange(q) for _ in range(8)]
    proj = [Pr.from_affine(p) for p in pts]
    r = B.msm(Pr, scalars, proj)
    assert Pr.is_equal(r, B.msm_direct(Pr, scalars, proj))
    # same point => (sum s) * P
    same = B.msm(Pr, scalars, [proj[0]] * 8)
    assert Pr.is_equal(same, Pr.scale(sum(scalars) % q, proj[0]))
    # adding -sum => zero
    z = B.msm(Pr, scalars + [(-sum(scalars)) % q], [proj[0]] * 9)
    assert Pr.to_affine(z)[2]
    # same scalar => s * sum P
    acc = Pr.zero
    for p in proj:
        acc = Pr.add(acc, p)
    assert Pr.is_equal(B.msm(Pr, [scalars[0]] * 8, proj), Pr.scale(scalars[0], acc))


@pytest.mark.parametrize("c", WEIER, ids=lambda c: c["label"])
def test_glv_decompose(c):
    """glv/glv-test.ts:83-125: valid decomposition with ~127-bit halves"""
    q, lam = c["order"], c["endomorphism"]["lambda_"]
    consts = B.glv_constants(q, lam)
    (v00, v01), (v10, v11) = consts["v"]
    assert (v00 + lam * v10) % q == 0 and (v01 + lam * v11) % q == 0
    rng = random.Random(3)
    for s in [0, 1, q - 1, lam] + [rng.randrange(q) for _ in range(2000)]:
        s0, s1 = B.glv_decompose(s, q, lam, consts)
        assert (s0 + s1 * lam - s) % q == 0
        assert abs(s0) < (1 << 128) and abs(s1) < (1 << 128)


def test_signed_digits():
    """msm-batched-affine.ts:180-199: sum of (+-l_k) 2^(ck) reproduces the scalar"""
    rng = random.Random(4)
    for c in (2, 5, 13, 16):
        L = 1 << (c - 1)
        for _ in range(200):
            s = rng.randrange(1 << 127)
            K = -(-(127 + 1) // c)
            d = B.signed_digits(s, c, K)
            assert all(0 <= l <= L for l, _ in d)
            assert sum((-l if neg else l) << (k * c) for k, (l, neg) in enumerate(d)) == s


def _rand_points(params, rng, n):
    if params["kind"] == "weierstrass":
        A = B.AffineWeierstrass(params)
        out = []
        for _ in range(n):
            x, y, z = A.scale(rng.randrange(1, 1 << 64), A.one)
            out.append({"x": x, "y": y, "isZero": z})
        return out
    T = B.TwistedEdwards(params)
    out = []
    for _ in range(n):
        x, y = T.to_affine(T.scale(rng.randrange(1, 1 << 64), T.one))
        out.append({"x": x, "y": y})
    return out


@pytest.mark.parametrize("c", WEIER + [P.ED_ON_BLS12_377], ids=lambda c: c["label"])
def test_c_oracle_equals_python_oracle(c):
    rng = random.Random(5)
    for n in (1, 2, 7, 33):
        pts = _rand_points(c, rng, n)
        scalars = [rng.randrange(c["order"]) for _ in range(n)]
        if c["kind"] == "weierstrass":
            Pr = B.ProjectiveWeierstrass(c)
            x, y, z = Pr.to_affine(B.msm(Pr, scalars, [Pr.from_affine((p["x"], p["y"], False)) for p in pts]))
            want = {"x": x, "y": y, "isZero": z}
        else:
            T = B.TwistedEdwards(c)
            x, y = T.to_affine(B.msm(T, scalars, [T.from_affine((p["x"], p["y"])) for p in pts]))
            want = {"x": x, "y": y, "isZero": False}
        assert c_oracle.msm(c, scalars, pts) == want
        assert c_oracle.msm(c, scalars, pts, threads=2) == want


def test_c_oracle_known_answers_and_edge_cases():
    c = P.BLS12_377
    q = c["order"]
    pt = dict(P.KAT_BLS12_377_POINT, isZero=False)
    assert c_oracle.msm(c, [2, q - 1], [pt, pt]) == pt
    assert c_oracle.msm(c, [1, q - 1], [pt, pt]) == {"x": 0, "y": 1, "isZero": True}
    inf = {"x": 0, "y": 0, "isZero": True}
    assert c_oracle.msm(c, [5, 1], [inf, pt]) == pt
    A = B.AffineWeierstrass(c)
    x, y, z = A.scale(12345678901234567890, (pt["x"], pt["y"], False))
    assert c_oracle.scale(c, 12345678901234567890, pt) == {"x": x, "y": y, "isZero": z}
    k = P.KAT_ED377_POINT
    te = P.ED_ON_BLS12_377
    assert c_oracle.msm(te, [2, te["order"] - 1], [k, k]) == {"x": k["x"], "y": k["y"], "isZero": False}


def test_sharded_c_oracle_equals_plain():
    """oracle_msm_sharded (bench.py's CPU baseline) == oracle_msm: MSM(A u B) = MSM(A) + MSM(B)"""
    from oracle import prng
    for c in (P.BLS12_377, P.ED_ON_BLS12_377):
        q, fb, n = c["order"], c["fe_bytes"], 700
        gen = {"x": c["generator"]["x"], "y": c["generator"]["y"]}
        pts = [c_oracle.scale(c, prng.point_multiplier(5, i), gen) for i in range(n)]
        sb = b"".join(int(prng.scalar(5, i, q)).to_bytes(32, "little") for i in range(n))
        pb = b"".join(int(p["x"]).to_bytes(fb, "little") + int(p["y"]).to_bytes(fb, "little") for p in pts)
        want = c_oracle.msm_bytes(c, sb, pb, n, None, 4)[0]
        for shards in (1, 2, 5, 700, 900):
            assert c_oracle.msm_bytes_sharded(c, sb, pb, n, shards, 4)[0] == want, (c["label"], shards)


@pytest.mark.parametrize("c", [2, 3, 4, 5, 6, 9])
@pytest.mark.parametrize("label", ["bls12-377", "pallas", "ed-on-bls12-377"])
def test_bucket_reduction_2d_equals_running_sum(label, c):
    """The engine's two-dimensional bucket reduction (reduce2d_kernels.h: row / column sums + two half-length weighted
    sums, the factor D applied as doublings) is the same group element as the reference's running sum
    (msm-batched-affine.ts:544-571) -- on random buckets incl. empty ones, for even and odd splits of the index bits."""
    params = P.CURVES[label]
    te = params["kind"] != "weierstrass"
    C = B.TwistedEdwards(params) if te else B.ProjectiveWeierstrass(params)
    rng = random.Random(100 * c + len(label))
    g = C.one if hasattr(C, "one") else None
    if g is None:
        gx, gy = params["generator"]["x"], params["generator"]["y"]
        g = C.from_affine((gx, gy)) if te else C.from_affine((gx, gy, False))
    L = 1 << (c - 1)
    buckets = [C.zero if rng.random() < 0.2 else C.scale(rng.randrange(1, 1 << 32), g) for _ in range(L)]
    want = B.reduce_buckets_running_sum(C, buckets)
    got = B.reduce_buckets_2d(C, buckets, c)
    assert C.is_equal(got, want)
    # and both are sum l * B_l
    direct = C.zero
    for l, bk in enumerate(buckets, start=1):
        direct = C.add(direct, C.scale(l, bk))
    assert C.is_equal(want, direct)
