"""Parity of the HIP MSM (through the C ABI) against the oracle -- the build's version of the
reference's integration test src/msm.test.ts:24-118 (for ed-on-bls12-377, pallas, bls12-377, bls12-381:
msmUnsafe == msmProjective == bigint msm for N = 2^0 .. 2^12) and of its known-answer smoke tests
(scripts/zprize23/submission-test-bls377.ts, submission-test.ts).  All comparisons are bit-exact on
the canonical affine result."""
import random

import pytest

from oracle import bigint_ref as B
from oracle import c_oracle
from oracle import params as P
from oracle import prng

pytestmark = pytest.mark.gpu

WEIER = ["bls12-377", "pallas", "bls12-381"]


@pytest.fixture(scope="module")
def curves():
    import msm_zprize_amd as m
    m.startThreads()
    cache = {}

    def get(label):
        if label not in cache:
            params = m.curves.BY_LABEL[label]
            cache[label] = (m.Weierstrass if params["kind"] == "weierstrass" else m.TwistedEdwards).create(params)
        return cache[label]

    yield get
    for c in cache.values():
        c.close()


def _strip(p):
    return {"x": p["x"], "y": p["y"], "isZero": bool(p.get("isZero", False))}


def _oracle(label, scalars, pts):
    return c_oracle.msm(P.CURVES[label], scalars, pts)


@pytest.mark.parametrize("label", WEIER + ["ed-on-bls12-377"])
def test_random_inputs_match_their_spec(curves, label):
    """device generators == their documented pure functions of (seed, index); points on the curve"""
    curve = curves(label)
    c = P.CURVES[label]
    n, seed = 300, 12345
    pts = curve.Parallel.randomPointsFast(n, seed)
    got = curve.Affine.toBigints(pts)
    if c["kind"] == "weierstrass":
        A = B.AffineWeierstrass(c)
        gen = {"x": c["generator"]["x"], "y": c["generator"]["y"], "isZero": False}
        for p in got:
            assert A.is_on_curve((p["x"], p["y"], p["isZero"]))
    else:
        T = B.TwistedEdwards(c)
        gen = {"x": c["generator"]["x"], "y": c["generator"]["y"]}
        for p in got:
            assert T.is_on_curve(T.from_affine((p["x"], p["y"])))
    for i in [0, 1, 2, 17, 299]:
        want = c_oracle.scale(c, prng.point_multiplier(seed, i), gen)
        assert _strip(got[i]) == _strip(want), i
    sc = curve.Parallel.randomScalars(n, seed)
    assert curve.Scalar.toBigints(sc) == [prng.scalar(seed, i, c["order"]) for i in range(n)]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 16, 64, 257, 1024, 4096])
@pytest.mark.parametrize("label", WEIER)
def test_weierstrass_msm_vs_oracle(curves, label, n):
    """msm.test.ts:44-82: msmUnsafe (GLV on/off), msm (safe) and msmProjective all equal the bigint MSM"""
    curve = curves(label)
    seed = 1000 + n
    pts = curve.Parallel.randomPointsFast(n, seed)
    sc = curve.Parallel.randomScalars(n, seed)
    want = _oracle(label, curve.Scalar.toBigints(sc), curve.Affine.toBigints(pts))
    for glv in (0, 1):
        for c in ([0] if n > 300 else [0, 2, 5]):
            got = curve.Parallel.msmUnsafe(sc, pts, n, True, {"glv": glv, "c": c})["result"]
            assert got == want, (n, glv, c)
        assert curve.Parallel.msm(sc, pts, n, False, {"glv": glv})["result"] == want
    assert curve.Parallel.msmProjective(sc, pts, n)["result"] == want
    assert curve.Parallel.msmProjective(sc, pts, n, {"c": 7})["result"] == want
    pts.free(); sc.free()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 16, 64, 257, 1024, 4096])
def test_twisted_edwards_msm_vs_oracle(curves, n):
    """msm.test.ts:85-118"""
    curve = curves("ed-on-bls12-377")
    seed = 2000 + n
    pts = curve.Parallel.randomPointsFast(n, seed)
    sc = curve.Parallel.randomScalars(n, seed)
    want = _oracle("ed-on-bls12-377", curve.Scalar.toBigints(sc), curve.Affine.toBigints(pts))
    for c in ([0] if n > 300 else [0, 3, 6]):
        assert _strip(curve.Parallel.msm(sc, pts, n, True, {"c": c})["result"]) == _strip(want), (n, c)
    pts.free(); sc.free()


def test_host_buffer_scalars_and_byte_routes(curves):
    """scalars handed over as a host buffer (msmz_msm) and points through pointsFromBytes (parallel.ts:97-133)"""
    curve = curves("bls12-377")
    n = 500
    pts = curve.Parallel.randomPointsFast(n, 77)
    sc = curve.Parallel.randomScalars(n, 78)
    scalars = curve.Scalar.toBigints(sc)
    points = curve.Affine.toBigints(pts)
    want = _oracle("bls12-377", scalars, points)
    raw = b"".join(s.to_bytes(32, "little") for s in scalars)
    assert curve.Parallel.msmUnsafe(raw, pts, n, False, {"glv": 0})["result"] == want
    up = curve.Parallel.pointsFromBigints(points)
    assert curve.Affine.toBigints(up) == points
    assert curve.Parallel.msmUnsafe(curve.Parallel.scalarsFromBytes(raw), up, n)["result"] == want


def test_known_answer_submission_bls377(curves):
    """scripts/zprize23/submission-test-bls377.ts:6-45: 2P + (q-1)P = P; 1000 x same point."""
    bls377 = curves("bls12-377")
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
    for glv in (0, 1):
        r2 = bls377.Parallel.msm(bls377.Parallel.scalarsFromBigints(scalars), same, n, False, {"glv": glv})["result"]
        one = bls377.Parallel.pointsFromBigints([pt])
        r3 = bls377.Parallel.msm(bls377.Parallel.scalarsFromBigints([sum(scalars) % q]), one, 1)["result"]
        assert r2 == r3
        assert r3 == c_oracle.scale(P.BLS12_377, sum(scalars) % q, pt)
    assert bls377.Parallel.msmProjective(bls377.Parallel.scalarsFromBigints(scalars), same, n)["result"] == r3


def test_known_answer_submission_ed377(curves):
    """scripts/zprize23/submission-test.ts:5-20"""
    curve = curves("ed-on-bls12-377")
    q = P.ED_ON_BLS12_377["order"]
    k = P.KAT_ED377_POINT
    pts = curve.Parallel.pointsFromBigints([k, k])
    sc = curve.Parallel.scalarsFromBigints([2, q - 1])
    r = curve.Parallel.msm(sc, pts, 2)["result"]
    assert (r["x"], r["y"]) == (k["x"], k["y"])


@pytest.mark.parametrize("label", WEIER)
def test_safe_path_edge_cases(curves, label):
    """batchAddNew semantics (curve-affine.ts:412-447): infinity inputs, P + P, P + (-P), zero scalars -- on every
    Weierstrass curve (different limb counts / fe_is_zero margins)"""
    curve = curves(label)
    c = P.CURVES[label]
    q = c["order"]
    g = c["generator"]
    pt = dict(P.KAT_BLS12_377_POINT, isZero=False) if label == "bls12-377" else \
        _strip(c_oracle.scale(c, 0xDEADBEEF12345, {"x": g["x"], "y": g["y"], "isZero": False}))
    neg = {"x": pt["x"], "y": c["modulus"] - pt["y"], "isZero": False}
    inf = {"x": 0, "y": 0, "isZero": True}
    cases = [
        ([pt, neg], [1, 1]),                      # P + (-P) = 0
        ([pt, pt, neg, neg], [5, 7, 5, 7]),       # cancels to 0
        ([inf, pt, inf], [3, 4, 5]),              # infinity inputs are ignored
        ([pt, pt, pt], [0, 0, 0]),                # all-zero scalars
        ([pt] * 7, [1] * 7),                      # pure doubling tree
        ([pt, neg, pt], [q - 1, q - 1, 1]),
    ]
    for points, scalars in cases:
        want = _oracle(label, scalars, points)
        pts = curve.Parallel.pointsFromBigints(points)
        sc = curve.Parallel.scalarsFromBigints(scalars)
        for glv in (0, 1):
            for cc in (0, 3):
                assert curve.Parallel.msm(sc, pts, len(points), False, {"glv": glv, "c": cc})["result"] == want
        assert curve.Parallel.msmProjective(sc, pts, len(points))["result"] == want


def test_golden_msm_vectors(curves):
    """tests/golden/msm_vectors.json: device-generated inputs (same seeds) and every MSM entry point against the
    committed expected results -- no oracle call at run time"""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm_vectors.json")) as f:
        vectors = json.load(f)["vectors"]
    assert len(vectors) == 16
    for v in vectors:
        curve = curves(v["curve"])
        n = v["n"]
        pts = curve.Parallel.randomPointsFast(n, v["seed"])
        sc = curve.Parallel.randomScalars(n, v["seed"])
        p0 = curve.Affine.toBigints(pts)[0]
        assert (hex(p0["x"]), hex(p0["y"])) == (v["first_point"]["x"], v["first_point"]["y"])
        assert hex(curve.Scalar.toBigints(sc)[0]) == v["first_scalar"]
        want = (v["result"]["x"], v["result"]["y"], v["result"]["isZero"])

        def key(r):
            return (hex(r["x"]), hex(r["y"]), bool(r.get("isZero", False)))

        if v["curve"] == "ed-on-bls12-377":
            assert key(curve.Parallel.msm(sc, pts, n)["result"]) == want
        else:
            for glv in (0, 1):
                assert key(curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": glv})["result"]) == want
                assert key(curve.Parallel.msm(sc, pts, n, False, {"glv": glv})["result"]) == want
            assert key(curve.Parallel.msmProjective(sc, pts, n)["result"]) == want
        pts.free(); sc.free()


@pytest.mark.parametrize("label", ["bls12-377", "pallas"])
def test_long_buckets(curves, label):
    """Heavily repeated scalars: a few buckets hold hundreds of (distinct) points, so the pair tree runs many
    rounds, stops short of the longest bucket and the reduction adds the leftover partial sums; the reference's
    counterpart is the same-scalar case of src/bigint/msm.test.ts:47-56."""
    curve = curves(label)
    c = P.CURVES[label]
    q = c["order"]
    rng = random.Random(77)
    n = 700
    pts = curve.Parallel.randomPointsFast(n, 4242)
    points = curve.Affine.toBigints(pts)
    s1, s2 = rng.randrange(q), rng.randrange(q)
    for scalars in ([s1] * n,
                    [s1 if i % 3 else s2 for i in range(n)],
                    [s1 if i < 500 else rng.randrange(q) for i in range(n)],
                    [q - 1] * n):
        want = _oracle(label, scalars, points)
        sc = curve.Parallel.scalarsFromBigints(scalars)
        for glv in (0, 1):
            for cc in (0, 4, 9):
                assert curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": glv, "c": cc})["result"] == want, (glv, cc)
            assert curve.Parallel.msm(sc, pts, n, False, {"glv": glv})["result"] == want
        assert curve.Parallel.msmProjective(sc, pts, n)["result"] == want
        sc.free()
    pts.free()


def test_unsafe_reports_degenerate_batch(curves):
    """msmUnsafe on equal points: the reference traps (inverse.ts:198-199); here a status is returned"""
    import msm_zprize_amd._native as nat
    curve = curves("bls12-377")
    pt = dict(P.KAT_BLS12_377_POINT, isZero=False)
    pts = curve.Parallel.pointsFromBigints([pt, pt])
    sc = curve.Parallel.scalarsFromBigints([1, 1])
    with pytest.raises(nat.MsmzError) as e:
        curve.Parallel.msmUnsafe(sc, pts, 2, False, {"glv": 0})
    assert e.value.status == 5


def test_range_errors(curves):
    import msm_zprize_amd._native as nat
    curve = curves("bls12-377")
    q, p = P.BLS12_377["order"], P.BLS12_377["modulus"]
    with pytest.raises(nat.MsmzError):
        curve.Parallel.scalarsFromBigints([q])
    with pytest.raises(nat.MsmzError):
        curve.Parallel.pointsFromBigints([{"x": p, "y": 1}])


@pytest.mark.parametrize("label", WEIER)
def test_glv_over_a_prefix_of_a_point_set(curves, label):
    """msm(scalars, points, N) with N < allocated (msm-batched-affine.ts:74-97; the warm-up of
    scripts/msm-weierstrass.ts:24 is exactly this): the endomorphism images sit behind the WHOLE set"""
    curve = curves(label)
    n_set = 1500
    pts = curve.Parallel.randomPointsFast(n_set, 909)
    sc = curve.Parallel.randomScalars(n_set, 910)
    scalars, points = curve.Scalar.toBigints(sc), curve.Affine.toBigints(pts)
    for n in (1, 7, 300, 1024, 1499):
        want = _oracle(label, scalars[:n], points[:n])
        for glv in (0, 1):
            assert curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": glv})["result"] == want, (n, glv)
            assert curve.Parallel.msm(sc, pts, n, False, {"glv": glv, "c": 6})["result"] == want, (n, glv)
        assert curve.Parallel.msmProjective(sc, pts, n)["result"] == want
    pts.free(); sc.free()


@pytest.mark.parametrize("label", ["bls12-377", "ed-on-bls12-377"])
def test_multi_engine_context_on_one_gpu(label):
    """msmz_create with n_devices > 1 (startThreads(n), parallel.ts:291-315): here three engines, all on GPU 0 --
    the scheduler, the block split of uploads / generators / prefixes and the host-side fold are the same code that
    drives several GPUs.  Inputs must equal the single-engine ones (generators are functions of the GLOBAL index)
    and every MSM must equal the oracle."""
    import msm_zprize_amd as m
    import msm_zprize_amd._native as nat
    params = m.curves.BY_LABEL[label]
    make = (m.Weierstrass if params["kind"] == "weierstrass" else m.TwistedEdwards).create
    m.startThreads()
    single = make(params)
    m.startThreads(devices=[0, 0, 0])
    multi = make(params)
    m.startThreads()
    assert nat.lib().msmz_ctx_n_devices(multi._ctx) == 3 and nat.lib().msmz_ctx_n_devices(single._ctx) == 1
    n_set = 3 * 65536 + 1000      # every engine holds a full block and the last block is ragged
    seed = 4711
    pts1, sc1 = single.Parallel.randomPointsFast(n_set, seed), single.Parallel.randomScalars(n_set, seed)
    ptsm, scm = multi.Parallel.randomPointsFast(n_set, seed), multi.Parallel.randomScalars(n_set, seed)
    for first, count in ((0, 50), (65530, 12), (2 * 65536 - 3, 65536 + 9), (n_set - 40, 40)):
        assert multi.Affine.toBigints(ptsm, first, count) == single.Affine.toBigints(pts1, first, count)
        assert multi.Scalar.toBigints(scm, first, count) == single.Scalar.toBigints(sc1, first, count)
    variants = [{"glv": 0}, {"glv": 1}] if params["kind"] == "weierstrass" else [{}]
    for n in (n_set, 65536 + 5, 100):            # whole set and prefixes (the small one lives on engine 0 only)
        for opt in variants:
            want = single.Parallel.msm(sc1, pts1, n, False, opt)["result"]
            got = multi.Parallel.msm(scm, ptsm, n, True, opt)
            assert got["result"] == want, (n, opt)
    # oracle check on a size it finishes quickly + host-buffer scalars + byte upload through the split
    n = 70000
    scalars, points = single.Scalar.toBigints(sc1, 0, n), single.Affine.toBigints(pts1, 0, n)
    want = _oracle(label, scalars, points)
    raw = b"".join(s.to_bytes(32, "little") for s in scalars)
    up = multi.Parallel.pointsFromBigints(points)
    assert _strip(multi.Parallel.msm(raw, up, n)["result"]) == _strip(want)
    assert _strip(multi.Parallel.msm(multi.Parallel.scalarsFromBytes(raw), ptsm, n)["result"]) == _strip(want)
    assert _strip(multi.Parallel.msm(scm, ptsm, n)["result"]) == _strip(want)
    multi.close(); single.close()


def test_debug_environment_cannot_change_a_result(monkeypatch):
    """MSMZ_DBG used to reach the kernels; no environment variable may alter a release build's result"""
    import msm_zprize_amd as m
    for k, v in {"MSMZ_DBG": "7", "MSMZ_FUSED": "1", "MSMZ_TAIL_SKIP": "0", "MSMZ_ATOMIC_SORT": "1"}.items():
        monkeypatch.setenv(k, v)
    m.startThreads()
    curve = m.Weierstrass.create(m.curves.bls12377Params)
    n = 2000
    pts, sc = curve.Parallel.randomPointsFast(n, 5), curve.Parallel.randomScalars(n, 6)
    want = _oracle("bls12-377", curve.Scalar.toBigints(sc), curve.Affine.toBigints(pts))
    assert curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": 0})["result"] == want
    curve.close()


def test_window_sizes_that_take_the_fallback_sort(curves):
    """a user-chosen window so large that its coarse bins do not fit the LDS-staged sort (c = 22: 2^21 buckets per
    window, 1024 bins per window): the one-pass atomic sort + the same plan / rounds / reduction must give the same
    result (the reference accepts any c: msm-batched-affine.ts:79-97)"""
    curve = curves("bls12-377")
    n = 3000
    pts = curve.Parallel.randomPointsFast(n, 31337)
    sc = curve.Parallel.randomScalars(n, 31338)
    want = _oracle("bls12-377", curve.Scalar.toBigints(sc), curve.Affine.toBigints(pts))
    for glv, c in ((0, 22), (1, 22), (0, 19)):
        assert curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": glv, "c": c})["result"] == want, (glv, c)
    assert curve.Parallel.msmProjective(sc, pts, n, {"c": 22})["result"] == want
    pts.free(); sc.free()


@pytest.mark.parametrize("label", WEIER)
def test_batched_affine_bucket_reduction(curves, label):
    """the reference's reduceBucketsAffine (msm-batched-affine-single-thread.ts:522-667,
    doc/zprize22.md:317-358) as an option -- first reduction level by lock-step batched-affine running sums; must give
    the same point as the default XYZZ reduction and the oracle for every window size (group sizes 2, 4, 8), with
    empty buckets, repeated points (doublings inside the chain) and the safe / unsafe / GLV variants"""
    curve = curves(label)
    c = P.CURVES[label]
    q = c["order"]
    rng = random.Random(21)
    for n, seed in ((1, 1), (5, 2), (300, 3), (2500, 4)):
        pts = curve.Parallel.randomPointsFast(n, 7000 + seed)
        sc = curve.Parallel.randomScalars(n, 7100 + seed)
        want = _oracle(label, curve.Scalar.toBigints(sc), curve.Affine.toBigints(pts))
        for glv in (0, 1):
            for cc in (0, 2, 3, 4, 6, 9, 12):
                got = curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": glv, "c": cc, "reduceAffine": 1})["result"]
                assert got == want, (n, glv, cc)
            assert curve.Parallel.msm(sc, pts, n, False, {"glv": glv, "reduceAffine": 1})["result"] == want
        pts.free(); sc.free()
    # equal points in neighbouring buckets: the chain R = R + E hits P + P (doubling) and P + (-P)
    g = c["generator"]
    pt = _strip(c_oracle.scale(c, 0xABCDEF, {"x": g["x"], "y": g["y"], "isZero": False}))
    neg = {"x": pt["x"], "y": c["modulus"] - pt["y"], "isZero": False}
    points = [pt, pt, pt, neg, pt]
    scalars = [1, 2, 3, 2, q - 1]
    want = _oracle(label, scalars, points)
    p2 = curve.Parallel.pointsFromBigints(points)
    s2 = curve.Parallel.scalarsFromBigints(scalars)
    for cc in (2, 3, 5):
        assert curve.Parallel.msm(s2, p2, 5, False, {"glv": 0, "c": cc, "reduceAffine": 1})["result"] == want, cc


@pytest.mark.parametrize("label", WEIER)
def test_glv_half_longer_than_assumed_is_redone(label):
    """The engine sizes the GLV windows for halves below 2^127 and redoes the MSM with the analytic bound
    (src/wasm/glv.ts:216-226 `maxBits`; tools/gen_constants.py glv_proven_bits) when the slicing kernel flags a
    longer half.  No real scalar is known to take that path, so the test hook shrinks the ASSUMED length: ordinary
    halves overflow, the flag is raised, and the redone MSM must still equal the oracle."""
    import msm_zprize_amd as m
    from msm_zprize_amd._native import lib
    m.startThreads()
    curve = m.Weierstrass.create(m.curves.BY_LABEL[label])
    try:
        n = 300
        pts = curve.Parallel.randomPointsFast(n, 91)
        sc = curve.Parallel.randomScalars(n, 92)
        want = _oracle(label, curve.Scalar.toBigints(sc), curve.Affine.toBigints(pts))
        assert lib().msmz_test_retries(curve._ctx) == 0
        assert curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": 1})["result"] == want
        assert lib().msmz_test_retries(curve._ctx) == 0            # the default bound holds for ordinary scalars
        for bits, c in ((100, 0), (64, 7), (110, 13)):   # K * c stays below the halves' ~126 bits
            assert lib().msmz_test_set_glv_bits(curve._ctx, bits) == 0
            before = lib().msmz_test_retries(curve._ctx)
            assert curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": 1, "c": c})["result"] == want, (bits, c)
            assert curve.Parallel.msm(sc, pts, n, False, {"glv": 1, "c": c})["result"] == want, (bits, c)
            assert lib().msmz_test_retries(curve._ctx) == before + 2, (bits, c)   # both MSMs took the redo path
        assert lib().msmz_test_set_glv_bits(curve._ctx, 0) == 0
        before = lib().msmz_test_retries(curve._ctx)
        assert curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": 1})["result"] == want
        assert lib().msmz_test_retries(curve._ctx) == before
        assert lib().msmz_test_set_glv_bits(curve._ctx, 4) == 1 and lib().msmz_test_set_glv_bits(curve._ctx, 128) == 1
    finally:
        curve.close()


def test_download_ranges_cannot_wrap(curves):
    """first + count is checked without 64-bit wrap-around (a C caller could pass first = 2^64 - 1, count = 2)"""
    import ctypes as C
    from msm_zprize_amd._native import lib
    curve = curves("bls12-377")
    pts = curve.Parallel.randomPointsFast(8, 5)
    sc = curve.Parallel.randomScalars(8, 5)
    buf = C.create_string_buffer(96 * 8)
    big = (1 << 64) - 1
    assert lib().msmz_download_points(curve._ctx, pts.handle, big, 2, buf, None) == 1
    assert lib().msmz_download_points(curve._ctx, pts.handle, 2, big, buf, None) == 1
    assert lib().msmz_download_points(curve._ctx, pts.handle, 17, 0, buf, None) == 1    # beyond the images too
    assert lib().msmz_download_points(curve._ctx, pts.handle, 8, 8, buf, None) == 0     # the endomorphism images
    assert lib().msmz_download_scalars(curve._ctx, sc.handle, big, 2, buf) == 1
    assert lib().msmz_download_scalars(curve._ctx, sc.handle, 7, 2, buf) == 1
    assert lib().msmz_download_scalars(curve._ctx, sc.handle, 7, 1, buf) == 0
    pts.free(); sc.free()


@pytest.mark.parametrize("label,glv,c", [("bls12-377", 0, 14), ("bls12-377", 0, 18), ("bls12-377", 0, 11), ("bls12-377", 1, 9),
                                         ("pallas", 0, 17), ("pallas", 0, 15), ("bls12-381", 0, 16)])
def test_thin_top_window_is_folded(curves, label, glv, c):
    """A user-chosen window size whose TOP window has only 1-2 significant bits: the engine folds that window into its
    own bucket set (copies of the digit's small range instead of a handful of long buckets; the two-dimensional
    reduction's column sums are the per-digit sums).  Same result as the oracle, safe and unsafe, and far fewer tree
    rounds than the longest bucket of the unfolded layout would need."""
    curve = curves(label)
    for n, seed in ((300, 31), (4096, 32)):
        pts = curve.Parallel.randomPointsFast(n, seed)
        sc = curve.Parallel.randomScalars(n, seed + 100)
        want = _oracle(label, curve.Scalar.toBigints(sc), curve.Affine.toBigints(pts))
        out = curve.Parallel.msmUnsafe(sc, pts, n, True, {"glv": glv, "c": c})
        assert out["result"] == want, (n, "unsafe")
        assert curve.Parallel.msm(sc, pts, n, False, {"glv": glv, "c": c})["result"] == want, (n, "safe")
        if n == 4096:
            # unfolded, the top window's <= 4 digit values x <= 8 sub-windows would hold >= n / 32 entries per bucket
            assert out["stats"].max_bucket < n // 32, out["stats"].max_bucket
        pts.free(); sc.free()


@pytest.mark.parametrize("label,c", [("bls12-377", 17), ("bls12-377", 16), ("pallas", 16), ("bls12-381", 0)])
def test_tile_boundaries_and_extreme_scalars(curves, label, c):
    """Sizes around the bucket sort's tile of 2048 scalars (one scalar short of a tile, a full tile, one scalar into the
    next, two tiles and one) at the window sizes whose sort kernels are specialized (16 / 17), with scalars that sit at
    the edges of the digit range: 0, 1, q - 1 (largest top-window digit), 2^k - 1 (all-ones windows: every digit carries),
    2^k (one non-zero window) -- among random ones.  msmUnsafe with GLV on / off and msm (safe) against the oracle."""
    curve = curves(label)
    q = P.CURVES[label]["order"]
    rng = random.Random(77)
    for n in (2047, 2048, 2049, 4097):
        pts = curve.Parallel.randomPointsFast(n, 500 + n)
        points = curve.Affine.toBigints(pts)
        edge = [0, 1, q - 1, q - 2, (1 << 200) - 1, 1 << 200, (1 << 17) - 1, 1 << 17, (1 << 16), (1 << 252) % q]
        scalars = [rng.randrange(q) for _ in range(n)]
        for i, e in enumerate(edge):
            scalars[(i * 211) % n] = e
            scalars[n - 1 - i] = e
        sc = curve.Parallel.scalarsFromBigints(scalars)
        want = _oracle(label, scalars, points)
        for glv in (0, 1):
            assert curve.Parallel.msmUnsafe(sc, pts, n, True, {"glv": glv, "c": c})["result"] == want, (n, glv)
        assert curve.Parallel.msm(sc, pts, n, False, {"glv": 0, "c": c})["result"] == want, (n, "safe")
        pts.free(); sc.free()
