"""Stage-level parity on the device: every routine of the hot path run in isolation through the test hooks of the
C ABI (include/msmz_test.h) and compared with the oracle -- the build's version of the reference's per-operation
checks wasm == bigint (src/field.test.ts:159-211, src/curve-projective.test.ts:77-209, src/glv/glv-test.ts:83-125,
src/testing/equivalent-wasm.ts:97-147).  All comparisons are bit-exact."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import bigint_ref as B
from oracle import params as P

pytestmark = pytest.mark.gpu

CURVES = ["bls12-377", "pallas", "bls12-381", "ed-on-bls12-377"]
# Montgomery radix of the device representation: N limbs of W bits (csrc/constants_gen.h)
RADIX_BITS = {"bls12-377": 14 * 28, "bls12-381": 14 * 28, "pallas": 9 * 29, "ed-on-bls12-377": 9 * 29}
(TF_MUL, TF_SQR, TF_ADD, TF_SUB, TF_INVERSE, TF_INVERSE_WAVE, TF_ROUNDTRIP, TF_IS_ZERO, TF_SLOT_ROUNDTRIP) = range(9)
TP_ADD, TP_ADD_X4, TP_DBL, TP_DBL_X4 = 0, 1, 3, 4


@pytest.fixture(scope="module")
def ctxs():
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


def _lib():
    from msm_zprize_amd import _native
    return _native.lib()


def _field(curve, op, a, b):
    fb = curve.fe_bytes
    n = len(a)
    ab = b"".join(int(x).to_bytes(fb, "little") for x in a)
    bb = b"".join(int(x).to_bytes(fb, "little") for x in b)
    out = C.create_string_buffer(fb * n)
    st = _lib().msmz_test_field(curve._ctx, op, ab, bb, n, out)
    assert st == 0, st
    return [int.from_bytes(out.raw[fb * i:fb * (i + 1)], "little") for i in range(n)]


def _field_inputs(p, fb, rng, n):
    """canonical values, the edge values of field.test.ts and LAZY residues in [p, 4p) (the device never assumes < p)"""
    top = min(4 * p, 1 << (8 * fb))
    vals = [0, 1, 2, p - 1, p - 2, p, p + 1, 2 * p - 1, 2 * p, 2 * p + 1, 3 * p - 1, 3 * p + 7, top - 1, (p + 1) // 2]
    vals += [rng.randrange(p) for _ in range(n - len(vals) - 20)]
    vals += [rng.randrange(p, top) for _ in range(20)]
    return vals


@pytest.mark.parametrize("label", CURVES)
def test_field_operations(ctxs, label):
    """fe_mul / fe_sqr / lazy add / sub / equality / memory + slot formats (field.test.ts:40-157)"""
    curve = ctxs(label)
    p = P.CURVES[label]["modulus"]
    rng = random.Random(hash(label) & 0xffff)
    n = 600
    a = _field_inputs(p, curve.fe_bytes, rng, n)
    b = list(a)
    rng.shuffle(b)
    Rinv = B.inverse(pow(2, RADIX_BITS[label], p), p)
    assert _field(curve, TF_MUL, a, b) == [x * y * Rinv % p for x, y in zip(a, b)]
    assert _field(curve, TF_SQR, a, b) == [x * x * Rinv % p for x in a]
    assert _field(curve, TF_ADD, a, b) == [(x + y) % p for x, y in zip(a, b)]
    assert _field(curve, TF_SUB, a, b) == [(x - y) % p for x, y in zip(a, b)]
    assert _field(curve, TF_ROUNDTRIP, a, b) == [x % p for x in a]
    assert _field(curve, TF_SLOT_ROUNDTRIP, a, b) == [(x + y) % p for x, y in zip(a, b)]
    # equality of lazy values: pairs that agree mod p but differ as integers must compare equal
    c = [(x % p) + rng.choice([0, p, 2 * p]) for x in a]
    c = [v if v < min(4 * p, 1 << (8 * curve.fe_bytes)) else v - p for v in c]
    assert _field(curve, TF_IS_ZERO, a, c) == [1] * n
    assert _field(curve, TF_IS_ZERO, a, b) == [1 if (x - y) % p == 0 else 0 for x, y in zip(a, b)]


@pytest.mark.parametrize("label", CURVES)
def test_field_inversion(ctxs, label):
    """fe_inverse (per lane) == fe_inverse_wave (the wave-wide form k_batch_add uses) == bigint inverse, incl. 0, 1,
    p - 1, lazy inputs (field.test.ts:159-211 batchInverse / inverse)"""
    curve = ctxs(label)
    p = P.CURVES[label]["modulus"]
    rng = random.Random(99)
    a = _field_inputs(p, curve.fe_bytes, rng, 200)
    R2 = pow(2, 2 * RADIX_BITS[label], p)
    want = [0 if x % p == 0 else R2 * B.inverse(x % p, p) % p for x in a]
    assert _field(curve, TF_INVERSE, a, a) == want
    assert _field(curve, TF_INVERSE_WAVE, a, a) == want


@pytest.mark.parametrize("label", ["bls12-377", "pallas", "bls12-381"])
def test_glv_decomposition(ctxs, label):
    """glv-test.ts:102-125: s0 + s1 * lambda = s (mod q) and both halves below 2^127 -- on random scalars and on the
    corners 0, 1, q - 1, lambda, q - lambda, powers of two"""
    curve = ctxs(label)
    c = P.CURVES[label]
    q, lam = c["order"], c["endomorphism"]["lambda_"]
    rng = random.Random(7)
    scalars = [0, 1, 2, q - 1, q - 2, lam, q - lam, (lam + 1) % q, q // 2, q // 3] + [1 << k for k in range(0, 250, 13)]
    scalars += [rng.randrange(q) for _ in range(3000)]
    n = len(scalars)
    raw = b"".join(s.to_bytes(32, "little") for s in scalars)
    s0b, s1b, neg = C.create_string_buffer(16 * n), C.create_string_buffer(16 * n), C.create_string_buffer(2 * n)
    assert _lib().msmz_test_glv(curve._ctx, raw, n, s0b, s1b, neg) == 0
    worst = 0
    for i, s in enumerate(scalars):
        s0 = int.from_bytes(s0b.raw[16 * i:16 * i + 16], "little") * (-1 if neg.raw[2 * i] else 1)
        s1 = int.from_bytes(s1b.raw[16 * i:16 * i + 16], "little") * (-1 if neg.raw[2 * i + 1] else 1)
        assert (s0 + s1 * lam - s) % q == 0, i
        worst = max(worst, abs(s0).bit_length(), abs(s1).bit_length())
    assert worst <= 127


@pytest.mark.parametrize("glv", [0, 1])
@pytest.mark.parametrize("label", ["bls12-377", "pallas", "ed-on-bls12-377"])
def test_signed_digits(ctxs, label, glv):
    """the device's window slicing == signed_digits of the oracle (msm-batched-affine.ts:180-199) for several c,
    incl. scalars with long runs of ones (carries ripple through every window)"""
    curve = ctxs(label)
    cv = P.CURVES[label]
    if glv and "endomorphism" not in cv:
        pytest.skip("no endomorphism")
    q = cv["order"]
    rng = random.Random(3)
    scalars = [0, 1, q - 1, (1 << 200) - 1, ((1 << 253) - 1) % q, q // 2] + [rng.randrange(q) for _ in range(500)]
    n = len(scalars)
    raw = b"".join(s.to_bytes(32, "little") for s in scalars)
    s0b, s1b, neg = C.create_string_buffer(16 * n), C.create_string_buffer(16 * n), C.create_string_buffer(2 * n)
    if glv:
        assert _lib().msmz_test_glv(curve._ctx, raw, n, s0b, s1b, neg) == 0
    bits = 128 if glv else (q - 1).bit_length()
    for c in (2, 7, 13, 16, 17, 21):
        K = -(-(bits + 1) // c)
        digits = np.zeros((2 if glv else 1) * n * K, dtype=np.uint32)
        assert _lib().msmz_test_digits(curve._ctx, raw, n, c, K, glv, digits.ctypes.data_as(C.c_void_p)) == 0
        digits = digits.reshape((2 if glv else 1), n, K)
        for i in list(range(8)) + [rng.randrange(n) for _ in range(40)]:
            halves = [scalars[i]]
            negs = [0]
            if glv:
                halves = [int.from_bytes(s0b.raw[16 * i:16 * i + 16], "little"), int.from_bytes(s1b.raw[16 * i:16 * i + 16], "little")]
                negs = [neg.raw[2 * i], neg.raw[2 * i + 1]]
            for h, (sv, sg) in enumerate(zip(halves, negs)):
                want = B.signed_digits(sv, c, K)
                got = [(int(d) & 0x7fffffff, int(d) >> 31) for d in digits[h, i]]
                assert [g[0] for g in got] == [w[0] for w in want], (label, c, i, h)
                assert [g[1] for g in got] == [(w[1] ^ sg) if w[0] else 0 for w in want], (label, c, i, h)
                # value check: sum of signed digits * 2^(ck) reproduces the half scalar
                assert sum((-l if ng else l) << (c * k) for k, (l, ng) in enumerate(want)) == sv


def _sort(curve, scalars, c, glv, fallback=0):
    n = len(scalars)
    raw = b"".join(s.to_bytes(32, "little") for s in scalars)
    geom = (C.c_uint32 * 8)()
    assert _lib().msmz_test_sort(curve._ctx, raw, n, c, glv, fallback, geom, None, 0, None, 0) == 0
    cc, K, Keff, L, nb, E, maxb, spread = list(geom)
    off = np.zeros(nb + 1, dtype=np.uint32)
    refs = np.zeros(max(E, 1), dtype=np.uint32)
    assert _lib().msmz_test_sort(curve._ctx, raw, n, c, glv, fallback, geom, off.ctypes.data_as(C.c_void_p), nb + 1,
                                 refs.ctypes.data_as(C.c_void_p), max(E, 1)) == 0
    return dict(c=cc, K=K, Keff=Keff, L=L, nb=nb, E=E, maxb=maxb, spread=spread), off, refs[:E]


@pytest.mark.parametrize("fallback", [0, 1])
@pytest.mark.parametrize("label,glv", [("bls12-377", 0), ("bls12-377", 1), ("pallas", 1), ("ed-on-bls12-377", 0)])
def test_bucket_sort_membership(ctxs, label, glv, fallback):
    """sortPoints (msm-batched-affine.ts:444-490): after the sort, bucket (window k, digit l) holds exactly the
    (index, sign) pairs whose k-th signed digit is l -- for the LDS-staged two-level sort and for the one-pass atomic
    fallback; also offsets, entry count and the largest bucket"""
    curve = ctxs(label)
    cv = P.CURVES[label]
    q = cv["order"]
    rng = random.Random(11)
    n = 3000
    scalars = [rng.randrange(q) for _ in range(n - 300)] + [rng.choice([5, q - 5, 1 << 77]) for _ in range(300)]
    lam = cv.get("endomorphism", {}).get("lambda_")
    for c in (5, 11, 16, 17):   # 16 / 17: the sort kernels specialized for the default window sizes (unrolled window loop)
        g, off, refs = _sort(curve, scalars, c, glv, fallback)
        K, L, Keff = g["K"], g["L"], g["Keff"]
        assert g["c"] == c and L == 1 << (c - 1) and g["nb"] == Keff * L and off[0] == 0 and off[-1] == g["E"]
        # expected bucket contents from the oracle
        halves = []
        for i, s in enumerate(scalars):
            if glv:
                s0, s1 = B.glv_decompose(s, q, lam)
                # any valid decomposition is fine: take the DEVICE's halves for the digit comparison
                halves.append(None)
            else:
                halves.append([(i, s, 0)])
        if glv:
            raw = b"".join(s.to_bytes(32, "little") for s in scalars)
            s0b, s1b, neg = C.create_string_buffer(16 * n), C.create_string_buffer(16 * n), C.create_string_buffer(2 * n)
            assert _lib().msmz_test_glv(curve._ctx, raw, n, s0b, s1b, neg) == 0
            halves = [[(i, int.from_bytes(s0b.raw[16 * i:16 * i + 16], "little"), neg.raw[2 * i]),
                       (n + i, int.from_bytes(s1b.raw[16 * i:16 * i + 16], "little"), neg.raw[2 * i + 1])] for i in range(n)]
        want = {}
        smask = (1 << g["spread"]) - 1
        for hs in halves:
            for entry, sv, sg in hs:
                for k, (l, ng) in enumerate(B.signed_digits(sv, c, K)):
                    if l == 0:
                        continue
                    kw = k + (entry & smask) if k == K - 1 else k
                    want.setdefault(kw * L + l - 1, []).append((entry, ng ^ sg))
        assert sum(len(v) for v in want.values()) == g["E"]
        assert max(len(v) for v in want.values()) == g["maxb"]
        for b in list(want.keys())[:400] + [0, g["nb"] - 1]:
            got = sorted((int(r) & 0x7fffffff, int(r) >> 31) for r in refs[off[b]:off[b + 1]])
            assert got == sorted(want.get(b, [])), (label, c, b)
        sizes = np.diff(off.astype(np.int64))
        assert int(sizes.sum()) == g["E"] and int((sizes > 0).sum()) == len(want)


@pytest.mark.parametrize("label", ["bls12-377", "pallas", "bls12-381"])
def test_xyzz_point_arithmetic(ctxs, label):
    """XYZZ add / 4-lane add / double against the affine oracle, with the edge cases of curve-projective.test.ts:
    P + 0, 0 + P, 0 + 0, P + P, P + (-P)"""
    curve = ctxs(label)
    cv = P.CURVES[label]
    A = B.AffineWeierstrass(cv)
    p = cv["modulus"]
    rng = random.Random(5)
    gen = (cv["generator"]["x"], cv["generator"]["y"], False)
    pts = [A.scale(rng.randrange(1, 1 << 64), gen) for _ in range(40)]
    a = pts + [pts[0], pts[1], pts[2], pts[3], pts[4]]
    b = pts[1:] + pts[:1] + [pts[0], A.negate(pts[1]), pts[2], pts[3], pts[4]]
    a_inf = [0] * 40 + [0, 0, 1, 0, 1]
    b_inf = [0] * 40 + [0, 0, 0, 1, 1]
    fb = curve.fe_bytes
    enc = lambda ps: b"".join(int(x).to_bytes(fb, "little") + int(y).to_bytes(fb, "little") for x, y, _ in ps)
    n = len(a)
    out = C.create_string_buffer(2 * fb * n)

    def run(op):
        assert _lib().msmz_test_point(curve._ctx, op, enc(a), bytes(a_inf), enc(b), bytes(b_inf), n, out) == 0
        res = []
        for i in range(n):
            x = int.from_bytes(out.raw[2 * fb * i:2 * fb * i + fb], "little")
            y = int.from_bytes(out.raw[2 * fb * i + fb:2 * fb * (i + 1)], "little")
            res.append((0, 0, True) if x == 0 and y == 0 else (x, y, False))
        return res

    zero = (0, 0, True)
    ops_a = [zero if fa else pa for pa, fa in zip(a, a_inf)]
    ops_b = [zero if fbb else pb for pb, fbb in zip(b, b_inf)]
    norm = lambda Q: zero if Q[2] else (Q[0] % p, Q[1] % p, False)
    want_add = [norm(A.add(x, y)) for x, y in zip(ops_a, ops_b)]
    assert run(TP_ADD) == want_add
    assert run(TP_ADD_X4) == want_add
    assert run(TP_DBL) == [norm(A.double(x)) for x in ops_a]
    assert run(TP_DBL_X4) == [norm(A.double(w)) for w in want_add]


def test_twisted_edwards_point_arithmetic(ctxs):
    """extended twisted-Edwards add / 4-lane add / double (curve-twisted-edwards.ts:84-165) incl. P + P and P + 0"""
    curve = ctxs("ed-on-bls12-377")
    cv = P.CURVES["ed-on-bls12-377"]
    T = B.TwistedEdwards(cv)
    rng = random.Random(6)
    g = T.from_affine((cv["generator"]["x"], cv["generator"]["y"]))
    pts = [T.to_affine(T.scale(rng.randrange(1, 1 << 64), g)) for _ in range(30)]
    ident = (0, 1)
    a = pts + [pts[0], pts[1], ident]
    b = pts[1:] + pts[:1] + [pts[0], ident, ident]
    fb = curve.fe_bytes
    enc = lambda ps: b"".join(int(x).to_bytes(fb, "little") + int(y).to_bytes(fb, "little") for x, y in ps)
    n = len(a)
    out = C.create_string_buffer(2 * fb * n)

    def run(op):
        assert _lib().msmz_test_point(curve._ctx, op, enc(a), None, enc(b), None, n, out) == 0
        return [(int.from_bytes(out.raw[2 * fb * i:2 * fb * i + fb], "little"),
                 int.from_bytes(out.raw[2 * fb * i + fb:2 * fb * (i + 1)], "little")) for i in range(n)]

    aff = lambda Q: tuple(T.to_affine(Q))
    want = [aff(T.add(T.from_affine(x), T.from_affine(y))) for x, y in zip(a, b)]
    assert run(TP_ADD) == want
    assert run(TP_ADD_X4) == want
    assert run(TP_DBL) == [aff(T.double(T.from_affine(x))) for x in a]
    assert run(TP_DBL_X4) == [aff(T.double(T.add(T.from_affine(x), T.from_affine(y)))) for x, y in zip(a, b)]
