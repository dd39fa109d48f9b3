"""Plain-integer restatement of the reference's `src/bigint` layer (TEST INFRASTRUCTURE ONLY).

Every function names the reference lines it follows (paths relative to /root/reference/src).
Points:  affine Weierstrass = (x, y, is_zero); projective = (X, Y, Z); twisted Edwards
extended = (X, Y, Z, T).
"""
from __future__ import annotations


def log2ceil(n: int) -> int:
    """util.ts:163-167 -- ceil(log2(n)), smallest k with n <= 2^k."""
    if n == 1:
        return 0
    return (n - 1).bit_length()


def inverse(x: int, p: int) -> int:
    """bigint/field.ts:117-122 (EGCD inverse; raises on 0)."""
    x %= p
    if x == 0:
        raise ZeroDivisionError("cannot invert 0")
    return pow(x, -1, p)


# --------------------------------------------------------------------------- affine Weierstrass
class AffineWeierstrass:
    """bigint/affine-weierstrass.ts:31-184 (a = 0: y^2 = x^3 + b)."""

    def __init__(self, params):
        assert params["a"] == 0
        self.p = params["modulus"]
        self.q = params["order"]
        self.b = params["b"]
        self.cofactor = params["cofactor"]
        self.zero = (0, 1, True)
        self.one = (params["generator"]["x"], params["generator"]["y"], False)

    def add(self, P1, P2):
        """affine-weierstrass.ts:44-67 -- complete addition."""
        p = self.p
        if P1[2]:
            return P2
        if P2[2]:
            return P1
        x1, y1, _ = P1
        x2, y2, _ = P2
        if (x1 - x2) % p == 0:
            if (y1 - y2) % p == 0:
                return self.double(P1)
            assert (y1 + y2) % p == 0
            return self.zero
        d = inverse(x2 - x1, p)
        m = (y2 - y1) * d % p
        x3 = (m * m - x1 - x2) % p
        y3 = (m * (x1 - x3) - y1) % p
        return (x3, y3, False)

    def double(self, P):
        """affine-weierstrass.ts:72-85."""
        p = self.p
        x, y, z = P
        if z:
            return self.zero
        d = inverse(2 * y, p)
        m = 3 * x * x * d % p
        x2 = (m * m - 2 * x) % p
        y2 = (m * (x - x2) - y) % p
        return (x2, y2, False)

    def negate(self, P):
        """affine-weierstrass.ts:90-93."""
        if P[2]:
            return self.zero
        return (P[0], (-P[1]) % self.p, False)

    def scale(self, s: int, P):
        """affine-weierstrass.ts:111-119 -- MSB-first double-and-add."""
        Q = self.zero
        for i in range(s.bit_length() - 1, -1, -1):
            Q = self.double(Q)
            if (s >> i) & 1:
                Q = self.add(Q, P)
        return Q

    def is_on_curve(self, P):
        """affine-weierstrass.ts:132-135."""
        if P[2]:
            return True
        x, y, _ = P
        return (y * y - x * x * x - self.b) % self.p == 0

    def is_in_subgroup(self, P):
        return self.scale(self.q, P)[2]

    def is_equal(self, P1, P2):
        if P1[2] or P2[2]:
            return P1[2] and P2[2]
        return (P1[0] - P2[0]) % self.p == 0 and (P1[1] - P2[1]) % self.p == 0


# --------------------------------------------------------------------------- projective Weierstrass
class ProjectiveWeierstrass:
    """bigint/projective-weierstrass.ts:19-240 (homogeneous, a = 0)."""

    def __init__(self, params):
        assert params["a"] == 0
        self.p = params["modulus"]
        self.q = params["order"]
        self.b = params["b"]
        self.zero = (0, 1, 0)
        self.one = (params["generator"]["x"], params["generator"]["y"], 1)
        self.scalar_bits = log2ceil(self.q)

    def add(self, P1, P2):
        """projective-weierstrass.ts:33-85 (add-1998-cmo-2 + zero/equal cases)."""
        p = self.p
        X1, Y1, Z1 = P1
        X2, Y2, Z2 = P2
        if Z1 % p == 0:
            return P2
        if Z2 % p == 0:
            return P1
        Y1Z2 = Y1 * Z2 % p
        X1Z2 = X1 * Z2 % p
        Z1Z2 = Z1 * Z2 % p
        u = (Y2 * Z1 - Y1Z2) % p
        uu = u * u % p
        v = (X2 * Z1 - X1Z2) % p
        if v == 0:
            if u == 0:
                return self.double(P1)
            return self.zero
        vv = v * v % p
        vvv = v * vv % p
        R = vv * X1Z2 % p
        A = (uu * Z1Z2 - vvv - 2 * R) % p
        X3 = v * A % p
        Y3 = (u * (R - A) - vvv * Y1Z2) % p
        Z3 = vvv * Z1Z2 % p
        return (X3, Y3, Z3)

    def double(self, P):
        """projective-weierstrass.ts:90-115 (dbl-1998-cmo-2, a = 0)."""
        p = self.p
        X1, Y1, Z1 = P
        if Z1 % p == 0:
            return self.zero
        w = 3 * X1 * X1 % p
        s = Y1 * Z1 % p
        ss = s * s % p
        sss = s * ss
        R = Y1 * s % p
        B = X1 * R % p
        h = (w * w - 8 * B) % p
        X3 = 2 * h * s % p
        Y3 = (w * (4 * B - h) - 8 * R * R) % p
        Z3 = 8 * sss % p
        return (X3, Y3, Z3)

    def negate(self, P):
        return (P[0], (-P[1]) % self.p, P[2])

    def is_equal(self, P1, P2):
        """projective-weierstrass.ts:124-139."""
        p = self.p
        X1, Y1, Z1 = P1
        X2, Y2, Z2 = P2
        if Z1 % p == 0:
            return Z2 % p == 0
        if Z2 % p == 0:
            return False
        return (X1 * Z2 - X2 * Z1) % p == 0 and (Y1 * Z2 - Y2 * Z1) % p == 0

    def scale(self, s: int, P):
        """projective-weierstrass.ts:148-156."""
        Q = self.zero
        for i in range(s.bit_length() - 1, -1, -1):
            Q = self.double(Q)
            if (s >> i) & 1:
                Q = self.add(Q, P)
        return Q

    def from_affine(self, A):
        """projective-weierstrass.ts:205-208."""
        if A[2]:
            return self.zero
        return (A[0], A[1], 1)

    def to_affine(self, P):
        """projective-weierstrass.ts:209-213 -- canonical affine (x, y, is_zero); zero = (0, 1, True)."""
        p = self.p
        X, Y, Z = P
        if Z % p == 0:
            return (0, 1, True)
        zi = inverse(Z, p)
        return (X * zi % p, Y * zi % p, False)

    def is_on_curve(self, P):
        p = self.p
        X, Y, Z = P
        return (Y * Y * Z - X * X * X - self.b * Z * Z * Z) % p == 0


# --------------------------------------------------------------------------- twisted Edwards (a = -1)
class TwistedEdwards:
    """bigint/twisted-edwards.ts:28-217 (extended coordinates, add-2008-hwcd-3, k = 2d)."""

    def __init__(self, params):
        self.p = params["modulus"]
        self.q = params["order"]
        self.d = params["d"]
        self.k = 2 * self.d
        self.cofactor = params["cofactor"]
        self.zero = (0, 1, 1, 0)
        self.one = self.from_affine((params["generator"]["x"], params["generator"]["y"]))
        self.scalar_bits = log2ceil(self.q)

    def from_affine(self, A):
        """twisted-edwards.ts:36-38."""
        x, y = A[0], A[1]
        return (x, y, 1, x * y % self.p)

    def to_affine(self, P):
        """twisted-edwards.ts:39-45."""
        p = self.p
        X, Y, Z, _ = P
        assert Z % p != 0
        zi = inverse(Z, p)
        return (X * zi % p, Y * zi % p)

    def add(self, P1, P2):
        """twisted-edwards.ts:52-85 (strongly unified)."""
        p = self.p
        X1, Y1, Z1, T1 = P1
        X2, Y2, Z2, T2 = P2
        A = (Y1 - X1) * (Y2 - X2) % p
        B = (Y1 + X1) * (Y2 + X2) % p
        C = T1 * T2 % p * self.k % p
        D = 2 * Z1 * Z2 % p
        E = (B - A) % p
        F = (D - C) % p
        G = (D + C) % p
        H = (B + A) % p
        return (E * F % p, G * H % p, F * G % p, E * H % p)

    def double(self, P):
        """twisted-edwards.ts:92-94."""
        return self.add(P, P)

    def negate(self, P):
        p = self.p
        return ((-P[0]) % p, P[1], P[2], (-P[3]) % p)

    def is_zero(self, P):
        """twisted-edwards.ts:117-124."""
        p = self.p
        X, Y, Z, T = P
        return Z % p != 0 and X % p == 0 and T % p == 0 and (Y - Z) % p == 0

    def is_equal(self, P1, P2):
        p = self.p
        return (
            P1[2] % p != 0
            and P2[2] % p != 0
            and (P1[0] * P2[2] - P2[0] * P1[2]) % p == 0
            and (P1[1] * P2[2] - P2[1] * P1[2]) % p == 0
        )

    def scale(self, s: int, P):
        Q = self.zero
        for i in range(s.bit_length() - 1, -1, -1):
            Q = self.double(Q)
            if (s >> i) & 1:
                Q = self.add(Q, P)
        return Q

    def is_on_curve(self, P):
        """twisted-edwards.ts:161-169."""
        p = self.p
        X, Y, Z, T = P
        if Z % p == 0:
            return False
        if (T * Z - X * Y) % p != 0:
            return False
        return (-X * X + Y * Y - Z * Z - self.d * T * T) % p == 0


# --------------------------------------------------------------------------- naive Pippenger
def msm(curve, scalars, points):
    """bigint/msm.ts:8-53 -- unsigned c-bit windows, c = max(log2(N) - 1, 1).

    `curve` needs .zero, .add, .double and .scalar_bits (Curve.Scalar.sizeInBits).
    """
    N = len(scalars)
    assert N == len(points)
    b = curve.scalar_bits
    c = max(log2ceil(N) - 1, 1)
    c_mask = (1 << c) - 1
    K = -(-b // c)
    L = 1 << c
    partition_sums = []
    for k in range(K):
        buckets = [curve.zero] * (L - 1)
        for i in range(N):
            l = (scalars[i] >> (k * c)) & c_mask
            if l == 0:
                continue
            buckets[l - 1] = curve.add(buckets[l - 1], points[i])
        running = curve.zero
        triangle = curve.zero
        for l in range(L - 2, -1, -1):
            running = curve.add(running, buckets[l])
            triangle = curve.add(triangle, running)
        partition_sums.append(triangle)
    result = partition_sums[K - 1]
    for k in range(K - 2, -1, -1):
        for _ in range(c):
            result = curve.double(result)
        result = curve.add(result, partition_sums[k])
    return result


def msm_direct(curve, scalars, points):
    """Definition of the MSM, sum_i [s_i] P_i by double-and-add (independent second opinion)."""
    acc = curve.zero
    for s, P in zip(scalars, points):
        acc = curve.add(acc, curve.scale(s, P))
    return acc


# --------------------------------------------------------------------------- GLV
def egcd_stop_early(l: int, p: int):
    """glv/glv.ts:21-50 -- lattice basis [[v00, v01], [v10, v11]] with v0j + l*v1j = 0 (mod p)."""
    assert l <= p
    r0, r1 = p, l
    s0, s1 = 1, 0
    t0, t1 = 0, 1
    while r1 * r1 > p:
        quot = r0 // r1
        r0, r1 = r1, r0 - quot * r1
        s0, s1 = s1, s0 - quot * s1
        t0, t1 = t1, t0 - quot * t1
    quot = r0 // r1
    r2 = r0 - quot * r1
    t2 = t0 - quot * t1
    v00, v10 = r1, -t1
    if max(r0, abs(t0)) <= max(r2, abs(t2)):
        v01, v11 = r0, -t0
    else:
        v01, v11 = r2, -t2
    return (v00, v01), (v10, v11)


def _tdiv(a: int, b: int) -> int:
    """JS bigint division truncates toward zero."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def glv_constants(q: int, lam: int, w: int = 29):
    """wasm/glv.ts:35-51 with montgomeryParams(q, w, 1) (scalar-glv.ts:36)."""
    n = -(-(log2ceil(q) + 1) // w)
    n0 = -(-n // 2)
    m = n0 * w
    k = (n - n0) * w
    (v00, v01), (v10, v11) = egcd_stop_early(lam, q)
    det = v00 * v11 - v10 * v01
    m0 = _tdiv((1 << (m + k)) * -v11, det)
    m1 = _tdiv((1 << (m + k)) * v10, det)
    return dict(n=n, n0=n0, m=m, k=k, v=((v00, v01), (v10, v11)), det=det, m0=m0, m1=m1)


def _div_pow2_round(x: int, m: int) -> int:
    """glv/glv-test.ts:144-149 -- round(x / 2^m) for x >= 0."""
    up = (x >> (m - 1)) & 1
    return (x >> m) + up


def glv_decompose(s: int, q: int, lam: int, consts=None):
    """Bigint formula of the wasm `decompose` (wasm/glv.ts:76-81; glv/glv-test.ts:96-100).

    Returns signed (s0, s1) with s0 + s1*lambda = s (mod q).
    """
    c = consts or glv_constants(q, lam)
    (v00, v01), (v10, v11) = c["v"]
    m, k, m0, m1 = c["m"], c["k"], c["m0"], c["m1"]
    sg0 = 1 if m0 >= 0 else -1
    sg1 = 1 if m1 >= 0 else -1
    x0 = sg0 * _div_pow2_round(abs(m0) * (s >> k), m)
    x1 = sg1 * _div_pow2_round(abs(m1) * (s >> k), m)
    s0 = v00 * x0 + v01 * x1 + s
    s1 = v10 * x0 + v11 * x1
    return s0, s1


def endomorphism(P, beta: int, p: int):
    """wasm/curve.ts:90-103 -- (x, y) -> (beta*x, y)."""
    return (P[0] * beta % p, P[1], P[2])


def signed_digits(s: int, c: int, K: int):
    """msm-batched-affine.ts:180-199 -- signed c-bit digits; returns [(l, negate)] per window."""
    L = 1 << (c - 1)
    out = []
    carry = 0
    for k in range(K):
        l = ((s >> (k * c)) & ((1 << c) - 1)) + carry
        if l > L:
            l = 2 * L - l
            carry = 1
        else:
            carry = 0
        out.append((l, carry))
    return out


# ------------------------------------------------------------------------------------------ bucket reduction
def reduce_buckets_running_sum(curve, buckets):
    """sum_{l=1..L} l * B_l by the reference's running sum (msm-batched-affine.ts:544-571 reduceBucketsColumnProjective,
    msm-basic.ts:192-223): buckets[l - 1] = B_l in the curve's projective / extended form (curve.zero = empty)."""
    running, total = curve.zero, curve.zero
    for B in reversed(buckets):
        running = curve.add(running, B)
        total = curve.add(total, running)
    return total


def reduce_buckets_2d(curve, buckets, c):
    """The same sum the way the HIP engine forms it (msm_zprize_amd/csrc/reduce2d_kernels.h): the weight j = l is split
    j = h * D + d with H = 2^ceil((c-1)/2) rows and D = L / H columns,
        sum_j j E_j = D * sum_h h R_h + sum_d d C_d,   R_h = sum_d E[h D + d],  C_d = sum_h E[h D + d],
    the bucket of weight L = H * D is added twice into row H / 2, and the factor D is applied as b = log2 D doublings
    (the host's Horner pass adds the row result b bit positions above the column result).  Returns the window sum."""
    L = 1 << (c - 1)
    assert len(buckets) == L
    a = (c - 1 + 1) // 2
    b = c - 1 - a
    H, D = 1 << a, 1 << b
    E = [curve.zero] + list(buckets[:L - 1])           # E[j] = bucket of weight j, j in [0, L)
    rows = [curve.zero] * H
    cols = [curve.zero] * D
    for j in range(1, L):
        h, d = divmod(j, D)
        rows[h] = curve.add(rows[h], E[j])
        cols[d] = curve.add(cols[d], E[j])
    for _ in range(2):
        rows[H // 2] = curve.add(rows[H // 2], buckets[L - 1])
    weighted = lambda xs: reduce_buckets_running_sum(curve, xs[1:]) if len(xs) > 1 else curve.zero   # sum_i i * xs[i]
    A, Bc = weighted(rows), weighted(cols)
    for _ in range(b):
        A = curve.double(A)
    return curve.add(A, Bc)
