#!/usr/bin/env python3
"""Generate msm_zprize_amd/csrc/constants_gen.h from msm_zprize_amd/curves.py.

This is the build-time analogue of the reference's module init:
`montgomeryParams` (bigint/field-util.ts:18-41), the constants `createMsmField` writes into
wasm memory (field-msm.ts:165-177: p, R, R^2, ...) and the GLV lattice basis `glvGeneral` bakes
into the scalar module (wasm/glv.ts:45-63, glv/glv.ts:21-50).  Limbs are 32-bit words of the
64-bit-limb little-endian layout (6x64 for 377/381-bit fields, 4x64 for 255-bit fields).

Run:  python tools/gen_constants.py   (output is committed; the build does not need Python)
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("curves", os.path.join(ROOT, "msm_zprize_amd", "curves.py"))
curves = importlib.util.module_from_spec(spec)
spec.loader.exec_module(curves)


def limbs(x, n):
    assert 0 <= x < (1 << (32 * n)), (hex(x), n)
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def arr(name, vals):
    body = ", ".join("0x%08xu" % v for v in vals)
    return "  static constexpr uint32_t %s[%d] = {%s};" % (name, len(vals), body)


def egcd_stop_early(l, p):
    """Truncated EGCD lattice basis (same construction as glv/glv.ts:21-50)."""
    r0, r1, t0, t1 = p, l, 0, 1
    while r1 * r1 > p:
        qt = r0 // r1
        r0, r1 = r1, r0 - qt * r1
        t0, t1 = t1, t0 - qt * t1
    qt = r0 // r1
    r2, t2 = r0 - qt * r1, t0 - qt * t1
    v00, v10 = r1, -t1
    if max(r0, abs(t0)) <= max(r2, abs(t2)):
        v01, v11 = r0, -t0
    else:
        v01, v11 = r2, -t2
    return (v00, v01), (v10, v11)


def tdiv(a, b):
    qt = abs(a) // abs(b)
    return qt if (a >= 0) == (b >= 0) else -qt


def glv_device_constants(q, lam):
    """Constants for the device decomposition:
         x_j = sign(m_j) * round(|m_j| * s / 2^256),  m0 = 2^256 * (-v11) / det, m1 = 2^256 * v10 / det
         s0 = s + v00*x0 + v01*x1,   s1 = v10*x0 + v11*x1
    """
    (v00, v01), (v10, v11) = egcd_stop_early(lam, q)
    det = v00 * v11 - v10 * v01
    assert abs(det) == q
    assert (v00 + lam * v10) % q == 0 and (v01 + lam * v11) % q == 0
    m0 = tdiv((1 << 256) * -v11, det)
    m1 = tdiv((1 << 256) * v10, det)
    sg = lambda x: 1 if x >= 0 else -1
    for v in (v00, v01, v10, v11):
        assert abs(v) < (1 << 128)
    assert abs(m0) < (1 << 160) and abs(m1) < (1 << 160)
    return dict(v=(v00, v01, v10, v11), m=(m0, m1),
                # sign of the product v_ij * x_j
                sgn=(sg(v00) * sg(m0), sg(v01) * sg(m1), sg(v10) * sg(m0), sg(v11) * sg(m1)))


def glv_proven_bits(q, c):
    """Analytic bound on the halves, the counterpart of maxS0 / maxS1 of the reference (src/wasm/glv.ts:216-226), in
    exact rational arithmetic.  The real solution of  V x = (-s, 0)  is x*_0 = -s v11 / det, x*_1 = s v10 / det.  The
    device computes x_j = round(|m_j| s / 2^256) with m_j = trunc(2^256 (-v11 | v10) / det): the truncation moves
    m_j by less than 1, hence x_j by less than s / 2^256 < q / 2^256, and the rounding by at most 1/2, so
        |x_j - x*_j| < E = 1/2 + q / 2^256,
        |s0| = |v00 e0 + v01 e1| < (|v00| + |v01|) E,   |s1| < (|v10| + |v11|) E.
    Returns the bit length that bound implies (so |s_j| < 2^bits for EVERY scalar < q)."""
    from fractions import Fraction
    import math
    v00, v01, v10, v11 = c["v"]
    E = Fraction(1, 2) + Fraction(q, 1 << 256)
    bound = max((abs(v00) + abs(v01)) * E, (abs(v10) + abs(v11)) * E)
    return int(math.ceil(bound)).bit_length()


def glv_decompose_model(s, q, lam, c):
    """Integer model of the device kernel (used to bound |s_i| and as a self-check)."""
    v00, v01, v10, v11 = c["v"]
    m0, m1 = c["m"]
    sg = lambda x: 1 if x >= 0 else -1
    rnd = lambda x: (x >> 256) + ((x >> 255) & 1)
    x0 = sg(m0) * rnd(abs(m0) * s)
    x1 = sg(m1) * rnd(abs(m1) * s)
    return s + v00 * x0 + v01 * x1, v10 * x0 + v11 * x1


def limbs_w(x, n, w):
    assert 0 <= x < (1 << (w * n)), (hex(x), n, w)
    return [(x >> (w * i)) & ((1 << w) - 1) for i in range(n)]


def iarr(name, vals):
    body = ", ".join(str(v) for v in vals)
    return "  static constexpr int32_t %s[%d] = {%s};" % (name, len(vals), body)


def field_struct(name, p, nwords, N, W, extra=()):
    """Base-field constants.  Arithmetic radix is 2^W with N signed lazy limbs (R = 2^(N*W));
    the memory format is `nwords` saturated 32-bit words (= nwords/2 64-bit limbs, little endian)."""
    R = 1 << (N * W)
    assert R > (1 << 6) * p, "need headroom for lazy values"
    assert 3 * p < (1 << (32 * nwords)), "memory format must hold lazy values < 3p"
    pinv = pow(p, -1, 1 << W)
    PL = limbs_w(p, N, W)
    out = ["struct %s {" % name,
           "  static constexpr int N = %d;    // limbs in registers" % N,
           "  static constexpr int W = %d;    // bits per limb" % W,
           "  static constexpr int NW = %d;   // 32-bit words in memory (%d x 64-bit limbs)" % (nwords, nwords // 2),
           "  static constexpr int BITS = %d;" % p.bit_length(),
           iarr("PL", PL),
           iarr("NPL", [-v for v in PL]),
           "  static constexpr uint32_t PINV = 0x%08xu;  // p^-1 mod 2^W" % pinv,
           arr("PW", limbs(p, nwords)),
           iarr("ONE", limbs_w(R % p, N, W)),          # Montgomery form of 1
           iarr("R2", limbs_w(R * R % p, N, W)),       # to-Montgomery factor
           iarr("R3", limbs_w(R * R * R % p, N, W)),   # fixes up a plain inverse of a Montgomery value
           iarr("P2", limbs_w(2 * p, N, W)),
           iarr("P4", limbs_w(4 * p, N, W))]
    for nm, val in extra:
        out.append(iarr(nm, limbs_w(val * R % p, N, W)))
    out.append("};")
    return "\n".join(out)


def main():
    import random
    rng = random.Random(1234)
    L = ["// GENERATED by tools/gen_constants.py -- do not edit.",
         "// Field / curve / GLV constants for the MSM kernels (32-bit words of 64-bit-limb LE layout).",
         "#pragma once", "#include <cstdint>", "", "namespace msmz {", ""]
    for c in curves.ALL_CURVES:
        p, q = c["modulus"], c["order"]
        n = c["fe_bytes"] // 4
        tag = {"bls12-377": "Bls377", "pallas": "Pallas", "bls12-381": "Bls381", "ed-on-bls12-377": "Ed377"}[c["label"]]
        R = 1 << (32 * n)
        extra = [("GX", c["generator"]["x"]), ("GY", c["generator"]["y"])]
        if c["kind"] == "weierstrass":
            extra += [("B", c["b"]), ("B3", 3 * c["b"] % p), ("BETA", c["endomorphism"]["beta"])]
        else:
            extra += [("D", c["d"]), ("K2D", 2 * c["d"] % p)]
        L.append("// %s base field: p = 0x%x" % (c["label"], p))
        N, W = {12: (14, 28), 8: (9, 29)}[n]
        L.append(field_struct(tag + "Fp", p, n, N, W, extra))
        L.append("")
        # scalar field
        L.append("struct %sFr {" % tag)
        L.append("  static constexpr int BITS = %d;" % q.bit_length())
        L.append(arr("Q", limbs(q, 8)))
        if c["kind"] == "weierstrass":
            lam = c["endomorphism"]["lambda_"]
            g = glv_device_constants(q, lam)
            mx = 0
            samples = [0, 1, 2, q - 1, q - 2, q // 2, q // 3, lam, lam + 1, q - lam] + [rng.randrange(q) for _ in range(20000)]
            for s in samples:
                s0, s1 = glv_decompose_model(s, q, lam, g)
                assert (s0 + s1 * lam - s) % q == 0
                mx = max(mx, abs(s0), abs(s1))
            assert mx < (1 << 127), mx.bit_length()
            proven = glv_proven_bits(q, g)
            assert mx.bit_length() <= proven <= 128, (mx.bit_length(), proven)   # 4-word halves can never truncate
            L.append("  static constexpr bool HAS_GLV = true;")
            L.append("  static constexpr int GLV_BITS = 128;  // windows cover GLV_BITS bits: halves below 2^127 (observed: %d bits) never overflow them" % mx.bit_length())
            L.append("  static constexpr int GLV_PROVEN_BITS = %d;  // analytic bound (tools/gen_constants.py glv_proven_bits; cf. src/wasm/glv.ts:216-226): |s0|, |s1| < 2^%d for every scalar" % (proven, proven))
            L.append("  static constexpr int GLV_TYP_BITS = %d;  // bit length of the largest half seen in 20000 samples (bucket-balance heuristic only)" % mx.bit_length())
            v00, v01, v10, v11 = g["v"]
            L.append(arr("GLV_V00", limbs(abs(v00), 4)))
            L.append(arr("GLV_V01", limbs(abs(v01), 4)))
            L.append(arr("GLV_V10", limbs(abs(v10), 4)))
            L.append(arr("GLV_V11", limbs(abs(v11), 4)))
            L.append(arr("GLV_M0", limbs(abs(g["m"][0]), 5)))
            L.append(arr("GLV_M1", limbs(abs(g["m"][1]), 5)))
            L.append("  // sign (+1 -> 0, -1 -> 1) of the products v00*x0, v01*x1, v10*x0, v11*x1")
            L.append("  static constexpr uint32_t GLV_NEG[4] = {%s};" % ", ".join("1u" if s < 0 else "0u" for s in g["sgn"]))
            L.append(arr("LAMBDA", limbs(lam, 8)))
        else:
            L.append("  static constexpr bool HAS_GLV = false;")
            L.append("  static constexpr int GLV_BITS = 0;")
            L.append("  static constexpr int GLV_PROVEN_BITS = 0;")
            L.append("  static constexpr int GLV_TYP_BITS = 0;")
        L.append("};")
        L.append("")
    L.append("}  // namespace msmz")
    path = os.path.join(ROOT, "msm_zprize_amd", "csrc", "constants_gen.h")
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
