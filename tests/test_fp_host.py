"""Field arithmetic of msm_zprize_amd/csrc/fp.h compiled for the HOST (same templates the
kernels instantiate) against Python integers -- mirrors the reference's wasm-vs-bigint
equivalence tests (src/field.test.ts:40-211: multiply, square, inverse, reduce, to/from
Montgomery) over the four base fields the MSM uses.  CPU only."""
import os
import random
import subprocess

import pytest

from oracle import params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "fp_host_test.cpp")
EXE = os.path.join(ROOT, "tests", "native", "fp_host_test")

FIELDS = {
    "bls377": (P.BLS12_377["modulus"], 14 * 28),
    "bls381": (P.BLS12_381["modulus"], 14 * 28),
    "pallas": (P.PALLAS["modulus"], 9 * 29),
    "ed377": (P.ED_ON_BLS12_377["modulus"], 9 * 29),
}


@pytest.fixture(scope="module")
def driver():
    deps = [SRC, os.path.join(ROOT, "msm_zprize_amd", "csrc", "fp.h"), os.path.join(ROOT, "msm_zprize_amd", "csrc", "scalar.h"), os.path.join(ROOT, "msm_zprize_amd", "csrc", "constants_gen.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", EXE, SRC])

    def run(lines):
        out = subprocess.run([EXE], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split()
        assert len(out) == len(lines)
        return out

    return run


def _vals(p, rng, n):
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << (p.bit_length() - 1)), p // 3]
    return edge + [rng.randrange(p) for _ in range(n)]


@pytest.mark.parametrize("field", list(FIELDS))
def test_mul_sqr_chain(driver, field):
    p, rbits = FIELDS[field]
    R = 1 << rbits
    Ri = pow(R, -1, p)
    rng = random.Random(sorted(FIELDS).index(field) + 1)
    xs, ys = _vals(p, rng, 200), _vals(p, rng, 200)
    rng.shuffle(ys)
    lines, want = [], []
    for x, y in zip(xs, ys):
        lines.append(f"{field} mul 2 {x:x} {y:x}"); want.append(x * y * Ri % p)
        lines.append(f"{field} sqr 1 {x:x}"); want.append(x * x * Ri % p)
        lines.append(f"{field} mul_lazy 2 {x:x} {y:x}"); want.append((x - y) * (x + y) * Ri % p)
        lines.append(f"{field} chain 2 {x:x} {y:x}"); want.append(((x * y * Ri - x - y) * x * Ri - y) % p)
        lines.append(f"{field} canon 1 {x:x}"); want.append(x)
        lines.append(f"{field} tomont 1 {x:x}"); want.append(x * R % p)
        lines.append(f"{field} frommont 1 {x:x}"); want.append(x * Ri % p)
    got = driver(lines)
    for l, g, w in zip(lines, got, want):
        assert int(g, 16) == w, l


@pytest.mark.parametrize("field", list(FIELDS))
def test_lazy_inputs_up_to_3p(driver, field):
    """memory residues are lazy: anything in [0, 3p) must behave like its class mod p"""
    p, rbits = FIELDS[field]
    Ri = pow(1 << rbits, -1, p)
    rng = random.Random(7)
    lines, want = [], []
    for _ in range(200):
        x, y = rng.randrange(p), rng.randrange(p)
        kx, ky = rng.randrange(3), rng.randrange(3)
        X, Y = x + kx * p, y + ky * p
        lines.append(f"{field} mul 2 {X:x} {Y:x}"); want.append(x * y * Ri % p)
        lines.append(f"{field} chain 2 {X:x} {Y:x}"); want.append(((x * y * Ri - x - y) * x * Ri - y) % p)
        lines.append(f"{field} canon 1 {X:x}"); want.append(x)
    got = driver(lines)
    for l, g, w in zip(lines, got, want):
        assert int(g, 16) == w, l


@pytest.mark.parametrize("field", list(FIELDS))
def test_store_ranges(driver, field):
    """fe_store / fe_store_mulout leave a residue of the right class inside [0, 3p)"""
    p, rbits = FIELDS[field]
    Ri = pow(1 << rbits, -1, p)
    rng = random.Random(11)
    lines, cls = [], []
    for _ in range(300):
        x, y = rng.randrange(3 * p), rng.randrange(3 * p)
        lines.append(f"{field} store 2 {x:x} {y:x}"); cls.append((x - 3 * y) % p)
        lines.append(f"{field} store_mulout 2 {x:x} {y:x}"); cls.append(x * y * Ri % p)
    for x, y in [(0, 3 * p - 1), (3 * p - 1, 0), (0, 0), (p, p), (3 * p - 1, 3 * p - 1)]:
        lines.append(f"{field} store 2 {x:x} {y:x}"); cls.append((x - 3 * y) % p)
    got = driver(lines)
    for l, g, w in zip(lines, got, cls):
        v = int(g, 16)
        assert v % p == w and 0 <= v < 3 * p, l


@pytest.mark.parametrize("field", list(FIELDS))
def test_inverse(driver, field):
    """inverse.ts semantics: Montgomery in, Montgomery out; 0 is reported, not trapped"""
    p, rbits = FIELDS[field]
    R = 1 << rbits
    rng = random.Random(3)
    xs = [1, 2, p - 1, R % p] + [rng.randrange(1, p) for _ in range(60)]
    got = driver([f"{field} inv 1 {x:x}" for x in xs])
    for x, g in zip(xs, got):
        # x = a R  ->  a^-1 R = R^2 / x
        assert int(g, 16) == R * R * pow(x, -1, p) % p
    assert driver([f"{field} inv 1 0", f"{field} inv 1 {p:x}"]) == ["ZERO", "ZERO"]


@pytest.mark.parametrize("field,params", [("bls377", P.BLS12_377), ("pallas", P.PALLAS), ("bls381", P.BLS12_381)])
def test_glv_decompose(driver, field, params):
    """scalar.h glv_decompose (the kernel's code, compiled for the host): s = (+-)s0 + (+-)s1*lambda mod q with
    |s0|, |s1| < 2^127 -- the relations glv/glv-test.ts:102-125 checks for the wasm `decompose`"""
    q, lam = params["order"], params["endomorphism"]["lambda_"]
    rng = random.Random(17)
    xs = [0, 1, 2, q - 1, q - 2, lam, lam + 1, q - lam, q // 2, (1 << 128) - 1, 1 << 128] + [rng.randrange(q) for _ in range(3000)]
    got = driver([f"{field} glv 1 {x:x}" for x in xs])
    for x, g in zip(xs, got):
        n0, s0, n1, s1 = g.split(":")
        s0, s1 = int(s0, 16), int(s1, 16)
        assert s0 < (1 << 127) and s1 < (1 << 127)
        v = (-s0 if n0 == "1" else s0) + (-s1 if n1 == "1" else s1) * lam
        assert (v - x) % q == 0, hex(x)


def test_host64_matches_limb_arithmetic():
    """csrc/host64.h (64-bit limbs, same Montgomery radix) == fp.h / curve.h on the host"""
    src = os.path.join(ROOT, "tests", "native", "host64_test.cpp")
    exe = os.path.join(ROOT, "tests", "native", "host64_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("mismatches 0") == 3, r.stdout
