"""ctypes wrapper of oracle/msm_oracle.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "msm_oracle.c")
LIB = os.path.join(HERE, "libmsm_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB)
        l.oracle_msm.restype = C.c_int
        l.oracle_msm.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_uint64, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p,
                                 C.c_uint64, C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint64)]
        l.oracle_msm_sharded.restype = C.c_int
        l.oracle_msm_sharded.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_uint64, C.c_int, C.c_char_p, C.c_char_p,
                                         C.c_char_p, C.c_uint64, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_int,
                                         C.POINTER(C.c_uint64)]
        l.oracle_scale.restype = C.c_int
        l.oracle_scale.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_uint64, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p,
                                   C.POINTER(C.c_int)]
        l.oracle_num_threads.restype = C.c_int
        _lib = l
    return _lib


def _curve_args(params):
    kind = 1 if params["kind"] == "twisted-edwards" else 0
    fb = params["fe_bytes"]
    return kind, params["modulus"].to_bytes(fb, "little"), fb // 8, params.get("d", 0)


def msm_bytes(params, scalars_le32: bytes, points_xy: bytes, n: int, is_inf: bytes = None, threads: int = 0):
    """bigint/msm.ts on raw wire-format buffers; returns ({x, y, isZero}, n_adds)."""
    kind, p_le, nl, d = _curve_args(params)
    fb = params["fe_bytes"]
    out = C.create_string_buffer(2 * fb)
    inf = C.c_int()
    adds = C.c_uint64()
    bits = (params["order"] - 1).bit_length()
    st = lib().oracle_msm(kind, p_le, nl, d, bits, scalars_le32, points_xy, is_inf, n, out, C.byref(inf), threads,
                          C.byref(adds))
    if st:
        raise RuntimeError(f"oracle_msm failed: {st}")
    r = {"x": int.from_bytes(out.raw[:fb], "little"), "y": int.from_bytes(out.raw[fb:], "little"), "isZero": inf.value != 0}
    if r["isZero"] and kind == 0:
        r["x"], r["y"] = 0, 1
    return r, adds.value


def msm_bytes_sharded(params, scalars_le32: bytes, points_xy: bytes, n: int, shards: int, threads: int = 0):
    """msm_bytes on `shards` contiguous index ranges in parallel (bench.py's CPU baseline); returns (point, n_adds)."""
    kind, p_le, nl, d = _curve_args(params)
    fb = params["fe_bytes"]
    out = C.create_string_buffer(2 * fb)
    inf = C.c_int()
    adds = C.c_uint64()
    bits = (params["order"] - 1).bit_length()
    st = lib().oracle_msm_sharded(kind, p_le, nl, d, bits, scalars_le32, points_xy, None, n, shards, out, C.byref(inf),
                                  threads, C.byref(adds))
    if st:
        raise RuntimeError(f"oracle_msm_sharded failed: {st}")
    r = {"x": int.from_bytes(out.raw[:fb], "little"), "y": int.from_bytes(out.raw[fb:], "little"), "isZero": inf.value != 0}
    return r, adds.value


def msm(params, scalars, points, threads=0):
    """scalars: ints; points: dicts {x, y, isZero?}"""
    fb = params["fe_bytes"]
    sb = b"".join(int(s).to_bytes(32, "little") for s in scalars)
    pb = b"".join(int(p["x"]).to_bytes(fb, "little") + int(p["y"]).to_bytes(fb, "little") for p in points)
    inf = bytes(1 if p.get("isZero") else 0 for p in points)
    return msm_bytes(params, sb, pb, len(scalars), inf if any(inf) else None, threads)[0]


def scale(params, s, point):
    kind, p_le, nl, d = _curve_args(params)
    fb = params["fe_bytes"]
    out = C.create_string_buffer(2 * fb)
    inf = C.c_int()
    pb = int(point["x"]).to_bytes(fb, "little") + int(point["y"]).to_bytes(fb, "little")
    st = lib().oracle_scale(kind, p_le, nl, d, int(s).to_bytes(32, "little"), pb, int(bool(point.get("isZero"))), out,
                            C.byref(inf))
    if st:
        raise RuntimeError(f"oracle_scale failed: {st}")
    r = {"x": int.from_bytes(out.raw[:fb], "little"), "y": int.from_bytes(out.raw[fb:], "little"), "isZero": inf.value != 0}
    if r["isZero"] and kind == 0:
        r["x"], r["y"] = 0, 1
    return r
