"""Host-side mirror of the reference's curve factories (`src/parallel.ts`).

    from msm_zprize_amd import Weierstrass, startThreads, stopThreads
    from msm_zprize_amd.curves import bls12377Params
    startThreads()                                   # parallel.ts:291-315  (here: pick the GPU)
    Curve = Weierstrass.create(bls12377Params)       # parallel.ts:40-177
    points = Curve.Parallel.randomPointsFast(N)      # handles to device-resident inputs
    scalars = Curve.Parallel.randomScalars(N)
    out = Curve.Parallel.msmUnsafe(scalars, points, N, True)       # {"result": ..., "log": [...]}
    Curve.Affine.toBigint(out["result"])             # {"x": .., "y": .., "isZero": ..}

Same names, argument order and error behaviour as the reference where Python allows (the reference's
"pointers" into wasm memory become handle objects for device memory; its promises become plain
return values).  Everything below the `Parallel` methods runs in HIP through the C ABI of
include/msmz.h -- this module contains no arithmetic.
"""
import ctypes as C

from . import _native
from ._native import MsmzLog, MsmzOpts, check, lib

_state = {"devices": None}

MAX_DEVICES = 8


def startThreads(n=None, device=None, devices=None):
    """parallel.ts:291-315.  The reference spawns n-1 workers that share one MSM; here the workers are GPUs:
    `n` = number of GPUs a curve context drives (devices 0..n-1; every input set is split over them and the
    partial sums are added on the host -- msmz_create with n_devices = n).  `device` picks one GPU (default:
    LOCAL_RANK or 0, the one-process-per-GPU launch of bench.py); `devices` lists ids explicitly (an id may
    repeat: several engines on one GPU, used to rehearse the scheduler)."""
    import os
    if devices is None:
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if n in (None, 1) else 0
        devices = [int(device) + i for i in range(int(n) if n else 1)]
    devices = [int(d) for d in devices]
    if not 1 <= len(devices) <= MAX_DEVICES:
        raise ValueError(f"startThreads: 1..{MAX_DEVICES} devices, got {len(devices)}")
    _state["devices"] = devices
    lib()  # fail early if the HIP library is not built
    return devices[0] if len(devices) == 1 else devices


def stopThreads():
    """parallel.ts:317-320."""
    _state["devices"] = None


class DeviceArray:
    """A device-resident input array (the reference's wasm-memory pointer lists)."""

    def __init__(self, curve, handle, n, kind):
        self.curve, self.handle, self.n, self.kind = curve, handle, n, kind

    def __len__(self):
        return self.n

    def free(self):
        if self.handle is not None:
            check(lib().msmz_free(self.curve._ctx, self.handle), "msmz_free")
            self.handle = None


class _Scalar:
    def __init__(self, curve):
        self._c = curve
        self.modulus = curve.params["order"]
        self.sizeInBits = (self.modulus - 1).bit_length()

    def toBigints(self, arr, first=0, count=None):
        """readBigint over a range (scripts/msm-weierstrass.ts:74-78)."""
        count = arr.n - first if count is None else count
        buf = C.create_string_buffer(32 * count)
        check(lib().msmz_download_scalars(self._c._ctx, arr.handle, first, count, buf), "msmz_download_scalars")
        raw = buf.raw
        return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(count)]


class _Affine:
    def __init__(self, curve):
        self._c = curve

    def toBigint(self, point):
        """curve-affine.ts:220-233 -- {x, y, isZero}; accepts an MSM result or a canonical record."""
        return dict(point)

    def toBigints(self, arr, first=0, count=None):
        count = arr.n - first if count is None else count
        fb = self._c.fe_bytes
        buf = C.create_string_buffer(2 * fb * count)
        inf = C.create_string_buffer(count)
        check(lib().msmz_download_points(self._c._ctx, arr.handle, first, count, buf, inf), "msmz_download_points")
        raw = buf.raw
        out = []
        for i in range(count):
            x = int.from_bytes(raw[2 * fb * i:2 * fb * i + fb], "little")
            y = int.from_bytes(raw[2 * fb * i + fb:2 * fb * (i + 1)], "little")
            out.append({"x": x, "y": y, "isZero": inf.raw[i] != 0})
        return out


class _Parallel:
    """Curve.Parallel (parallel.ts:135-145 / 250-258)."""

    def __init__(self, curve):
        self._c = curve

    # -- inputs ---------------------------------------------------------------------------------
    def randomPointsFast(self, n, seed=0x6D736D7A):
        h = C.c_uint64()
        check(lib().msmz_random_points(self._c._ctx, n, seed, C.byref(h)), "msmz_random_points")
        return DeviceArray(self._c, h.value, n, "points")

    def randomScalars(self, n, seed=0x6D736D7A):
        h = C.c_uint64()
        check(lib().msmz_random_scalars(self._c._ctx, n, seed, C.byref(h)), "msmz_random_scalars")
        return DeviceArray(self._c, h.value, n, "scalars")

    def pointsFromBytes(self, data, n=None, is_inf=None):
        """parallel.ts:97-112: x||y little-endian canonical, 2*fe_bytes per point."""
        fb = self._c.fe_bytes
        n = len(data) // (2 * fb) if n is None else n
        if n <= 0 or len(data) < 2 * fb * n:
            raise ValueError(f"pointsFromBytes: {len(data)} bytes for {n} points of {2 * fb} bytes")
        if is_inf is not None and len(is_inf) < n:
            raise ValueError(f"pointsFromBytes: {len(is_inf)} infinity flags for {n} points")
        h = C.c_uint64()
        check(lib().msmz_upload_points(self._c._ctx, bytes(data), None if is_inf is None else bytes(is_inf), n,
                                       C.byref(h)), "msmz_upload_points")
        return DeviceArray(self._c, h.value, n, "points")

    def scalarsFromBytes(self, data, n=None):
        """parallel.ts:114-133: 32 bytes little-endian per scalar."""
        n = len(data) // 32 if n is None else n
        if n <= 0 or len(data) < 32 * n:
            raise ValueError(f"scalarsFromBytes: {len(data)} bytes for {n} scalars of 32 bytes")
        h = C.c_uint64()
        check(lib().msmz_upload_scalars(self._c._ctx, bytes(data), n, C.byref(h)), "msmz_upload_scalars")
        return DeviceArray(self._c, h.value, n, "scalars")

    def pointsFromBigints(self, points):
        """Affine.writeBigints route (scripts/zprize23/submission-bls377.ts:90-93)."""
        fb = self._c.fe_bytes
        data = b"".join(int(p["x"]).to_bytes(fb, "little") + int(p["y"]).to_bytes(fb, "little") for p in points)
        inf = bytes(1 if p.get("isZero") else 0 for p in points)
        return self.pointsFromBytes(data, len(points), inf if any(inf) else None)

    def scalarsFromBigints(self, scalars):
        """Scalar.writeBigint route (submission-bls377.ts:95-102)."""
        return self.scalarsFromBytes(b"".join(int(s).to_bytes(32, "little") for s in scalars), len(scalars))

    # -- the MSM --------------------------------------------------------------------------------
    def _msm(self, scalars, points, N, verbose, options, safe, buckets):
        options = dict(options or {})
        opts = MsmzOpts()
        opts.c = int(options.get("c") or 0)
        opts.glv = int(options.get("glv", self._c.default_glv))
        opts.safe = int(options.get("useSafeAdditions", safe))
        opts.buckets = buckets
        opts.timing = 1 if verbose else 0
        opts.reserved[0] = int(options.get("reduceAffine", 0))   # 1: batched-affine first reduction level (reduceBucketsAffine)
        fb = self._c.fe_bytes
        out = C.create_string_buffer(2 * fb)
        inf = C.c_int()
        log = MsmzLog()
        if N <= 0 or N > len(points):
            raise ValueError(f"msm: N = {N} but the point set holds {len(points)}")
        if N > (len(scalars) if isinstance(scalars, DeviceArray) else len(scalars) // 32):
            raise ValueError(f"msm: N = {N} but fewer scalars were given")
        if isinstance(scalars, DeviceArray):
            st = lib().msmz_msm_resident(self._c._ctx, points.handle, scalars.handle, N, C.byref(opts), out,
                                         C.byref(inf), C.byref(log))
        else:
            st = lib().msmz_msm(self._c._ctx, points.handle, bytes(scalars), N, C.byref(opts), out, C.byref(inf),
                                C.byref(log))
        check(st, "msmz_msm")
        raw = out.raw
        result = {"x": int.from_bytes(raw[:fb], "little"), "y": int.from_bytes(raw[fb:], "little"),
                  "isZero": inf.value != 0}
        if result["isZero"] and self._c.kind == "weierstrass":
            result["x"], result["y"] = 0, 1   # bigint/projective-weierstrass.ts:210 toAffine of zero
        return {"result": result, "log": _format_log(log), "stats": log}

    def msm(self, scalars, points, N, verbose=False, options=None):
        """Safe additions (msm-batched-affine.ts:74-328 with useSafeAdditions = true)."""
        return self._msm(scalars, points, N, verbose, options, 1, 0)

    def msmUnsafe(self, scalars, points, N, verbose=False, options=None):
        """msm-batched-affine.ts:574-586."""
        return self._msm(scalars, points, N, verbose, options, 0, 0)

    def msmProjective(self, scalars, points, N, options=None):
        """parallel.ts:69-87: no GLV, projective buckets (msm-basic.ts)."""
        options = dict(options or {})
        options["glv"] = 0
        return self._msm(scalars, points, N, True, options, 1, 1)


def _format_log(log):
    """Same shape as the reference's deferred log lines (msm-common.ts:192-230)."""
    lines = [[{"n_entries": int(log.n_entries), "K": log.K, "c": log.c}]]
    for i, name in enumerate(_native.STAGE_NAMES):
        lines.append([f"{name}... {log.stage_ms[i]:.3f}ms"])
    for r in range(log.rounds):
        if log.batch_add_ms[r] > 0:
            lines.append([f"batch add round {r}: {log.batch_add_ms[r]:.3f}ms"])
    return lines


class _Curve:
    def __init__(self, params, kind):
        if params["kind"] != kind:
            raise ValueError(f"{params['label']} is not a {kind} curve")
        if _state["devices"] is None:
            startThreads()
        self.params = params
        self.kind = kind
        self.fe_bytes = params["fe_bytes"]
        self.default_glv = -1 if kind == "weierstrass" else 0   # -1: GLV below 2^21 points (include/msmz.h)
        ctx = C.c_void_p()
        devs = _state["devices"]
        self.devices = list(devs)
        check(lib().msmz_create(C.byref(ctx), params["curve_id"], (C.c_int * len(devs))(*devs), len(devs)), "msmz_create")
        self._ctx = ctx
        self.Scalar = _Scalar(self)
        self.Affine = _Affine(self)
        self.Parallel = _Parallel(self)

    def close(self):
        if self._ctx is not None:
            lib().msmz_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pointAdd(self, a, b):
        """Host-side group addition of two affine results (combining per-GPU partial sums)."""
        fb = self.fe_bytes
        enc = lambda p: int(p["x"]).to_bytes(fb, "little") + int(p["y"]).to_bytes(fb, "little")
        za, zb = bool(a.get("isZero")), bool(b.get("isZero"))
        out = C.create_string_buffer(2 * fb)
        inf = C.c_int()
        check(lib().msmz_point_add(self.params["curve_id"], None if za else enc(a), int(za), None if zb else enc(b),
                                   int(zb), out, C.byref(inf)), "msmz_point_add")
        r = {"x": int.from_bytes(out.raw[:fb], "little"), "y": int.from_bytes(out.raw[fb:], "little"),
             "isZero": inf.value != 0}
        if r["isZero"] and self.kind == "weierstrass":
            r["x"], r["y"] = 0, 1
        return r


class Weierstrass:
    """`Weierstraß.create(params)` (parallel.ts:33-34, 40-177)."""

    @staticmethod
    def create(params):
        return _Curve(params, "weierstrass")


class TwistedEdwards:
    """`TwistedEdwards.create(params)` (parallel.ts:36-37, 179-289)."""

    @staticmethod
    def create(params):
        return _Curve(params, "twisted-edwards")
