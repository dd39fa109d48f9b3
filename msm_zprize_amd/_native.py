"""ctypes binding of the C ABI in include/msmz.h (libmsmz.so, built in-tree by msm_zprize_amd.build).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, every entry point
raises.  The library is loaded lazily so that CPU-only tooling (tests of the host logic) can import
the package.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSMZ_LIB") or os.path.join(HERE, "libmsmz.so")   # MSMZ_LIB: development builds

MSMZ_N_STAGES = 8
STAGE_NAMES = ["digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"]


class MsmzOpts(C.Structure):
    _fields_ = [("c", C.c_int32), ("glv", C.c_int32), ("safe", C.c_int32), ("buckets", C.c_int32),
                ("timing", C.c_int32), ("reserved", C.c_int32 * 3)]


class MsmzLog(C.Structure):
    _fields_ = [("stage_ms", C.c_float * MSMZ_N_STAGES), ("c", C.c_int32), ("K", C.c_int32), ("rounds", C.c_int32),
                ("glv", C.c_int32), ("n_entries", C.c_uint64), ("n_pairs", C.c_uint64), ("max_bucket", C.c_uint32),
                ("scatter_launches", C.c_uint32), ("scatter_kernel_ms", C.c_float), ("batch_add_ms", C.c_float * 32)]


EXPORTS = {
    # name: (restype, argtypes)
    "msmz_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int]),
    "msmz_destroy": (None, [C.c_void_p]),
    "msmz_strerror": (C.c_char_p, [C.c_int]),
    "msmz_curve_fe_bytes": (C.c_int, [C.c_int]),
    "msmz_ctx_fe_bytes": (C.c_int, [C.c_void_p]),
    "msmz_ctx_n_devices": (C.c_int, [C.c_void_p]),
    "msmz_upload_points": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "msmz_upload_scalars": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "msmz_random_points": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "msmz_random_scalars": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "msmz_download_points": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p]),
    "msmz_download_scalars": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p]),
    "msmz_free": (C.c_int, [C.c_void_p, C.c_uint64]),
    "msmz_msm": (C.c_int, [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(MsmzOpts), C.c_char_p,
                           C.POINTER(C.c_int), C.POINTER(MsmzLog)]),
    "msmz_msm_resident": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(MsmzOpts), C.c_char_p,
                                    C.POINTER(C.c_int), C.POINTER(MsmzLog)]),
    "msmz_point_add": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    # stage-level test hooks (include/msmz_test.h)
    "msmz_test_set_glv_bits": (C.c_int, [C.c_void_p, C.c_int]),
    "msmz_test_retries": (C.c_int, [C.c_void_p]),
    "msmz_test_field": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_uint64, C.c_char_p]),
    "msmz_test_glv": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_char_p]),
    "msmz_test_digits": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "msmz_test_sort": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_uint64, C.c_void_p, C.c_uint64]),
    "msmz_test_point": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64,
                                  C.c_char_p]),
}

_lib = None


class MsmzError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().msmz_strerror(status).decode() if _lib is not None else "?"
        super().__init__(f"{where}: msmz status {status} ({msg})")


def lib():
    """Load libmsmz.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m msm_zprize_amd.build` "
                               "(the MSM has no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, where):
    if status != 0:
        raise MsmzError(status, where)
