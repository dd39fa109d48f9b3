"""C-ABI checks that need no GPU: the library builds for gfx950, loads, exports every symbol that
include/msmz.h declares, and fails loudly (no CPU fallback) when asked for a context without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from msm_zprize_amd import build
    build.build(verbose=False)
    from msm_zprize_amd import _native
    return _native.lib()


def test_exports_match_header(lib):
    from msm_zprize_amd import _native
    header = open(os.path.join(ROOT, "include", "msmz.h")).read()
    declared = set(re.findall(r"\b(msmz_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"libmsmz.so does not export {name}"
    assert declared == set(_native.EXPORTS), "ctypes binding and header disagree"


def test_struct_layouts_match_header():
    from msm_zprize_amd._native import MsmzLog, MsmzOpts
    assert ctypes.sizeof(MsmzOpts) == 32
    # msmz_log: 8 floats, 4 int32, 2 uint64, 2 uint32, 1 float, 32 floats (+ padding to 8)
    assert ctypes.sizeof(MsmzLog) == 8 * 4 + 4 * 4 + 2 * 8 + 2 * 4 + 4 + 32 * 4 + 4


def test_strerror_and_curve_table(lib):
    assert lib.msmz_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.msmz_strerror(2)
    assert [lib.msmz_curve_fe_bytes(i) for i in range(5)] == [48, 32, 48, 32, -1]


def test_no_cpu_fallback(lib):
    """without a visible GPU msmz_create must fail with MSMZ_ERR_NO_DEVICE, never fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    ctx = ctypes.c_void_p()
    dev = (ctypes.c_int * 1)(0)
    assert lib.msmz_create(ctypes.byref(ctx), 0, dev, 1) == 2
    assert lib.msmz_create(ctypes.byref(ctx), 0, dev, 0) == 2
    assert lib.msmz_create(ctypes.byref(ctx), 99, dev, 1) == 1
    import msm_zprize_amd as m
    with pytest.raises(Exception):
        m.Weierstrass.create(m.curves.bls12377Params)


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under msm_zprize_amd/ may reference it"""
    pkg = os.path.join(ROOT, "msm_zprize_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".js", ".cc")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("# oracle", ""), os.path.join(dirpath, f)
