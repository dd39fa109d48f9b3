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
    header = open(os.path.join(ROOT, "include", "msmz.h")).read() + open(os.path.join(ROOT, "include", "msmz_test.h")).read()
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
    # a multi-device context (startThreads(n), parallel.ts:291-315) is accepted by the ABI: without GPUs it is
    # NO_DEVICE like the single-device one, not ARG; more than MSMZ_MAX_DEVICES ids is an argument error
    dev2 = (ctypes.c_int * 2)(0, 1)
    assert lib.msmz_create(ctypes.byref(ctx), 0, dev2, 2) == 2
    dev9 = (ctypes.c_int * 9)(*range(9))
    assert lib.msmz_create(ctypes.byref(ctx), 0, dev9, 9) == 1
    assert lib.msmz_ctx_n_devices(None) == -1 and lib.msmz_ctx_fe_bytes(None) == -1
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


def test_host_layer_validates_buffer_lengths():
    """a short host buffer must raise before the native library reads past its end"""
    from msm_zprize_amd import parallel

    class FakeCurve:
        fe_bytes = 48
        default_glv = 1
        kind = "weierstrass"
        _ctx = None

    par = parallel._Parallel(FakeCurve())
    with pytest.raises(ValueError):
        par.pointsFromBytes(b"\0" * 95, 1)
    with pytest.raises(ValueError):
        par.pointsFromBytes(b"\0" * 96 * 4, 4, is_inf=b"\0" * 3)
    with pytest.raises(ValueError):
        par.scalarsFromBytes(b"\0" * 63, 2)
    pts = parallel.DeviceArray(FakeCurve(), 1, 4, "points")
    with pytest.raises(ValueError):
        par.msmUnsafe(b"\0" * (32 * 3), pts, 4)
    with pytest.raises(ValueError):
        par.msmUnsafe(b"\0" * (32 * 8), pts, 5)


def test_shard_block_split_is_a_prefix_partition():
    """msm_zprize_amd/csrc/multi.h: blocks of 2^16 entries dealt round-robin; the first n entries of a set are a
    prefix of every device's local array and the local counts add up to n (python restatement of shard_count)"""
    from msm_zprize_amd.sharding import block_shard_count, block_local_index
    for G in (1, 2, 3, 8):
        for n in (1, 65535, 65536, 65537, 3 * 65536 + 17, 1 << 20, (1 << 23) + 5):
            counts = [block_shard_count(n, g, G) for g in range(G)]
            assert sum(counts) == n
            for i in (0, n - 1, n // 2, min(n - 1, 65536 * G)):
                g, li = block_local_index(i, G)
                assert li < counts[g]
                # prefix property: the entry just past n on the same device (if any) has a local index >= counts[g]
            for g in range(G):
                nxt = [j for j in range(n, n + 65536 * G + 1, 4099) if block_local_index(j, G)[0] == g]
                assert all(block_local_index(j, G)[1] >= counts[g] for j in nxt)
