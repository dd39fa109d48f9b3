"""Development aid: runs only the bucket sort (test hook msmz_test_sort) on 2^log2n random BLS12-377 scalars, for the
workgroup timelines of a -DMSMZ_TRACE build (tools/wg_timeline.py).  usage: sort_probe.py [log2n] [c] [glv]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np

import msm_zprize_amd as m
from msm_zprize_amd import _native

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c = int(sys.argv[2]) if len(sys.argv) > 2 else 17
glv = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 1 << log2n
m.startThreads()
curve = m.Weierstrass.create(m.curves.BY_LABEL["bls12-377"])
rng = np.random.default_rng(5)
raw = rng.integers(0, 256, size=n * 32, dtype=np.uint8)
raw.reshape(n, 32)[:, 31] &= 0x0F   # below the group order (top byte of r is 0x12)
geom = (C.c_uint32 * 8)()
lib = _native.lib()
for _ in range(3):
    st = lib.msmz_test_sort(curve._ctx, raw.ctypes.data_as(C.c_char_p), n, c, glv, 0, geom, None, 0, None, 0)
    assert st == 0, st
print("geometry c K Keff L nb E maxb spread:", list(geom))
