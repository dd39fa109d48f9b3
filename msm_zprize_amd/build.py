"""Build the HIP shared library (libmsmz.so) in-tree for gfx950.  `python -m msm_zprize_amd.build`.

One translation unit per kernel family, compiled in parallel; objects are cached under csrc/_obj/.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmsmz.so")
SOURCES = ["msmz.hip", "kern_batch.hip", "kern_reduce.hip", "kern_misc.hip", "kern_gen.hip"]
# (source, curve id) translation units; curve 3 (twisted Edwards) has no batched-affine kernels
UNITS = [("msmz.hip", None)] + [(f, c) for c in (0, 1, 2, 3) for f in SOURCES[1:] if not (f == "kern_batch.hip" and c == 3)]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))   # every header: a change to any of them rebuilds
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value"]


def _deps_mtime():
    deps = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(HERE, "..", "include", "msmz.h")]
    return max(os.path.getmtime(d) for d in deps)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return _deps_mtime() > t or any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = _deps_mtime()

    def compile_one(unit):
        src, curve = unit
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", "" if curve is None else f"_c{curve}") + ".o")
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(hdr_t, os.path.getmtime(s)):
            return o
        cmd = [hipcc] + FLAGS + ([] if curve is None else [f"-DMSMZ_CURVE={curve}"]) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_napi(force=False, verbose=True):
    """The N-API addon used by the JavaScript/TypeScript host (js/parallel.mjs)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "napi", "msmz_napi.c")
    out = os.path.join(root, "js", "msmz_napi.node")
    inc = "/usr/include/node"
    if not os.path.exists(os.path.join(inc, "node_api.h")):
        raise RuntimeError("node_api.h not found: cannot build the N-API addon")
    if not force and os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return out
    cmd = ["gcc", "-O2", "-shared", "-fPIC", f"-I{inc}", src, f"-L{HERE}", "-lmsmz",
           "-Wl,-rpath,$ORIGIN/../msm_zprize_amd", "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_napi(force="--force" in sys.argv)
