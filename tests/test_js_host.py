"""The JavaScript/TypeScript host (js/parallel.mjs over the N-API addon napi/msmz_napi.c)."""
import json
import os
import shutil
import subprocess

import pytest

from oracle import c_oracle
from oracle import params as P
from oracle import prng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


@pytest.fixture(scope="module")
def addon():
    from msm_zprize_amd import build
    build.build(verbose=False)
    return build.build_napi(verbose=False)


def test_addon_loads_and_has_no_fallback(addon):
    """CPU: the addon loads under node, exposes the binding, and refuses to create a context without a GPU"""
    js = ("const a=require(%r); const names=Object.keys(a).sort(); let err=null;"
          "try{a.create(0,0)}catch(e){err={code:e.code,msg:e.message}}"
          "console.log(JSON.stringify({names,err,fe:[a.feBytes(0),a.feBytes(1),a.feBytes(9)]}))" % addon)
    out = json.loads(subprocess.check_output([NODE, "-e", js], text=True))
    assert set(out["names"]) >= {"create", "destroy", "uploadPoints", "uploadScalars", "randomPoints", "randomScalars",
                                 "downloadPoints", "downloadScalars", "free", "msm", "pointAdd"}
    assert out["fe"] == [48, 32, -1]
    import torch
    if not torch.cuda.is_available():
        assert out["err"] and out["err"]["code"] == "2" and "no CPU fallback" in out["err"]["msg"]


@pytest.mark.gpu
@pytest.mark.parametrize("glv", [0, 1])
def test_run_msm_377_script_matches_closed_form(addon, glv):
    """GPU: node js/scripts/run-msm-377.mjs (the mirror of scripts/run-msm-377.ts) == (sum s_i a_i) G"""
    n = 14
    out = subprocess.check_output([NODE, os.path.join(ROOT, "js", "scripts", "run-msm-377.mjs"), str(n), "--json",
                                   "--glv", str(glv)], text=True, cwd=ROOT)
    got = json.loads(out.strip().splitlines()[-1])
    c = P.BLS12_377
    q = c["order"]
    N = 1 << n
    t = prng.sum_of_products_mod(prng.scalars_np(2, N, q), prng.multipliers_np(1, N), q)
    want = c_oracle.scale(c, t, {"x": c["generator"]["x"], "y": c["generator"]["y"], "isZero": False})
    assert (int(got["x"]), int(got["y"]), got["isZero"]) == (want["x"], want["y"], want["isZero"])
