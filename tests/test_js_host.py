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


def _node(script, *args, timeout=600):
    out = subprocess.run([NODE, os.path.join(ROOT, "js", script)] + [str(a) for a in args], capture_output=True, text=True,
                         cwd=ROOT, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_all_js_sources_parse():
    """CPU: every ECMAScript source of the host mirror parses under the image's node (no GPU needed)"""
    files = [os.path.join(ROOT, "js", f) for f in ("parallel.mjs", "msm.test.mjs")]
    files += [os.path.join(ROOT, "js", "scripts", f) for f in sorted(os.listdir(os.path.join(ROOT, "js", "scripts")))]
    assert len(files) >= 14
    for f in files:
        subprocess.run([NODE, "--check", f], check=True, capture_output=True)


@pytest.mark.gpu
def test_submission_compute_msm_known_answers(addon):
    """GPU: the mirror of the ZPrize entry (scripts/zprize23/submission-bls377.ts:20-65 compute_msm, incl. the
    equal-points -> `msm` switch, the pointer-style routes and the byte route) passes the reference's own known-answer
    script (submission-test-bls377.ts:6-45); the n-copies result also equals (sum of scalars) * P from the oracle"""
    rep = _node("scripts/submission-test-bls377.mjs", "--json")
    assert rep["twoPoints"] and rep["samePoints"] and rep["byteRoute"]
    c = P.BLS12_377
    point = {"x": 111871295567327857271108656266735188604298176728428155068227918632083036401841336689521497731900230387779623820740,
             "y": 76860045326390600098227152997486448974650822224305058012700629806287380625419427989664237630603922765089083164740,
             "isZero": False}
    want = c_oracle.scale(c, int(rep["sum"]["scalar"]), point)
    assert (int(rep["sum"]["x"]), int(rep["sum"]["y"])) == (want["x"], want["y"])


@pytest.mark.gpu
def test_js_msm_test_mirror_against_fixtures(addon):
    """GPU: js/msm.test.mjs = src/msm.test.ts:24-118 (4 curves x 2^0..2^12; msmUnsafe, msmProjective, TE msm) against
    tests/golden/js_msm_fixtures.json"""
    rep = _node("msm.test.mjs", "--json")
    assert rep == {"ok": True, "checked": 7 + 3 * 7 * 2}


@pytest.mark.gpu
@pytest.mark.parametrize("script,label", [("scripts/run-msm-ed-377.mjs", "ed-on-bls12-377"),
                                          ("scripts/run-msm-pallas.mjs", "pallas"),
                                          ("scripts/run-msm-pallas-projective.mjs", "pallas")])
def test_js_run_scripts_match_closed_form(addon, script, label):
    """GPU: the mirrors of scripts/run-msm-ed-377.ts / run-msm-pallas.ts / run-msm-pallas-projective.ts (runMsm of
    msm-twisted-edwards.ts, msm-weierstrass.ts, msm-weierstrass-projective.ts) == (sum s_i a_i) G"""
    n = 10
    got = _node(script, n, "--json")
    c = P.CURVES[label]
    q = c["order"]
    N = 1 << n
    t = prng.sum_of_products_mod(prng.scalars_np(2, N, q), prng.multipliers_np(1, N), q)
    want = c_oracle.scale(c, t, {"x": c["generator"]["x"], "y": c["generator"]["y"], "isZero": False})
    assert (int(got["x"]), int(got["y"])) == (want["x"], want["y"])


@pytest.mark.gpu
def test_js_benchmark_and_window_sweep_scripts(addon):
    """GPU: benchmarkMsm's protocol (15 runs, 5 dropped) and the window sweep of scripts/evaluate-msm-377.ts:15-62"""
    r = _node("scripts/run-msm-pallas.mjs", 12, "--evaluate", "--json")
    assert len(r["times"]) == 10 and r["median_ms"] > 0
    s = _node("scripts/evaluate-msm-377.mjs", 12, "--json")
    best = s["best"]["12"]
    assert len(s["times"]["12"]) == 3 and abs(best["c"] - best["chosen"]) <= 1 and best["time"] > 0
