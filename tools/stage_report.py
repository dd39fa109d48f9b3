"""Per-stage times of the Pallas (projective buckets) and ed-on-bls12-377 configurations (development aid)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import msm_zprize_amd as m
m.startThreads()
names = ["digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"]
def show(label, st):
    print(label, f"c={st.c} K={st.K}", "  ".join(f"{nm}={st.stage_ms[i]:.3f}" for i, nm in enumerate(names)), flush=True)
C = m.Weierstrass.create(m.curves.pallasParams)
n = 1 << 22
pts = C.Parallel.randomPointsFast(n, 1)
for rep in range(2):
    sc = C.Parallel.randomScalars(n, 5 + rep)
    out = C.Parallel.msmProjective(sc, pts, n, {})
    sc.free()
show("pallas 2^22 projective", out["stats"])
for c in (15, 17, 18):
    sc = C.Parallel.randomScalars(n, 9)
    out = C.Parallel.msmProjective(sc, pts, n, {"c": c}); sc.free()
    show(f"pallas 2^22 projective c={c}", out["stats"])
sc = C.Parallel.randomScalars(n, 7)
out = C.Parallel.msmUnsafe(sc, pts, n, True, {"glv": 1}); show("pallas 2^22 glv affine", out["stats"])
out = C.Parallel.msmUnsafe(sc, pts, n, True, {"glv": 0}); show("pallas 2^22 noglv affine", out["stats"])
C.close()
T = m.TwistedEdwards.create(m.curves.edOnBls12377Params) if hasattr(m.curves, "edOnBls12377Params") else None
if T:
    n = 1 << 24
    pts = T.Parallel.randomPointsFast(n, 1)
    for rep in range(2):
        sc = T.Parallel.randomScalars(n, 5 + rep)
        out = T.Parallel.msm(sc, pts, n, True); sc.free()
    show("ed377 2^24", out["stats"])
    for c in (16, 18):
        sc = T.Parallel.randomScalars(n, 9)
        out = T.Parallel.msm(sc, pts, n, True, {"c": c}); sc.free()
        show(f"ed377 2^24 c={c}", out["stats"])
