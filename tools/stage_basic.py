"""Stage times of the msmBasic path (msmProjective on a Weierstrass curve, msm on the twisted Edwards curve):
   python tools/stage_basic.py [curve=pallas] [log2n=22] [runs=5] [c=0]"""
import os, statistics, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msm_zprize_amd as m
label = sys.argv[1] if len(sys.argv) > 1 else "pallas"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 22
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
c = int(sys.argv[4]) if len(sys.argv) > 4 else 0
m.startThreads()
params = m.curves.BY_LABEL[label]
te = params["kind"] != "weierstrass"
C = (m.TwistedEdwards if te else m.Weierstrass).create(params)
n = 1 << log2n
pts = C.Parallel.randomPointsFast(n, 1)
names = ["digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"]
acc = []
for i in range(runs + 1):
    sc = C.Parallel.randomScalars(n, 50 + i)
    out = C.Parallel.msm(sc, pts, n, True, {"c": c}) if te else C.Parallel.msmProjective(sc, pts, n, {"c": c})
    sc.free()
    if i >= 1:
        acc.append([out["stats"].stage_ms[j] for j in range(8)])
last = out["stats"]
print(f"{label} 2^{log2n} msmBasic c={last.c} K={last.K} entries={last.n_entries}")
print("  " + "  ".join(f"{nm}={statistics.mean(a[j] for a in acc):.3f}" for j, nm in enumerate(names)))
C.close()
