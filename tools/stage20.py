"""Per-stage and per-round times of one configuration (development aid):
   python tools/stage20.py [log2n=20] [glv=0] [c=0] [runs=6] [curve=bls12-377]"""
import os, statistics, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msm_zprize_amd as m
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
glv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = int(sys.argv[3]) if len(sys.argv) > 3 else 0
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 6
label = sys.argv[5] if len(sys.argv) > 5 else "bls12-377"
m.startThreads()
params = m.curves.BY_LABEL[label]
C = (m.Weierstrass if params["kind"] == "weierstrass" else m.TwistedEdwards).create(params)
n = 1 << log2n
pts = C.Parallel.randomPointsFast(n, 1)
names = ["digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"]
acc, wall, last = [], [], None
for i in range(runs + 2):
    sc = C.Parallel.randomScalars(n, 50 + i)
    t0 = time.perf_counter()
    try:
        out = C.Parallel.msmUnsafe(sc, pts, n, True, {"glv": glv, "c": c}) if params["kind"] == "weierstrass" else C.Parallel.msm(sc, pts, n, True, {"c": c})
    except Exception as e:   # timing-experiment builds return garbage (degenerate batches): kernel times come from rocprof
        print("msm failed:", e); sc.free(); continue
    dt = (time.perf_counter() - t0) * 1e3
    sc.free()
    if i >= 2:
        acc.append([out["stats"].stage_ms[j] for j in range(8)]); wall.append(dt); last = out["stats"]
if not acc:
    sys.exit(0)
mean = [statistics.mean(a[j] for a in acc) for j in range(8)]
print(f"{label} 2^{log2n} glv={glv} c={last.c} K={last.K} rounds={last.rounds} max_bucket={last.max_bucket} entries={last.n_entries} pairs={last.n_pairs}")
print("  wall median %.3f ms   " % statistics.median(wall) + "  ".join(f"{nm}={mean[j]:.3f}" for j, nm in enumerate(names)))
print("  coarse kernel %.4f ms   rounds: " % last.scatter_kernel_ms + " ".join(f"{last.batch_add_ms[r]:.3f}" for r in range(last.rounds)))
C.close()
