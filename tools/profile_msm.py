"""Print per-stage / per-round timings of one MSM configuration (development aid)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import msm_zprize_amd as m
ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--c", type=int, default=0)
ap.add_argument("--glv", type=int, default=0)
ap.add_argument("--safe", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
m.startThreads()
C = m.Weierstrass.create(m.curves.bls12377Params)
n = 1 << a.log2n
pts = C.Parallel.randomPointsFast(n, 1)
for rep in range(a.reps):
    sc = C.Parallel.randomScalars(n, 2 + rep)
    f = C.Parallel.msm if a.safe else C.Parallel.msmUnsafe
    out = f(sc, pts, n, True, {"glv": a.glv, "c": a.c})
    st = out["stats"]
    sc.free()
print(f"n=2^{a.log2n} c={st.c} K={st.K} rounds={st.rounds} entries={st.n_entries} pairs={st.n_pairs} maxb={st.max_bucket}")
names = ["digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"]
print("  ".join(f"{nm}={st.stage_ms[i]:.3f}" for i, nm in enumerate(names)))
print(f"scatter coarse kernel = {st.scatter_kernel_ms:.4f} ms -> {st.K * n * (2 if a.glv else 1) * 8 / st.scatter_kernel_ms / 1e6:.1f} GB/s algorithmic")
print("rounds ms:", " ".join(f"{st.batch_add_ms[r]:.3f}" for r in range(st.rounds)))
