"""Concurrency probe: a 2^k MSM on ONE GPU through 1, 2 and 4 engines (streams + host threads) of the library's
multi-device scheduler -- every engine runs the whole pipeline on its share of the points, concurrently.
   python tools/two_engines_one_gpu.py [log2n=20]"""
import os, statistics, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msm_zprize_amd as m
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
for engines in (1, 2, 4):
    m.startThreads(devices=[0] * engines)
    C = m.Weierstrass.create(m.curves.bls12377Params)
    pts = C.Parallel.randomPointsFast(n, 1)
    ts = []
    for i in range(12):
        sc = C.Parallel.randomScalars(n, 50 + i)
        t0 = time.perf_counter()
        C.Parallel.msmUnsafe(sc, pts, n, False, {"glv": 0})
        ts.append((time.perf_counter() - t0) * 1e3)
        sc.free()
    print(f"2^{log2n} engines on GPU 0: {engines}  median {statistics.median(ts[2:]):.3f} ms")
    pts.free(); C.close(); m.stopThreads()
