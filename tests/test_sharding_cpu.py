"""The N > 1 path on CPU: two gloo ranks, each with its input shard; the exchange step
(all_gather of the partial sums + host-side msmz_point_add) must reproduce the whole MSM.
The per-shard MSM itself is computed by the oracle here (no GPU in this test); on the GPU box the
same combine code runs after the HIP MSM (bench.py --gpus N)."""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, random
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from oracle import c_oracle, params as P, bigint_ref as B
from msm_zprize_amd import sharding
from msm_zprize_amd import curves
label = sys.argv[1]
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
oc, pc = P.CURVES[label], curves.BY_LABEL[label]
rng = random.Random(99)
n = 37
if oc["kind"] == "weierstrass":
    A = B.AffineWeierstrass(oc)
    pts = []
    for _ in range(n):
        x, y, z = A.scale(rng.randrange(1, 1 << 64), A.one)
        pts.append({"x": x, "y": y, "isZero": z})
else:
    T = B.TwistedEdwards(oc)
    pts = []
    for _ in range(n):
        x, y = T.to_affine(T.scale(rng.randrange(1, 1 << 64), T.one))
        pts.append({"x": x, "y": y})
scalars = [rng.randrange(oc["order"]) for _ in range(n)]
lo, hi = sharding.shard_range(n, rank, world)
partial = c_oracle.msm(oc, scalars[lo:hi], pts[lo:hi])
total = sharding.combine_partials(pc, partial)
whole = c_oracle.msm(oc, scalars, pts)
ok = (total["x"], total["y"]) == (whole["x"], whole["y"]) and bool(total.get("isZero")) == bool(whole.get("isZero"))
print(json.dumps({"rank": rank, "ok": ok, "range": [lo, hi]}), flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("label", ["bls12-377", "pallas", "ed-on-bls12-377"])
def test_two_rank_gloo_combine(tmp_path, label):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + random.Random(label).randrange(2000)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), label],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('"ok": true') == 2, out.stdout


def test_shard_range_partition():
    from msm_zprize_amd.sharding import shard_range
    for n in (1, 7, 8, 1 << 20, (1 << 26) + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_point_add_host_only():
    """msmz_point_add is pure host code: works without a GPU, matches the oracle's group law"""
    from msm_zprize_amd import curves, sharding
    from oracle import bigint_ref as B, params as P
    c = P.BLS12_377
    A = B.AffineWeierstrass(c)
    G = A.one
    p2, p3 = A.scale(2, G), A.scale(3, G)
    d = lambda t: {"x": t[0], "y": t[1], "isZero": t[2]}
    pc = curves.bls12377Params
    assert sharding.point_add(pc, d(G), d(p2)) == d(p3)
    assert sharding.point_add(pc, d(G), d(G)) == d(p2)
    assert sharding.point_add(pc, d(G), d(A.negate(G)))["isZero"]
    assert sharding.point_add(pc, sharding.identity(pc), d(p3)) == d(p3)
