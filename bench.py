#!/usr/bin/env python3
"""Benchmark of the MSM hot path.  `python bench.py --gpus N --steps K --warmup W`

A step = one MSM over one batch of fresh synthetic scalars; points and scalars are resident in HBM
when the timed region starts (the reference keeps its points resident and times only the msm call:
scripts/msm-weierstrass.ts:19-35).

  N = 1   BASELINE.json configs[1]: BLS12-377 G1, 2^20 points, no GLV, affine buckets.
  N > 1   BASELINE.json configs[4] as stated: ONE MSM of 2^26 points split N ways (2^26 / N per GPU).  Every rank
          owns a contiguous input shard, runs the whole single-GPU pipeline on it, and the N partial sums are gathered
          (RCCL all_gather of 100-byte records) and added on the host -- no data-path collective.  "scaling": "strong"
          (total work fixed over N = 2, 4, 8).  The line also carries per_rank_ms, ranks_seen, backend and
          same_size_1gpu_ms (rank 0 runs an MSM of its own shard size while the other ranks wait) so the efficiency of
          the sharded run against one undisturbed GPU doing the same per-GPU work can be read off one line;
          `--log2n-total 26 --gpus 1` gives the one-GPU time of the whole 2^26 MSM.
  --route ctx   one process drives all N GPUs through ONE context (msmz_create(n_devices = N), the startThreads(n)
          route of src/parallel.ts:291-315); `--devices 0,0` rehearses it on one GPU.  Default route "proc" = one process
          per GPU under torch.distributed.run (what the driver launches).

Prints ONE JSON line (rank 0).  value = non-zero signed digits (= bucket insertions, "point-adds",
SURVEY.md section 8d) of all ranks per second.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=0, help="log2 of points PER GPU (overrides --log2n-total)")
    ap.add_argument("--log2n-total", type=int, default=0,
                    help="log2 of the points of the whole MSM, split over the GPUs (default: 20 at 1 GPU, 26 otherwise)")
    ap.add_argument("--glv", type=int, default=0)
    ap.add_argument("--c", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks share GPU 0)")
    ap.add_argument("--route", default="proc", choices=["proc", "ctx"],
                    help="proc: one process per GPU (torch.distributed.run); ctx: one process, one multi-device context")
    ap.add_argument("--devices", default="", help="--route ctx: comma-separated device ids (default 0..N-1; ids may repeat)")
    ap.add_argument("--cpu-log2n", type=int, default=20)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ctx_route = args.route == "ctx"
    if ctx_route:
        if world != 1:
            raise SystemExit("--route ctx runs as ONE process (no torch.distributed.run)")
    elif world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the MSM has no CPU path)")
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = 0          # all ranks share the one GPU; the exchange step runs over gloo on CPU tensors
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    ranks_seen = dist.get_world_size() if world > 1 else 1

    import msm_zprize_amd as m
    from msm_zprize_amd.curves import bls12377Params as params

    gpus = args.gpus
    if args.log2n:
        log2n, log2n_total = args.log2n, None
        n = 1 << log2n                                   # points per GPU
    else:
        log2n_total = args.log2n_total or (20 if gpus == 1 else 26)
        if (1 << log2n_total) % gpus:
            raise SystemExit(f"2^{log2n_total} points do not split evenly over {gpus} GPUs")
        n = (1 << log2n_total) // gpus
        log2n = n.bit_length() - 1 if n & (n - 1) == 0 else None
    if ctx_route:
        devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(gpus))
        if len(devs) != gpus:
            raise SystemExit(f"--devices lists {len(devs)} ids for --gpus {gpus}")
        m.startThreads(devices=devs)
        n_ctx = n * gpus                                 # the context splits the set over its devices itself
    else:
        m.startThreads(device=local_rank)
        n_ctx = n
    curve = m.Weierstrass.create(params)
    par = curve.Parallel
    seed = 0x6D736D7A + 1   # config index 1
    # shard = contiguous index range [rank*n, (rank+1)*n): generator index is global via the seed offset
    points = par.randomPointsFast(n_ctx, seed + 1000003 * rank)
    nsets = args.steps + args.warmup
    # distinct scalar sets per step while they fit comfortably (32 B per scalar); beyond that the sets are reused round-robin
    n_distinct = max(2, min(nsets, (8 << 30) // (32 * n_ctx)))
    scalar_sets = [par.randomScalars(n_ctx, seed + 7919 * (s + 1) + 1000003 * rank) for s in range(n_distinct)]
    opts = {"glv": args.glv, "c": args.c}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from msm_zprize_amd import sharding

    def one_step(s, verbose=True):
        out = par.msmUnsafe(scalar_sets[s % n_distinct], points, n_ctx, verbose, opts)
        # exchange step (N > 1): gather the per-GPU partial sums over RCCL, add them on the host
        res = sharding.combine_partials(params, out["result"], device=None if rehearsal else torch.device("cuda", local_rank))
        return out, res

    for s in range(args.warmup):
        one_step(s)
    # the same per-GPU work on ONE undisturbed GPU: rank 0 alone, every other rank waits at the barrier
    same_size_1gpu_ms = None
    if world > 1:
        barrier()
        if rank == 0:
            ts = []
            for s in range(3):
                t0 = time.perf_counter()
                par.msmUnsafe(scalar_sets[s % n_distinct], points, n_ctx, False, opts)
                ts.append((time.perf_counter() - t0) * 1e3)
            same_size_1gpu_ms = statistics.median(ts)
    barrier()
    t0 = time.perf_counter()
    stats = []
    for s in range(args.warmup, nsets):
        out, _ = one_step(s)
        stats.append(out["stats"])
    t_local = time.perf_counter() - t0      # this rank's own K steps, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    red_dev = "cpu" if rehearsal else "cuda"
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    entries = torch.tensor([float(sum(int(st.n_entries) for st in stats))], dtype=torch.float64, device=red_dev)
    per_rank = torch.zeros(max(world, 1), dtype=torch.float64, device=red_dev)
    per_rank[rank] = t_local / args.steps * 1e3
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(entries, op=dist.ReduceOp.SUM)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_entries = float(entries.item())
    per_rank_ms = [float(v) for v in per_rank.cpu().tolist()]

    if rank == 0:
        st0 = stats[-1]
        K, c = st0.K, st0.c
        ms_per_step = elapsed / args.steps * 1e3
        # Roofline of the HBM-bound kernel the north_star names, the bucket scatter k_coarse: it
        # reads every scalar once (32 B) and writes one packed (bucket bits | sign | index) word per non-zero digit
        # (4 B); digits are never materialized.  Algorithmic bytes per launch = 32 n + 4 E.  Duration: HIP events
        # recorded on the library's own stream around that kernel, averaged over the timed steps.
        entries = statistics.mean(float(s.n_entries) for s in stats) / (gpus if ctx_route else 1)   # per device
        scatter_ms = statistics.mean(float(s.scatter_kernel_ms) for s in stats)
        scatter_bytes = 32 * n + 4 * entries
        achieved = scatter_bytes / (scatter_ms * 1e-3) / 1e9 if scatter_ms > 0 else 0.0
        # the whole sort (scalars -> sorted references + bucket offsets): same algorithmic bytes
        # (the references are written once more, as the sort's output) over histogram + scan + coarse + fine
        sort_ms = statistics.mean(float(s.stage_ms[0]) + float(s.stage_ms[1]) + float(s.stage_ms[2]) for s in stats)
        acc_ms = statistics.mean(float(s.stage_ms[4]) for s in stats)
        pairs = statistics.mean(float(s.n_pairs) for s in stats)
        per_step = [float(s.stage_ms[7]) for s in stats]   # host wall clock of each msmz_msm_resident call
        extra = extra_measurements(par, points, n, args) if (world == 1 and not ctx_route) else {}
        result = {
            "metric": "Mpoint-adds/s (ms per MSM in ms_per_step)",
            "value": total_entries / elapsed / 1e6,
            "unit": "Mpoint-adds/s",
            "n_gpus": gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_msm_median": statistics.median(per_step),
            "ms_per_msm_stdev": statistics.stdev(per_step) if len(per_step) > 1 else 0.0,
            "higher_is_better": True, "scaling": "weak" if gpus == 1 else "strong", "vs_baseline": None,
            "route": args.route, "backend": ("single context, host fold" if ctx_route else (args.backend if world > 1 else "none")),
            "ranks_seen": ranks_seen, "per_rank_ms": per_rank_ms, "same_size_1gpu_ms": same_size_1gpu_ms,
            "dtype": "u32 limbs (28-bit lazy Montgomery, i64 accumulate)", "data": "synthetic",
            "config": {"workload": (f"BLS12-377 G1 MSM 2^{log2n_total} split over {gpus} GPU" if log2n_total else
                                    f"BLS12-377 G1 MSM 2^{log2n} per GPU x {gpus} GPU") +
                                   f" ({n} points per GPU), {'GLV' if args.glv else 'no GLV'}, affine buckets (batched-affine), msmUnsafe",
                       "baseline_config": "configs[1]" if (gpus == 1 and n == 1 << 20) else ("configs[4]" if n * gpus == 1 << 26 else "other"),
                       "points_per_gpu": n, "log2n_total": log2n_total, "c": c, "K": K, "glv": bool(args.glv),
                       "point_adds_per_msm": total_entries / args.steps,
                       "sharding": f"input-split x{gpus}" + (" (blocks of 2^16 dealt round-robin inside one context)" if ctx_route else "")},
            "roofline": {"kernel": "k_coarse (bucket scatter: scalars -> per-bin runs of packed references)", "bound": "hbm",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         # PMC bytes were collected on the 2^20 no-GLV workload only
                         "traffic": pmc_traffic("k_coarse") if (n == 1 << 20 and gpus == 1 and not args.glv) else None,
                         "bytes_per_launch": scatter_bytes, "avg_launch_ms": scatter_ms},
            "sort_roofline": {"stage": "whole bucket sort: k_hist + k_bin_scan + k_coarse + k_fine", "bound": "hbm",
                              "achieved": scatter_bytes / (sort_ms * 1e-3) / 1e9 if sort_ms > 0 else 0.0, "peak": 8000.0,
                              "unit": "GB/s", "frac": scatter_bytes / (sort_ms * 1e-3) / 1e9 / 8000.0 if sort_ms > 0 else 0.0,
                              "traffic": pmc_traffic("sort") if (n == 1 << 20 and gpus == 1 and not args.glv) else None,
                              "bytes": scatter_bytes, "avg_ms": sort_ms},
            "batch_add_roofline": {"kernel": "k_batch_add (all tree rounds)", "bound": "hbm (measured: memory-bound, DESIGN.md)",
                                   "algorithmic_bytes_per_addition": 496,
                                   "achieved": pairs * 496 / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0, "peak": 8000.0,
                                   "unit": "GB/s", "frac": pairs * 496 / (acc_ms * 1e-3) / 1e9 / 8000.0 if acc_ms > 0 else 0.0,
                                   "traffic": pmc_traffic("k_batch_add") if (n == 1 << 20 and gpus == 1 and not args.glv) else None,
                                   "additions": pairs, "avg_ms": acc_ms},
            "valu_roofline": {"kernel": "k_batch_add (all rounds)", "bound": "int32 VALU (v_mad_i64_i32)",
                              "achieved": pairs * 6 / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0,
                              "peak": 72.0, "unit": "Gmodmul/s",
                              "note": "6 field mults per affine add; peak = measured fe_mul loop (profiles/r01_ubench_fp_modmul.txt)",
                              "avg_ms": acc_ms},
            "stage_ms": {name: statistics.mean(float(s.stage_ms[i]) for s in stats)
                         for i, name in enumerate(["digits", "scan", "scatter", "plan", "accumulate", "reduce", "final", "total"])},
        }
        result.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(params, args.cpu_log2n)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(key):
    """HBM bytes per launch (k_coarse) / per MSM (sort, k_batch_add) from the rocprofv3 PMC passes committed under
    profiles/ (FETCH_SIZE and WRITE_SIZE, separate passes; see profiles/README.md for the corrections applied)."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)[key]["hbm_bytes"]
    except Exception:
        return None


def extra_measurements(par, points, n, args):
    """Numbers SURVEY.md section 8(d) asks for beside the resident-scalar rate (never the bench `value`):
    h2d_inclusive_ms  one MSM with the 32-byte scalars handed over as a HOST buffer (msmz_msm: H2D of the scalars +
                      all kernels + D2H of the result) -- section 8(d)'s definition of the metric;
    byte_route_ms     the ZPrize `compute_msm(points, scalars)` entry (scripts/zprize23/submission-bls377.ts:20-65):
                      canonical little-endian point bytes are uploaded and converted on the device (pointsFromBytes),
                      then the MSM runs with host scalars."""
    import msm_zprize_amd as m
    from msm_zprize_amd import _native
    curve = par._c
    opts = {"glv": args.glv, "c": args.c}
    sc = par.randomScalars(n, 424242)
    sbuf = ctypes.create_string_buffer(32 * n)
    _native.check(_native.lib().msmz_download_scalars(curve._ctx, sc.handle, 0, n, sbuf), "download")
    raw = sbuf.raw
    times = []
    for _ in range(6):
        t0 = time.perf_counter()
        par.msmUnsafe(raw, points, n, False, opts)
        times.append((time.perf_counter() - t0) * 1e3)
    out = {"h2d_inclusive_ms": statistics.median(times[1:])}
    if n <= (1 << 22):
        fb = curve.fe_bytes
        pbuf = ctypes.create_string_buffer(2 * fb * n)
        _native.check(_native.lib().msmz_download_points(curve._ctx, points.handle, 0, n, pbuf, None), "download")
        praw = pbuf.raw
        t_up, t_all = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            up = par.pointsFromBytes(praw, n)
            t1 = time.perf_counter()
            par.msmUnsafe(raw, up, n, False, opts)
            t2 = time.perf_counter()
            up.free()
            t_up.append((t1 - t0) * 1e3)
            t_all.append((t2 - t0) * 1e3)
        out["byte_route_ms"] = {"upload_points": statistics.median(t_up), "upload_plus_msm": statistics.median(t_all)}
    sc.free()
    return out


def effective_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box of this pool
    shows 256 logical CPUs but grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(params, log2n):
    """The oracle's C restatement of src/bigint/msm.ts timed on this host's cores, on a bounded sample
    (2^log2n points of the same distribution).  A reported baseline, not the target."""
    from oracle import c_oracle
    from oracle import params as OP
    import msm_zprize_amd as m
    oparams = OP.CURVES[params["label"]]
    n = 1 << log2n
    curve = m.Weierstrass.create(params)
    pts = curve.Parallel.randomPointsFast(n, 99)
    sc = curve.Parallel.randomScalars(n, 99)
    fb = curve.fe_bytes
    pbuf = ctypes.create_string_buffer(2 * fb * n)
    sbuf = ctypes.create_string_buffer(32 * n)
    from msm_zprize_amd import _native
    _native.check(_native.lib().msmz_download_points(curve._ctx, pts.handle, 0, n, pbuf, None), "download")
    _native.check(_native.lib().msmz_download_scalars(curve._ctx, sc.handle, 0, n, sbuf), "download")
    threads = min(c_oracle.lib().oracle_num_threads(), effective_cpus())
    # bigint/msm.ts on one index range has only ~14 windows to run in parallel; shard the input so that
    # (ranges x windows) covers every core that is reported
    shards = max(1, threads // 14)
    t0 = time.perf_counter()
    res, adds = c_oracle.msm_bytes_sharded(oparams, sbuf.raw, pbuf.raw, n, shards, threads)
    dt = time.perf_counter() - t0
    gpu = curve.Parallel.msmUnsafe(sc, pts, n, False, {"glv": 0})["result"]
    curve.close()
    return {"value": adds / dt / 1e6, "unit": "Mpoint-adds/s", "cores": threads, "kind": "port",
            "sample": f"oracle/msm_oracle.c (restated src/bigint/msm.ts) on 2^{log2n} points in {shards} index ranges, "
                      f"{dt:.2f} s, OpenMP over ranges x windows; result {'==' if gpu == res else '!='} GPU result",
            "ms_per_msm": dt * 1e3}


if __name__ == "__main__":
    main()
