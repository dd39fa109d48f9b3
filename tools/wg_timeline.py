"""Development aid: reads the workgroup time stamps a -DMSMZ_TRACE build writes (MSMZ_TRACE_OUT=file) and prints, per
instrumented kernel, when workgroups start and end (100 MHz wall clock -> microseconds) and how long each phase takes.

  tools/build_variant.sh trace -DMSMZ_DEV -DMSMZ_TRACE
  MSMZ_LIB=variants/libmsmz_trace.so MSMZ_TRACE_OUT=gpurun_out/trace.bin python tools/stage20.py 20 0 0 3
  python tools/wg_timeline.py gpurun_out/trace.bin
"""
import sys
import numpy as np

LABELS = {
    "k_coarse": {1: "loads + scan", 2: "window 0", 3: "windows 1..K-2", 4: "last window + copy"},
    "k_fine": {1: "zero + loads in flight", 2: "atomics (hist + rank)", 3: "scan + offsets", 4: "place", 5: "copy out"},
    "k_plan_emit": {1: "bucket offsets", 2: "chunk totals", 3: "per-round scans", 4: "round 0", 5: "round 1", 6: "round 2",
                    7: "round 3", 8: "round 4", 9: "round 5", 10: "later rounds + bucket records"},
}
for _r in range(12):
    LABELS[f"k_batch_add round {_r}"] = {1: "forward pass", 2: "product tree up", 3: "inversion", 4: "down-sweep", 5: "backward pass"}


def report(name, t):
    t = t.astype(np.int64)
    hw = t[:, 15]
    used = [j for j in range(15) if (t[:, j] != 0).mean() > 0.5]
    ok = np.all(t[:, used] != 0, axis=1)
    print(f"{name}: {len(t)} workgroups ({int((~ok).sum())} with a stamp missing: other code path)")
    t = t[ok]
    hw = hw[ok]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    start, end = us[:, used[0]], us[:, used[-1]]
    print(f"  first start -> last end {end.max():.1f} us")
    for nm, v in (("start", start), ("end", end), ("life", end - start)):
        print(f"  {nm:6s} min/median/p90/max  {v.min():.1f} {np.median(v):.1f} {np.percentile(v, 90):.1f} {v.max():.1f}")
    for a, b in zip(used[:-1], used[1:]):
        d = us[:, b] - us[:, a]
        lab = LABELS.get(name, {}).get(b, f"slot {a} -> {b}")
        print(f"    {lab:34s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}")
    key = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 0x7) * 16 + ((hw >> 8) & 0xF)
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"  distinct (xcc, se, cu) ids: {len(uniq)}, workgroups per id min/max {cnt.min()}/{cnt.max()}")
    hist, edges = np.histogram(start, bins=12)
    print("  start histogram (us:count):", " ".join(f"{e:.0f}:{h}" for h, e in zip(hist, edges)))


def main():
    raw = open(sys.argv[1], "rb").read()
    pos = 0
    while pos + 40 <= len(raw):
        name = raw[pos:pos + 32].split(b"\0")[0].decode()
        n = int(np.frombuffer(raw, dtype=np.uint64, count=1, offset=pos + 32)[0])
        t = np.frombuffer(raw, dtype=np.uint64, count=n * 16, offset=pos + 40).reshape(n, 16)
        pos += 40 + n * 128
        report(name, t)


main()
