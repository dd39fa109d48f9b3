"""Development aid: reads the workgroup time stamps a -DMSMZ_TRACE build writes (MSMZ_TRACE_OUT=file) and prints, per
sort kernel, when workgroups start and end (100 MHz wall clock -> microseconds) and how long each phase takes.

  tools/build_variant.sh trace -DMSMZ_DEV -DMSMZ_TRACE
  MSMZ_LIB=variants/libmsmz_trace.so MSMZ_TRACE_OUT=gpurun_out/trace.bin python tools/stage20.py 20 0 0 3
  python tools/wg_timeline.py gpurun_out/trace.bin
"""
import sys
import numpy as np


def report(name, t, nslots, labels):
    t = t.astype(np.int64)
    hw = t[:, 15]
    t0 = t[:, 0].min()
    us = (t[:, :nslots] - t0) / 100.0
    start, end = us[:, 0], us[:, nslots - 1]
    print(f"{name}: {len(t)} workgroups, first start -> last end {end.max():.1f} us")
    print(f"  start  min/median/p90/max  {start.min():.1f} {np.median(start):.1f} {np.percentile(start, 90):.1f} {start.max():.1f}")
    print(f"  end    min/median/p90/max  {end.min():.1f} {np.median(end):.1f} {np.percentile(end, 90):.1f} {end.max():.1f}")
    dur = end - start
    print(f"  life   min/median/p90/max  {dur.min():.1f} {np.median(dur):.1f} {np.percentile(dur, 90):.1f} {dur.max():.1f}")
    for j in range(1, nslots):
        d = us[:, j] - us[:, j - 1]
        print(f"    {labels[j - 1]:34s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}")
    late = start > np.median(dur) * 0.5
    print(f"  workgroups starting after {np.median(dur) * 0.5:.1f} us (second pass): {int(late.sum())}")
    cu = (hw & 0xFFFF) >> 8 & 0xF
    se = (hw >> 13) & 0x7
    xcc = (hw >> 32) & 0xF
    key = xcc * 1000 + se * 16 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"  distinct (xcc, se, cu) ids: {len(uniq)}, workgroups per id min/max {cnt.min()}/{cnt.max()}")
    hist, edges = np.histogram(start, bins=12)
    print("  start histogram:", " ".join(f"{e:.0f}:{h}" for h, e in zip(hist, edges)))


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    tiles, nbins = int(raw[0]), int(raw[1])
    tc = raw[2:2 + tiles * 16].reshape(tiles, 16)
    tf = raw[2 + tiles * 16:2 + (tiles + nbins) * 16].reshape(nbins, 16)
    report("k_coarse", tc, 5, ["loads + reserve + scan", "window 0", "windows 1..K-2", "last window + copy"])
    report("k_fine", tf, 6, ["zero + loads in flight", "atomics (hist + rank)", "scan + offsets", "place", "copy out"])
    t0c, t0f = tc[:, 0].min(), tf[:, 0].min()
    print(f"k_fine first start - k_coarse first start: {(int(t0f) - int(t0c)) / 100.0:.1f} us")


main()
