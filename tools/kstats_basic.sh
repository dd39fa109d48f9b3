#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_ks_te24
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/stage_basic.py ed-on-bls12-377 24 3 > $OUT/run.txt 2>&1
cp $OUT/*/*kernel_stats.csv $OUT/kernel_stats.csv
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*agent_info.csv
python3 - "$OUT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/kernel_stats.csv")):
    nm = r["Name"]
    short = nm.split("(")[0].replace("void msmz::", "").replace("msmz::", "")[:60]
    print(f"{short:62s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} {r['Percentage']:>6s}%")
PY
