#!/bin/bash
# Run on the GPU box (via gpurun): one rocprofv3 counter pass (no --stats / trace domains beside kernel-trace).
# usage: tools/kpmc.sh <out-name> "<COUNTER1 COUNTER2 ...>" <tools/stage20.py args...>
NAME=$1; CTRS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/stage20.py "$@" > $OUT/run.txt 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void msmz::", "").replace("msmz::", "")[:44]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); calls[name].add(r["Dispatch_Id"])
with open(sys.argv[1] + "/summary.txt", "w") as fo:
    for name, v in agg.items():
        n = len(calls[name])
        line = f"{name:46s} calls={n:3d} per-call: " + " ".join(f"{k}={x / n:.4g}" for k, x in sorted(v.items()))
        print(line); fo.write(line + "\n")
PY
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*agent_info.csv $OUT/*/*counter_collection.csv
