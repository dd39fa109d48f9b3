#!/bin/bash
# Run on the GPU box (via gpurun): per-kernel time of one MSM configuration (tools/profile_msm.py) under rocprofv3.
# usage: tools/kernel_stats.sh <out-name> [profile_msm.py args...]
set -e
NAME=${1:-stats}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/profile_msm.py "$@" > $OUT/profile.txt 2>&1
cp $OUT/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
# durations of the reduce kernels of the LAST MSM, in launch order
red = [r for r in rows if "k_reduce" in r["Kernel_Name"]]
per = collections.OrderedDict()
last = red[-12:]
with open(out + "/reduce_levels.txt", "w") as fo:
    for r in last:
        nm = r["Kernel_Name"].split("<")[0].split("::")[-1]
        fo.write(f"{nm:18s} grid={r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','?'):>9s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:8.1f} us\n")
PY
rm -f $OUT/*/*kernel_trace.csv
echo done
