#!/bin/bash
# Run on the GPU box (via gpurun): collects the round's rocprofv3 evidence into gpurun_out/profiles_rNN/.
# usage: tools/collect_profiles.sh r02
set -e
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel-trace + stats of the bench command (the number bench.py prints comes from the same run shape)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
# 2. PMC passes (separate: FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/profile_msm.py --reps 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/profile_msm.py --reps 2 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_fetch > $OUT/${TAG}_pmc_fetch_summary.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_write > $OUT/${TAG}_pmc_write_summary.txt
python3 $GRAFT_REPO_ROOT/tools/make_pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json 2 20
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
echo collected
