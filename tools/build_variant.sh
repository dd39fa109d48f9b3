#!/bin/bash
# Development aid: link a variant libmsmz.so whose host TU and batch-add TUs are compiled with extra defines,
# e.g.  tools/build_variant.sh occ4 -DMSMZ_BATCH_OCC=4 ; run with MSMZ_LIB=variants/libmsmz_occ4.so
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/msm_zprize_amd/csrc/_obj
mkdir -p $ROOT/variants
FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -Wno-unused-value"
for c in 0 1 2; do
  /opt/rocm/bin/hipcc $FL -DMSMZ_CURVE=$c "$@" -c $ROOT/msm_zprize_amd/csrc/kern_batch.hip -o $ROOT/variants/kern_batch_c${c}_$TAG.o &
done
# curve 0 (and the twisted-Edwards reduce / accumulate kernels): sort / reduce kernels of the variant (the other curves keep the release objects)
/opt/rocm/bin/hipcc $FL -DMSMZ_CURVE=0 "$@" -c $ROOT/msm_zprize_amd/csrc/kern_misc.hip -o $ROOT/variants/kern_misc_c0_$TAG.o &
/opt/rocm/bin/hipcc $FL -DMSMZ_CURVE=0 "$@" -c $ROOT/msm_zprize_amd/csrc/kern_reduce.hip -o $ROOT/variants/kern_reduce_c0_$TAG.o &
/opt/rocm/bin/hipcc $FL -DMSMZ_CURVE=3 "$@" -c $ROOT/msm_zprize_amd/csrc/kern_reduce.hip -o $ROOT/variants/kern_reduce_c3_$TAG.o &
/opt/rocm/bin/hipcc $FL "$@" -c $ROOT/msm_zprize_amd/csrc/msmz.hip -o $ROOT/variants/msmz_$TAG.o &
wait
OBJS=$(ls $OBJ/*.o | grep -v "kern_batch_c" | grep -v "/msmz.o" | grep -v "kern_misc_c0" | grep -v "kern_reduce_c0" | grep -v "kern_reduce_c3")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/variants/libmsmz_$TAG.so $OBJS $ROOT/variants/kern_batch_c?_$TAG.o $ROOT/variants/kern_misc_c0_$TAG.o $ROOT/variants/kern_reduce_c0_$TAG.o $ROOT/variants/kern_reduce_c3_$TAG.o $ROOT/variants/msmz_$TAG.o
echo built variants/libmsmz_$TAG.so
