#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_f16exp3; mkdir -p $OUT; rm -f $OUT/prio.txt
L=$ROOT/open-hummingbird-eval_amd/lib
for rep in 1 2; do
  for lib in libhbird_hip.so abl/libhbird_hip_prio.so; do
    HBIRD_HIP_LIB=$L/$lib EXP_CL="0,0,-1" python tools/exp_f16_abl.py 10000000 768 21904 0 2>&1 | grep cluster >> $OUT/prio.txt
  done
done
cat $OUT/prio.txt
