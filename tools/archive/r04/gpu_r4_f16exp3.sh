#!/bin/bash
# Round 4, experiment 3: (a) static priority for waves 4-7 of the fp16 candidate kernel; (b) where its automatic clusters should start.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_f16exp3; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
for rep in 1 2; do
  for lib in libhbird_hip.so abl/libhbird_hip_prio.so; do
    HBIRD_HIP_LIB=$L/$lib EXP_CL="0,0,-1" python tools/exp_f16_abl.py 10000000 768 21904 0 2>&1 | grep cluster | sed "s/^/$lib /" >> $OUT/prio.txt
  done
done
cat $OUT/prio.txt
# cluster threshold: forced 8x1 (shared dealing is automatic) vs none, stages per workgroup in brackets
for shape in "2074072 384 12544" "5000000 768 12544" "2500000 768 21904" "1250000 768 21904" "5000000 384 21904"; do set -- $shape
  EXP_ROWS=$1 EXP_DIM=$2 EXP_NQ=$3 EXP_MODES=f16 EXP_ROUNDS=2 EXP_CFGS="1,1,0;8,1,-1;4,2,-1" EXP_OUT=r4_f16exp3/thr_$1_$2.json python tools/exp_cluster.py 2>&1 | grep same_bits | sed "s/^/$1 x $2 x $3: /" >> $OUT/threshold.txt
done
cat $OUT/threshold.txt
