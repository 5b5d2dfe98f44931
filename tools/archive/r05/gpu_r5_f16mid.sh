#!/bin/bash
# experiment: fp16 searches calibrated from 4,000 stages per workgroup with share-following cuts from 100 pairs (this build) / shipped rules (prev)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_f16mid; mkdir -p $OUT
P=open-hummingbird-eval_amd/lib/abl/libhbird_hip_prev.so; N=open-hummingbird-eval_amd/lib/libhbird_hip.so
for shape in "300000 768 12544 30" "1000000 384 12544 30" "600000 768 12544 30" "2074072 384 12544 30" "150000 768 21904 30"; do
  echo "== $shape use_fp16"; AB_FP16=1 AB_ROUNDS=10 AB_MS2=1 AB_WALL=1 timeout 900 python tools/ab_lib.py $shape $P $N 2>&1 | grep -v amdgpu | tail -2
done > $OUT/f16_mid_ab.txt 2>&1
cat $OUT/f16_mid_ab.txt
