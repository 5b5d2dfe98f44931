#!/bin/bash
# A/B (same box, interleaved): XCD calibration from 150,000 stages per workgroup (prev) / from 30,000 (this tree)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_thr; mkdir -p $OUT
P=open-hummingbird-eval_amd/lib/abl/libhbird_hip_prev.so; N=open-hummingbird-eval_amd/lib/libhbird_hip.so
for shape in "2074072 384 12544 30" "1000000 384 12544 30" "600000 768 12544 30" "2074072 384 12544 90" "1250000 768 12544 30" "4000000 384 21904 30"; do
  echo "== $shape"; AB_ROUNDS=6 timeout 900 python tools/ab_lib.py $shape $P $N 2>&1 | grep -v amdgpu | tail -2
done > $OUT/threshold_ab.txt 2>&1
cat $OUT/threshold_ab.txt
