#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_ol3; mkdir -p $OUT
L=$PWD/open-hummingbird-eval_amd/lib
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -k "one_launch or phased" > $OUT/pytest_ol.log 2>&1; tail -5 $OUT/pytest_ol.log
{
for cfg in "50176 384 12544 30" "300000 768 12544 30" "2074072 384 12544 30" "50176 384 12544 90" "300000 768 21904 90"; do
  echo "== fp32 $cfg"; AB_WALL=1 timeout 600 python tools/ab_lib.py $cfg $L/abl/libhbird_hip_prev.so $L/libhbird_hip.so@perphase $L/libhbird_hip.so
  echo "== fp16 $cfg"; AB_FP16=1 AB_WALL=1 timeout 600 python tools/ab_lib.py $cfg $L/abl/libhbird_hip_prev.so $L/libhbird_hip.so@perphase $L/libhbird_hip.so
done
echo "== fp32 k=90 5M x 768 (big pool search)"; timeout 900 python tools/ab_lib.py 5000000 768 21904 90 $L/abl/libhbird_hip_prev.so $L/libhbird_hip.so@perphase $L/libhbird_hip.so
echo "== fp16 5M x 768"; AB_FP16=1 timeout 900 python tools/ab_lib.py 5000000 768 21904 30 $L/abl/libhbird_hip_prev.so $L/libhbird_hip.so@perphase $L/libhbird_hip.so
} > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout 600 python tools/exp_ol_trace.py 50176 384 12544 30 f32 50176 384 12544 30 f16 300000 768 12544 30 f16 > $OUT/trace.txt 2>&1; grep -E "kernel ms|boundary [0-9]+:" $OUT/trace.txt | cut -c1-330
