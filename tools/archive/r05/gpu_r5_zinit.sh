#!/bin/bash
# A/B (same box, interleaved): accumulators of a tile read from LDS (prev) / started by MFMAs with C = 0 where the init values are zero
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_zinit; mkdir -p $OUT
P=open-hummingbird-eval_amd/lib/abl/libhbird_hip_prev.so; N=open-hummingbird-eval_amd/lib/libhbird_hip.so
for shape in "10000000 768 21904 30" "5000000 768 21904 30" "1250000 768 21904 30" "5000000 384 21904 30" "2074072 384 12544 30" "5000000 256 21904 30" "50176 384 12544 30" "5000000 768 21904 90"; do
  echo "== $shape"; timeout 900 python tools/ab_lib.py $shape $P $N 2>&1 | grep -v amdgpu | tail -2
done > $OUT/zero_init_ab.txt 2>&1
echo "== 2074072 384 12544 30 use_fp16" >> $OUT/zero_init_ab.txt; AB_FP16=1 timeout 600 python tools/ab_lib.py 2074072 384 12544 30 $P $N 2>&1 | grep -v amdgpu | tail -2 >> $OUT/zero_init_ab.txt
cat $OUT/zero_init_ab.txt
timeout 1500 python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py tests/test_edge_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
