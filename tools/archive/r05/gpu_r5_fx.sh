#!/bin/bash
# A/B (same box, interleaved): floor exchange of the small-search list kernel at every tile (prev) / throttled (this tree)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r5_fx; mkdir -p $OUT
P=open-hummingbird-eval_amd/lib/abl/libhbird_hip_prev.so; N=open-hummingbird-eval_amd/lib/libhbird_hip.so
for shape in "1250000 768 21904 30" "2500000 768 21904 30" "1250000 768 21904 5" "3000000 384 21904 30" "800000 1024 21904 30" "2074072 384 12544 30" "300000 768 12544 30" "1250000 768 21904 32"; do
  echo "== $shape"; timeout 600 python tools/ab_lib.py $shape $P $N 2>&1 | grep -v amdgpu | tail -4
done > $OUT/floor_exchange_ab.txt 2>&1
cat $OUT/floor_exchange_ab.txt
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
