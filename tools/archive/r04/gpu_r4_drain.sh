#!/bin/bash
# Round 4: pool drain in one pass (both lane halves, all queue levels) against the previous commit's library: parity tests of the pool
# paths, then kernel ms, use_fp16 and fp32 pools.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_drain; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
timeout 1500 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not headline and not full_size" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
for shape in "50176 384 12544 30" "50176 384 12544 90" "300000 768 21904 30" "300000 768 21904 90" "1250000 768 21904 30" "2074072 384 12544 30" "10000000 768 21904 30" "10000000 768 21904 90"; do
  AB_MS2=1 AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $L/abl/libhbird_hip_prev.so $L/libhbird_hip.so 2>&1 | grep same | sed "s/^/fp16 $shape: /" | tee -a $OUT/t.txt
done
for shape in "50176 384 12544 30" "50176 384 12544 90" "300000 768 21904 90" "2074072 384 12544 90"; do
  AB_MS2=1 timeout 900 python tools/ab_lib.py $shape $L/abl/libhbird_hip_prev.so $L/libhbird_hip.so 2>&1 | grep same | sed "s/^/fp32 $shape: /" | tee -a $OUT/t.txt
done
