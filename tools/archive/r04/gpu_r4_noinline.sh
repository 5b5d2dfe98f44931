#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_noinline; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not headline and not full_size" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
L=$ROOT/open-hummingbird-eval_amd/lib
python tools/exp_variant.py 10000000 768 21904 90 "0" fp16 2>&1 | grep variant | sed 's/^/fp16 k90 10M: /' | tee $OUT/t.txt
python tools/exp_variant.py 10000000 768 21904 30 "0" fp16 2>&1 | grep variant | sed 's/^/fp16 k30 10M: /' | tee -a $OUT/t.txt
for shape in "5000000 768 21904 90" "2074072 384 12544 90" "50176 384 12544 30"; do
  timeout 900 python tools/ab_lib.py $shape $L/abl/libhbird_hip_prelean.so $L/libhbird_hip.so 2>&1 | tail -2 | sed "s/^/$shape: /" | tee -a $OUT/t.txt
done
python tools/exp_phases.py 50176 384 12544 30 f16 2074072 384 12544 30 f16 2>&1 | grep phases | tee -a $OUT/t.txt
