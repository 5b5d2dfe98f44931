#!/bin/bash
# Round 4: the exact re-rank on a row-major copy of the bank: fp16 parity tests, then whole use_fp16 searches against the previous
# commit's library (lib/abl/libhbird_hip_nocarry.so), same box.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_rows32; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
timeout 1200 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp16 or f16 or nan or overflow or config or cfg or zero or denormal or outlier or random" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
for shape in "50176 384 12544 30" "300000 768 12544 30" "300000 768 21904 90" "300000 384 21904 90" "1250000 768 21904 30" "1250000 768 21904 90" "2074072 384 12544 30" "10000000 768 21904 30" "10000000 768 21904 90"; do
  AB_WALL=1 AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $L/abl/libhbird_hip_nocarry.so $L/libhbird_hip.so 2>&1 | grep same | sed "s/^/$shape: /" | tee -a $OUT/t.txt
done
