#!/bin/bash
# Round 4: fp16 candidate kernel with the bank pieces six stages ahead (query fragments four) against the previous commit's library
# (lib/abl/libhbird_hip_prev.so): fp16 parity tests, then kernel ms: one query tile against big banks, unclustered mid sizes, the headline.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_ahead; mkdir -p $OUT
L=$ROOT/open-hummingbird-eval_amd/lib
timeout 1200 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp16 or f16 or nan or overflow or config or cfg or zero or denormal or outlier or random or cluster" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
for shape in "10000000 768 196 30" "10000000 768 1369 30" "2074072 384 196 30" "50176 384 12544 30" "300000 768 21904 30" "1250000 768 12544 30" "2074072 384 12544 30" "2074072 128 12544 30" "10000000 768 21904 30" "10000000 768 21904 90"; do
  AB_MS2=1 AB_FP16=1 timeout 900 python tools/ab_lib.py $shape $L/abl/libhbird_hip_prev.so $L/libhbird_hip.so 2>&1 | grep same | sed "s/^/fp16 $shape: /" | tee -a $OUT/t.txt
done
