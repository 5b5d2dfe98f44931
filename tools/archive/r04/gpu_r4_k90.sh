#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/r4_k90; mkdir -p $OUT
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp16 or f16 or pool or wide or k90 or random" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
python tools/exp_variant.py 10000000 768 21904 90 "0,2" fp16 2>&1 | grep variant | tee $OUT/k90.txt
python tools/exp_variant.py 2074072 384 12544 90 "0,2" fp16 2>&1 | grep variant | tee -a $OUT/k90.txt
python tools/exp_variant.py 10000000 768 21904 50 "0,2" fp16 2>&1 | grep variant | tee -a $OUT/k90.txt
