#!/bin/bash
# Round 4, experiment 2: the fp16 candidate kernel with the lean stage loop -- parity tests of every fp16 path, then kernel ms at
# 10 M x 768 / cfg-2 / cfg-1 / cfg-4-ish under the cluster configurations of experiment 1.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
OUT=gpurun_out/${1:-r4_f16lean}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_edge_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "fp16 or f16 or cluster or random or pool or phase or nan or overflow or config or cfg" > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
F16="8,1,-1;8,1,-1,2;4,1,-1,2,256;4,2,-1,2,256;1,1,0"
EXP_MODES=f16 EXP_ROUNDS=3 EXP_CFGS="$F16" EXP_OUT=${1:-r4_f16lean}/f16_10M.json timeout 900 python tools/exp_cluster.py > $OUT/f16_10M.txt 2>&1; grep same_bits $OUT/f16_10M.txt
python tools/exp_phases.py 50176 384 12544 30 f16 300000 768 12544 30 f16 2074072 384 12544 30 f16 5000000 1024 21904 30 f16 > $OUT/phases.txt 2>&1; grep phases $OUT/phases.txt
